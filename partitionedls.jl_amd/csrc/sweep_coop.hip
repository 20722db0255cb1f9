// sweep_coop.hip — ONE large subproblem solved by many workgroups (cooperative launch, grid-wide barriers).
//
// Used for single "node" solves whose tableau does not fit the register-resident kernel (n > 272): the α-step of
// fit(Alt) at BASELINE config 4 (n = 513, Alt.jl:80-90) and the winner re-solve of fit(Opt) at such sizes.  Same
// algorithm and decisions as sweep_generic.hip; the (n+1)^2 tableau lives in global memory (2.1 MB at n = 513,
// L2 resident) and every workgroup owns a slice of its ROWS.  All control state (basis flags, rejections, the violator
// masks) is replicated per workgroup and evolves identically everywhere because every workgroup reads the same
// pivot row / rhs row after each grid barrier — no cross-workgroup messages besides the tableau itself.
// Per pivot: copy row k (= column k by symmetry) to LDS -> grid.sync() -> rank-1 update of the owned rows -> grid.sync().
#include "common.h"
#include <hip/hip_cooperative_groups.h>

namespace cg = cooperative_groups;

namespace partls {

static constexpr int COOP_THREADS = 256;
static constexpr int COOP_MAXWORDS = 16;

__device__ __forceinline__ int coop_sign_of_var(uint64_t m, uint64_t pat) { return 2 * __popcll(m & pat) - __popcll(m); }

__global__ __launch_bounds__(COOP_THREADS) void sweep_coop_kernel(SweepParams p)
{
    cg::grid_group grid = cg::this_grid();
    const int n = p.n, ld = n + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    extern __shared__ double smem[];
    double *r = smem;                                             // pivot row, ld doubles
    uint8_t *s_basic = reinterpret_cast<uint8_t *>(smem + ld);
    uint8_t *s_blocked = s_basic + n;
    __shared__ unsigned long long s_inf[COOP_MAXWORDS];

    double *T = p.scratch;
    const int nwg = gridDim.x, wg = blockIdx.x;
    const int rows_per = (ld + nwg - 1) / nwg;
    const int row0 = wg * rows_per, row1 = (row0 + rows_per < ld) ? row0 + rows_per : ld;
    const int nwords = (n + 63) >> 6;

    uint8_t *flagbuf = reinterpret_cast<uint8_t *>(T + (size_t)ld * ld);       // basis flags kept across launches
    if (!p.resume) {
        for (int i = row0 + wave; i < row1; i += COOP_THREADS / 64)
            for (int j = lane; j < ld; j += 64) T[(size_t)i * ld + j] = p.T0[(size_t)i * ld + j];
        for (int i = tid; i < n; i += COOP_THREADS) { s_basic[i] = 0; s_blocked[i] = 0; }
    } else {
        for (int i = tid; i < n; i += COOP_THREADS) { s_basic[i] = flagbuf[i]; s_blocked[i] = 0; }
    }
    const uint64_t pat = p.node_pat[0], gfree = p.node_free[0], gzero = p.node_zero[0];
    __threadfence();
    grid.sync();
    __threadfence();

    unsigned long long npiv = 0, nunconv = 0;
    int ninf_best = n + 1, patience = 3, rounds = 0;
    bool progress = false;
    for (;;) {
        if (progress) { for (int i = tid; i < n; i += COOP_THREADS) s_blocked[i] = 0; __syncthreads(); }
        progress = false;
        // ---- KKT scan of the rhs row (every workgroup, identically) ------------------------------------------------------
        for (int base = 0; base < nwords * 64; base += COOP_THREADS) {
            const int v = base + tid;
            bool bad = false;
            if (v < n) {
                const double q = __builtin_nontemporal_load(&T[(size_t)n * ld + v]);
                const uint64_t vm = p.mask[v];
                const int f = (vm & gzero) ? 0 : coop_sign_of_var(vm, pat);
                const double fq = (f > 0) ? q : ((f < 0) ? -q : 0.0);
                if (vm & gfree) bad = !s_basic[v] && !s_blocked[v] && (fabs(q) > p.tol);
                else if (s_basic[v]) bad = (f == 0) || (fq < -p.tol);
                else bad = (fq > p.tol) && !s_blocked[v];
            }
            const unsigned long long b = __ballot(bad);
            if (lane == 0 && (v >> 6) < nwords) s_inf[v >> 6] = b;
        }
        __syncthreads();
        int count = 0;
        for (int w = 0; w < nwords; ++w) count += __popcll(s_inf[w]);
        if (count == 0) break;
        bool all;
        if (count < ninf_best) { ninf_best = count; patience = 3; all = true; }
        else if (patience > 0) { --patience; all = true; }
        else all = false;
        if (++rounds > p.max_rounds) { ++nunconv; break; }

        int w = all ? 0 : nwords - 1;
        unsigned long long bits = s_inf[w];
        for (;;) {
            int k;
            if (all) {
                while (bits == 0 && w + 1 < nwords) { ++w; bits = s_inf[w]; }
                if (bits == 0) break;
                k = (w << 6) + __builtin_ctzll(bits);
                bits &= bits - 1;
            } else {
                while (bits == 0 && w > 0) { --w; bits = s_inf[w]; }
                if (bits == 0) break;
                k = (w << 6) + 63 - __builtin_clzll(bits);
                bits = 0; w = 0;
            }
            // ---- pivot k -------------------------------------------------------------------------------------------------
            for (int i = tid; i < ld; i += COOP_THREADS) r[i] = __builtin_nontemporal_load(&T[(size_t)k * ld + i]);
            __syncthreads();
            grid.sync();                                   // everybody holds row k before its owner rewrites it
            __threadfence();
            const double d = r[k];
            if (!s_basic[k] && !(d > p.piv_eps)) {         // dependent column: rejected for the current basis
                __syncthreads();
                if (tid == 0) s_blocked[k] = 1;
                __syncthreads();
                if (!all) break;
                continue;
            }
            const double inv = 1.0 / d, ainv = 1.0 / fabs(d);
            for (int i = row0 + wave; i < row1; i += COOP_THREADS / 64) {
                const double ri = r[i], mi = -ri * inv;
                double *row = T + (size_t)i * ld;
                if (i == k) {
                    for (int j = lane; j < ld; j += 64) row[j] = (j == k) ? -inv : r[j] * ainv;
                } else {
                    for (int j = lane; j < ld; j += 64) row[j] = (j == k) ? ri * ainv : fma(mi, r[j], row[j]);
                }
            }
            __syncthreads();
            if (tid == 0) s_basic[k] ^= 1;
            __threadfence();
            grid.sync();                                   // the whole tableau is updated before the next row is read
            __threadfence();                               // agent-scope acquire on this CU (per-XCD L2s are not coherent)
            progress = true;
            ++npiv;
            if (!all) break;
        }
        __syncthreads();
    }
    if (wg == 0) {
        for (int i = tid; i < n; i += COOP_THREADS) flagbuf[i] = s_basic[i];
        for (int i = tid; i < n; i += COOP_THREADS)
            p.node_sol[i] = s_basic[i] ? __builtin_nontemporal_load(&T[(size_t)n * ld + i]) : 0.0;
        if (tid == 0) {
            p.node_obj2[0] = __builtin_nontemporal_load(&T[(size_t)n * ld + n]);
            p.best_obj[0] = 0.0; p.best_pat[0] = (int64_t)pat;
            if (p.n_pivots && npiv) atomicAdd(p.n_pivots, npiv);
            if (p.n_unconverged && nunconv) atomicAdd(p.n_unconverged, nunconv);
        }
    }
}

hipError_t launch_sweep_coop(const SweepParams &p, int nwg, hipStream_t s)
{
    const size_t shmem = (size_t)(p.n + 1) * sizeof(double) + 2 * (size_t)p.n + 16;
    SweepParams pc = p;
    void *args[] = {&pc};
    return hipLaunchCooperativeKernel(reinterpret_cast<const void *>(&sweep_coop_kernel), dim3(nwg), dim3(COOP_THREADS), args,
                                      (unsigned)shmem, s);
}

}  // namespace partls
