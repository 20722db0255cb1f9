// sweep_generic.hip — sign-pattern sweep with the tableau in global memory (one workgroup per Gray-code chain).
//
// Replaces the loop body of fit(Opt) — indextobeta + bmatrix + nonneg_lsq + objective, Opt.jl:87-90 — for a whole
// range of patterns, and the node bound of BnB (BnB.jl:69-92) / the α-step of Alt (Alt.jl:80-90) for single patterns.
//
// Algorithm.  The subproblem of pattern s is  min ||Xo w - y||  s.t.  f_m w_m >= 0  with
// f_m = sum_k Po[m,k] s_k (Opt.jl:28-29); it depends on the data only through the Gram block.  We keep the symmetric
// principal-pivot tableau of the current passive ("basic") set B:
//        T = sweep_B [[G, c], [c', yy]]   =>   T[i][n] = w_i (i in B),  T[i][n] = c_i - G_iB w_B (i not in B),  T[n][n] = obj^2
// which does NOT depend on the signs.  KKT for pattern s reads off the rhs column:  f_i w_i >= 0 (i in B),
// f_i (c - G w)_i <= 0 (i not in B).  Consecutive patterns of a Gray-code walk differ in one group, so only that
// group's variables (plus a few neighbours) violate KKT: they are exchanged by block principal pivoting
// (all violators at once; Kim & Park's finite-termination backup rule), one symmetric rank-1 sweep per variable:
//        T_ij -= T_ik T_kj / d ,  T_ik = T_ik / |d| ,  T_kk = -1/d        (d = T_kk; the same formula enters and removes).
// This kernel is the fully general form: one tableau per workgroup in global scratch, of which only the UPPER TRIANGLE is kept (entries
// (i, c) with c >= i in an (n+1)^2 image; the rhs is its column n): the tableau is symmetric, and the bytes of the fused update are
// what bounds this kernel (round 3: 0.84 -> see DESIGN.md §6 M solves/s at D = 340).  With 256 concurrent
// tableaus of 2 MB (n = 513) the passes over the tableau are HBM-bound, so the violators of a scan are exchanged in BLOCKS of up
// to 16 pivots (gj_panel.h: panel of the pivot columns eliminated in LDS, then ONE fused rank-m pass over the tableau) instead
// of one full pass per pivot.  The register-resident production kernel is sweep_blk.hip; all share
// SweepParams and agree in their decisions up to rounding.
#include "gj_panel.h"

namespace partls {

static constexpr int GEN_THREADS = 1024;   // 16 waves per CU: the fused update is bound by memory latency, not by issue
static constexpr int GEN_MAXWORDS = 16;     // n <= 1024

__device__ __forceinline__ int sign_of_var(uint64_t m, uint64_t pat)
{
    // f = sum_k P[v,k] * s_k with s_k = +1 if bit k of pat else -1  ==  2*popc(m & pat) - popc(m)
    return 2 * __popcll(m & pat) - __popcll(m);
}

__global__ __launch_bounds__(GEN_THREADS) void sweep_generic_kernel(SweepParams p, int mb)
{
    const int n = p.n, ld = n + 1;
    const int tid = threadIdx.x, lane = tid & 63;
    extern __shared__ double smem[];
    double *Pn = smem;                                            // [mb][ld] panel: the block's pivot columns, all rows
    double *Zn = Pn + (size_t)mb * ld;                            // [mb][ld] column s as of its own step
    double *dinv = Zn + (size_t)mb * ld;                          // [GJ_MB] 1/d_s (0: rejected)
    double *uj = dinv + GJ_MB;                                    // [2][GJ_MB] pivot-row entries of the current / next step
    double *red = uj + 2 * GJ_MB;                                     // [GEN_THREADS / 64] reduction scratch
    uint8_t *s_basic = reinterpret_cast<uint8_t *>(red + GEN_THREADS / 64);   // n bytes
    uint8_t *s_blocked = s_basic + n;                             // n bytes
    __shared__ unsigned long long s_inf[GEN_MAXWORDS];
    __shared__ int s_viol[GEN_MAXWORDS * 64];
    __shared__ int s_nv;

    double *T = p.scratch + (size_t)blockIdx.x * (size_t)ld * (size_t)ld;
    const int nwords = (n + 63) >> 6;

    double best_obj = __builtin_inf();
    long long best_pat = -1;
    double second_obj = __builtin_inf();                           // runner-up of this workgroup (near-tie re-rank on the host)
    long long second_pat = -1;
    unsigned long long npiv = 0, nunconv = 0;

    const int64_t total = p.g_end - p.g_begin;
    const int64_t nchains = (total + p.chain_len - 1) / p.chain_len;

    for (int64_t chain = blockIdx.x; chain < nchains; chain += gridDim.x) {
        const int64_t g0 = p.g_begin + chain * p.chain_len;
        const int64_t g1 = (g0 + p.chain_len < p.g_end) ? g0 + p.chain_len : p.g_end;
        for (int idx = tid; idx < ld * ld; idx += GEN_THREADS) T[idx] = p.T0[idx];
        for (int i = tid; i < n; i += GEN_THREADS) { s_basic[i] = 0; s_blocked[i] = 0; }
        __syncthreads();

        for (int64_t g = g0; g < g1; ++g) {
            uint64_t pat = (uint64_t)g ^ ((uint64_t)g >> 1);
            // node mode: per-variable codes; a chain of nodes (bit-order calibration) warm-starts node i from node i - 1
            const int8_t *code = p.node_code ? p.node_code + ((size_t)chain * p.chain_len + (size_t)(g - g0)) * p.node_ld : nullptr;
            if (code) pat = (uint64_t)chain;
            for (int i = tid; i < n; i += GEN_THREADS) s_blocked[i] = 0;
            __syncthreads();
            int ninf_best = n + 1, patience = 3, rounds = 0;
            bool progress = false;
            for (;;) {
                if (progress) {                                    // rejections hold for the basis they were tested against only
                    for (int i = tid; i < n; i += GEN_THREADS) s_blocked[i] = 0;
                    __syncthreads();
                }
                progress = false;
                // ---- KKT scan of the rhs column -----------------------------------------------------------------
                for (int base = 0; base < nwords * 64; base += GEN_THREADS) {
                    const int v = base + tid;
                    bool bad = false;
                    if (v < n) {
                        const double q = T[(size_t)v * ld + n];               // the rhs is COLUMN n of the stored upper triangle
                        const int cd = code ? (int)code[v] : 0;
                        const int f = code ? (cd == 2 ? 0 : cd) : sign_of_var(p.mask[v], pat);
                        const double fq = (f > 0) ? q : ((f < 0) ? -q : 0.0);
                        if (cd == 2) bad = !s_basic[v] && !s_blocked[v] && (fabs(q) > p.tol);     // free: stationarity only
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
                else all = false;                                  // backup rule: single pivot, largest index
                if (++rounds > p.max_rounds) { ++nunconv; break; }
                // the violator list of this round (ascending; the backup rule keeps only the last one), exchanged in blocks
                if (tid == 0) {
                    int nv = 0;
                    for (int w = 0; w < nwords; ++w) {
                        unsigned long long bits = s_inf[w];
                        while (bits) { s_viol[nv++] = (w << 6) + __builtin_ctzll(bits); bits &= bits - 1; }
                    }
                    if (!all) { s_viol[0] = s_viol[nv - 1]; nv = 1; }
                    s_nv = nv;
                }
                __syncthreads();
                const int nv = s_nv;
                for (int b0 = 0; b0 < nv; b0 += mb) {
                    const int m = (nv - b0 < mb) ? nv - b0 : mb;
                    const int *ks = s_viol + b0;
                    gj_panel_load<GEN_THREADS, true>(T, ld, ks, m, Pn, tid);
                    const int acc = gj_panel_eliminate<GEN_THREADS>(Pn, Zn, dinv, uj, red, ks, m, ld, s_basic, p.piv_eps, tid);
                    gj_apply<GEN_THREADS, true>(T, ld, 0, ld, Pn, Zn, dinv, ks, m, tid);
                    if (tid == 0) {
                        for (int j = 0; j < m; ++j) {
                            if (dinv[j] == 0.0) s_blocked[ks[j]] = 1;        // accepted pivots flipped s_basic in the panel
                        }
                    }
                    if (acc) { npiv += (unsigned)acc; progress = true; }
                    __threadfence_block();
                    __syncthreads();
                }
            }
            const double obj2 = T[(size_t)n * ld + n];
            const double obj = sqrt(obj2 > 0.0 ? obj2 : 0.0);
            if (p.all_opt && tid == 0) p.all_opt[pat] = obj;
            if (obj < best_obj || (obj == best_obj && best_pat >= 0 && ref_index_less(pat, (unsigned long long)best_pat, p.rbit.gbit))) {
                second_obj = best_obj; second_pat = best_pat;
                best_obj = obj; best_pat = (long long)pat;
            } else if (obj < second_obj) { second_obj = obj; second_pat = (long long)pat; }
            if (p.node_piv && code && tid == 0) {                  // [pivots, blocks, scans] so far; this kernel reports pivots only
                unsigned *o = p.node_piv + 3 * ((size_t)chain * p.chain_len + (size_t)(g - g0));
                o[0] = (unsigned)npiv; o[1] = 0; o[2] = 0;
            }
            __syncthreads();
        }
        if (p.node_sol) {
            for (int i = tid; i < n; i += GEN_THREADS)
                p.node_sol[(size_t)chain * p.node_ld + i] = s_basic[i] ? T[(size_t)i * ld + n] : 0.0;
            if (tid == 0) p.node_obj2[chain] = T[(size_t)n * ld + n];
        }
        __syncthreads();
    }
    if (tid == 0) {
        p.best_obj[blockIdx.x] = best_obj;
        p.best_pat[blockIdx.x] = best_pat;
        if (p.second_obj) { p.second_obj[blockIdx.x] = second_obj; p.second_pat[blockIdx.x] = second_pat; }
        if (p.n_pivots && npiv) atomicAdd(p.n_pivots, npiv);
        if (p.n_unconverged && nunconv) atomicAdd(p.n_unconverged, nunconv);
    }
}

hipError_t launch_sweep_generic(const SweepParams &p, int grid, hipStream_t s)
{
    const int ld = p.n + 1;
    int mb = gj_block_size(ld, (size_t)136 * 1024);
    const size_t shmem = (size_t)2 * mb * ld * sizeof(double) + (3 * GJ_MB + GEN_THREADS / 64) * sizeof(double) + 2 * (size_t)p.n + 16;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_generic_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sweep_generic_kernel, dim3(grid), dim3(GEN_THREADS), shmem, s, p, mb);
    return hipGetLastError();
}

}  // namespace partls
