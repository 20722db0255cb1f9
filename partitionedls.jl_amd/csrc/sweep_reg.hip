// sweep_reg.hip — the production sign-pattern sweep: the principal-pivot tableau lives in REGISTERS.
//
// Replaces the loop body of fit(Opt), Opt.jl:87-90 (indextobeta + bmatrix + nonneg_lsq + objective), for every pattern of
// a Gray-code chain.  Same algorithm and decisions as sweep_generic.hip (see its header and DESIGN.md §4); what differs is
// where the tableau lives and how a pivot is executed.
//
// Data layout (one workgroup = 512 threads = 8 waves = 2 waves/SIMD on one CU, one tableau, one chain):
//   * variables are padded to 16*T; the symmetric tableau is cut into 16x16 tiles (rho, gamma), rho <= gamma only;
//   * thread (a, b) = (t & 15, (t >> 4) & 15) owns element (16 rho + a, 16 gamma + b) of EVERY stored tile — a tile-cyclic
//     layout, so every thread holds the same set of tile slots and a pivot is an outer-product update in registers:
//         S(rho,gamma) -= x[rho] * y[gamma],   x[rho] = r[16 rho + a],  y[gamma] = r[16 gamma + b] / d
//     with static register indices only (r = pivot column, staged through LDS once per pivot);
//   * v_fma_f64 can address only the 256 architectural VGPRs, so the T(T+1)/2 tile slots (153 at n = 257) are split
//     between the two halves of the workgroup by tile COLUMN: threads 0..255 hold columns gamma < G, threads 256..511
//     hold gamma >= G (78 / 75 slots at T = 17).  A half therefore needs only its own x/y ranges (24 / 22 LDS reads);
//   * the pivot column is gathered from registers by the owning threads under a wave-uniform switch on the tile index
//     (static register indices per case), written to LDS in a transposed, bank-conflict-free image R[c][rho] = r[16 rho + c];
//   * rhs column q: thread v owns q_v (v < n); the objective (corner) is replicated; KKT scan = one ballot per wave.
// One barrier per pivot (the LDS image is double buffered), one per KKT scan.  No global traffic inside a chain except the
// optional 8-byte objective per pattern.  Bound: fp64 FMA issue (NS*256 FMAs per pivot), not HBM.
#include "common.h"

namespace partls {
namespace regk {

static constexpr int THREADS = 512;
static constexpr int MAXT = 17;                 // n <= 272

constexpr int nslots(int T) { return T * (T + 1) / 2; }
constexpr int tri(int g) { return g * (g + 1) / 2; }
constexpr int split(int T)                      // tile columns [0, G) -> half 0, [G, T) -> half 1; balance the FMA count
{
    int best = 1, bestmax = 1 << 30;
    for (int g = 1; g < T; ++g) {
        int a = tri(g), b = nslots(T) - tri(g);
        int m = a > b ? a : b;
        if (m < bestmax) { bestmax = m; best = g; }
    }
    return T == 1 ? 1 : best;
}
constexpr int rstride(int T)                    // LDS image row stride (doubles): >= T and == 2 (mod 4) -> conflict-free
{
    int r = T;
    while (r % 4 != 2) ++r;
    return r;
}
constexpr int rbuf_doubles(int T) { return 16 * rstride(T) + 2; }

__device__ __forceinline__ double fast_rcp(double d)
{
    double y = __builtin_amdgcn_rcp(d);
    y = fma(fma(-d, y, 1.0), y, y);
    y = fma(fma(-d, y, 1.0), y, y);
    return y;
}

__device__ __forceinline__ int sign_of_var(uint64_t m, uint64_t pat)
{
    return 2 * __popcll(m & pat) - __popcll(m);
}

template <int T, int H>
struct Half {
    static constexpr int G = split(T);
    static constexpr int GLO = H ? G : 0;
    static constexpr int GHI = H ? T : G;
    static constexpr int OFF = H ? tri(G) : 0;
    static constexpr int CNT = (H ? nslots(T) - tri(G) : tri(G)) > 0 ? (H ? nslots(T) - tri(G) : tri(G)) : 1;
    static constexpr int XN = GHI;              // rows rho < GHI occur in this half
    static constexpr int YN = GHI - GLO > 0 ? GHI - GLO : 1;
    static constexpr int RS = rstride(T);
    __device__ static constexpr int idx(int rho, int gam) { return tri(gam) + rho - OFF; }
};

// ---- column gather (case KAPPA of the switch) -----------------------------------------------------------------------
template <int T, int H, int KAPPA, class SA>
__device__ __forceinline__ void gather_case(const SA &S, double *R, int a, int b, int beta)
{
    using L = Half<T, H>;
    if constexpr (KAPPA >= L::GLO && KAPPA < L::GHI) {
        if (b == beta) {
#pragma unroll
            for (int rho = 0; rho <= KAPPA; ++rho) R[a * L::RS + rho] = S[L::idx(rho, KAPPA)];
        }
    }
    if (a == beta) {
#pragma unroll
        for (int gam = (KAPPA + 1 > L::GLO ? KAPPA + 1 : L::GLO); gam < L::GHI; ++gam)
            R[b * L::RS + gam] = S[L::idx(KAPPA, gam)];
    }
}

template <int T, int H, int KAPPA, class SA>
__device__ __forceinline__ void fixup_case(SA &S, const double *R, double ainv, double ninv, int a, int b, int beta)
{
    using L = Half<T, H>;
    if constexpr (KAPPA >= L::GLO && KAPPA < L::GHI) {
        if (b == beta) {
#pragma unroll
            for (int rho = 0; rho <= KAPPA; ++rho) S[L::idx(rho, KAPPA)] = R[a * L::RS + rho] * ainv;
        }
    }
    if (a == beta) {
#pragma unroll
        for (int gam = (KAPPA > L::GLO ? KAPPA : L::GLO); gam < L::GHI; ++gam)
            S[L::idx(KAPPA, gam)] = R[b * L::RS + gam] * ainv;
    }
    if constexpr (KAPPA >= L::GLO && KAPPA < L::GHI) {
        if (a == beta && b == beta) S[L::idx(KAPPA, KAPPA)] = ninv;
    }
}

#define PARTLS_CASES(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16)

template <int T, int H>
__device__ __forceinline__ void sweep_body(const SweepParams &p, double *lds)
{
    using L = Half<T, H>;
    constexpr int RS = L::RS, RB = rbuf_doubles(T);
    const int tid = threadIdx.x, t8 = tid & 255, a = t8 & 15, b = t8 >> 4, lane = tid & 63, wave = tid >> 6;
    const int n = p.n;
    const int nwords = (n + 63) >> 6;

    unsigned long long *s_inf = reinterpret_cast<unsigned long long *>(lds + 2 * RB);   // [2][8]
    unsigned long long *s_bas = s_inf + 16;                                               // [2][8]

    double S[L::CNT];
    double q = 0.0, corner = 0.0;
    const bool has_var = tid < n;
    const uint64_t vmask = has_var ? p.mask[tid] : 0ULL;
    bool basic = false, blocked = false;

    double best_obj = __builtin_inf();
    long long best_pat = -1;
    unsigned long long npiv = 0, nunconv = 0;
    unsigned pc = 0, sc = 0;                                       // pivot / scan counters (double-buffer parity)

    const int64_t total = p.g_end - p.g_begin;
    const int64_t nchains = (total + p.chain_len - 1) / p.chain_len;

    for (int64_t chain = blockIdx.x; chain < nchains; chain += gridDim.x) {
        const int64_t g0 = p.g_begin + chain * p.chain_len;
        const int64_t g1 = (g0 + p.chain_len < p.g_end) ? g0 + p.chain_len : p.g_end;
        // ---- fresh tableau: the shared Gram block, coalesced (slot-major, 256 consecutive doubles per slot) ----------
#pragma unroll
        for (int s = 0; s < L::CNT; ++s) S[s] = p.T0[(size_t)(s + L::OFF) * 256 + t8];
        q = (tid < 16 * T) ? p.T0[(size_t)nslots(T) * 256 + tid] : 0.0;
        corner = p.T0[(size_t)nslots(T) * 256 + 16 * T];
        basic = false;

        for (int64_t g = g0; g < g1; ++g) {
            uint64_t pat = (uint64_t)g ^ ((uint64_t)g >> 1);
            bool isfree = false;
            int f;
            if (p.node_pat) {
                pat = p.node_pat[chain];
                isfree = (vmask & p.node_free[chain]) != 0;
                f = (vmask & p.node_zero[chain]) ? 0 : sign_of_var(vmask, pat);
            } else {
                f = sign_of_var(vmask, pat);
            }
            blocked = false;
            int ninf_best = n + 1, patience = 3, rounds = 0;
            bool progress = false;
            for (;;) {
                if (progress) blocked = false;                 // rejections hold for the basis they were tested against only
                progress = false;
                // ---- KKT scan of the rhs column (registers) ------------------------------------------------------------
                const int par = sc & 1;
                ++sc;
                bool bad = false;
                if (has_var) {
                    const double fq = (f > 0) ? q : ((f < 0) ? -q : 0.0);
                    if (isfree) bad = !basic && !blocked && (fabs(q) > p.tol);
                    else if (basic) bad = (f == 0) || (fq < -p.tol);
                    else bad = (fq > p.tol) && !blocked;
                }
                const unsigned long long bb = __ballot(bad), bs = __ballot(basic);
                if (lane == 0 && wave < nwords) { s_inf[par * 8 + wave] = bb; s_bas[par * 8 + wave] = bs; }
                __syncthreads();
                int count = 0;
                for (int w = 0; w < nwords; ++w) count += __popcll(s_inf[par * 8 + w]);
                if (count == 0) break;
                bool all;
                if (count < ninf_best) { ninf_best = count; patience = 3; all = true; }
                else if (patience > 0) { --patience; all = true; }
                else all = false;
                if (++rounds > p.max_rounds) { ++nunconv; break; }

                int w = all ? 0 : nwords - 1;
                unsigned long long bits = s_inf[par * 8 + w];
                for (;;) {
                    // next pivot index (uniform): ascending over all violators, or the single largest one (backup rule)
                    int k;
                    if (all) {
                        while (bits == 0 && w + 1 < nwords) { ++w; bits = s_inf[par * 8 + w]; }
                        if (bits == 0) break;
                        k = (w << 6) + __builtin_ctzll(bits);
                        bits &= bits - 1;
                    } else {
                        while (bits == 0 && w > 0) { --w; bits = s_inf[par * 8 + w]; }
                        if (bits == 0) break;
                        k = (w << 6) + 63 - __builtin_clzll(bits);
                        bits = 0; w = 0;
                    }
                    k = __builtin_amdgcn_readfirstlane(k);
                    const bool k_basic = (s_bas[par * 8 + (k >> 6)] >> (k & 63)) & 1ULL;
                    const int kappa = k >> 4, beta = k & 15;
                    double *R = lds + (pc & 1) * RB;                  // keep the LDS address space (no flat accesses)
                    ++pc;
                    // ---- gather column k into the LDS image ------------------------------------------------------------
#define PARTLS_G(i) if constexpr (i < T) { if (kappa == i) gather_case<T, H, i>(S, R, a, b, beta); }
                    PARTLS_CASES(PARTLS_G)
#undef PARTLS_G
                    if (tid == k) R[16 * RS] = q;
                    __syncthreads();
                    const double d = R[beta * RS + kappa];
                    if (!k_basic && !(d > p.piv_eps)) {            // dependent column (Lawson–Hanson's rejection): skip
                        if (tid == k) blocked = true;
                        if (!all) break;
                        continue;
                    }
                    const double rq = R[16 * RS];
                    const double inv = fast_rcp(d);
                    const double ainv = fabs(inv);
                    double x[L::XN];
#pragma unroll
                    for (int rho = 0; rho < L::XN; ++rho) x[rho] = R[a * RS + rho];
                    // ---- rank-1 update of the owned tile slots (y streamed from LDS, one tile column at a time) -------
#pragma unroll
                    for (int gam = L::GLO; gam < L::GHI; ++gam) {
                        const double yg = -R[b * RS + gam] * inv;
#pragma unroll
                        for (int rho = 0; rho <= gam; ++rho)
                            S[L::idx(rho, gam)] = fma(x[rho], yg, S[L::idx(rho, gam)]);
                    }
                    const double rqi = rq * inv;
                    if (tid < 16 * T) q = (tid == k) ? rq * ainv : fma(-R[(tid & 15) * RS + (tid >> 4)], rqi, q);
                    corner = fma(-rq, rqi, corner);
                    // ---- row / column k of the swept tableau -----------------------------------------------------------
#define PARTLS_F(i) if constexpr (i < T) { if (kappa == i) fixup_case<T, H, i>(S, R, ainv, -inv, a, b, beta); }
                    PARTLS_CASES(PARTLS_F)
#undef PARTLS_F
                    if (tid == k) basic = !basic;
                    progress = true;
                    ++npiv;
                    if (!all) break;
                }
            }
            const double obj = sqrt(corner > 0.0 ? corner : 0.0);
            if (p.all_opt && tid == 0) p.all_opt[pat] = obj;
            if (obj < best_obj || (obj == best_obj && (long long)pat < best_pat)) { best_obj = obj; best_pat = (long long)pat; }
        }
        if (p.node_sol) {
            if (has_var) p.node_sol[(size_t)chain * p.node_ld + tid] = basic ? q : 0.0;
            if (tid == 0) p.node_obj2[chain] = corner;
        }
    }
    if (tid == 0) {
        p.best_obj[blockIdx.x] = best_obj;
        p.best_pat[blockIdx.x] = best_pat;
        if (p.n_pivots && npiv) atomicAdd(p.n_pivots, npiv);
        if (p.n_unconverged && nunconv) atomicAdd(p.n_unconverged, nunconv);
    }
}

template <int T>
__global__ __launch_bounds__(THREADS, 2) void sweep_reg_kernel(SweepParams p)
{
    extern __shared__ double lds[];
    const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));   // wave-uniform: scalar branch
    if (half == 0) sweep_body<T, 0>(p, lds);
    else sweep_body<T, 1>(p, lds);
}

// Tfull ((n+1)^2) -> tile-cyclic initial state: [slot = tri(gamma) + rho][256 = a + 16 b], then q0[16 T], then the corner
__global__ void layout_reg_kernel(const double *__restrict__ Tfull, int n, int T, double *__restrict__ out)
{
    const int ld = n + 1;
    const int ns = T * (T + 1) / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int tot = ns * 256 + 16 * T + 1;
    if (idx >= tot) return;
    if (idx < ns * 256) {
        const int s = idx >> 8, t8 = idx & 255, a = t8 & 15, b = t8 >> 4;
        int gam = 0;
        while ((gam + 1) * (gam + 2) / 2 <= s) ++gam;
        const int rho = s - gam * (gam + 1) / 2;
        const int i = 16 * rho + a, j = 16 * gam + b;
        out[idx] = (i < n && j < n) ? Tfull[(size_t)i * ld + j] : ((i == j) ? 1.0 : 0.0);
    } else if (idx < ns * 256 + 16 * T) {
        const int v = idx - ns * 256;
        out[idx] = (v < n) ? Tfull[(size_t)v * ld + n] : 0.0;
    } else {
        out[idx] = Tfull[(size_t)n * ld + n];
    }
}

}  // namespace regk

bool sweep_reg_supported(int n) { return n >= 1 && n <= 16 * regk::MAXT; }
int sweep_reg_tiles(int n) { return (n + 15) / 16; }
size_t sweep_reg_t0_doubles(int T) { return (size_t)T * (T + 1) / 2 * 256 + 16 * (size_t)T + 8; }

hipError_t launch_layout_reg(const double *Tfull, int n, int T, double *T0reg, hipStream_t s)
{
    const int tot = T * (T + 1) / 2 * 256 + 16 * T + 1;
    hipLaunchKernelGGL(regk::layout_reg_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, Tfull, n, T, T0reg);
    return hipGetLastError();
}

template <int T>
static hipError_t launch_T(const SweepParams &p, int grid, hipStream_t s)
{
    const size_t shmem = (size_t)2 * regk::rbuf_doubles(T) * sizeof(double) + 32 * sizeof(unsigned long long);
    hipLaunchKernelGGL(regk::sweep_reg_kernel<T>, dim3(grid), dim3(regk::THREADS), shmem, s, p);
    return hipGetLastError();
}

hipError_t launch_sweep_reg(const SweepParams &p, int T, int grid, hipStream_t s)
{
    switch (T) {
#define PARTLS_L(i) case i + 1: return launch_T<i + 1>(p, grid, s);
        PARTLS_CASES(PARTLS_L)
#undef PARTLS_L
        default: return hipErrorInvalidValue;
    }
}

}  // namespace partls
