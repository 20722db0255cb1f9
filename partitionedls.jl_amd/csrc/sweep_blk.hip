// sweep_blk.hip — production sign-pattern sweep, BLOCKED principal pivots on a register-resident tableau.
//
// Replaces the loop body of fit(Opt), Opt.jl:87-90, for every pattern of a Gray-code chain (same mathematics and the
// same per-pattern decisions as sweep_generic.hip / sweep_reg.hip).  What is new relative to sweep_reg.hip:
// the violators of one KKT scan are exchanged one 16x16 TILE COLUMN at a time as a block pivot
//     T'_RR = T_RR - sum_s z_s z_s' / d_s          (z_s = pivot column as of its own step, d_s its pivot)
// so that the expensive part — the update of the register tableau — is ONE fused rank-m pass of uninterrupted FMAs, and
// the per-pivot overheads of the rank-1 kernel (column gather under a scalar if-chain, barrier, LDS round trip, fix-up
// if-chain) are paid once per tile instead of once per pivot:
//   1. gather   : every thread writes its slots of tile column kappa to an LDS panel P[16 cols][rows] (+ rhs row);
//   2. panel    : thread t owns ROW t of the panel in registers (<= 16 pivot columns); m sequential Gauss–Jordan steps,
//                 each exchanging only the m pivot-row entries through LDS (one barrier per step); z_s and 1/d_s recorded;
//   3. update   : all 512 threads: S(rho,gamma) -= x_s[rho] * y_s[gamma] for s = 1..m (x resident, y streamed from LDS);
//   4. scatter  : rows/columns of the pivoted variables are overwritten from the final panel (one if-chain per tile).
// Layout of the register tableau, thread grid, half split and LDS image are those of sweep_reg.hip.
#include "blk_common.h"

namespace partls {
namespace blk {

template <int T, int H>
__device__ __forceinline__ void sweep_body(const SweepParams &p, double *lds)
{
    using L = Half<T, H>;
    constexpr int RS = L::RS, CW = L::CW, RHSPOS = 16 * RS;
    const int tid = threadIdx.x, t8 = tid & 255, a = t8 & 15, b = t8 >> 4, lane = tid & 63, wave = tid >> 6;
    const int n = p.n;
    const int nwords = (n + 63) >> 6;

    double *Pbase = lds;                                  // [2][MB][CW]
    double *Z = lds + 2 * MB * CW;                        // [MB][CW]
    double *U = Z + MB * CW;                              // [2][MB+64]
    double *Dinv = U + 2 * (MB + 64);                     // [MB+64]
    unsigned long long *s_inf = reinterpret_cast<unsigned long long *>(Dinv + MB + 64);  // [2][8]
    unsigned long long *s_bas = s_inf + 16;                                            // [2][8]

    for (int i = tid; i < lds_doubles(T) + 32; i += THREADS) lds[i] = 0.0;            // padding rows are never gathered
    __syncthreads();

    double S[L::CNT];
    double q = 0.0, corner = 0.0;
    const bool has_var = tid < n;
    const uint64_t vmask = has_var ? p.mask[tid] : 0ULL;
    bool basic = false, blocked = false;
    const int mypos = (tid & 15) * RS + (tid >> 4);       // panel row position of variable `tid` (tid < 16 T)
    // panel phase: this thread owns row position `tid`, i.e. variable 16*rowrho + rowc (valid when rowc < 16)
    const int rowc = tid / RS, rowrho = tid - rowc * RS;
    const bool idle_wave = __builtin_amdgcn_readfirstlane((tid & ~63) > RHSPOS ? 1 : 0) != 0;   // no panel row in this wave

    double best_obj = __builtin_inf();
    long long best_pat = -1;
    unsigned long long npiv = 0, nunconv = 0;
    unsigned bc = 0, sc = 0;                              // block / scan counters (double-buffer parity)

    const int64_t total = p.g_end - p.g_begin;
    const int64_t nchains = (total + p.chain_len - 1) / p.chain_len;

    STAMP_DECL
    for (int64_t chain = blockIdx.x; chain < nchains; chain += gridDim.x) {
        const int64_t g0 = p.g_begin + chain * p.chain_len;
        const int64_t g1 = (g0 + p.chain_len < p.g_end) ? g0 + p.chain_len : p.g_end;
        STAMP(5);
#pragma unroll
        for (int s = 0; s < L::CNT; ++s) S[s] = p.T0[(size_t)(s + L::OFF) * 256 + t8];
        q = (tid < 16 * T) ? p.T0[(size_t)nslots(T) * 256 + tid] : 0.0;
        corner = p.T0[(size_t)nslots(T) * 256 + 16 * T];
        basic = false;
        STAMP(6);

        for (int64_t g = g0; g < g1; ++g) {
            uint64_t pat = (uint64_t)g ^ ((uint64_t)g >> 1);
            bool isfree = false;
            int f;
            if (p.node_pat) {                                       // node mode: chain index = node index
                pat = p.node_pat[chain];
                isfree = (vmask & p.node_free[chain]) != 0;
                f = (vmask & p.node_zero[chain]) ? 0 : sign_of_var(vmask, pat);
            } else {
                f = sign_of_var(vmask, pat);
            }
            blocked = false;
            int ninf_best = n + 1, patience = 3, rounds = 0;
            bool progress = false;                                    // did the previous round change the basis?
            for (;;) {
                // ---- KKT scan of the rhs column (registers) ------------------------------------------------------------
                // a column rejected as dependent is only dependent on the basis it was tested against (Lawson–Hanson
                // re-examines it after any exchange): forget the rejections once the basis has changed
                if (progress) blocked = false;
                progress = false;
                const int par = sc & 1;
                ++sc;
                bool bad = false;
                if (has_var) {
                    const double fq = (f > 0) ? q : ((f < 0) ? -q : 0.0);
                    if (isfree) bad = !basic && !blocked && (fabs(q) > p.tol);     // free: stationarity only
                    else if (basic) bad = (f == 0) || (fq < -p.tol);
                    else bad = (fq > p.tol) && !blocked;
                }
                const unsigned long long bb = __ballot(bad), bs = __ballot(basic);
                if (lane == 0 && wave < nwords) { s_inf[par * 8 + wave] = bb; s_bas[par * 8 + wave] = bs; }
                STAMP(9);
                __syncthreads();
                STAMP(10);
                // mask words are read once, reduced to (count, tile mask, largest violator) and dropped
                int count = 0, single_k = -1;
                unsigned tiles = 0;
#pragma unroll
                for (int w = 0; w < 5; ++w) {
                    unsigned long long ww = (w < nwords) ? s_inf[par * 8 + w] : 0ULL;
                    ww = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ww >> 32)) << 32) |
                         (unsigned)__builtin_amdgcn_readfirstlane((int)ww);             // uniform -> SALU bookkeeping
                    count += __popcll(ww);
                    if (ww) single_k = (w << 6) + 63 - __builtin_clzll(ww);
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub)
                        if ((ww >> (16 * sub)) & 0xFFFFull) tiles |= 1u << (4 * w + sub);
                }
                STAMP(0);
                if (count == 0) break;
                bool all;
                if (count < ninf_best) { ninf_best = count; patience = 3; all = true; }
                else if (patience > 0) { --patience; all = true; }
                else all = false;                                     // backup rule: only the largest violator
                if (++rounds > p.max_rounds) { ++nunconv; break; }
                if (!all) tiles = 1u << (single_k >> 4);
                tiles = (unsigned)__builtin_amdgcn_readfirstlane((int)tiles);

                while (tiles) {
                    const int kappa = __builtin_ctz(tiles);
                    const int wsel = kappa >> 2, sh = 16 * (kappa & 3);
                    unsigned pmall = (unsigned)((s_inf[par * 8 + wsel] >> sh) & 0xFFFFull);
                    if (!all) pmall = 1u << (single_k & 15);
                    const unsigned basm = (unsigned)__builtin_amdgcn_readfirstlane((int)((s_bas[par * 8 + wsel] >> sh) & 0xFFFFull));
                    pmall = (unsigned)__builtin_amdgcn_readfirstlane((int)pmall);
                    while (pmall) {
                        // ---- block: the lowest <= MB violators of tile column kappa -------------------------------------
                        unsigned rest = pmall;
#pragma unroll
                        for (int i = 0; i < MB; ++i) rest &= rest - 1;
                        const unsigned pm = pmall & ~rest;
                        pmall = rest;
                        const int m = __builtin_popcount(pm);
                        double *P = Pbase + (bc & 1) * MB * CW;
                        ++bc;
                        // ---- 1. gather the pivot columns (compacted) into the LDS panel ------------------------------------
#define PARTLS_G(i) if constexpr (i < T) { if (__builtin_expect(kappa == i, 0)) gather_tile<T, H, i, CW>(S, P, a, b, pm); }
                        PARTLS_CASES(PARTLS_G)                 // flat chain of independent ifs: the only form the register allocator keeps spill-free
#undef PARTLS_G
                        if (tid < 16 * T && (tid >> 4) == kappa && ((pm >> (tid & 15)) & 1u))
                            P[__builtin_popcount(pm & ((1u << (tid & 15)) - 1u)) * CW + RHSPOS] = q;
                        // is this thread's panel row a pivot row?  row position t <-> variable 16*rowrho + rowc
                        int myj = -1;
                        bool my_basic = false;
                        if (rowrho == kappa && rowc < 16 && ((pm >> rowc) & 1u)) {
                            myj = __builtin_popcount(pm & ((1u << rowc) - 1u));
                            my_basic = (basm >> rowc) & 1u;
                        }
                        STAMP(8);
                        __syncthreads();
                        STAMP(1);
                        // ---- 2. panel elimination -----------------------------------------------------------------------
                        bool blk_ok = false;
                        switch (m) {
                            case 1: blk_ok = panel_block<1, CW>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid, idle_wave); break;
                            case 2: blk_ok = panel_block<2, CW>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid, idle_wave); break;
                            case 3: blk_ok = panel_block<3, CW>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid, idle_wave); break;
                            case 4: blk_ok = panel_block<4, CW>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid, idle_wave); break;
                            case 5: blk_ok = panel_block<5, CW>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid, idle_wave); break;
                            case 6: blk_ok = panel_block<6, CW>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid, idle_wave); break;
                            case 7: blk_ok = panel_block<7, CW>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid, idle_wave); break;
                            default: blk_ok = panel_block<8, CW>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid, idle_wave); break;
                        }
                        STAMP(11);
                        __syncthreads();
                        STAMP(2);
                        // ---- 3. fused rank-m update of the register tableau ------------------------------------------------
#pragma unroll PARTLS_UPD_UNROLL
                        for (int s = 0; s < m; ++s) {
                            const double inv = Dinv[s];
                            const double *Zs = Z + s * CW;
                            double x[L::XN];
#pragma unroll
                            for (int rho = 0; rho < L::XN; ++rho) x[rho] = Zs[a * RS + rho];
#pragma unroll
                            for (int gam = L::GLO; gam < L::GHI; ++gam) {
                                const double yg = -Zs[b * RS + gam] * inv;
#pragma unroll
                                for (int rho = 0; rho <= gam; ++rho)
                                    S[L::idx(rho, gam)] = fma(x[rho], yg, S[L::idx(rho, gam)]);
                            }
                            const double zr = Zs[RHSPOS], zri = zr * inv;
                            if (tid < 16 * T) q = fma(-Zs[mypos], zri, q);
                            corner = fma(-zr, zri, corner);
                        }
                        STAMP(3);
                        // ---- 4. rows / columns of the pivoted variables come from the final panel ----------------------------
#define PARTLS_F(i) if constexpr (i < T) { if (__builtin_expect(kappa == i, 0)) scatter_tile<T, H, i, CW>(S, P, a, b, pm); }
                        PARTLS_CASES(PARTLS_F)
#undef PARTLS_F
                        if (tid < 16 * T && (tid >> 4) == kappa && ((pm >> (tid & 15)) & 1u)) {
                            const int j = __builtin_popcount(pm & ((1u << (tid & 15)) - 1u));
                            q = P[j * CW + RHSPOS];
                            if (Dinv[j] != 0.0) basic = !basic;
                            else blocked = true;
                        }
                        progress = progress || blk_ok;
                        npiv += (unsigned)m;
                        STAMP(4);
                    }
                    tiles &= tiles - 1;
                }
            }
            // patterns are ranked on objective^2 (the tableau corner; sqrt is monotone): one sqrt per workgroup instead of one per
            // pattern, unless every pattern's objective is wanted
            const double obj2 = corner > 0.0 ? corner : 0.0;
            if (p.all_opt && tid == 0) p.all_opt[pat] = sqrt(obj2);
            if (obj2 < best_obj || (obj2 == best_obj && (long long)pat < best_pat)) { best_obj = obj2; best_pat = (long long)pat; }
        }
        if (p.node_sol) {
            if (has_var) p.node_sol[(size_t)chain * p.node_ld + tid] = basic ? q : 0.0;
            if (tid == 0) p.node_obj2[chain] = corner;
        }
    }
    STAMP_FLUSH;
    if (tid == 0) {
        p.best_obj[blockIdx.x] = sqrt(best_obj);
        p.best_pat[blockIdx.x] = best_pat;
        if (p.n_pivots && npiv) atomicAdd(p.n_pivots, npiv);
        if (p.n_unconverged && nunconv) atomicAdd(p.n_unconverged, nunconv);
    }
}

template <int T>
__global__ __launch_bounds__(THREADS, 2) void sweep_blk_kernel(SweepParams p)
{
    extern __shared__ double lds[];
    const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    if (half == 0) sweep_body<T, 0>(p, lds);
    else sweep_body<T, 1>(p, lds);
}

}  // namespace blk

template <int T>
static hipError_t launch_blk_T(const SweepParams &p, int grid, hipStream_t s)
{
    const size_t shmem = (size_t)blk::lds_doubles(T) * sizeof(double) + 32 * sizeof(unsigned long long);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&blk::sweep_blk_kernel<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(blk::sweep_blk_kernel<T>, dim3(grid), dim3(blk::THREADS), shmem, s, p);
    return hipGetLastError();
}

hipError_t launch_sweep_blk(const SweepParams &p, int T, int grid, hipStream_t s)
{
    switch (T) {
#define PARTLS_L(i) case i + 1: return launch_blk_T<i + 1>(p, grid, s);
        PARTLS_CASES(PARTLS_L)
#undef PARTLS_L
        default: return hipErrorInvalidValue;
    }
}

}  // namespace partls
