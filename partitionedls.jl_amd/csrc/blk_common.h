// blk_common.h — device templates shared by the blocked sweep kernels (sweep_blk.hip, sweep_two.hip): the split register
// tableau, the tile-column gather/scatter and the exact-M panel elimination.  See sweep_blk.hip for the algorithm.
#pragma once
#include "common.h"
#include <type_traits>

namespace partls {
namespace blk {

static constexpr int THREADS = 512;
static constexpr int MAXT = 17;                 // n <= 272
#ifndef PARTLS_UPD_UNROLL
#define PARTLS_UPD_UNROLL 1
#endif
static constexpr int MB = 8;                    // pivots per block (a tile column with more violators takes two blocks)

constexpr int nslots(int T) { return T * (T + 1) / 2; }
constexpr int tri(int g) { return g * (g + 1) / 2; }
constexpr int split(int T)
{
    int best = 1, bestmax = 1 << 30;
    for (int g = 1; g < T; ++g) {
        int a = tri(g), b = nslots(T) - tri(g);
        int m = a > b ? a : b;
        if (m < bestmax) { bestmax = m; best = g; }
    }
    return T == 1 ? 1 : best;
}
constexpr int rstride(int T) { return (T % 2) ? T : T + 1; }      // odd row stride: 16 lanes x 8 B hit 32 distinct banks
constexpr int colw(int T) { return 513; }                          // panel column: one slot per thread (16*RS rows + rhs + dummies), odd
// LDS doubles: P[2][MB][COLW], Z[MB][COLW], U[2][MB+64], Dinv[MB+64] (64 per-lane dummy slots each); then 32 x u64 masks
constexpr int lds_doubles(int T) { return 3 * MB * colw(T) + 3 * (MB + 64); }

__device__ __forceinline__ double fast_rcp(double d)
{
    double y = __builtin_amdgcn_rcp(d);
    y = fma(fma(-d, y, 1.0), y, y);
    y = fma(fma(-d, y, 1.0), y, y);
    return y;
}
__device__ __forceinline__ int sign_of_var(uint64_t m, uint64_t pat) { return 2 * __popcll(m & pat) - __popcll(m); }

// Diagnostic build only (-DPARTLS_STAMPS): per-phase cycle shares of workgroup 0 / thread 0, written to p.scratch[0..7]
// (a buffer no other code of the kernel reads).  Never quote this build's run time (cdna_hip_programming.md §7).
#ifdef PARTLS_STAMPS
#define STAMP_DECL unsigned long long st_acc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t = __builtin_amdgcn_s_memtime();
#define STAMP(ph) do { __builtin_amdgcn_sched_barrier(0); unsigned long long _n = __builtin_amdgcn_s_memtime(); \
                       st_acc[ph] += _n - st_t; st_t = _n; __builtin_amdgcn_sched_barrier(0); } while (0)
#ifndef PARTLS_STAMP_TID
#define PARTLS_STAMP_TID 0
#endif
#define STAMP_FLUSH do { if (tid == PARTLS_STAMP_TID && blockIdx.x == 0 && p.scratch) for (int _i = 0; _i < 24; ++_i) p.scratch[_i] = (double)st_acc[_i]; } while (0)
#else
#define STAMP_DECL
#define STAMP(ph) do { } while (0)
#define STAMP_FLUSH do { } while (0)
#endif

template <int T, int H, int CWP = colw(T)>
struct Half {
    static constexpr int G = split(T);
    static constexpr int GLO = H ? G : 0;
    static constexpr int GHI = H ? T : G;
    static constexpr int OFF = H ? tri(G) : 0;
    static constexpr int CNT = (H ? nslots(T) - tri(G) : tri(G)) > 0 ? (H ? nslots(T) - tri(G) : tri(G)) : 1;
    static constexpr int XN = GHI;
    static constexpr int RS = rstride(T);
    static constexpr int CW = CWP;
    __device__ static constexpr int idx(int rho, int gam) { return tri(gam) + rho - OFF; }
};

// ---- tile-column gather ------------------------------------------------------------------------------------------------
// Element (16 rho + a, 16 KAPPA + b) belongs to column b of the tile, row position a*RS + rho.  Only the pivot columns are
// gathered, COMPACTED: the j-th pivot of the block (ascending local index) becomes panel column j = popc(pm below it).
template <int T, int H, int KAPPA, int CW, class SA>
__device__ __forceinline__ void gather_tile(const SA &S, double *P, int a, int b, unsigned pm)
{
    using L = Half<T, H, CW>;
    if constexpr (KAPPA >= L::GLO && KAPPA < L::GHI) {
        if ((pm >> b) & 1u) {
            double *col = P + __builtin_popcount(pm & ((1u << b) - 1u)) * L::CW + a * L::RS;
#pragma unroll
            for (int rho = 0; rho <= KAPPA; ++rho) col[rho] = S[L::idx(rho, KAPPA)];
        }
    }
    // row KAPPA of the stored triangle = column (16 KAPPA + a) by symmetry, row position b*RS + gamma
    if ((pm >> a) & 1u) {
        double *col = P + __builtin_popcount(pm & ((1u << a) - 1u)) * L::CW + b * L::RS;
#pragma unroll
        for (int gam = (KAPPA + 1 > L::GLO ? KAPPA + 1 : L::GLO); gam < L::GHI; ++gam) col[gam] = S[L::idx(KAPPA, gam)];
    }
}

template <int T, int H, int KAPPA, int CW, class SA>
__device__ __forceinline__ void scatter_tile(SA &S, const double *P, int a, int b, unsigned pm)
{
    using L = Half<T, H, CW>;
    if constexpr (KAPPA >= L::GLO && KAPPA < L::GHI) {
        if ((pm >> b) & 1u) {
            const double *col = P + __builtin_popcount(pm & ((1u << b) - 1u)) * L::CW + a * L::RS;
#pragma unroll
            for (int rho = 0; rho <= KAPPA; ++rho) S[L::idx(rho, KAPPA)] = col[rho];
        }
    }
    if ((pm >> a) & 1u) {
        const double *col = P + __builtin_popcount(pm & ((1u << a) - 1u)) * L::CW + b * L::RS;
#pragma unroll
        for (int gam = (KAPPA > L::GLO ? KAPPA : L::GLO); gam < L::GHI; ++gam) S[L::idx(KAPPA, gam)] = col[gam];
    }
}

// ---- panel elimination for a block of exactly M pivots (M = 1..MB); straight-line code, no guards ------------------------
// Thread t owns row position `prow` (= t for the real positions) of the M (compacted) pivot columns in registers pv[0..M);
// threads beyond the rhs row work on dummy positions: computed, stored, never read.  Step s: the pivot-row threads publish their entry of column s through U, the
// thread that IS pivot row s also publishes 1/d (0 for a dependent column, Lawson–Hanson's rejection) — both as
// unconditional stores, non-owners hit a dummy slot — one barrier, one batch of broadcast reads, then every thread
// updates its own row.
template <int M, int CW>
__device__ __forceinline__ bool panel_block(double *P, double *Z, double *U, double *Dinv, int myj, bool my_basic,
                                            double piv_eps, int tid, int prow, bool idle_wave)
{
    bool any_ok = false;                                    // uniform: was any pivot of the block carried out?
    // A wave whose 64 row positions all lie beyond the rhs row owns no panel row.  Two waves share a SIMD's issue slots, so
    // such a wave only keeps the barrier count and the uniform result instead of competing with a real wave.
    if (idle_wave) {
#pragma unroll
        for (int s = 0; s < M; ++s) {
            __syncthreads();
            any_ok = any_ok || (Dinv[s] != 0.0);
        }
        return any_ok;
    }
    constexpr int US = MB + 64;                             // U row stride; slots MB.. are per-lane dummies (no same-address stores)
    const int dummy = MB + (tid & 63);
    double pv[M];
#pragma unroll
    for (int j = 0; j < M; ++j) pv[j] = P[j * CW + prow];
    const int uslot = myj >= 0 ? myj : dummy;
#pragma unroll
    for (int s = 0; s < M; ++s) {
        Z[s * CW + prow] = pv[s];
        U[(s & 1) * US + uslot] = pv[s];
        {
            const double d = pv[s];
            const double r = (my_basic || d > piv_eps) ? fast_rcp(d) : 0.0;
            Dinv[myj == s ? s : dummy] = r;
        }
        __syncthreads();
        const double inv = Dinv[s], ainv = fabs(inv);
        double u[M];
#pragma unroll
        for (int j = 0; j < M; ++j) u[j] = U[(s & 1) * US + j];
        const bool ok = inv != 0.0, isrow = (myj == s);
        any_ok = any_ok || ok;
        const double fz = pv[s] * inv;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            if (j == s) continue;
            const double upd = isrow ? u[j] * ainv : fma(-fz, u[j], pv[j]);
            pv[j] = ok ? upd : pv[j];
        }
        pv[s] = ok ? (isrow ? -inv : pv[s] * ainv) : pv[s];
    }
#pragma unroll
    for (int j = 0; j < M; ++j) P[j * CW + prow] = pv[j];
    return any_ok;
}

// wave-uniform dispatch on the tile index as a binary decision tree: ~log2(T) scalar branches, two-input joins only
template <int LO, int HI, class F>
__device__ __forceinline__ void tile_dispatch(int kappa, F &&f)
{
    if constexpr (HI - LO == 1) {
        f(std::integral_constant<int, LO>{});
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (kappa < MID) tile_dispatch<LO, MID>(kappa, f);
        else tile_dispatch<MID, HI>(kappa, f);
    }
}

#define PARTLS_CASES_LO(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define PARTLS_CASES_HI(M) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16)
#define PARTLS_CASES(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16)

}  // namespace blk
}  // namespace partls
