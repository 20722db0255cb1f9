// sweep_blk.hip — production sign-pattern sweep, BLOCKED principal pivots on a register-resident tableau (n <= 320).
//
// Replaces the loop body of fit(Opt), Opt.jl:87-90 (indextobeta + bmatrix + nonneg_lsq + objective), for every pattern of a
// Gray-code chain, and serves the node solves of Alt (Alt.jl:80-90) and BnB (BnB.jl:69-92); same mathematics and the same
// per-pattern decisions as sweep_generic.hip.
//
// Data layout (one workgroup = 512 threads = 8 waves = 2 waves/SIMD on one CU, one tableau, one chain):
//   * variables are padded to 16*T; the symmetric tableau is cut into 16x16 tiles (rho, gamma), rho <= gamma only;
//   * thread (a, b) = (t & 15, (t >> 4) & 15) owns element (16 rho + a, 16 gamma + b) of EVERY stored tile — a tile-cyclic
//     layout, so a pivot is an outer-product update in registers with static register indices only;
//   * v_fma_f64 can address only the 256 architectural VGPRs, so the T(T+1)/2 tile slots (153 at n = 257) are split
//     between the two halves of the workgroup by tile COLUMN (threads 0..255: gamma < G, threads 256..511: gamma >= G);
//   * rhs column q: thread v owns q_v (v < n); the objective (corner) is replicated; KKT scan = one ballot per wave.
// The violators of one KKT scan are exchanged one TILE COLUMN at a time as a block pivot of m <= 8 variables
//     T'_RR = T_RR - sum_s z_s z_s' / d_s          (z_s = pivot column as of its own step, d_s its pivot):
//   1. gather   : the pivot columns go from registers to an LDS panel P[m cols][rows] (+ rhs row), compacted;
//   2. panel    : thread t owns ROW t of the panel in registers (<= 8 pivot columns); m sequential Gauss–Jordan steps, each
//                 exchanging only the m pivot-row entries of the CURRENT pivot column and 1/d through LDS (one barrier per
//                 step); z_s and 1/d_s are recorded for the update;
//   3. update   : all 512 threads: S(rho,gamma) -= x_s[rho] * y_s[gamma] for s = 1..m (x resident, y streamed from LDS);
//   4. scatter  : rows/columns of the pivoted variables are overwritten from the final panel (one if-chain per tile).
//
// Dependent columns.  An entering variable k is refused (Lawson–Hanson's rejection; re-examined after any exchange) when the
// basis B + k would be numerically dependent at the 1e-11 level on the unit-diagonal scale: with c_j = T[j][k] the regression
// coefficients of column k on the basic columns, the leave-one-out pivot of EVERY member of B + k must stay above piv_eps —
//     d_k > piv_eps                      (k itself:  relative residual norm of x_k after regression on B)
//     d_k > piv_eps * c_j^2  (j in B)    (1 / d_j(B + k) = 1 / d_j(B) + c_j^2 / d_k)
// The second line is what a Gram-based method needs on top of the textbook rule: the computed d_k carries an error of
// eps * (1 + |c|)^2, so with large coefficients (an exactly dependent column next to a nearly collinear pair) a true zero
// comes out at 1e-11..1e-10 and the fixed threshold alone accepts it (round-1 fuzz blocks 9 and 24).  The panel is computed
// optimistically; a thread that sees the second condition violated in its own row reports the step, the block is abandoned
// before the register tableau is touched, the offender is marked rejected and the rest of the block is redone.
// The test needs no basis bookkeeping: for a NONBASIC row j the Schur complement is positive semidefinite, T_jk^2 <= T_jj T_kk
// <= d_k, so d_k > piv_eps * T_jk^2 holds with a margin of 1e11 and every row can simply test its own entry; a leaving pivot
// has 1/d < 0 and a rejected one 1/d = 0, which fail the test by sign.  Only the rhs row (no variable) must not report.
#include "common.h"
#include <atomic>
#include <type_traits>

namespace partls {
namespace blk {

static constexpr int THREADS = 512;
static constexpr int MAXT = 20;                 // n <= 320 compiled (T = 21: 1500 spilled VGPRs); the library switches to sweep_lazy.hip beyond T = 18 (ctx.h: reg_maxt)
#ifndef PARTLS_UPD_UNROLL
#define PARTLS_UPD_UNROLL 1
#endif
static constexpr int MB = 8;                    // pivots per block (a tile column with more violators takes two blocks): the 256-thread kernel
#ifndef PARTLS_EXPORT_BIG
#define PARTLS_EXPORT_BIG 1      // the 512-thread kernel leaves the solution of every workgroup's best pattern behind (round 4: C3 finish 0.54 -> 0.16 ms, sweep 49.1 -> 48.6)
#endif
#ifndef PARTLS_MB_BIG
#define PARTLS_MB_BIG 8
#endif
static constexpr int EXPORT_MAXT = 17;          // ... up to this tile count (beyond it the two live registers of the export start spilling: 22 -> 32 spilled VGPRs at T = 18)
static constexpr int MBB = PARTLS_MB_BIG;       // ... of the 512-thread kernel (8..16; experiments: tools/experiments/README.md)
static constexpr int NO_VETO = 99;

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
// Small tableaus (T <= MAXT_S tile columns, n <= 160): ONE group of 256 threads owns every tile slot (<= 55 doubles per thread), so a
// workgroup is 4 waves and a CU holds three of them — three independent chains whose latency chains (panel steps, scans, barriers)
// interleave on the same SIMDs instead of one chain leaving them idle (BASELINE config 2: n = 128).
static constexpr int MAXT_S = 10;
static constexpr int CW_S = 257;                                   // one panel slot per thread of the 256-thread workgroup, odd
__device__ __forceinline__ double fast_rcp(double d)
{
    double y = __builtin_amdgcn_rcp(d);
    y = fma(fma(-d, y, 1.0), y, y);
    y = fma(fma(-d, y, 1.0), y, y);
    return y;
}
template <class PT> __device__ __forceinline__ PT *uniform_ptr(PT *p)     // a wave-uniform pointer the compiler cannot prove uniform -> SGPR pair
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (PT *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int sign_of_var(uint64_t m, uint64_t pat) { return 2 * __popcll(m & pat) - __popcll(m); }
__device__ __forceinline__ double readlane_f64(double v, int lane)      // lane: wave-uniform
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// Diagnostic build only (-DPARTLS_STAMPS): per-phase cycle shares of workgroup 0 / thread 0, written to p.scratch[0..23]
// (a buffer no other code of the kernel reads).  Never quote this build's run time (cdna_hip_programming.md §7).
#ifdef PARTLS_STAMPS
#define STAMP_DECL unsigned long long st_acc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t = __builtin_amdgcn_s_memtime();
#define STAMP(ph) do { __builtin_amdgcn_sched_barrier(0); unsigned long long _n = __builtin_amdgcn_s_memtime(); \
                       st_acc[ph] += _n - st_t; st_t = _n; __builtin_amdgcn_sched_barrier(0); } while (0)
#ifndef PARTLS_STAMP_TID
#define PARTLS_STAMP_TID 0
#endif
#define STAMP_FLUSH do { if (tid == PARTLS_STAMP_TID && blockIdx.x == 0 && p.scratch) { for (int _i = 0; _i < 24; ++_i) p.scratch[_i] = (double)st_acc[_i]; \
                                p.scratch[24] = (double)bc; p.scratch[25] = (double)sc; p.scratch[26] = (double)npiv; } } while (0)   /* + blocks, scans, pivots of workgroup 0 */
#else
#define STAMP_DECL
#define STAMP(ph) do { } while (0)
#define STAMP_FLUSH do { } while (0)
#endif

// LDS image, statically allocated at namespace scope (one copy per kernel, shared by the two half instantiations): every
// address is a link-time constant that folds into the ds_* immediates — with a dynamic base the compiler kept ~20 slot
// addresses in SGPRs and spilled them to VGPR lanes inside the panel steps.  ~100 KB of the CU's 160 KB.
static constexpr int CWMAX = colw(MAXT);
template <int CWX, int MAXTX>
struct LdsImageT {
    double Z[MB * CWX];
    double U[2 * (MB + 64)];
    double Dinv[MB + 64];
    unsigned long long s_inf[16];
    unsigned long long s_bas[16];
    unsigned s_sum[16];
    int s_veto[2];
    double s_best[4];
    unsigned char s_rbit[64];
    unsigned long long s_vmask[16 * MAXTX];
    double Pbase[2 * MB * CWX];
};
// One struct, so that the order is ours: everything the update and the panel steps address with per-thread offsets (Z, U, Dinv,
// the masks) sits in the first 64 KB, where base + element offset fits the 16-bit immediate of the ds_* instructions; with Z
// behind the two panel buffers every operand read of the update loop needed its own v_add for the address.
struct LdsImage {
    double Z[MBB * CWMAX];                     // [MB][CW] pivot column s as of its own step
    double U[2 * (MBB + 64)];                  // [2][MB+64] pivot-row entries of the current column (+ per-lane dummies)
    double Dinv[MBB + 64];                     // 1/d_s
    unsigned long long s_inf[16];              // [2][8] violator mask of the scan
    unsigned long long s_bas[16];              // [2][8] basis mask of the scan
    unsigned s_sum[16];                        // [2][8] per-wave summary of s_inf: count | tile bits << 8
    int s_veto[2];                             // first vetoed step of a block (by block parity)
    // per-thread state that is touched once per pattern lives in LDS, not in VGPRs (the register file holds the tableau):
    double s_best[4];                          // running minimum: obj^2, pattern (as bits); runner-up: obj^2, pattern
    unsigned char s_rbit[64];                  // reference bit of internal pattern bit b (exact ties only)
    unsigned long long s_vmask[16 * MAXT];     // group mask of variable v
    double Pbase[2 * MBB * CWMAX];             // [2][MB][CW] panel, double buffered by block parity
};
__shared__ LdsImage lds_image;
__shared__ LdsImageT<CW_S, MAXT_S> lds_small;              // the 256-thread kernel's image (~51 KB: three workgroups per CU)

template <int W>
__device__ __forceinline__ auto *image_of()
{
    if constexpr (W == 1) return &lds_small; else return &lds_image;
}

// W = 2: the tile slots are split between the two halves H = 0 / 1 of a 512-thread workgroup; W = 1: one group of 256 threads owns all
template <int T, int H, int W = 2>
struct Half {
    static constexpr int G = W == 1 ? T : split(T);
    static constexpr int GLO = H ? G : 0;
    static constexpr int GHI = H ? T : G;
    static constexpr int OFF = H ? tri(G) : 0;
    static constexpr int CNT = (H ? nslots(T) - tri(G) : tri(G)) > 0 ? (H ? nslots(T) - tri(G) : tri(G)) : 1;
    static constexpr int XN = GHI;
    static constexpr int RS = rstride(T);
    static constexpr int CW = W == 1 ? CW_S : colw(T);
    static constexpr int NT = W == 1 ? 256 : THREADS;              // threads of the workgroup
    __device__ static constexpr int idx(int rho, int gam) { return tri(gam) + rho - OFF; }
};

// ---- tile-column gather ------------------------------------------------------------------------------------------------
// Element (16 rho + a, 16 KAPPA + b) belongs to column b of the tile, row position a*RS + rho.  Only the pivot columns are
// gathered, COMPACTED: the j-th pivot of the block (ascending local index) becomes panel column j = popc(pm below it).
template <int T, int H, int W, int KAPPA, class SA>
__device__ __forceinline__ void gather_tile(const SA &S, double *P, int a, int b, unsigned pm)
{
    using L = Half<T, H, W>;
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

template <int T, int H, int W, int KAPPA, class SA>
__device__ __forceinline__ void scatter_tile(SA &S, const double *P, int a, int b, unsigned pm)
{
    using L = Half<T, H, W>;
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
// threads beyond the rhs row work on dummy positions: computed, stored, never read.  Step s: the pivot-row threads publish
// their entry of column s through U, the thread that IS pivot row s also publishes 1/d (0 for a dependent column, Lawson–
// Hanson's rejection) — both as unconditional stores, non-owners hit a dummy slot — one barrier, one batch of broadcast
// reads, then every thread updates its own row with  T_ij -= T_is T_sj / d,  T_is = T_is / |d|,  T_ss = -1/d.
// NUMERICS: the factor T_sj is taken from the pivot COLUMN s (its entry at pivot row j, u[j]) — never from column j at pivot
// row s, which is the same number only in exact arithmetic.  With u from column s the panel receives exactly the symmetric
// rank-1 term z_s z_s' / d_s that the register tableau receives in the update; mixing the two breaks that consistency and
// loses the solution on ill-conditioned data (pivots ~1e-8: measured, tools/tableau_emul.py --block).
// `my_basic`: this thread is a pivot row and its variable is basic (leaves).
// Returns the first step whose entering pivot this thread's row vetoes under the leave-one-out rule (file header), M if none.
template <int M, int CW, int MBK>
__device__ __forceinline__ int panel_block(double *P, double *Z, double *U, double *Dinv, int myj, bool my_basic,
                                           double piv_eps, int tid, int prow, bool idle_wave)
{
    // A wave whose 64 row positions all lie beyond the rhs row owns no panel row.  Two waves share a SIMD's issue slots, so
    // such a wave only keeps the barrier count instead of competing with a real wave.
    if (idle_wave) {
#pragma unroll
        for (int s = 0; s < M; ++s) __syncthreads();
        return M;
    }
    constexpr int US = MBK + 64;                            // U row stride; slots MB.. are per-lane dummies (no same-address stores)
    const int dummy = MBK + (tid & 63);
    double pv[M];
#pragma unroll
    for (int j = 0; j < M; ++j) pv[j] = P[j * CW + prow];
    const int uslot = myj >= 0 ? myj : dummy;
    int veto = M;
#pragma unroll
    for (int s = 0; s < M; ++s) {
        Z[s * CW + prow] = pv[s];
        U[(s & 1) * US + uslot] = pv[s];
        if (myj == s) {                                     // the one pivot-row thread (its wave only: the others branch over)
            const double d = pv[s];
            Dinv[s] = (my_basic || d > piv_eps) ? fast_rcp(d) : 0.0;
        }
        __syncthreads();
        double inv = Dinv[s];
        double u[M];
#pragma unroll
        for (int j = 0; j < M; ++j) u[j] = U[(s & 1) * US + j];
        // all reads of the step are issued together, right after the barrier: left alone, the compiler sinks the u[j] loads into the
        // "pivot accepted" branch, i.e. behind a second LDS round trip (wait for 1/d, branch, then load and wait again)
        asm volatile("" : "+v"(inv));
#pragma unroll
        for (int j = 0; j < M; ++j) asm volatile("" : "+v"(u[j]));
        // leave-one-out veto: T_js^2 >= d / piv_eps  (inv = 1/d; negative for a leaving pivot, 0 for a rejected one)
        if ((pv[s] * pv[s]) * (inv * piv_eps) >= 1.0) veto = veto < s ? veto : s;
        // inv is the same in every lane: a scalar branch skips a rejected pivot (1/d = 0: exponent field 0)
        if (__builtin_amdgcn_readfirstlane(__double2hiint(inv)) & 0x7ff00000) {
            const double ainv = fabs(inv), fz = -pv[s] * inv;
            if (myj == s) {                                 // pivot row: T_sj / |d| from the pivot COLUMN's entries, T_ss = -1/d
#pragma unroll
                for (int j = 0; j < M; ++j) pv[j] = (j == s) ? -inv : u[j] * ainv;
            } else {
#pragma unroll
                for (int j = 0; j < M; ++j) pv[j] = (j == s) ? pv[s] * ainv : fma(fz, u[j], pv[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < M; ++j) P[j * CW + prow] = pv[j];
    return veto;
}

#define PARTLS_CASES(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19)

// NODE: node mode (Alt alpha-steps, BnB bounds: one subproblem per chain, free / zero groups) is a separate instantiation, so
// that the chain-mode sweep carries neither the node pointers nor the per-thread `free` flag through its loops.
template <int T, int H, bool NODE, int W = 2>
__device__ __forceinline__ void sweep_body(const SweepParams &p)
{
    using L = Half<T, H, W>;
    constexpr int RS = L::RS, CW = L::CW, RHSPOS = 16 * RS, NW = (T + 3) / 4;      // NW: 64-bit mask words that can be non-empty
    constexpr int THREADS = L::NT;                                                  // shadows the namespace constant
    constexpr int MB = W == 1 ? blk::MB : MBB;                                      // pivots per block of this kernel (shadows the namespace constant)
    static_assert(W == 2 || RHSPOS < 256, "the 256-thread kernel needs every panel row position (and the rhs row) below 256");
    auto &lds_image = *image_of<W>();
    const int tid = threadIdx.x, t8 = tid & 255, a = t8 & 15, b = t8 >> 4, lane = tid & 63, wave = tid >> 6;
    const int n = p.n;
    const int nwords = (n + 63) >> 6;

    // link-time constant addresses into the LDS image
    double *const Pbase = lds_image.Pbase, *const Z = lds_image.Z, *const U = lds_image.U, *const Dinv = lds_image.Dinv;
    unsigned long long *const s_inf = lds_image.s_inf, *const s_bas = lds_image.s_bas, *const s_vmask = lds_image.s_vmask;
    unsigned *const s_sum = lds_image.s_sum;
    int *const s_veto = lds_image.s_veto;
    double *const s_best = lds_image.s_best;
    for (int i = tid; i < 2 * MB * CW; i += THREADS) Pbase[i] = 0.0;                   // padding rows are never gathered
    for (int i = tid; i < MB * CW; i += THREADS) Z[i] = 0.0;
    if (tid < 2 * (MB + 64)) U[tid] = 0.0;
    if (tid < MB + 64) Dinv[tid] = 0.0;
    if (tid < 16) { s_inf[tid] = 0; s_bas[tid] = 0; s_sum[tid] = 0; }
    if (tid < 2) s_veto[tid] = NO_VETO;
    if (tid == 0) {
        s_best[0] = __builtin_inf(); reinterpret_cast<long long *>(s_best)[1] = -1;
        s_best[2] = __builtin_inf(); reinterpret_cast<long long *>(s_best)[3] = -1;
    }
    if (tid < 16 * T) s_vmask[tid] = tid < p.n ? p.mask[tid] : 0ULL;
    if (tid < 64) lds_image.s_rbit[tid] = tid < 40 ? p.rbit.gbit[tid] : 0;
    __syncthreads();

    double S[L::CNT];
    double q = 0.0, corner = 0.0;
    double mybest = __builtin_inf();                      // chain mode: objective^2 of the pattern whose solution sits in p.best_sol
    const bool has_var = tid < n;
    bool basic = false, blocked = false;
    const int mypos = (tid & 15) * RS + (tid >> 4);       // panel row position of variable `tid` (tid < 16 T)
    // panel phase: this thread owns row position `tid`, i.e. variable 16*rowrho + rowc (valid when rowc < 16)
    const int rowc = tid / RS, rowrho = tid - rowc * RS;
    const bool idle_wave = __builtin_amdgcn_readfirstlane((tid & ~63) > RHSPOS ? 1 : 0) != 0;   // no panel row in this wave

    unsigned npiv = 0, nunconv = 0, nveto = 0;            // per workgroup: far below 2^32
    unsigned bc = 0, sc = 0;                              // block / scan counters (double-buffer parity)

    // 32-bit loop state (the host guarantees nchains < 2^31 and chain_len < 2^31): every 64-bit scalar held across the pattern loop is
    // an SGPR pair the allocator spills to a VGPR lane and reloads
    const int clen = (int)p.chain_len;
    const int nchains = (int)((p.g_end - p.g_begin + clen - 1) / clen);

    STAMP_DECL
    for (int chain = blockIdx.x; chain < nchains; chain += gridDim.x) {
        const int64_t g0 = p.g_begin + (int64_t)chain * clen;
        const int glen = (int)((g0 + clen < p.g_end) ? clen : p.g_end - g0);
        STAMP(5);
        const double *src = p.T0;
        if constexpr (NODE) {                                         // warm start from a snapshot (BnB children), wave-uniform
            if (p.node_src) {
                const double *snap = p.node_src[chain];
                if (snap) src = snap;
            }
        }
        // One wave-uniform base per PAIR of slots, opaque to the optimiser, + the thread's constant element offset + an immediate: left
        // alone, LLVM hoists the CNT per-slot address offsets out of the chain loop as CNT live VGPRs, spills them, and every load / store of
        // a chain start or snapshot then waits for its own scratch reload (`scratch_load; s_waitcnt vmcnt(0); global_store` 76 times: 35 us
        // per snapshot, 85 % of a warm-started BnB node — found in the round-4 kernel trace of the node batches)
        {
            const double *bp = uniform_ptr(src) + (size_t)L::OFF * 256;
#pragma unroll
            for (int s = 0; s < L::CNT; ++s) {
                if ((s & 1) == 0) asm volatile("" : "+s"(bp));
                S[s] = bp[(s & 1) * 256 + t8];
                if (s & 1) bp += 512;
            }
        }
        q = (tid < 16 * T) ? src[(size_t)nslots(T) * 256 + tid] : 0.0;
        corner = src[(size_t)nslots(T) * 256 + 16 * T];
        basic = false;
        if constexpr (NODE) {
            if (src != p.T0 && tid < 16 * T) basic = reinterpret_cast<const int8_t *>(src + (nslots(T) * 256 + 16 * T + 8))[tid] != 0;
        }
        STAMP(6);

        for (int gi = 0; gi < glen; ++gi) {
            const uint64_t g = (uint64_t)g0 + (unsigned)gi;
            uint64_t pat = g ^ (g >> 1);
            const uint64_t vmask = s_vmask[tid < 16 * T ? tid : 0] & (has_var ? ~0ULL : 0ULL);
            bool isfree = false;
            int f;
            if constexpr (NODE) {                                   // node mode: chain index = node index, per-variable codes
                pat = (uint64_t)chain;
                const int code = has_var ? (int)p.node_code[((size_t)chain * clen + gi) * p.node_ld + tid] : 0;   // a chain of nodes: warm start
                isfree = code == 2;
                f = isfree ? 0 : code;
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
                    const double fq = q * (double)(f > 1 ? 1 : (f < -1 ? -1 : f));     // sign(f) * q  (|f| = 2: feature of two groups)
                    if (NODE && isfree) bad = !basic && !blocked && (fabs(q) > p.tol);     // free: stationarity only
                    else if (basic) bad = (f == 0) || (fq < -p.tol);
                    else bad = (fq > p.tol) && !blocked;
                }
                const unsigned long long bb = __ballot(bad), bs = __ballot(basic);
                // every wave condenses ITS mask word before the barrier (count in bits 0..7, one bit per 16-variable tile column in
                // bits 8..11), so that after it each wave combines NW small words instead of re-deriving everything from NW masks
                unsigned summ = (unsigned)__popcll(bb);
#pragma unroll
                for (int sub = 0; sub < 4; ++sub)
                    if ((bb >> (16 * sub)) & 0xFFFFull) summ |= 0x100u << sub;
                if (lane == 0 && wave < nwords) { s_inf[par * 8 + wave] = bb; s_bas[par * 8 + wave] = bs; s_sum[par * 8 + wave] = summ; }
                STAMP(9);
                __syncthreads();
                STAMP(10);
                int count = 0;
                unsigned tiles = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const unsigned sw = (unsigned)__builtin_amdgcn_readfirstlane((int)s_sum[par * 8 + w]);    // 0 for w >= nwords
                    count += (int)(sw & 0xFFu);
                    tiles |= (sw >> 8) << (4 * w);
                }
                STAMP(0);
                if (count == 0) break;
                bool all;
                if (count < ninf_best) { ninf_best = count; patience = 3; all = true; }
                else if (patience > 0) { --patience; all = true; }
                else all = false;                                     // backup rule: only the largest violator
                if (++rounds > p.max_rounds) { ++nunconv; break; }
                int single_k = -1;
                if (!all) {                                           // rare: the largest violator, from the mask words themselves
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        unsigned long long ww = s_inf[par * 8 + w];
                        ww = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ww >> 32)) << 32) |
                             (unsigned)__builtin_amdgcn_readfirstlane((int)ww);
                        if (ww) single_k = (w << 6) + 63 - __builtin_clzll(ww);
                    }
                    tiles = 1u << (single_k >> 4);
                }
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
                        unsigned rest = 0;
                        if (__builtin_expect(__builtin_popcount(pmall) > MB, 0)) {
                            rest = pmall;
#pragma unroll
                            for (int i = 0; i < MB; ++i) rest &= rest - 1;
                        }
                        const unsigned pm = pmall & ~rest;
                        pmall = rest;
                        const int m = __builtin_popcount(pm);
                        const int bpar = bc & 1;
                        double *P = Pbase + bpar * MB * CW;
                        ++bc;
                        // the tile index is invariant in this loop: left alone, the compiler hoists all 17 `kappa == i` tests out of it
                        // as 64-bit lane masks (34 SGPRs, spilled to VGPR lanes and reloaded per block); the opaque copy keeps each
                        // test a plain s_cmp at its use (C3 sweep 93.0 -> 85.3 ms)
                        int kap = kappa;
                        asm volatile("" : "+s"(kap));
                        STAMP(12);
                        // ---- 1. gather the pivot columns (compacted) into the LDS panel ------------------------------------
#define PARTLS_G(i) if constexpr (i < T) { if (__builtin_expect(kap == i, 0)) gather_tile<T, H, W, i>(S, P, a, b, pm); }
                        PARTLS_CASES(PARTLS_G)                 // flat chain of independent ifs: the only form the register allocator keeps spill-free
#undef PARTLS_G
                        STAMP(13);
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
                        if (tid == 0) s_veto[bpar ^ 1] = NO_VETO;              // the other slot: last read before this barrier
                        // ---- 2. panel elimination -----------------------------------------------------------------------
                        {
                            int veto;
                            switch (m) {
#define PARTLS_PB(i) case i: if constexpr (i <= MB) veto = panel_block<(i <= MB ? i : 1), CW, MB>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid, idle_wave); else veto = 0; break;
                                PARTLS_PB(1) PARTLS_PB(2) PARTLS_PB(3) PARTLS_PB(4) PARTLS_PB(5) PARTLS_PB(6) PARTLS_PB(7) PARTLS_PB(8)
                                PARTLS_PB(9) PARTLS_PB(10) PARTLS_PB(11) PARTLS_PB(12) PARTLS_PB(13) PARTLS_PB(14) PARTLS_PB(15)
#undef PARTLS_PB
                                default: veto = panel_block<MB, CW, MB>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid, idle_wave); break;
                            }
                            if (veto < m && tid != RHSPOS) atomicMin(&s_veto[bpar], veto);     // the rhs row is no variable
                        }
                        STAMP(11);
                        __syncthreads();
                        STAMP(2);
                        // leave-one-out veto at step vs: nothing of this block has touched the register tableau yet.  The offender
                        // is rejected for the current basis and the other pivots of the block are redone; the rest of the block body
                        // runs with ZERO pivots (no extra control-flow edge in the block loop: the allocator keeps the tableau put)
                        const int vs = __builtin_amdgcn_readfirstlane(s_veto[bpar]);
                        unsigned pmx = pm;
                        int mx = m;
                        if (__builtin_expect(vs < m, 0)) {
                            unsigned r = pm;
                            for (int i = 0; i < vs; ++i) r &= r - 1;
                            const unsigned vbit = r & (0u - r);
                            blocked = blocked || (tid == 16 * kappa + __builtin_ctz(vbit));
                            pmall |= pm & ~vbit;
                            pmx = 0;
                            mx = 0;
                            ++nveto;
                        }
                        // ---- 3. fused rank-m update of the register tableau ------------------------------------------------
                        bool blk_ok = false;
#pragma unroll PARTLS_UPD_UNROLL
                        for (int s = 0; s < mx; ++s) {
                            const double inv = Dinv[s];
                            blk_ok = blk_ok || (inv != 0.0);
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
#define PARTLS_F(i) if constexpr (i < T) { if (__builtin_expect(kap == i, 0)) scatter_tile<T, H, W, i>(S, P, a, b, pmx); }
                        PARTLS_CASES(PARTLS_F)
#undef PARTLS_F
                        if (tid < 16 * T && (tid >> 4) == kappa && ((pmx >> (tid & 15)) & 1u)) {
                            const int j = __builtin_popcount(pmx & ((1u << (tid & 15)) - 1u));
                            q = P[j * CW + RHSPOS];
                            if (Dinv[j] != 0.0) basic = !basic;
                            else blocked = true;
                        }
                        progress = progress || blk_ok;
                        npiv += (unsigned)mx;
                        STAMP(4);
                    }
                    tiles &= tiles - 1;
                }
            }
            // patterns are ranked on objective^2 (the tableau corner; sqrt is monotone): one sqrt per workgroup instead of one per
            // pattern, unless every pattern's objective is wanted
            const double obj2 = corner > 0.0 ? corner : 0.0;
            // bookkeeping of the finished pattern: by the last thread — its wave owns no panel row and has slack, wave 0 has none
            if constexpr (NODE) {                                     // chain of nodes (bit-order calibration): pivots, blocks, scans so far
                if (p.node_piv && tid == 0) {
                    unsigned *o = p.node_piv + 3 * ((size_t)chain * clen + gi);
                    o[0] = npiv; o[1] = bc; o[2] = sc;
                }
            }
            if (p.all_opt && tid == THREADS - 1) p.all_opt[pat] = sqrt(obj2);
            if constexpr (!NODE && (W == 1 || (PARTLS_EXPORT_BIG && T <= EXPORT_MAXT))) {
                // (round 3: 256-thread kernel only — in the 512-thread kernel the two extra live registers moved 8 spills and cost 1.8 % of the
                // C3 sweep.  Round 4: the spills turned out to be hoisted address offsets and are gone; -DPARTLS_EXPORT_BIG=1 measures it again)
                // the workgroup's best pattern so far leaves its solution behind (rhs column of the basic variables, as node mode's
                // node_sol): the host takes the winner's from here instead of solving that pattern again from the empty basis.  Every
                // thread decides for itself on the replicated corner; on exact objective ties the FIRST pattern's solution stays (the
                // host checks the solution against the winning pattern's signs and re-solves if they disagree).
                if (p.best_sol && obj2 < mybest) {
                    mybest = obj2;
                    if (has_var) p.best_sol[(size_t)blockIdx.x * p.node_ld + tid] = basic ? q : 0.0;
                }
            }
            if (tid == THREADS - 1) {                                // lexicographic (objective, pattern) minimum: argmin's first-index rule
                const double bo = s_best[0];
                const long long bp = reinterpret_cast<long long *>(s_best)[1];
                if (obj2 < bo || (obj2 == bo && ref_index_less(pat, (unsigned long long)bp, lds_image.s_rbit))) {
                    s_best[2] = bo; reinterpret_cast<long long *>(s_best)[3] = bp;             // the old minimum becomes the runner-up
                    s_best[0] = obj2; reinterpret_cast<long long *>(s_best)[1] = (long long)pat;
                } else if (obj2 < s_best[2]) { s_best[2] = obj2; reinterpret_cast<long long *>(s_best)[3] = (long long)pat; }
            }
        }
        if constexpr (NODE) {
            if (has_var) p.node_sol[(size_t)chain * p.node_ld + tid] = basic ? q : 0.0;
            if (tid == 0) p.node_obj2[chain] = corner;
            if (p.node_dst) {                                     // snapshot of the final state: what a child node starts from
                double *snap = p.node_dst[chain];
                if (snap && (H == 0 || T > 1)) {
                    {
                        double *bp = uniform_ptr(snap) + (size_t)L::OFF * 256;  // (addresses as at the chain start: see there)
#pragma unroll
                        for (int s = 0; s < L::CNT; ++s) {
                            if ((s & 1) == 0) asm volatile("" : "+s"(bp));
                            bp[(s & 1) * 256 + t8] = S[s];
                            if (s & 1) bp += 512;
                        }
                    }
                    if (tid < 16 * T) {
                        snap[(size_t)nslots(T) * 256 + tid] = q;
                        reinterpret_cast<int8_t *>(snap + (nslots(T) * 256 + 16 * T + 8))[tid] = basic ? 1 : 0;
                    }
                    if (tid == 0) snap[(size_t)nslots(T) * 256 + 16 * T] = corner;
                }
            }
            if (p.node_tab && (H == 0 || T > 1)) {                // final tableau + basis, same layout as T0 (half 1 of T = 1 owns nothing)
                double *tab = p.node_tab + (size_t)chain * (nslots(T) * 256 + 16 * T + 8);
                {
                    double *bp = uniform_ptr(tab) + (size_t)L::OFF * 256;
#pragma unroll
                    for (int s = 0; s < L::CNT; ++s) {
                        if ((s & 1) == 0) asm volatile("" : "+s"(bp));
                        bp[(s & 1) * 256 + t8] = S[s];
                        if (s & 1) bp += 512;
                    }
                }
                if (tid < 16 * T) p.node_basic[(size_t)chain * 16 * T + tid] = basic ? 1 : 0;
            }
        }
    }
    STAMP_FLUSH;
    __syncthreads();                                      // s_best was last written by thread THREADS - 1
    if (tid == 0) {
        p.best_obj[blockIdx.x] = sqrt(s_best[0]);
        p.best_pat[blockIdx.x] = reinterpret_cast<long long *>(s_best)[1];
        if (p.second_obj) { p.second_obj[blockIdx.x] = sqrt(s_best[2]); p.second_pat[blockIdx.x] = reinterpret_cast<long long *>(s_best)[3]; }
        if (p.n_pivots && npiv) atomicAdd(p.n_pivots, (unsigned long long)npiv);
        if (p.n_unconverged && nunconv) atomicAdd(p.n_unconverged, (unsigned long long)nunconv);
        if (p.n_vetoes && nveto) atomicAdd(p.n_vetoes, (unsigned long long)nveto);
    }
}

// Barrier invariant.  The two halves of the workgroup run DIFFERENT instantiations (sweep_body<T,0> / <T,1>) and meet at
// workgroup barriers issued from different program counters.  That is only sound because every barrier of the body is reached
// under wave-uniform control flow whose conditions (count, tiles, pmall, m, vs, rounds, patience, chain and pattern counters)
// are computed from LDS words or kernel arguments that are identical for all 512 threads — never from a half's own registers —
// so both instantiations execute the same barrier sequence: per scan 1, per block 2 + m (the m panel steps; idle waves only
// count them).  Any edit that makes a barrier conditional on per-half or per-wave data deadlocks the CU.
template <int T, bool NODE>
__global__ __launch_bounds__(THREADS, 2) void sweep_blk_kernel(SweepParams p)
{
    const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    if (half == 0) sweep_body<T, 0, NODE>(p);
    else sweep_body<T, 1, NODE>(p);
}

#ifndef PARTLS_SMALL_OCC
#define PARTLS_SMALL_OCC(T) ((T) <= 8 ? 3 : 2)          // waves per SIMD = workgroups per CU the allocator is asked to leave room for
#endif
template <int T, bool NODE>
__global__ __launch_bounds__(256, PARTLS_SMALL_OCC(T)) void sweep_small_kernel(SweepParams p)
{
    sweep_body<T, 0, NODE, 1>(p);
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

}  // namespace blk

bool sweep_reg_supported(int n) { return n >= 1 && n <= 16 * blk::MAXT; }
int sweep_reg_tiles(int n) { return (n + 15) / 16; }
bool sweep_reg_exports(int T) { return PARTLS_EXPORT_BIG != 0 && T <= blk::EXPORT_MAXT; }
bool sweep_reg_small(int T) { return T <= blk::MAXT_S; }   // the 256-thread kernel (several chains per CU) runs this tile count
size_t sweep_reg_t0_doubles(int T) { return (size_t)T * (T + 1) / 2 * 256 + 16 * (size_t)T + 8; }

hipError_t launch_layout_reg(const double *Tfull, int n, int T, double *T0reg, hipStream_t s)
{
    const int tot = T * (T + 1) / 2 * 256 + 16 * T + 1;
    hipLaunchKernelGGL(blk::layout_reg_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, Tfull, n, T, T0reg);
    return hipGetLastError();
}

template <int T>
static hipError_t launch_blk_T(const SweepParams &p, int grid, hipStream_t s)
{
    if constexpr (T <= blk::MAXT_S) {                        // small tableau: 256-thread workgroups, several per CU
        if (p.node_code) hipLaunchKernelGGL((blk::sweep_small_kernel<T, true>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((blk::sweep_small_kernel<T, false>), dim3(grid), dim3(256), 0, s, p);
    } else {
        if (p.node_code) hipLaunchKernelGGL((blk::sweep_blk_kernel<T, true>), dim3(grid), dim3(blk::THREADS), 0, s, p);     // ~100 KB of static LDS
        else hipLaunchKernelGGL((blk::sweep_blk_kernel<T, false>), dim3(grid), dim3(blk::THREADS), 0, s, p);
    }
    return hipGetLastError();
}

// chains (workgroups) of the chain-mode sweep a CU runs at the same time: 1 for the 512-thread kernel, the occupancy of the 256-thread one
template <int T>
static int concurrency_T()
{
    if constexpr (T <= blk::MAXT_S) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(&blk::sweep_small_kernel<T, false>), 256, 0) != hipSuccess || n < 1) n = 1;
        return n;
    }
    return 1;
}
int sweep_reg_concurrency_query(int T);
int sweep_reg_concurrency(int T)
{
    // the occupancy query costs tens of microseconds — as much as a tenth of a whole C2-sized fit: asked once per tile count
    // (atomic: the rank threads of partls_fit_opt_multi call this concurrently; every thread would store the same value)
    static std::atomic<int> cached[blk::MAXT + 2];
    if (T >= 0 && T <= blk::MAXT) { const int v = cached[T].load(std::memory_order_relaxed); if (v > 0) return v; }
    const int v = sweep_reg_concurrency_query(T);
    if (T >= 0 && T <= blk::MAXT) cached[T].store(v, std::memory_order_relaxed);
    return v;
}
int sweep_reg_concurrency_query(int T)
{
    switch (T) {
#ifdef PARTLS_ONLY_T
        case PARTLS_ONLY_T: return concurrency_T<PARTLS_ONLY_T>();
#else
#define PARTLS_L(i) case i + 1: return concurrency_T<i + 1>();
        PARTLS_CASES(PARTLS_L)
#undef PARTLS_L
#endif
        default: return 1;
    }
}

hipError_t launch_sweep_blk(const SweepParams &p, int T, int grid, hipStream_t s)
{
    switch (T) {
#ifdef PARTLS_ONLY_T
        case PARTLS_ONLY_T: return launch_blk_T<PARTLS_ONLY_T>(p, grid, s);
#else
#define PARTLS_L(i) case i + 1: return launch_blk_T<i + 1>(p, grid, s);
        PARTLS_CASES(PARTLS_L)
#undef PARTLS_L
#endif
        default: return hipErrorInvalidValue;
    }
}

}  // namespace partls
