// sweep_lazy.hip — sign-pattern sweep beyond the register kernel (n > 320): the tableau in global memory, its updates DEFERRED.
//
// Same algorithm, same decisions and same SweepParams as sweep_generic.hip (block principal pivoting on the symmetric
// principal-pivot tableau, Opt.jl:87-90 per pattern; node mode for BnB / Alt), one workgroup per Gray-code chain.  What differs is
// where the bytes go.  sweep_generic.hip applies every block of <= 16 pivots to the whole (n+1)^2 / 2 triangle at once: 2 x 0.47 MB
// through the memory system per block at n = 341 for ~0.7 MFLOP of work — it is bound by that traffic and its latency (1.06 M solves/s at D = 340).
// But a block only ever READS m columns of the tableau (its pivot columns) and the rhs column.  So here
//   * the rhs column q (and the objective corner) lives in LDS and follows every block at once;
//   * the rank-1 terms of a block (z_s = pivot column s as of its own step, 1/d_s) are appended to a PENDING list in LDS (R terms);
//   * a pivot column k is MATERIALISED on demand:  T[i][k] = base[i][k] - sum_t z_t[i] z_t[k] / d_t over the pending terms (MFMA);
//   * the rows / columns of the pivoted variables, which a block REPLACES (T_ik = T_ik / |d|, T_kk = -1/d) rather than updates, are
//     written to the base image at once (2 m n entries), and what the pending terms hold at those indices is ZEROED: an entry of the
//     base is always "as of the last replacement of its row or column", and the terms older than that contribute exact zeros;
//   * only when the pool is full is the base image brought up to date: ONE pass with a rank-R update (v_mfma_f64_16x16x4 on 16 x 16
//     tiles of the upper triangle) — once per ~40 pivots instead of once per block.
// Base-image bytes per pivot drop ~5x (measured: profiles/r03_d340_traffic.json), and the pass itself runs on the matrix pipe with
// 16 x fewer LDS operand reads than the FMA form.  The panel of a block is eliminated in two phases with three barriers per block
// (lz_panel_eliminate).  D = 340: 1.06 -> 4.0-4.5 M solves/s.  DESIGN.md §4 "Beyond n = 320".
#include "gj_panel.h"

namespace partls {

static constexpr int LZ_MAXWORDS = 16;      // n <= 1024
static constexpr int LZ_MAXR = 64;          // rows of the LDS pool at most
#ifndef LZ_SPIN_SLEEP
#define LZ_SPIN_SLEEP 1                     // s_sleep argument of the phase-2 waves' poll of the panel's progress word
#endif
#ifndef LZ_TWO_PHASE_MAX_NT
#define LZ_TWO_PHASE_MAX_NT 1024           // workgroup sizes up to this run the two-phase panel (512: the 1024-thread plan keeps the step-by-step one)
#endif
#ifndef LZ_PANEL_STEPS
#define LZ_PANEL_STEPS 1                    // 1: the 512-thread plan runs the block panel in the register kernel's one-barrier-per-step form (lz_panel_steps); 0: two-phase everywhere (round 3)
#endif
#ifndef LZ_MIN_SPLIT
#define LZ_MIN_SPLIT 6                      // a block is cut short to fill the pool when at least this many pivots still fit
#endif

typedef double lz_double4 __attribute__((ext_vector_type(4)));

#ifdef PARTLS_LZ_STAMPS     // diagnostic build: cycles of thread 0 of workgroup 0 per phase, printed at the end
#define LZ_STAMP(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); lz_cyc[slot] += now_ - lz_last; lz_last = now_; ++lz_cnt[slot]; } while (0)
#else
#define LZ_STAMP(slot) do { } while (0)
#endif
#ifdef PARTLS_LZ_STAMPS
#define LZ_STK lz_cyc
#else
#define LZ_STK nullptr
#endif

__device__ __forceinline__ int lz_sign_of_var(uint64_t m, uint64_t pat) { return 2 * __popcll(m & pat) - __popcll(m); }

// entry (i, k) of the stored upper triangle
__device__ __forceinline__ size_t lz_tri(int i, int k, int ld) { return i <= k ? (size_t)i * ld + k : (size_t)k * ld + i; }

// The base image brought up to date: T[i][c] -= sum_t z_t[i] z_t[c] / d_t over the pending terms, for the upper triangle (c >= i), rows
// and columns < n (column n, the rhs, lives in LDS).  No masks: entries of a term that must not be applied any more (row / column
// replaced since — see the kernel) were zeroed in LDS when that happened, rejected pivots and the padding up to a multiple of 4 have
// 1/d = 0.  Work item = a strip of FOUR 16 x 16 tiles of one tile row (four independent accumulators per wave, shared A fragment),
// k = 4 pending terms per MFMA: A[i][k] = -z_t[i] / d_t, B[k][c] = z_t[c]; C/D map: col = lane & 15, row = (lane >> 4) + 4 reg.
// The strip's 16 loads are issued FIRST and only consumed after the MFMAs (the products are summed from zero and added to the loaded
// entries at the end), so the memory latency of a strip hides behind its own arithmetic without any cross-iteration pipelining (a
// prefetch of the next strip into a second register set ends in s_waitcnt vmcnt(0) in front of the MFMAs: the compiler cannot count
// across the loop's back edge).  Diagonal tiles are updated whole (their lower halves are never read).  Floor: n^2 flop per pending
// term on a 0.3 TFLOP/s CU — ~1k cycles per pivot at n = 341.
template <int NT>
__device__ __forceinline__ void lz_flush(double *T, int ld_, int n_, const double *Zp, const double *dp, int Rcur_, int tid, unsigned long long *stk = nullptr)
{
#ifdef PARTLS_LZ_STAMPS
    unsigned long long fl_t0 = __builtin_readcyclecounter(), fl_k = 0, fl_l = 0;
#endif
    const int ld = __builtin_amdgcn_readfirstlane(ld_), n = __builtin_amdgcn_readfirstlane(n_);
    const int Rcur = __builtin_amdgcn_readfirstlane(Rcur_);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int NTl = (n + 15) >> 4;                                          // tile rows / columns
    const int ksteps = (Rcur + 3) >> 2;
    constexpr int NW = NT / 64;
    constexpr int SW = NT <= 512 ? 4 : 2;                                   // tiles per strip (2 x 8 x SW VGPRs: the 1024-thread plan has 128 in all)
    int nitems = 0;
    for (int I = 0; I < NTl; ++I) nitems += (NTl - I + SW - 1) / SW;
    for (int w = wave; w < nitems; w += NW) {
        int I = 0, rem = w;
        for (;;) { const int c = (NTl - I + SW - 1) / SW; if (rem < c) break; rem -= c; ++I; }
        const int J0 = I + SW * rem;
        // EDGE: the strip reaches beyond row / column n - 1 (last tile row / column): clamped loads and operand reads (out-of-range results
        // are computed on valid data and not stored).  Interior strips — most — run without any of that.
        auto strip = [&](auto edge_c) {
            constexpr bool EDGE = decltype(edge_c)::value;
            const int ra = I * 16 + fr, ra_c = (EDGE && ra >= n) ? n - 1 : ra;
            int cb[SW], cb_c[SW], off[4];
#pragma unroll
            for (int q = 0; q < SW; ++q) { cb[q] = (J0 + q) * 16 + fr; cb_c[q] = (EDGE && cb[q] >= n) ? n - 1 : cb[q]; }
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int row = I * 16 + fk + 4 * r; off[r] = ((EDGE && row >= n) ? n - 1 : row) * ld; }
            lz_double4 old[SW], acc[SW];
#pragma unroll
            for (int q = 0; q < SW; ++q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { old[q][r] = T[off[r] + cb_c[q]]; acc[q][r] = 0.0; }
            }
#ifdef PARTLS_LZ_STAMPS
            fl_l += __builtin_readcyclecounter() - fl_t0; fl_t0 = __builtin_readcyclecounter();
#endif
            // operands of k-step ks: term t = 4 ks + fk; the reads of step ks + 1 are in flight under the MFMAs of step ks
            const double *zr = Zp + fk * ld;
            const double *dr = dp + fk;
            double za = zr[ra_c], zb[SW], di = dr[0];
#pragma unroll
            for (int q = 0; q < SW; ++q) zb[q] = zr[cb_c[q]];
            for (int ks = 0; ks < ksteps; ++ks) {
                const double a = -za * di;
                double b[SW];
#pragma unroll
                for (int q = 0; q < SW; ++q) b[q] = zb[q];
                zr += 4 * ld; dr += 4;                                      // one step beyond the last: rows of the pool / qs — read, never used
                za = zr[ra_c]; di = dr[0];
#pragma unroll
                for (int q = 0; q < SW; ++q) zb[q] = zr[cb_c[q]];
#pragma unroll
                for (int q = 0; q < SW; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[q], acc[q], 0, 0, 0);
            }
#ifdef PARTLS_LZ_STAMPS
            asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[SW - 1][0]));
            fl_k += __builtin_readcyclecounter() - fl_t0; fl_t0 = __builtin_readcyclecounter();
#endif
#pragma unroll
            for (int q = 0; q < SW; ++q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (!EDGE || (cb[q] < n && I * 16 + fk + 4 * r < n)) T[off[r] + cb_c[q]] = old[q][r] + acc[q][r];
                }
            }
        };
        if (I * 16 + 16 <= n && (J0 + SW) * 16 <= n) strip(std::false_type{});
        else strip(std::true_type{});
    }
#ifdef PARTLS_LZ_STAMPS
    fl_l += __builtin_readcyclecounter() - fl_t0;
    if (stk) { stk[6] += fl_k; stk[7] += fl_l; }
#endif
}

__device__ __forceinline__ double lz_readlane(double v, int l)            // l: wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double lz_bperm(double v, int srclane)        // v of lane `srclane` (any lane -> any lane; no LDS memory)
{
    const int lo = __builtin_amdgcn_ds_bpermute(srclane << 2, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(srclane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// acc += y[lane 16 (lane / 16) + g] * x   (gfx90a+; semantics and rate checked by tools/ubench/fmac_dpp.hip)
__device__ __forceinline__ void lz_fmac_bcast(double &acc, double y, double x, int g)
{
    switch (g) {
#define LZ_FB(G) case G: asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #G " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(y), "v"(x)); break;
        LZ_FB(0) LZ_FB(1) LZ_FB(2) LZ_FB(3) LZ_FB(4) LZ_FB(5) LZ_FB(6) LZ_FB(7) LZ_FB(8) LZ_FB(9) LZ_FB(10) LZ_FB(11) LZ_FB(12) LZ_FB(13) LZ_FB(14) LZ_FB(15)
#undef LZ_FB
        default: break;
    }
}

// The block's panel, eliminated in LDS — the recurrence of gj_panel_eliminate (gj_panel.h: same operations in the same order on every
// row, same acceptance rule, same outputs: BIT-IDENTICAL results), reorganised so that it needs three barriers per BLOCK instead of two
// per PIVOT.  Row i of the panel only ever needs, at step s, the entries of pivot column s at the m pivot rows (u_s[j]), d_s and 1/d_s:
// the recurrence restricted to the m pivot ROWS is self-contained.  So
//   phase 1: ONE wave runs those m rows (lane j = pivot row k_j; the u_s[j] of a step are DPP row broadcasts of the lanes' own entries)
//            and leaves the table (u_s[j], d_s, 1/d_s) in LDS — m dependent steps, no barrier, no LDS round trip;
//   phase 2: every other row runs its m steps against the table — no communication at all.
// MT (8 or 16) is the compiled panel width: columns / lanes m..MT-1 are zero and stay zero, so no step carries per-column guards.
// The leave-one-out veto (an entering pivot is refused when ANY basic row j has T_jk^2 eps >= d_k) is only known after phase 2: the
// steps are taken optimistically, every row raises the flag of a step it would have vetoed, and in the rare case that one is raised
// the block is redone from its (still untouched) panel image with that step marked "refused" — exactly what the step-by-step form does.
// tab: [GJ_MB][GJ_MB] u, [GJ_MB][GJ_MB] final pivot rows, [GJ_MB] d, [GJ_MB] 1/d;  red: [GJ_MB] veto flags.
template <int NT, int MT>
__device__ __forceinline__ int lz_panel_eliminate(double *__restrict__ Pn, double *__restrict__ Zn, double *__restrict__ dinv,
                                                  double *__restrict__ tab, double *__restrict__ red, const int *__restrict__ ks, int m_,
                                                  int ld_, uint8_t *__restrict__ s_basic, int myj, unsigned basm_, double piv_eps, int tid,
                                                  unsigned long long &nveto, unsigned long long *stk = nullptr, bool drop_progress = false)
{
#ifdef PARTLS_LZ_STAMPS
    unsigned long long pt0 = __builtin_readcyclecounter();
#define LZ_PSTAMP(i) do { if (stk) { const unsigned long long n_ = __builtin_readcyclecounter(); stk[i] += n_ - pt0; pt0 = n_; } } while (0)
#else
#define LZ_PSTAMP(i) do { } while (0)
#endif
    const int m = __builtin_amdgcn_readfirstlane(m_), ld = __builtin_amdgcn_readfirstlane(ld_);
    const unsigned basm = (unsigned)__builtin_amdgcn_readfirstlane((int)basm_);   // bit s: pivot s LEAVES the basis (no acceptance test)
    const int lane = tid & 63;
    const bool has_row = tid < ld;
    const bool var_row = tid < ld - 1;                                   // the last row is the rhs: no variable, never vetoes
    double *__restrict__ tabU = tab, *__restrict__ tabF = tab + GJ_MB * GJ_MB, *__restrict__ tabD = tabF + GJ_MB * GJ_MB,
           *__restrict__ tabI = tabD + GJ_MB;
    const int krow = ks[lane < m ? lane : 0];                            // phase 1: the pivot row this lane of wave 0 stands for
    unsigned skip = 0;                                                   // steps refused by the veto (found in earlier trips)
    unsigned accm = 0;
    LZ_PSTAMP(8);
    double pv[MT];                                                       // phase 2: this thread's own row
    // phase 1 runs on the LAST wave (at n <= 447 it owns no tableau row, so nobody waits for its phase 2), the other waves follow it step
    // by step: `prog` (LDS) = steps whose table row is complete; a row's step s starts as soon as prog > s.  The last wave takes its own
    // rows (if any) when it is through.  Every wait is on the last wave only, which never waits: no cycle.
    constexpr int P1W = NT / 64 - 1;
    const bool p1wave = (tid >> 6) == P1W;
    int *prog = reinterpret_cast<int *>(red + GJ_MB);                    // red: [GJ_MB] flags, this word, a failure flag
    for (;;) {
        if (tid < GJ_MB) red[tid] = 0.0;
        if (tid == 0) __hip_atomic_store(prog, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        if (p1wave) {                                                    // ---- phase 1 -------------------------------------------
            // All 64 lanes work: lane 16 g + j holds the entries of pivot row k_j in the four panel columns 4 g .. 4 g + 3.  A step needs, in
            // every lane, column s at its own pivot row (zi: the multiplier) and at the pivot rows of its four columns (u): two wave
            // permutes (ds_bpermute, no LDS memory) bring column s from the DPP row that owns it into every row — once as it is, once
            // rotated by 4 g, so that u_{4g+c} sits in lane c of row g and v_fmac_f64_dpp row_newbcast:c hands it to the whole row.
            // ~45 instructions per step instead of ~150 with all 16 columns in one lane (the step is bound by ONE wave's issue rate).
            const int g = lane >> 4, jl = lane & 15;
            const int krow1 = ks[jl < m ? jl : 0];
            double pp[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) pp[c] = (4 * g + c < m && jl < m) ? Pn[(4 * g + c) * ld + krow1] : 0.0;
#pragma unroll
            for (int s = 0; s < GJ_MB; ++s) {
                if (s >= m) break;                                       // uniform
                const int gs = s >> 2, cs = s & 3;                       // the DPP row and register that hold column s
                const double zi = lz_bperm(pp[cs], 16 * gs + jl);        // column s at pivot row j
                const double zr = lz_bperm(pp[cs], 16 * gs + ((jl + 4 * g) & 15));   // ... at pivot row j + 4 g: lane c of row g = u_{4g+c}
                if (lane < GJ_MB) tabU[s * GJ_MB + lane] = zi;           // phase 2 reads it as broadcasts
                const double d = lz_readlane(zi, s);
                const bool bas = (basm >> s) & 1u, skp = (skip >> s) & 1u;
                const bool pre = !skp && (bas || d > piv_eps);           // uniform
                if (!bas && !skp && lane < m && lane != s && (zi * zi) * piv_eps >= d) red[s] = 1.0;
                const double inv = pre ? gj_rcp(d) : 0.0, ainv = fabs(inv);    // refused: 1/d = 0 makes the step a no-op below
                if (lane == 0) { tabD[s] = d; tabI[s] = inv; }
                // the LDS executes one wave's operations in order: the progress word cannot become visible before the table row written
                // above it — no s_waitcnt needed, only the compiler must keep the order
                asm volatile("" ::: "memory");
                // (drop_progress: fault injection, PARTLS_LZ_FAULT — the word is never written, every follower runs into the bound of its wait)
                if (lane == 0 && !drop_progress) __hip_atomic_store(prog, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (lane < m) Zn[s * ld + krow1] = zi;
                // row update  pp[c] = fma(u_{4g+c}, -zi / d, pp[c]); the pivot row itself becomes u |1/d| = fma(u, |1/d|, 0): the same
                // instruction with its own multiplier and a zeroed addend
                const bool piv = pre && jl == s;
                double mi = piv ? ainv : -zi * inv;
                if (piv) { pp[0] = 0.0; pp[1] = 0.0; pp[2] = 0.0; pp[3] = 0.0; }
                double zrr = zr;
                asm volatile("s_nop 1" : "+v"(mi), "+v"(zrr));           // a DPP read of a fresh VALU result needs two wait states the assembler cannot see
                lz_fmac_bcast(pp[0], zrr, mi, 0);
                lz_fmac_bcast(pp[1], zrr, mi, 1);
                lz_fmac_bcast(pp[2], zrr, mi, 2);
                lz_fmac_bcast(pp[3], zrr, mi, 3);
                if (pre && g == gs) pp[cs] = piv ? -inv : zi * ainv;     // column s itself (what the fmac left there is not used)
            }
            if (jl < m) {                                                // the pivot rows' final entries wait in LDS until the flags are known
#pragma unroll
                for (int c = 0; c < 4; ++c) tabF[jl * GJ_MB + 4 * g + c] = pp[c];
            }
        }
        LZ_PSTAMP(9);
        if (has_row && myj < 0) {                                        // ---- phase 2: all other rows ---------------------------
#pragma unroll
            for (int j = 0; j < MT; ++j) pv[j] = (j < m) ? Pn[j * ld + tid] : 0.0;
#pragma unroll
            for (int s = 0; s < MT; ++s) {
                if (s >= m) break;                                       // uniform
                if (!p1wave) {                                           // (uniform) wait for the table row of step s
                    int spins = 0;
                    while (__hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= s && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(LZ_SPIN_SLEEP);
                    // the wave polled sits in the same workgroup and never waits itself, so the bound (seconds) is never reached; should it
                    // be, the solve must not pass for converged: the flag ends up in n_unconverged (PARTLS_ERR_NOT_CONVERGED)
                    if (spins >= (1 << 22)) red[GJ_MB + 1] = 1.0;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
                // the whole table row in ONE batch of LDS reads, ahead of this step's LDS stores (the compiler cannot move a load across a
                // store that may alias it: read in program order behind them, every ds_read2 pays its own round trip — 5 per step)
                const double inv = tabI[s];                              // 0: refused — the row does not move
                const double ds_ = tabD[s];
                double u[MT];
#pragma unroll
                for (int j = 0; j < MT; ++j) u[j] = tabU[s * GJ_MB + j];
                const double zi = pv[s];
                if (!((basm >> s) & 1u) && var_row && (zi * zi) * piv_eps >= ds_) red[s] = 1.0;   // only counted when 1/d != 0
                Zn[s * ld + tid] = zi;
                const double mi = -zi * inv;
#pragma unroll
                for (int j = 0; j < MT; ++j) if (j != s) pv[j] = fma(mi, u[j], pv[j]);
                pv[s] = (inv != 0.0) ? zi * fabs(inv) : zi;
            }
        }
        LZ_PSTAMP(11);
        __syncthreads();
        LZ_PSTAMP(12);
        const bool live = lane < m && tabI[lane] != 0.0;
        const unsigned veto = (unsigned)__ballot(live && red[lane] != 0.0);
        accm = (unsigned)__ballot(live);
        if (veto == 0) {
            // the final panel: rows / columns of the pivoted variables
            if (has_row && myj < 0) {
#pragma unroll
                for (int j = 0; j < MT; ++j) if (j < m) Pn[j * ld + tid] = pv[j];
            }
            if (p1wave && lane < m) {
#pragma unroll
                for (int j = 0; j < MT; ++j) if (j < m) Pn[j * ld + krow] = tabF[lane * GJ_MB + j];
                if (live) s_basic[krow] ^= 1;                            // accepted pivots change sides
                dinv[lane] = tabI[lane];
            }
            break;
        }
        skip |= veto & (0u - veto);                                      // the first refused step changes every later one: one at a time
        ++nveto;
        __syncthreads();
    }
    LZ_PSTAMP(13);
    __syncthreads();
    LZ_PSTAMP(14);
    return __popc(accm);
}

// NODE: node mode (BnB bounds, Alt alpha-steps, calibration walks) is its own instantiation: only there can the base image live in a
// snapshot slot, so only there is T a per-chain variable (as a run-time choice it cost the chain-mode sweep 4 %: D = 340 56.7 -> 59.0 ms)
// ---- the panel in the register kernel's form (round 4; LZ_PANEL_STEPS) ------------------------------------------------------------------
// sweep_blk.hip's panel_block: thread t owns ROW t of the block's M (compile-time) pivot columns in registers; M Gauss–Jordan steps, ONE
// barrier each: the pivot-row threads publish their entry of the current pivot column (U, double buffered by step parity), the one
// pivot-row thread of the step computes 1/d, everybody reads both back and updates its own row — straight-line code per M, no guards.
// 585 cycles per step in tools/ubench/panel_two_phase.hip against the ~1.8k per pivot the two-phase form above costs HERE (its phase 1
// needs two wave permutes per step to serve up to 16 columns from 4 per lane).  Same operations in the same order on every row as
// lz_panel_eliminate / gj_panel_eliminate; the leave-one-out veto is raised per row and step and resolved by the caller (redo from the
// untouched panel image with the offender skipped).  U: [2][16 + 64] doubles (slots 16.. are per-lane dummies).
template <int M>
__device__ __forceinline__ int lz_panel_steps(double *__restrict__ Pn, double *__restrict__ Zn, double *__restrict__ U, double *__restrict__ dn,
                                              int *__restrict__ vflag, int ld, int myj, bool my_basic, unsigned skip, double piv_eps, int tid,
                                              bool has_row, bool var_row, bool idle_wave, unsigned long long *stk = nullptr)
{
#ifdef PARTLS_LZ_STAMPS
    unsigned long long pt0 = __builtin_readcyclecounter();
#endif
    if (idle_wave) {
#pragma unroll
        for (int s = 0; s < M; ++s) __syncthreads();
        __syncthreads();                                     // the veto word's barrier
        return __builtin_amdgcn_readfirstlane(*vflag);
    }
    constexpr int US = GJ_MB + 64;
    const int row = has_row ? tid : ld - 1;                  // threads beyond the rhs row shadow it: computed, never stored
    const int uslot = myj >= 0 ? myj : GJ_MB + (tid & 63);
    // The row lives in an array of exactly M doubles that never leaves this function (round 4: as a 16-wide array handed back to the caller
    // of the switch over M, every join of the per-lane `pivot row?` branch below copied all 16 registers — 30-40 v_mov_b64 per step, 1.5k
    // cycles per step against the 585 of tools/ubench/panel_two_phase.hip), and the row update is ONE instruction stream for pivot and
    // other rows (multiplier and addend selected per lane, as phase 1 of lz_panel_eliminate does).
    double pv[M];
#pragma unroll
    for (int j = 0; j < M; ++j) pv[j] = Pn[(size_t)j * ld + row];
    int veto = M;
    LZ_PSTAMP(8);
#pragma unroll
    for (int s = 0; s < M; ++s) {
        if (has_row) Zn[(size_t)s * ld + tid] = pv[s];
        U[(s & 1) * US + uslot] = pv[s];
        if (myj == s) {                                      // the one pivot-row thread (its wave only: the others branch over)
            const double d = pv[s];
            dn[s] = (!((skip >> s) & 1u) && (my_basic || d > piv_eps)) ? gj_rcp(d) : 0.0;
        }
        __syncthreads();
        // ONE LDS read brings the pivot rows' entries of column s into every wave: lane l of every row of 16 lanes holds u_l, and the row
        // update takes u_j from lane j of its own row (v_fmac_f64_dpp row_newbcast:j).  As M broadcast reads per thread the step kept the
        // LDS pipe busy ~300 cycles (6 waves x (M / 2 + 1) ds_read2_b64, each a full 64-lane transfer): the per-step stamps showed 600
        // cycles between the barrier and the data
        double inv = dn[s];
        double uvec = U[(s & 1) * US + (tid & 15)];
        // leave-one-out veto: T_js^2 >= d / piv_eps  (inv = 1/d; negative for a leaving pivot, 0 for a rejected one); the rhs row is no variable
        if (var_row && (pv[s] * pv[s]) * (inv * piv_eps) >= 1.0) veto = veto < s ? veto : s;
        if (__builtin_expect(__builtin_amdgcn_readfirstlane(__double2hiint(inv)) & 0x7ff00000, 1)) {
            const bool piv = myj == s;
            const double ainv = fabs(inv);
            double mult = piv ? ainv : -pv[s] * inv;         // pivot row: u |1/d| = fma(|1/d|, u, 0);  other rows: fma(-z / d, u, own entry)
            const double ps = piv ? -inv : pv[s] * ainv;
            asm volatile("s_nop 1" : "+v"(mult), "+v"(uvec));   // (a DPP read of a fresh VALU result needs two wait states the assembler cannot see)
#pragma unroll
            for (int j = 0; j < M; ++j) {
                if (j == s) continue;
                double acc = piv ? 0.0 : pv[j];
                lz_fmac_bcast(acc, uvec, mult, j);
                pv[j] = acc;
            }
            pv[s] = ps;
        }
    }
    LZ_PSTAMP(9);
    if (veto < M) atomicMin(vflag, veto);
    __syncthreads();
    LZ_PSTAMP(11);
    const int vs = __builtin_amdgcn_readfirstlane(*vflag);
    if (vs >= M && has_row) {                                // no step refused: the final panel = rows / columns of the pivoted variables
#pragma unroll
        for (int j = 0; j < M; ++j) Pn[(size_t)j * ld + tid] = pv[j];
    }
    return vs;
}

// the block's panel through lz_panel_steps: redo on a veto (the panel image is untouched until no step is refused), flip the basis flags
// of the accepted pivots; returns the number of accepted pivots
template <int NT>
__device__ __forceinline__ int lz_panel_stepwise(double *__restrict__ Pn, double *__restrict__ Zn, double *__restrict__ dn, double *__restrict__ U,
                                                 int *__restrict__ vflag, int m_, int ld_, uint8_t *__restrict__ s_basic, int myj, double piv_eps,
                                                 int tid, unsigned long long &nveto, unsigned long long *stk = nullptr)
{
#ifdef PARTLS_LZ_STAMPS
    unsigned long long pt0 = __builtin_readcyclecounter();
#endif
    const int m = __builtin_amdgcn_readfirstlane(m_), ld = __builtin_amdgcn_readfirstlane(ld_);
    const bool has_row = tid < ld, var_row = tid < ld - 1;
    const bool idle_wave = __builtin_amdgcn_readfirstlane((tid & ~63) >= ld ? 1 : 0) != 0;
    const bool my_basic = myj >= 0 && s_basic[tid] != 0;
    const int lane = tid & 63;
    unsigned skip = 0;
    for (;;) {
        if (tid == 0) *vflag = GJ_MB;
        // (the barrier of the first step orders this store before every atomicMin: a veto is only raised after step 0's barrier)
        int vs;
        switch (m) {
#define LZ_PS(i) case i: vs = lz_panel_steps<i>(Pn, Zn, U, dn, vflag, ld, myj, my_basic, skip, piv_eps, tid, has_row, var_row, idle_wave, stk); break;
            LZ_PS(1) LZ_PS(2) LZ_PS(3) LZ_PS(4) LZ_PS(5) LZ_PS(6) LZ_PS(7) LZ_PS(8) LZ_PS(9) LZ_PS(10) LZ_PS(11) LZ_PS(12) LZ_PS(13) LZ_PS(14) LZ_PS(15)
#undef LZ_PS
            default: vs = lz_panel_steps<GJ_MB>(Pn, Zn, U, dn, vflag, ld, myj, my_basic, skip, piv_eps, tid, has_row, var_row, idle_wave, stk); break;
        }
        LZ_PSTAMP(10);                                        // (entry, the whole variant)
        if (vs >= m) break;
        skip |= 1u << vs;                                     // the first refused step changes every later one: one at a time
        ++nveto;
        __syncthreads();                                      // everybody has read the flag before thread 0 resets it
    }
    const double dmy = dn[lane < m ? lane : 0];
    const unsigned accm = (unsigned)__ballot(lane < m && dmy != 0.0);
    if (myj >= 0 && dn[myj] != 0.0) s_basic[tid] ^= 1;         // accepted pivots change sides
    LZ_PSTAMP(12);
    __syncthreads();
    LZ_PSTAMP(13);
    return __popc(accm);
}

template <int NT, bool NODE>
__global__ __launch_bounds__(NT) void sweep_lazy_kernel(SweepParams p, int mb, int rows)
{
    const int n = p.n, ld = n + 1;
    const int tid = threadIdx.x, lane = tid & 63;
    extern __shared__ double smem[];
    // ONE pool of `rows` tableau-row images: the pending terms grow from row 0, the panel of the current block (m <= mb columns) sits in
    // the last m rows — a block fits while Rcur + 2 m <= rows (a fixed mb-row panel would idle half its rows on the typical block)
    double *pool = smem;                                          // [rows + 4][ld]  (+4: the flush reads one k-step beyond the last)
    double *Zp = pool;                                            // pending terms: column s as of its own step
    double *qs = pool + (size_t)(rows + 4) * ld;                  // [ld] the rhs column, up to date; qs[n] = objective^2
    double *dp = qs + ld;                                         // [rows + GJ_MB] 1/d of the pending terms (0: rejected, never applied)
    double *Cj = dp + rows + GJ_MB;                               // [rows + 4][GJ_MB] z_t[k_j] / d_t of the current block's pivot columns
    double *tab = Cj + (size_t)(rows + 4) * GJ_MB;                      // 2 [GJ_MB][GJ_MB] + 2 [GJ_MB]: the panel's pivot-row tables (lz_panel_eliminate)
    double *red = tab + 2 * GJ_MB * GJ_MB + 2 * GJ_MB;                // [16] veto flags + the panel's progress word
    uint8_t *s_basic = reinterpret_cast<uint8_t *>(red + 18);     // n bytes
    uint8_t *s_blocked = s_basic + n;                             // n bytes
    int8_t *rowj = reinterpret_cast<int8_t *>(s_blocked + n);    // [ld] row i is pivot row rowj[i] of the current block (-1: none)
    __shared__ unsigned s_basm;
    __shared__ unsigned long long s_inf[LZ_MAXWORDS];
    __shared__ int s_viol[LZ_MAXWORDS * 64];

    double *const Tscratch = p.scratch + (size_t)blockIdx.x * (size_t)ld * (size_t)ld;
    double *T = Tscratch;
    const int nwords = (n + 63) >> 6;
    const bool has_row = tid < ld;
    const uint64_t mymask = (tid < n && !p.node_code) ? p.mask[tid] : 0;   // group membership of this thread's variable (chain mode)

    double best_obj = __builtin_inf();
    long long best_pat = -1;
    double second_obj = __builtin_inf();
    long long second_pat = -1;
    unsigned long long npiv = 0, nunconv = 0, nveto = 0;
    double wg_best2 = __builtin_inf();                            // objective^2 of the pattern whose solution sits in p.best_sol (the same in every thread)
    bool lz_fault = p.coop_fault == 77 && blockIdx.x == 0;       // test hook: workgroup 0's first block loses its progress word (see lz_panel_eliminate)

#ifdef PARTLS_LZ_STAMPS
    unsigned long long lz_cyc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, lz_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lz_last = __builtin_readcyclecounter();
#endif
    for (int i = tid; i < (rows + 4) * ld; i += NT) pool[i] = 0.0;   // everything the flush may read is finite from the start
    for (int i = tid; i < rows + GJ_MB; i += NT) dp[i] = 0.0;
    for (int i = tid; i < ld; i += NT) rowj[i] = -1;
    if (tid == 0) red[GJ_MB + 1] = 0.0;                            // set if a wait on the panel's progress word ever ran into its bound

    const int64_t total = p.g_end - p.g_begin;
    const int64_t nchains = (total + p.chain_len - 1) / p.chain_len;

    for (int64_t chain = blockIdx.x; chain < nchains; chain += gridDim.x) {
        const int64_t g0 = p.g_begin + chain * p.chain_len;
        const int64_t g1 = (g0 + p.chain_len < p.g_end) ? g0 + p.chain_len : p.g_end;
        // Tableau SNAPSHOTS (node mode; round 4: BnB warm starts beyond the register kernel, BnB.jl:120-124).  A snapshot is the state a
        // node solve ends in, with every pending term applied: [ld x ld] base image (upper triangle of the n x n block), [ld] rhs column +
        // corner, [n] basis flags.  A node with a destination slot works IN that slot (its base image lives there from the start: one copy
        // in, none out); its source is its parent's slot, or the fresh tableau.
        const double *src = p.T0;
        bool from_snap = false;
        if constexpr (NODE) T = Tscratch;
        if constexpr (NODE) {                                         // wave-uniform (kernel arguments and uniform loads)
            if (p.node_dst && p.node_dst[chain]) T = p.node_dst[chain];
            if (p.node_src && p.node_src[chain]) { src = p.node_src[chain]; from_snap = true; }
        }
        for (int i = tid >> 6; i < n; i += NT / 64)                  // upper triangle only, whole 64-chunks
            for (int c = (i & ~63) + lane; c < n; c += 64) T[(size_t)i * ld + c] = src[(size_t)i * ld + c];
        if (has_row) qs[tid] = from_snap ? src[(size_t)ld * ld + tid] : src[(size_t)tid * ld + n];
        {
            const uint8_t *fl = reinterpret_cast<const uint8_t *>(src + (size_t)ld * ld + ld);
            for (int i = tid; i < n; i += NT) { s_basic[i] = from_snap ? fl[i] : 0; s_blocked[i] = 0; }
        }
        int Rcur = 0;                                                         // pending terms (uniform)
        __threadfence_block();
        __syncthreads();
        LZ_STAMP(0);                                                          // chain start

        for (int64_t g = g0; g < g1; ++g) {
            uint64_t pat = (uint64_t)g ^ ((uint64_t)g >> 1);
            const int8_t *code = p.node_code ? p.node_code + ((size_t)chain * p.chain_len + (size_t)(g - g0)) * p.node_ld : nullptr;
            if (code) pat = (uint64_t)chain;
            for (int i = tid; i < n; i += NT) s_blocked[i] = 0;
            __syncthreads();
            int ninf_best = n + 1, patience = 3, rounds = 0;
            bool progress = false;
            for (;;) {
                if (progress) {                                    // rejections hold for the basis they were tested against only
                    for (int i = tid; i < n; i += NT) s_blocked[i] = 0;
                    __syncthreads();
                }
                progress = false;
                // ---- KKT scan of the rhs column (LDS); every violator finds its own place in the list ------------------
                bool bad = false;
                if (tid < n) {
                    const int v = tid;
                    const double q = qs[v];
                    const int cd = code ? (int)code[v] : 0;
                    const int f = code ? (cd == 2 ? 0 : cd) : lz_sign_of_var(mymask, pat);
                    const double fq = (f > 0) ? q : ((f < 0) ? -q : 0.0);
                    if (cd == 2) bad = !s_basic[v] && !s_blocked[v] && (fabs(q) > p.tol);     // free: stationarity only
                    else if (s_basic[v]) bad = (f == 0) || (fq < -p.tol);
                    else bad = (fq > p.tol) && !s_blocked[v];
                }
                const unsigned long long bal = __ballot(bad);
                if (lane == 0 && (tid >> 6) < nwords) s_inf[tid >> 6] = bal;
                __syncthreads();
                int count = 0, before = 0;
                for (int w = 0; w < nwords; ++w) {
                    const int c = __popcll(s_inf[w]);
                    if (w < (tid >> 6)) before += c;
                    count += c;
                }
                if (count == 0) break;
                bool all;
                if (count < ninf_best) { ninf_best = count; patience = 3; all = true; }
                else if (patience > 0) { --patience; all = true; }
                else all = false;                                  // backup rule: single pivot, largest index
                if (++rounds > p.max_rounds) { ++nunconv; break; }
                if (bad) {                                         // ascending list; the backup rule keeps only the last one
                    const int pos = before + __popcll(bal & ((1ULL << lane) - 1ULL));
                    if (all) s_viol[pos] = tid;
                    else if (pos == count - 1) s_viol[0] = tid;
                }
                __syncthreads();
                const int nv = all ? count : 1;
                LZ_STAMP(1);                                                  // scan + violator list
                for (int b0 = 0; b0 < nv;) {
                    int m = (nv - b0 < mb) ? nv - b0 : mb;
                    if (Rcur + 2 * m > rows) {
                        // no room for m more terms below an m-row panel.  A smaller block that fills the pool first (one more block,
                        // but ~40 % more terms per pass over the base image), or the pass now
                        const int m2 = (rows - Rcur) >> 1;
                        if (m2 >= LZ_MIN_SPLIT) m = m2;
                        else {
                            if (tid < 4) dp[Rcur + tid] = 0.0;                // padding of the last k-step
                            __syncthreads();
                            #ifdef PARTLS_LZ_STAMPS
                            lz_flush<NT>(T, ld, n, Zp, dp, Rcur, tid, lz_cyc);
#else
                            lz_flush<NT>(T, ld, n, Zp, dp, Rcur, tid);
#endif
                            __threadfence_block();
                            __syncthreads();
                            Rcur = 0;
                            LZ_STAMP(2);                                      // flush
                        }
                    }
                    m = __builtin_amdgcn_readfirstlane(m);
                    const int *ks = s_viol + b0;
                    double *Pn = pool + (size_t)(rows - m) * ld;   // the block's panel: its pivot columns, all rows
                    // ---- coefficients of the pending terms for this block's columns (rows padded with zeros to a multiple of 4) ------
                    for (int e = tid; e < ((Rcur + 3) & ~3) * GJ_MB; e += NT) {
                        const int t = e / GJ_MB, j = e % GJ_MB;
                        Cj[e] = (j < m && t < Rcur) ? Zp[(size_t)t * ld + ks[j]] * dp[t] : 0.0;
                    }
                    if (tid < 64) {                                // who is a pivot row of this block, and which pivots leave the basis
                        const bool mine = tid < m;
                        const int k = ks[mine ? tid : 0];
                        if (mine) rowj[k] = (int8_t)tid;
                        const unsigned long long bm = __ballot(mine && s_basic[k] != 0);
                        if (tid == 0) s_basm = (unsigned)bm;
                    }
                    __syncthreads();
                    const int myj = has_row ? (int)rowj[tid] : -1;
                    const unsigned basm = s_basm;
                    // ---- the block's pivot columns: base entries (in flight) minus the pending terms ------------------
                    // P[j][i] = base[i][k_j] - sum_t z_t[i] (z_t[k_j] / d_t): a (rows x terms) x (terms x 16) product on the matrix pipe, 16
                    // rows per MFMA tile (A[i][k] = z_t[i], B[k][j] = Cj[t][j]); C/D map: column j = lane & 15, row = (lane >> 4) + 4 reg.
                    {
                        constexpr int TPW = NT <= 512 ? 3 : 1;     // tiles per wave and turn: their base loads are all in flight together
                        const int fr = lane & 15, fk = lane >> 4, wv = tid >> 6;
                        const int ksteps = (Rcur + 3) >> 2, NTl = (n + 15) >> 4;
                        const int kcol = ks[fr < m ? fr : 0];
                        for (int t0 = wv * TPW; t0 < NTl; t0 += (NT / 64) * TPW) {
                            double base[TPW][4];
                            lz_double4 acc[TPW];
#pragma unroll
                            for (int u = 0; u < TPW; ++u) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int row = (t0 + u) * 16 + fk + 4 * r, rc = row < n ? row : n - 1;
                                    base[u][r] = (fr < m) ? T[lz_tri(rc, kcol, ld)] : 0.0;
                                    acc[u][r] = 0.0;
                                }
                            }
                            int ra[TPW];
#pragma unroll
                            for (int u = 0; u < TPW; ++u) { const int x = (t0 + u) * 16 + fr; ra[u] = x < n ? x : n - 1; }
                            const double *zr = Zp + fk * ld, *cr = Cj + fk * GJ_MB + fr;
                            for (int q = 0; q < ksteps; ++q) {
                                const double b = cr[0];
#pragma unroll
                                for (int u = 0; u < TPW; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(zr[ra[u]], b, acc[u], 0, 0, 0);
                                zr += 4 * ld; cr += 4 * GJ_MB;
                            }
#pragma unroll
                            for (int u = 0; u < TPW; ++u) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int row = (t0 + u) * 16 + fk + 4 * r;
                                    if (fr < m && row < n) Pn[(size_t)fr * ld + row] = base[u][r] - acc[u][r];
                                }
                            }
                        }
                        if (tid < m) Pn[(size_t)tid * ld + n] = qs[ks[tid]];         // row n of column k = entry k of the rhs column
                    }
                    __syncthreads();
                    LZ_STAMP(3);                                              // gather + materialise
                    double *Zn = Zp + (size_t)Rcur * ld, *dn = dp + Rcur;
                    int acc_piv;
                    if constexpr (LZ_PANEL_STEPS && NT <= 512) {
                        // one barrier per step, the register kernel's panel: 2-4 % ahead of the two-phase form on the 512-thread plan (D = 340:
                        // 4.64 -> 4.74 M solves/s, D = 500: 608 -> 634 k), and no polled progress word; on the 1024-thread plan a barrier
                        // meets 16 waves and the form is 2x slower (D = 600: 12.5 -> 26.9 ms per 4096 patterns): there the two-phase form stays
                        acc_piv = lz_panel_stepwise<NT>(Pn, Zn, dn, tab, reinterpret_cast<int *>(red), m, ld, s_basic, myj, p.piv_eps, tid, nveto, LZ_STK);
                        (void)lz_fault; (void)basm;
                    } else if constexpr (NT <= 512) {
                        acc_piv = m <= 8 ? lz_panel_eliminate<NT, 8>(Pn, Zn, dn, tab, red, ks, m, ld, s_basic, myj, basm, p.piv_eps, tid, nveto, LZ_STK, lz_fault)
                                         : lz_panel_eliminate<NT, GJ_MB>(Pn, Zn, dn, tab, red, ks, m, ld, s_basic, myj, basm, p.piv_eps, tid, nveto, LZ_STK, lz_fault);
                        lz_fault = false;
                    } else if constexpr (NT <= LZ_TWO_PHASE_MAX_NT) {         // 128 VGPRs per thread: blocks of at most 8 pivots (lazy_plan), the narrow panel only
                        acc_piv = lz_panel_eliminate<NT, 8>(Pn, Zn, dn, tab, red, ks, m, ld, s_basic, myj, basm, p.piv_eps, tid, nveto, LZ_STK, lz_fault);
                        lz_fault = false;
                    } else {                                                  // 128 VGPRs per thread: the step-by-step form (two barriers per pivot, few registers)
                        acc_piv = gj_panel_eliminate<NT>(Pn, Zn, dn, tab, red, ks, m, ld, s_basic, p.piv_eps, tid);
                    }
                    LZ_STAMP(4);                                              // panel
                    // ---- after the panel: rhs column, replaced rows / columns, bookkeeping -----------------------------
                    // All LDS reads of this phase go out in one batch, with clamped indices instead of loops bounded by m: a load behind a
                    // store that may alias it waits for its own round trip (~130 cycles each, m of them in a row).
                    const double dmy = dn[lane < m ? lane : 0];
                    const unsigned accm = (unsigned)__ballot(lane < m && dmy != 0.0);
                    int kmy = ks[lane < m ? lane : 0];                       // lane j of every wave: pivot variable j and 1/d_j
                    double dmy_ = dmy;
                    // both are read with v_readlane inside `if (has_row)`: pin their computation HERE, where every lane is active — the
                    // compiler may otherwise sink the loads into the branch, and a wave whose last rows end below lane 15 (ld % 64 < 16)
                    // would read lanes that never executed them
                    asm volatile("" : "+v"(kmy), "+v"(dmy_));
                    if constexpr (NT <= 512) {
                        if (has_row) {
                            double zs[GJ_MB], zn[GJ_MB], pj[GJ_MB];
    #pragma unroll
                            for (int j = 0; j < GJ_MB; ++j) {
                                const int jc = j < m ? j : m - 1;
                                zs[j] = Zn[(size_t)jc * ld + tid]; zn[j] = Zn[(size_t)jc * ld + n]; pj[j] = Pn[(size_t)jc * ld + tid];
                            }
                            const double qn = myj >= 0 ? Pn[(size_t)myj * ld + n] : 0.0;
                            double a = qs[tid];
    #pragma unroll
                            for (int j = 0; j < GJ_MB; ++j) {
                                const double dj = lz_readlane(dmy_, j);          // 0 beyond m and for refused pivots
                                if ((accm >> j) & 1u) a = fma(-zs[j] * dj, zn[j], a);
                            }
                            qs[tid] = myj >= 0 ? qn : a;
                            if (tid < n) {
                                // entry (k_a, k_b), a < b, is taken from panel column a: every entry of the base has exactly one writer
    #pragma unroll
                                for (int j = 0; j < GJ_MB; ++j) {
                                    const int kj = __builtin_amdgcn_readlane(kmy, j);
                                    if (j < m && (myj < 0 || j <= myj)) T[lz_tri(tid, kj, ld)] = pj[j];
                                }
                            }
                        }
                    } else if (has_row) {                                     // 128 VGPRs per thread: no room for the batch (48 doubles)
                        if (myj >= 0) qs[tid] = Pn[(size_t)myj * ld + n];
                        else {
                            double a = qs[tid];
                            for (int s = 0; s < m; ++s)
                                if ((accm >> s) & 1u) a = fma(-Zn[(size_t)s * ld + tid] * dn[s], Zn[(size_t)s * ld + n], a);
                            qs[tid] = a;
                        }
                        if (tid < n) {
                            for (int j = 0; j < m; ++j)
                                if (myj < 0 || j <= myj) T[lz_tri(tid, ks[j], ld)] = Pn[(size_t)j * ld + tid];
                        }
                    }
                    // the rows / columns of the pivoted variables now hold ALL terms up to this block: what the pending terms (this
                    // block's included) say about them must never be applied again — zero it where it is stored (the entries a
                    // non-pivot row reads in the rhs update above are its own and entry n: not touched)
                    for (int e = tid; e < (Rcur + m) * GJ_MB; e += NT) {
                        const int t = e / GJ_MB, j = e % GJ_MB;
                        if (j < m) Zp[(size_t)t * ld + ks[j]] = 0.0;
                    }
                    if (tid < m) {
                        const int k = ks[tid];
                        if (dn[tid] == 0.0) s_blocked[k] = 1;
                        rowj[k] = -1;
                    }
                    Rcur += m;
                    b0 += m;
                    if (acc_piv) { npiv += (unsigned)acc_piv; progress = true; }
                    __threadfence_block();
                    __syncthreads();
                    LZ_STAMP(5);                                              // rhs, replaced rows / columns
                }
            }
            const double obj2 = qs[n];
            const double obj = sqrt(obj2 > 0.0 ? obj2 : 0.0);
            if (p.all_opt && tid == 0) p.all_opt[pat] = obj;
            if (obj < best_obj || (obj == best_obj && best_pat >= 0 && ref_index_less(pat, (unsigned long long)best_pat, p.rbit.gbit))) {
                second_obj = best_obj; second_pat = best_pat;
                best_obj = obj; best_pat = (long long)pat;
            } else if (obj < second_obj) { second_obj = obj; second_pat = (long long)pat; }
            // the workgroup's best pattern so far leaves its solution behind (as sweep_blk.hip's 256-thread kernel does): the host takes
            // the winner's from here instead of solving that pattern again — at n > 320 a 1.2 ms solve on the many-workgroup kernel
            if (p.best_sol && !code && obj2 < wg_best2) {
                wg_best2 = obj2;
                for (int i = tid; i < n; i += NT) p.best_sol[(size_t)blockIdx.x * p.node_ld + i] = s_basic[i] ? qs[i] : 0.0;
            }
            if (p.node_piv && code && tid == 0) {
                unsigned *o = p.node_piv + 3 * ((size_t)chain * p.chain_len + (size_t)(g - g0));
                o[0] = (unsigned)npiv; o[1] = 0; o[2] = 0;
            }
            __syncthreads();
        }
        if (p.node_sol) {
            for (int i = tid; i < n; i += NT)
                p.node_sol[(size_t)chain * p.node_ld + i] = s_basic[i] ? qs[i] : 0.0;
            if (tid == 0) p.node_obj2[chain] = qs[n];
        }
        if (NODE && T != Tscratch) {                                 // the node leaves a snapshot: apply what is still pending, add q and the flags
            if (Rcur > 0) {
                if (tid < 4) dp[Rcur + tid] = 0.0;                    // padding of the last k-step
                __syncthreads();
                lz_flush<NT>(T, ld, n, Zp, dp, Rcur, tid);
                Rcur = 0;
            }
            if (has_row) T[(size_t)ld * ld + tid] = qs[tid];
            uint8_t *fl = reinterpret_cast<uint8_t *>(T + (size_t)ld * ld + ld);
            for (int i = tid; i < n; i += NT) fl[i] = s_basic[i];
        }
        __syncthreads();
    }
#ifdef PARTLS_LZ_STAMPS
    if (tid == NT - 64 && blockIdx.x == 0)
        printf("lazy stamps, last wave: panel setup %llu p1 %llu p2 %llu bar %llu tail %llu\n", lz_cyc[8], lz_cyc[9], lz_cyc[11], lz_cyc[12], lz_cyc[13]);
    if (tid == 0 && blockIdx.x == 0)
        printf("lazy stamps (cycles, count): start %llu %llu | scan %llu %llu | flush %llu %llu | materialise %llu %llu | panel %llu %llu | post %llu %llu | flush k-loops %llu other %llu | panel: setup %llu p1 %llu bar %llu p2 %llu bar %llu tail %llu end %llu | pivots %llu mb %d rows %d\n",
               lz_cyc[0], lz_cnt[0], lz_cyc[1], lz_cnt[1], lz_cyc[2], lz_cnt[2], lz_cyc[3], lz_cnt[3], lz_cyc[4], lz_cnt[4], lz_cyc[5], lz_cnt[5], lz_cyc[6], lz_cyc[7], lz_cyc[8], lz_cyc[9], lz_cyc[10], lz_cyc[11], lz_cyc[12], lz_cyc[13], lz_cyc[14], npiv, mb, rows);
#endif
    if (tid == 0) {
        p.best_obj[blockIdx.x] = best_obj;
        p.best_pat[blockIdx.x] = best_pat;
        if (p.second_obj) { p.second_obj[blockIdx.x] = second_obj; p.second_pat[blockIdx.x] = second_pat; }
        if (p.n_pivots && npiv) atomicAdd(p.n_pivots, npiv);
        if (red[GJ_MB + 1] != 0.0) ++nunconv;
        if (p.n_unconverged && nunconv) atomicAdd(p.n_unconverged, nunconv);
        if (p.n_vetoes && nveto) atomicAdd(p.n_vetoes, nveto);           // entering pivots refused by the leave-one-out rule (512-thread plan)
    }
}

// LDS plan for leading dimension ld: pivots per block and rows of the pool (pending terms + panel).  false: no plan fits (never for n <= 1023)
bool lazy_plan(int ld, int *mb_out, int *rows_out, size_t *shmem_out)
{
    const size_t budget = (size_t)151 * 1024;
    const size_t fixed = (size_t)ld * 8 /* qs */ + 4 * (size_t)ld * 8 /* read-ahead rows */ + 3 * (size_t)ld /* flags, rowj */ + (2 * GJ_MB * GJ_MB + 8 * GJ_MB + 18) * 8 + 64;
    const size_t row = (size_t)ld * 8;
    if (budget < fixed + 6 * (row + 8 + GJ_MB * 8)) return false;
    int rows = (int)((budget - fixed) / (row + 8 + GJ_MB * 8));              // each row: z or panel column, 1/d, Cj
    if (rows > LZ_MAXR) rows = LZ_MAXR;
    int mb = rows / 3;
    if (mb > GJ_MB) mb = GJ_MB;
    if (ld > 512 && LZ_TWO_PHASE_MAX_NT >= 1024 && mb > 8) mb = 8;          // the 1024-thread plan compiles the 8-column panel only
    if (mb < 2) mb = 2;
    *mb_out = mb; *rows_out = rows;
    *shmem_out = ((size_t)(rows + 4) * ld + ld + rows + GJ_MB + (size_t)(rows + 4) * GJ_MB + 2 * GJ_MB * GJ_MB + 2 * GJ_MB + 18) * 8 + 3 * (size_t)ld + 16;
    return true;
}

template <int NT>
static hipError_t launch_lazy_nt(const SweepParams &p, int grid, int mb, int rows, size_t shmem, hipStream_t s)
{
    if (p.node_code) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_lazy_kernel<NT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((sweep_lazy_kernel<NT, true>), dim3(grid), dim3(NT), shmem, s, p, mb, rows);
    } else {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_lazy_kernel<NT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((sweep_lazy_kernel<NT, false>), dim3(grid), dim3(NT), shmem, s, p, mb, rows);
    }
    return hipGetLastError();
}

// one thread per tableau row in the panel: 512 threads up to n = 511 (half the waves at every barrier of the m panel steps), 1024 beyond
hipError_t launch_sweep_lazy(const SweepParams &p, int grid, hipStream_t s)
{
    int mb = 0, rows = 0;
    size_t shmem = 0;
    if (!lazy_plan(p.n + 1, &mb, &rows, &shmem)) return hipErrorInvalidValue;
    return p.n + 1 <= 512 ? launch_lazy_nt<512>(p, grid, mb, rows, shmem, s) : launch_lazy_nt<1024>(p, grid, mb, rows, shmem, s);
}

}  // namespace partls
