// sweep_two.hip — EXPERIMENTAL two-level sign-pattern sweep (opt-in: PARTLS_KERNEL=two; the product default is sweep_blk.hip).
// The blocked register-tableau machinery of sweep_blk.hip walks only the HIGH groups; the LOW groups (at most two: those whose
// variables all lie in tile columns 0-1) are enumerated on small register-resident tableaus.
//
// Replaces the loop body of fit(Opt), Opt.jl:87-90, like sweep_blk.hip (same subproblems, same KKT conditions, same
// tolerances), but changes which tableau pays for a sign flip.  In Gray order the lowest groups flip most often, and in
// sweep_blk.hip every flip costs ~|P_k| exchanges on the (n+1)^2 register tableau.  Here, for a block of 2^v consecutive
// Gray indices (v = p.low_groups <= 2; the HIGH part of the pattern is constant inside a block):
//   1. frozen big solve : the register tableau is brought to the optimum of the HIGH configuration with every LOW
//      variable held nonbasic (sign code 0) — ordinary blocked pivots, once per 2^v patterns;
//   2. low panel        : tile columns 0 and 1 of that tableau (the LOW variables' columns, all rows) are gathered into a
//      persistent LDS panel LP[slot][row position];
//   3. small solves     : for each of the 2^v LOW sign choices the subproblem restricted to the small set (slots = variables
//      0..31 + discovered ones) is a principal-pivoting problem on a 41 x 41 tableau whose entries are read from LP (rows of
//      the small set): the Schur complement of the big basis.  The 2^v tableaus are solved CONCURRENTLY, one per team of two
//      waves (lane = column, rows interleaved over the two waves in registers), all teams in lockstep with one LDS barrier
//      per pivot; a pivot there costs 41^2 FMAs instead of (n+1)^2;
//   4. verification     : the frozen variables' rhs under a small solution is q_r - sum_j LP[j][r] * (+-u_j) over the
//      small variables j whose status changed (u = small rhs, + entered / - left): an exact KKT check of the FULL
//      subproblem.  A frozen violator is "discovered": its column is gathered from the registers (the shared gather code
//      of the block loop, which then runs with a zero pivot count) and appended to LP, and the affected patterns are solved
//      again.  If more than ECAP variables are discovered in a block, its pending patterns are solved the classical way on
//      the register tableau (the low variables are simply unfrozen), so the result never depends on the heuristic split.
// The objective of a pattern is the corner of its small tableau (or of the register tableau in the classical fallback).
// Chain mode only (node mode stays on sweep_blk.hip).  v = 0 degenerates to the algorithm of sweep_blk.hip.
// Status: bit-compatible with sweep_blk.hip on all tests, 2.9x fewer big-tableau pivots, but 121.9 ms vs 97.6 ms
// on C3 — an unblocked small pivot is a ~1 k-cycle latency chain; variants and measurements in tools/experiments/.
#include "blk_common.h"

namespace partls {
namespace two {

using namespace blk;

static constexpr int TLV = 32;                 // small-set slots taken by tile columns 0 and 1
static constexpr int ECAP = 8;                 // discovered variables per block
static constexpr int NSMAX = TLV + ECAP;       // small variables; the rhs has index NSMAX

constexpr int nrows(int T) { return 16 * rstride(T) + 1; }          // panel row positions incl. the rhs row (odd)
constexpr int cw2(int T) { return nrows(T) + 32; }                  // block panel column: + 32 dummy slots (odd)
constexpr int cwl(int T) { return nrows(T); }                       // low panel column
static constexpr int NTEAM = 4;                // small solves in flight: one per pair of waves
// doubles after the low panel: rowbuf[2][NTEAM][64], ub[NTEAM][64], objb[NTEAM], smask[NSMAX] (u64), cmk[NTEAM] (u64);
// then ints: spos[NSMAX], sbflag[NSMAX], actf[2][NTEAM], failf[NTEAM], vflag[8]
constexpr int low_doubles() { return 3 * NTEAM * 64 + NTEAM + NSMAX + NTEAM + (2 * NSMAX + 3 * NTEAM + 8 + 1) / 2; }
constexpr int lds_doubles2(int T) { return 3 * MB * cw2(T) + 3 * (MB + 64) + NSMAX * cwl(T) + low_doubles(); }

__device__ __forceinline__ double uniform_f64(double x)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// Workgroup barrier for exchanges that go through LDS only: waits for this wave's LDS traffic, not for outstanding global /
// scratch stores (which __syncthreads() would, at a full memory round trip per lockstep step when a spill store is in flight).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ double readlane_f64(double x, int l)          // l must be wave-uniform
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long u)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

template <int T, int H>
__device__ __forceinline__ void sweep_body2(const SweepParams &p, double *lds)
{
    constexpr int CW = cw2(T), CWL = cwl(T);
    using L = Half<T, H, CW>;
    constexpr int RS = L::RS, RHSPOS = 16 * RS, NR = RHSPOS + 1;
    const int tid = threadIdx.x, t8 = tid & 255, a = t8 & 15, b = t8 >> 4, lane = tid & 63, wave = tid >> 6;
    const int n = p.n;
    const int nwords = (n + 63) >> 6;

    double *Pbase = lds;                                  // [2][MB][CW]
    double *Z = lds + 2 * MB * CW;                        // [MB][CW]
    double *U = Z + MB * CW;                              // [2][MB+64]
    double *Dinv = U + 2 * (MB + 64);                     // [MB+64]
    double *LP = Dinv + MB + 64;                          // [NSMAX][CWL]   low panel
    double *rowbuf = LP + NSMAX * CWL;                    // [2][NTEAM][64] pivot rows of the small tableaus (double-buffered)
    double *ub = rowbuf + 2 * NTEAM * 64;                 // [NTEAM][64]    signed solution change of each team's pattern
    double *objb = ub + NTEAM * 64;                       // [NTEAM]        objective^2 of each team's pattern
    unsigned long long *smask = reinterpret_cast<unsigned long long *>(objb + NTEAM);   // [NSMAX] group mask of a small slot
    unsigned long long *cmk = smask + NSMAX;              // [NTEAM]        small variables whose status changed
    int *spos = reinterpret_cast<int *>(cmk + NTEAM);     // [NSMAX] panel row position of a small slot
    int *sbflag = spos + NSMAX;                           // [NSMAX] is the slot's variable basic in the register tableau?
    int *actf = sbflag + NSMAX;                           // [2][NTEAM] team still pivoting?
    int *failf = actf + 2 * NTEAM;                        // [NTEAM] small solve hit the round cap
    int *vflag = failf + NTEAM;                           // [8] per wave: patterns with frozen violators
    unsigned long long *s_inf = reinterpret_cast<unsigned long long *>(lds + lds_doubles2(T));  // [2][8]
    unsigned long long *s_bas = s_inf + 16;                                                    // [2][8]

    for (int i = tid; i < lds_doubles2(T) + 32; i += THREADS) lds[i] = 0.0;            // padding rows are never gathered
    __syncthreads();
    if (tid < TLV) spos[tid] = (tid & 15) * RS + (tid >> 4);

    double S[L::CNT];
    double q = 0.0, corner = 0.0;
    const bool has_var = tid < n;
    const uint64_t vmask = has_var ? p.mask[tid] : 0ULL;
    bool basic = false, blocked = false;
    const int mypos = (tid & 15) * RS + (tid >> 4);       // panel row position of variable `tid` (tid < 16 T)
    const int rowc = tid / RS, rowrho = tid - rowc * RS;
    const bool idle_wave = __builtin_amdgcn_readfirstlane((tid & ~63) > RHSPOS ? 1 : 0) != 0;   // no panel row in this wave

    const int v = (T >= 3) ? (p.low_groups < 2 ? p.low_groups : 2) : 0;      // 2^v <= NTEAM patterns per block
    const int64_t bl = (int64_t)1 << v;
    const int ecap = p.low_ecap < ECAP ? (p.low_ecap < 0 ? 0 : p.low_ecap) : ECAP;
    const bool lowvar = (vmask & ((1ULL << v) - 1ULL)) != 0;
    bool insmall = false;

    double best_obj = __builtin_inf();
    long long best_pat = -1;
    unsigned long long npiv = 0, nunconv = 0, nsmall = 0;
    unsigned bc = 0, sc = 0, np = 0;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), team = wv >> 1, pr = wv & 1;

    const int64_t total = p.g_end - p.g_begin;
    const int64_t nchains = (total + p.chain_len - 1) / p.chain_len;

    STAMP_DECL
    unsigned long long st_steps = 0;
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

        for (int64_t gb = g0; gb < g1;) {
            // ---- a block of patterns with the same HIGH configuration ------------------------------------------------------
            int64_t ge;
            bool frozen;
            if (v > 0 && (gb & (bl - 1)) == 0 && gb + bl <= g1) { ge = gb + bl; frozen = true; }
            else { ge = (gb | (bl - 1)) + 1; if (ge > g1) ge = g1; frozen = false; }
            bool lphase = false;
            int nE = 0;
            insmall = tid < TLV;
            unsigned pend = (1u << (int)(ge - gb)) - 1u;          // patterns gb + t of the block that are not recorded yet
            int64_t g = gb;
            uint64_t pat = (uint64_t)g ^ ((uint64_t)g >> 1);
            int f = (frozen && lowvar) ? 0 : sign_of_var(vmask, pat);
            blocked = false;
            int ninf_best = n + 1, patience = 3, rounds = 0;
            bool progress = false;

            // publish the violator / basis masks of the 512 threads and reduce them to (count, tile mask, largest violator)
            int par = 0, count = 0, single_k = -1;
            unsigned tiles = 0;
            auto publish = [&](bool bad) {
                par = sc & 1;
                ++sc;
                const unsigned long long bb = __ballot(bad), bs = __ballot(basic);
                if (lane == 0 && wave < nwords) { s_inf[par * 8 + wave] = bb; s_bas[par * 8 + wave] = bs; }
                STAMP(9);
                __syncthreads();
                STAMP(10);
                count = 0; single_k = -1; tiles = 0;
#pragma unroll
                for (int w = 0; w < 5; ++w) {
                    unsigned long long ww = (w < nwords) ? s_inf[par * 8 + w] : 0ULL;
                    ww = uniform_u64(ww);
                    count += __popcll(ww);
                    if (ww) single_k = (w << 6) + 63 - __builtin_clzll(ww);
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub)
                        if ((ww >> (16 * sub)) & 0xFFFFull) tiles |= 1u << (4 * w + sub);
                }
                STAMP(0);
            };
            auto record = [&](double o2, uint64_t pt) {
                const double obj = sqrt(o2 > 0.0 ? o2 : 0.0);
                if (p.all_opt && tid == 0) p.all_opt[pt] = obj;
                if (obj < best_obj || (obj == best_obj && (long long)pt < best_pat)) { best_obj = obj; best_pat = (long long)pt; }
            };
            // classical mode: pattern g is done, go to the next pending one of the block (false: none left)
            auto next_pending = [&]() -> bool {
                pend &= ~(1u << (int)(g - gb));
                if (pend == 0) return false;
                g = gb + __builtin_ctz(pend);
                pat = (uint64_t)g ^ ((uint64_t)g >> 1);
                f = sign_of_var(vmask, pat);
                blocked = false; ninf_best = n + 1; patience = 3; rounds = 0;
                return true;
            };

            for (;;) {
                bool all = true;
                if (!lphase) {
                    // ---- KKT scan of the big rhs column (registers) ---------------------------------------------------------
                    if (progress) blocked = false;
                    progress = false;
                    bool bad = false;
                    if (has_var) {
                        const double fq = (f > 0) ? q : ((f < 0) ? -q : 0.0);
                        if (basic) bad = (f == 0) || (fq < -p.tol);
                        else bad = (fq > p.tol) && !blocked;
                    }
                    publish(bad);
                    if (count == 0) {
                        if (frozen) {
                            // ---- the HIGH configuration is solved: gather the low panel and start the small solves -----------
#ifndef PARTLS_X1
                            if constexpr (T >= 3) {
                                gather_tile<T, H, 0, CWL>(S, LP, a, b, 0xFFFFu);
                                gather_tile<T, H, 1, CWL>(S, LP + 16 * CWL, a, b, 0xFFFFu);
                            }
#endif
                            if (tid < TLV) { LP[tid * CWL + RHSPOS] = q; smask[tid] = vmask; sbflag[tid] = basic ? 1 : 0; }
                            STAMP(15);
                            lphase = true;
                            continue;
                        }
                        record(corner, pat);                              // pattern g is solved on the register tableau
                        if (!next_pending()) break;
                        continue;
                    }
                    if (count < ninf_best) { ninf_best = count; patience = 3; }
                    else if (patience > 0) --patience;
                    else all = false;                                     // backup rule: only the largest violator
                    if (++rounds > p.max_rounds) {
                        ++nunconv;
                        if (frozen) { frozen = false; f = sign_of_var(vmask, pat); rounds = 0; continue; }
                        record(corner, pat);                              // given up (reported through n_unconverged)
                        if (!next_pending()) break;
                        continue;
                    }
                    if (!all) tiles = 1u << (single_k >> 4);
                } else {
                    // ---- low phase: the pending patterns of the block are solved CONCURRENTLY, one per team (pair of waves), on
                    //      register-resident small tableaus; all teams step in lockstep with one barrier per pivot ------------------
                    bool finished = false, overflow = false;
#ifndef PARTLS_X3
                    for (;;) {
                        const int ns = TLV + nE;
                        STAMP(7);
                        __syncthreads();                                  // low panel / slot tables complete
                        const bool active = (pend >> team) & 1u;
                        const uint64_t patt = (uint64_t)(gb + team) ^ ((uint64_t)(gb + team) >> 1);
                        // build: lane = column (NSMAX = rhs); this wave holds rows 2c + pr (c < 20) in R, and its own copies of the
                        // rhs row (Rr, lane NSMAX = objective^2) and of the diagonal (Dg).
                        // R and Rr are stored RAW: the true entry is R[c] * cs with the per-column scale cs (lane = column), so that
                        // rescaling the pivot column is one multiply of cs instead of one per row
                        double R[20], Rr, Dg, cs = 1.0;
                        {
                            const bool colv = lane < ns || lane == NSMAX;
                            const int coff = (lane == NSMAX) ? 0 : (lane < ns ? lane : 0) * CWL;
#pragma unroll
                            for (int c = 0; c < 20; ++c) {
                                const int i = 2 * c + pr;
                                const int ii = i < ns ? i : 0;
                                const double x = (lane == NSMAX) ? LP[ii * CWL + RHSPOS] : LP[coff + spos[ii]];
                                R[c] = (i < ns && colv) ? x : 0.0;
                            }
                            const int ll = lane < ns ? lane : 0;
                            const double x = LP[ll * CWL + RHSPOS];
                            Rr = (lane < ns) ? x : ((lane == NSMAX) ? corner : 0.0);
                            Dg = LP[ll * CWL + spos[ll]];
                        }
                        int fi = 0;
                        bool sb0 = false;
                        if (lane < ns) { fi = sign_of_var(smask[lane], patt); sb0 = sbflag[lane] != 0; }
                        const unsigned long long sbase = uniform_u64(__ballot(sb0));
                        unsigned long long sbasic = sbase, sblocked = 0ULL, todo = 0ULL;
                        int sbest = ns + 1, spat = 3, srounds = 0;
                        bool sprog = false, sfail = false, conv = !active;
                        STAMP(12);
                        for (;;) {
                            int k = -1;
                            if (!conv) {
                                if (todo == 0ULL) {                               // new round: KKT scan of the small rhs (registers)
                                    if (sprog) sblocked = 0ULL;
                                    sprog = false;
                                    bool sbad = false;
                                    if (lane < ns) {
                                        const double qs = Rr * cs;
                                        const double fq = (fi > 0) ? qs : ((fi < 0) ? -qs : 0.0);
                                        if ((sbasic >> lane) & 1ULL) sbad = (fi == 0) || (fq < -p.tol);
                                        else sbad = (fq > p.tol) && !((sblocked >> lane) & 1ULL);
                                    }
                                    todo = uniform_u64(__ballot(sbad));
                                    const int cnt = __popcll(todo);
                                    if (cnt == 0) conv = true;
                                    else {
                                        if (cnt < sbest) { sbest = cnt; spat = 3; }
                                        else if (spat > 0) --spat;
                                        else todo = 1ULL << (63 - __builtin_clzll(todo));
                                        if (++srounds > p.max_rounds) { sfail = true; conv = true; }
                                    }
                                }
                                if (!conv) { k = __builtin_ctzll(todo); todo &= todo - 1; }
                            }
                            STAMP(16);
                            const int slot = (int)(np & 1u) * NTEAM + team;
                            const int abase = (int)(np & 1u) * NTEAM;
                            ++np;
                            if (k >= 0 && pr == (k & 1)) {                        // the wave that owns row k publishes it (raw)
                                const int co = k >> 1;
                                double rowv = R[0];
#define PARTLS_SEL(c) rowv = (co == c) ? R[c] : rowv;
                                PARTLS_SEL(1) PARTLS_SEL(2) PARTLS_SEL(3) PARTLS_SEL(4) PARTLS_SEL(5) PARTLS_SEL(6) PARTLS_SEL(7)
                                PARTLS_SEL(8) PARTLS_SEL(9) PARTLS_SEL(10) PARTLS_SEL(11) PARTLS_SEL(12) PARTLS_SEL(13)
                                PARTLS_SEL(14) PARTLS_SEL(15) PARTLS_SEL(16) PARTLS_SEL(17) PARTLS_SEL(18) PARTLS_SEL(19)
#undef PARTLS_SEL
                                rowbuf[slot * 64 + lane] = rowv;
                            }
                            if (lane == 0 && pr == 0) actf[slot] = conv ? 0 : 1;
                            STAMP(17);
                            lds_barrier();
                            STAMP(18);
                            const int any = __builtin_amdgcn_readfirstlane(actf[abase] | actf[abase + 1] | actf[abase + 2] | actf[abase + 3]);
                            if (k >= 0) {
                                const double d = readlane_f64(Dg, k);
                                const bool colk = lane == k;
                                const double raw = rowbuf[slot * 64 + lane];
                                if (!((sbasic >> k) & 1ULL) && !(d > p.piv_eps)) sblocked |= 1ULL << k;      // dependent column
                                else {
                                    const double inv = fast_rcp(d), ainv = fabs(inv);
                                    const double r = colk ? d : raw * cs;        // true pivot row
                                    // multipliers: row i gets -r_i / d; the pivot row itself (read back from lane k) gets 1/|d| - 1,
                                    // which turns it into r / |d|
                                    const double rs = colk ? ainv - 1.0 : -r * inv;
                                    const double rz = colk ? 0.0 : raw;
                                    // row 2c + pr needs multiplier lane 2c + pr: the odd wave shifts the vector down one lane so that both
                                    // parities read lane 2c with an IMMEDIATE select (an SGPR select costs a reload + hazard stall per row)
                                    const double rsh = __shfl_down(rs, 1);
                                    const double rsx = pr ? rsh : rs;
#pragma unroll
                                    for (int c = 0; c < 20; ++c) R[c] = fma(readlane_f64(rsx, 2 * c), rz, R[c]);
                                    Rr = fma(readlane_f64(rs, NSMAX), rz, Rr);
                                    Dg = colk ? -inv : fma(rs, r, Dg);          // lanes != k: rs = -r / d
                                    cs = colk ? cs * ainv : cs;                   // the pivot column becomes column / |d|
                                    sbasic ^= 1ULL << k;
                                    sprog = true;
                                    ++nsmall;
                                }
                            }
                            STAMP(19);
                            ++st_steps;
                            if (!any) break;
                        }
                        STAMP(13);
                        // ---- verification: every thread checks its (frozen) variable under each pending pattern's small solution -----
                        if (pr == 0) {
                            const unsigned long long chm = sbasic ^ sbase;
                            double uv = 0.0;
                            if (lane < ns && ((chm >> lane) & 1ULL)) uv = ((sbasic >> lane) & 1ULL) ? Rr * cs : -Rr * cs;
                            ub[lane * NTEAM + team] = uv;                      // [small variable][team]
                            const double o2 = readlane_f64(Rr, NSMAX);
                            if (lane == 0) { cmk[team] = chm; objb[team] = o2; failf[team] = sfail ? 1 : 0; }
                        }
                        __syncthreads();
                        // one pass over the union of the changed small variables updates all four candidate rhs values
                        double qn[NTEAM] = {q, q, q, q};
                        {
                            unsigned long long ch = uniform_u64(cmk[0] | cmk[1] | cmk[2] | cmk[3]);
                            while (ch) {
#pragma unroll
                                for (int z = 0; z < 2; ++z) {
                                    const int j = ch ? __builtin_ctzll(ch) : 0;
                                    const double lp = ch ? LP[j * CWL + mypos] : 0.0;
                                    ch &= ch - 1;
#pragma unroll
                                    for (int t = 0; t < NTEAM; ++t) qn[t] = fma(-lp, ub[j * NTEAM + t], qn[t]);
                                }
                            }
                        }
                        bool badU = false;
                        unsigned myviol = 0;
#pragma unroll
                        for (int t = 0; t < NTEAM; ++t) {
                            const uint64_t pt = (uint64_t)(gb + t) ^ ((uint64_t)(gb + t) >> 1);
                            const int ft = sign_of_var(vmask, pt);
                            bool bad = false;
                            if (has_var && !insmall && ((pend >> t) & 1u)) {
                                const double fq = (ft > 0) ? qn[t] : -qn[t];
                                bad = basic ? (fq < -p.tol) : (fq > p.tol);
                            }
                            if (__ballot(bad)) myviol |= 1u << t;
                            badU = badU || bad;
                        }
                        if (lane == 0) vflag[wv] = (int)myviol;
                        STAMP(14);
                        publish(badU);
                        unsigned vb = 0, fb = 0;
#pragma unroll
                        for (int w = 0; w < 8; ++w) vb |= (unsigned)vflag[w];
#pragma unroll
                        for (int t = 0; t < NTEAM; ++t) fb |= failf[t] ? (1u << t) : 0u;
                        vb = (unsigned)__builtin_amdgcn_readfirstlane((int)vb);
                        fb = (unsigned)__builtin_amdgcn_readfirstlane((int)fb) & pend;
                        for (int t = 0; t < NTEAM; ++t) {
                            if (!((pend >> t) & 1u) || ((vb | fb) >> t) & 1u) continue;
                            record(objb[t], (uint64_t)(gb + t) ^ ((uint64_t)(gb + t) >> 1));
                            pend &= ~(1u << t);
                        }
                        if (pend == 0) finished = true;
                        else if (fb != 0 || nE + count > ecap) overflow = true;
                        break;
                    }
#else
                    finished = true;
#endif
                    if (finished) break;
                    if (overflow) {
                        // ---- too many unstable frozen variables (or a small solve gave up): the pending patterns of the block
                        //      are solved classically on the register tableau -------------------------------------------------------
                        frozen = false; lphase = false;
                        g = gb + __builtin_ctz(pend);
                        pat = (uint64_t)g ^ ((uint64_t)g >> 1);
                        f = sign_of_var(vmask, pat);
                        blocked = false; ninf_best = n + 1; patience = 3; rounds = 0; progress = false;
                        continue;
                    }
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
                        PARTLS_CASES(PARTLS_G)
#undef PARTLS_G
                        const bool mine = tid < 16 * T && (tid >> 4) == kappa && ((pm >> (tid & 15)) & 1u);
                        const int myslot = __builtin_popcount(pm & ((1u << (tid & 15)) - 1u));
                        if (mine) P[myslot * CW + RHSPOS] = q;
                        int myj = -1;
                        bool my_basic = false;
                        if (rowrho == kappa && rowc < 16 && ((pm >> rowc) & 1u)) {
                            myj = __builtin_popcount(pm & ((1u << rowc) - 1u));
                            my_basic = (basm >> rowc) & 1u;
                        }
                        STAMP(8);
                        __syncthreads();
                        STAMP(1);
                        // the phase flag is laundered through an empty asm so that the optimiser cannot unswitch this loop on it
                        // (two copies of the block body do not survive register allocation)
                        int lp = __builtin_amdgcn_readfirstlane(lphase ? 1 : 0);
                        asm volatile("" : "+s"(lp));
#ifndef PARTLS_X2
                        // ---- discovery (low phase): the gathered columns join the low panel.  The rest of the block body runs
                        // on the scratch buffers only (update and scatter see m = 0): no extra control-flow edges in this loop,
                        // which the register allocator does not survive.
                        if (lp) {
                            const int slot0 = TLV + nE;
                            if (tid < NR)
                                for (int j = 0; j < m; ++j) LP[(slot0 + j) * CWL + tid] = P[j * CW + tid];
                            if (mine) { spos[slot0 + myslot] = mypos; smask[slot0 + myslot] = vmask; sbflag[slot0 + myslot] = basic ? 1 : 0; insmall = true; }
                            nE += m;
                        }
#endif
                        const int mu = lp ? 0 : m;                       // pivots applied to the register tableau
                        const unsigned pmscat = lp ? 0u : pm;           // nothing is scattered in discovery mode
                        // ---- 2. panel elimination -----------------------------------------------------------------------
                        bool blk_ok = false;
                        switch (m) {
                            case 1: blk_ok = panel_block<1, CW>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid < NR ? tid : NR + (tid & 31), idle_wave); break;
                            case 2: blk_ok = panel_block<2, CW>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid < NR ? tid : NR + (tid & 31), idle_wave); break;
                            case 3: blk_ok = panel_block<3, CW>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid < NR ? tid : NR + (tid & 31), idle_wave); break;
                            case 4: blk_ok = panel_block<4, CW>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid < NR ? tid : NR + (tid & 31), idle_wave); break;
                            case 5: blk_ok = panel_block<5, CW>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid < NR ? tid : NR + (tid & 31), idle_wave); break;
                            case 6: blk_ok = panel_block<6, CW>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid < NR ? tid : NR + (tid & 31), idle_wave); break;
                            case 7: blk_ok = panel_block<7, CW>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid < NR ? tid : NR + (tid & 31), idle_wave); break;
                            default: blk_ok = panel_block<8, CW>(P, Z, U, Dinv, myj, my_basic, p.piv_eps, tid, tid < NR ? tid : NR + (tid & 31), idle_wave); break;
                        }
                        STAMP(11);
                        __syncthreads();
                        STAMP(2);
                        // ---- 3. fused rank-m update of the register tableau ------------------------------------------------
#pragma unroll PARTLS_UPD_UNROLL
                        for (int s = 0; s < mu; ++s) {
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
#define PARTLS_F(i) if constexpr (i < T) { if (__builtin_expect(kappa == i, 0)) scatter_tile<T, H, i, CW>(S, P, a, b, pmscat); }
                        PARTLS_CASES(PARTLS_F)
#undef PARTLS_F
                        if (mine && !lp) {
                            q = P[myslot * CW + RHSPOS];
                            if (Dinv[myslot] != 0.0) basic = !basic;
                            else blocked = true;
                        }
                        progress = progress || blk_ok;
                        npiv += (unsigned)mu;
                        STAMP(4);
                    }
                    tiles &= tiles - 1;
                }
            }
            gb = ge;
        }
    }
    STAMP_FLUSH;
#ifdef PARTLS_STAMPS
    if (tid == PARTLS_STAMP_TID && blockIdx.x == 0 && p.scratch) p.scratch[24] = (double)st_steps;
#endif
    (void)st_steps;
    if (tid == 0) {
        p.best_obj[blockIdx.x] = best_obj;
        p.best_pat[blockIdx.x] = best_pat;
        if (p.n_pivots && npiv) atomicAdd(p.n_pivots, npiv);
        if (p.n_small_pivots && nsmall) atomicAdd(p.n_small_pivots, nsmall);
        if (p.n_unconverged && nunconv) atomicAdd(p.n_unconverged, nunconv);
    }
}

template <int T>
__global__ __launch_bounds__(THREADS, 2) void sweep_two_kernel(SweepParams p)
{
    extern __shared__ double lds[];
    const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    if (half == 0) sweep_body2<T, 0>(p, lds);
    else sweep_body2<T, 1>(p, lds);
}

}  // namespace two

template <int T>
static hipError_t launch_two_T(const SweepParams &p, int grid, hipStream_t s)
{
    const size_t shmem = (size_t)two::lds_doubles2(T) * sizeof(double) + 32 * sizeof(unsigned long long);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&two::sweep_two_kernel<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(two::sweep_two_kernel<T>, dim3(grid), dim3(blk::THREADS), shmem, s, p);
    return hipGetLastError();
}

hipError_t launch_sweep_two(const SweepParams &p, int T, int grid, hipStream_t s)
{
    switch (T) {
#ifdef PARTLS_ONLY_T                       // compile-time experiments: one instantiation only
        case PARTLS_ONLY_T: return launch_two_T<PARTLS_ONLY_T>(p, grid, s);
#else
#define PARTLS_L(i) case i + 1: return launch_two_T<i + 1>(p, grid, s);
        PARTLS_CASES(PARTLS_L)
#undef PARTLS_L
#endif
        default: return hipErrorInvalidValue;
    }
}

int sweep_two_small_vars() { return two::TLV; }

}  // namespace partls
