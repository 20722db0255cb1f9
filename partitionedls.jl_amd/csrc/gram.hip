// gram.hip — the one dense contraction of the path: G_aug = Z'Z with Z = [X 1 y]  (N x (M+2)), fp64 MFMA.
//
// Replaces the data passes that every nonneg_lsq / norm call of the reference repeats per pattern
// (Opt.jl:88-90: bmatrix copy, NNLS workspace copy, Householder sweeps, objective GEMM): X is read ONCE here and all
// 2^K' subproblems then work on the (M+2)^2 Gram block.  G = Xo'Xo is G_aug[0..M, 0..M],
// c = Xo'y is column M+1, yy = y'y the corner; the ones column and y are virtual (never materialised).
//
// Kernel: 128x128 output macro-tile per workgroup of 512 threads (8 waves in a 2 x 4 grid, each 64 x 32 = 8
// v_mfma_f64_16x16x4_f64 accumulators; two workgroups per CU = 4 waves per SIMD), samples split into row chunks, each chunk staged
// through a double-buffered LDS image in 16-sample panels ([col][17] layout: conflict-free fragment reads, see GLD), one barrier per
// panel, the staging of the next panels interleaved INTO the MFMA stream.  Only tile pairs I <= J (SYRK).
// What bounds it (measured, round 2): not memory — with every column aliased onto the same few pages (all hits) the time is the same
// — but MFMA issue: one wave issues a v_mfma_f64_16x16x4_f64 at most every ~140 cycles while the pipe is busy 64, so the pipe is
// full only while at least two waves of a SIMD are inside their MFMA streams (tools/ubench/mfma_f64_peak.hip: 35 TFLOP/s with one
// wave per SIMD, 77.5 with two or more).  Hence: 4 waves per SIMD, no MFMA-free load/store phase, 64 x 32 per wave.  History at C4
// (N = 1M, D = 512): 64 x 64 tiles, 4 workgroups of 4 waves, separate phases 5.56 ms; 128 x 128, 4 waves, separate phases 5.9-6.4;
// + LDS double buffer 5.77; + staging interleaved 5.40; 8 waves per workgroup 5.16 ms (MFMA pipe 76 % busy).
// Partial tiles go to a tile-packed per-slice slab and are summed in a fixed order by gram_reduce (bitwise reproducible; no float
// atomics).  Roofline: fp64 MFMA bound — 2*N*(M+2)^2/2 flops over 8*N*M bytes (intensity ~ M/8 flop/B).
#include "common.h"
#include <cstdlib>

namespace partls {

typedef double double4_t __attribute__((ext_vector_type(4)));

static constexpr int GT = 128;                 // macro tile edge (features)
static constexpr int NB = GT / 16;             // 16-wide blocks per tile edge
#ifndef PARTLS_GRAM_GK
#define PARTLS_GRAM_GK 16
#endif
static constexpr int GK = PARTLS_GRAM_GK;      // samples per LDS panel
// Padded panel stride (doubles).  The fragment reads (16 columns x 2 or 4 k-values per wave instruction) must be conflict-free under
// BOTH bankings the LDS applies: ds_read_b64 (32-lane groups, 64 banks) and ds_read2_b64 (16-lane groups, 32 banks) — the compiler
// merges the reads of two k-steps of a column into ds_read2_b64.  17 doubles = 34 dwords: banks 34c mod 64 are the 16 distinct
// multiples of 2 {0,34,4,38,...} (+2 for the second k-value lands on the complementary set), and 2c mod 32 are distinct.
// (GK + 2 = 18 or 34 is conflict-free only for plain ds_read_b64: the merged reads hit every bank twice, 3.6 extra LDS cycles per
// instruction on the counters.)
static constexpr int GLD = 17;
static_assert(GK == 16, "panel depth: the stride above and the slot schedule of gram_body are written for 16");
static constexpr int NWV = 8;                  // waves per workgroup (512 threads): two workgroups per CU = 4 waves per SIMD
static constexpr int CPW = 64 / GK;            // columns one wave-wide load instruction covers (GK lanes per column)
static constexpr int NL = GT / (NWV * CPW);    // loads per thread, panel and matrix

// Tiles cover the FEATURES only (nt = ceil(M / 128) tile columns, pairs I <= J).  The two virtual columns of Z = [X 1 y] — the
// ones column and y — ride on the DIAGONAL tiles: tile (I, I) carries eight extra 16 x 16 accumulators (one per row block) whose
// B fragment is [1, y, 0, ...], which yields X_I' 1 and X_I' y for its 128 columns, and the workgroups of tile (0, 0) also
// accumulate the 2 x 2 corner (N, 1'y, y'y) as plain sums.
//
// Loading one feature element is split in two so that the global loads stay in flight under the MFMAs:
//   z_load  : unconditional load from a clamped address (issued one panel ahead, result untouched),
//   z_value : zero for the padding columns of the last tile (EDGE) and for the row tail, applied at the LDS store.
// Addresses: wave-uniform 64-bit base (tile column block, panel row: SGPRs) + ONE per-lane 32-bit element offset shared by all the
// loads of a panel (csub * ldX + row in panel), so the 2 x NL loads in flight cost no address registers (element offsets, not bytes: 3 * ldX + 15 must fit 32 bits; the host refuses ldX >= 2^30).
template <bool EDGE>
__device__ __forceinline__ double z_load(const double *__restrict__ Xrow, int64_t ldX, int M, int colu, int csub, unsigned lane_off, unsigned rowoff)
{
    if constexpr (!EDGE) {
        return (Xrow + (int64_t)colu * ldX)[lane_off];
    } else {                                                     // clamp the column to M - 1 (value discarded by z_value)
        const int cb = colu < M - 1 ? colu : M - 1;
        const int cl = colu + csub < M - 1 ? colu + csub : M - 1;
        return (Xrow + (int64_t)cb * ldX)[(unsigned)(cl - cb) * (unsigned)ldX + rowoff];
    }
}
template <bool EDGE>
__device__ __forceinline__ double z_value(double x, bool rv, int M, int col)
{
    double v = x;
    if constexpr (EDGE) v = col < M ? x : 0.0;
    return rv ? v : 0.0;
}

// Diagonal tiles compute the upper triangle only, at 16 x 16 granularity: 36 sub-tiles (r <= c) + the 8 virtual ones (r, V)
// = 44 MFMAs per k-step, 6 or 5 per wave (an off-diagonal tile issues 8 per wave).  Item t of wave W: row block DR[W][t], column
// block DC[W][t] (NB = V).  On a diagonal tile the A fragment of block r and the B fragment of block c are the same LDS words
// (A[row = lane & 15][k = lane >> 4], B[k][col = lane & 15], both from sA), so one fragment per block serves both sides.
static constexpr int DNI = 6;                   // items per wave: 6 for waves 0..3, 5 for waves 4..7 (DCNT)
__device__ constexpr int DCNT[NWV] = {6, 6, 6, 6, 5, 5, 5, 5};
__device__ constexpr int DR[NWV][DNI] = {{0, 0, 0, 0, 0, 0}, {0, 0, 1, 1, 1, 1}, {1, 1, 1, 2, 2, 2}, {2, 2, 2, 3, 3, 3},
                                         {3, 3, 4, 4, 4, 4}, {4, 5, 5, 5, 6, 6}, {6, 7, 0, 1, 2, 2}, {3, 4, 5, 6, 7, 7}};
__device__ constexpr int DC[NWV][DNI] = {{0, 1, 2, 3, 4, 5}, {6, 7, 1, 2, 3, 4}, {5, 6, 7, 2, 3, 4}, {5, 6, 7, 3, 4, 5},
                                         {6, 7, 4, 5, 6, 6}, {7, 5, 6, 7, 6, 6}, {7, 7, 8, 8, 8, 8}, {8, 8, 8, 8, 8, 8}};
constexpr unsigned diag_blocks(int W)
{
    unsigned m = 0;
    for (int t = 0; t < DCNT[W]; ++t) m |= (1u << DR[W][t]) | (1u << DC[W][t]);
    return m;
}

// Both panel routines call `slot(g)`, g = 0..3 (diagonal) / 0..7 (off-diagonal), at evenly spaced points of their MFMA stream: the
// caller hangs one piece of the staging work for the NEXT panels on each call (see gram_body).  Why: one wave can issue a
// v_mfma_f64_16x16x4_f64 only every ~140 cycles although the pipe is busy for 64 (tools/ubench/mfma_f64_peak.hip: 35 TFLOP/s with
// one wave per SIMD, 77.5 with two or more), so the pipe is full only while BOTH waves of a SIMD are inside their MFMA streams —
// a wave that leaves the stream for a separate load/store phase halves the rate of its partner.  The ~70 idle issue cycles
// between a wave's own MFMAs are where the staging instructions go.
template <int W, class F>
__device__ __forceinline__ void diag_panel(const double *sA, const double *sV, int fr, int fk, double4_t (&acc)[8], F &&slot)
{
    constexpr unsigned BM = diag_blocks(W);
    auto load = [&](double (&f)[NB + 1], int ks) {
#pragma unroll
        for (int r = 0; r < NB; ++r)
            if ((BM >> r) & 1u) f[r] = sA[(r * 16 + fr) * GLD + ks * 4 + fk];
        if ((BM >> NB) & 1u) f[NB] = fr < 2 ? sV[fr * GK + ks * 4 + fk] : 0.0;
    };
    double f_n[NB + 1] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    load(f_n, 0);
#pragma unroll
    for (int ks = 0; ks < GK / 4; ++ks) {
        double f[NB + 1];
#pragma unroll
        for (int r = 0; r <= NB; ++r) f[r] = f_n[r];
        if (ks + 1 < GK / 4) load(f_n, ks + 1);
#pragma unroll
        for (int t = 0; t < DCNT[W]; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[DR[W][t]], f[DC[W][t]], acc[t], 0, 0, 0);
            if (t == 2) slot(ks);
        }
    }
}

// off-diagonal tile: wave (wr, wc) of the 2 x 4 grid owns rows wr*64.. of A and columns wc*32.. of B: 4 + 2 fragments, 8 MFMAs per
// k-step; the fragments of k-step ks+1 are read from LDS while the MFMAs of k-step ks are issued
template <class F>
__device__ __forceinline__ void full_panel(const double *sA, const double *sB, int wr, int wc, int fr, int fk, double4_t (&acc)[8], F &&slot)
{
    // a[i] is dead after row i's two MFMAs, so the A fragment of k-step ks+1 is read into the same register right behind them;
    // only the B fragments (live until the last row) need a second set
    double a[4], b_n[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = sA[((wr * 4 + i) * 16 + fr) * GLD + fk];
#pragma unroll
    for (int j = 0; j < 2; ++j) b_n[j] = sB[((wc * 2 + j) * 16 + fr) * GLD + fk];
#pragma unroll
    for (int ks = 0; ks < GK / 4; ++ks) {
        double b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = b_n[j];
        if (ks + 1 < GK / 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j) b_n[j] = sB[((wc * 2 + j) * 16 + fr) * GLD + (ks + 1) * 4 + fk];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
            if (ks + 1 < GK / 4) a[i] = sA[((wr * 4 + i) * 16 + fr) * GLD + (ks + 1) * 4 + fk];
            if (i & 1) slot(ks * 2 + (i >> 1));
        }
    }
}

// Work decomposition (XCD-aware): the samples are cut into row chunks of `chunk_rows`.  Workgroups are dealt round-robin
// over the 8 XCDs, so `blockIdx.x & 7` labels the XCD group; group x, slice s walks the chunks x + 8*(s + S*j), and the np
// tile-pair workgroups of one (x, s) walk the SAME chunk sequence, so panels are re-read from that XCD's L2 by the other tile
// pairs (speed only — correctness does not depend on placement).  Each workgroup accumulates all its chunks in registers and
// writes one partial tile to its slot of slice (x*S + s): slice layout = np tiles of 128 x 128, then the two virtual columns
// [2][ldv] (ones, y), whose entries ldv-2.. hold the 2 x 2 corner.  Inside, the next panel is prefetched into registers
// (independent, unconditional global loads) while the MFMAs of the current one run from LDS.
__host__ __device__ constexpr size_t slice_doubles(int np, int nt) { return (size_t)np * GT * GT + 2 * ((size_t)nt * GT + 2); }

template <bool EA, bool EB, bool DIAG>
__device__ __forceinline__ void gram_body(const double *__restrict__ X, int64_t N, int M, int64_t ldX,
                                          const double *__restrict__ y, double *__restrict__ slab, int np, int nt, int pair,
                                          int chunk_rows, int S, int I, int J, int xg, int sl, double *sA, double *sB, double *sV)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double4_t acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};

    const int smp = lane & (GK - 1), csub = lane / GK;  // loader: GK lanes cover one column's GK contiguous samples
    const int fr = lane & 15, fk = lane >> 4;          // MFMA fragment coordinates
    const int wr = wave >> 2, wc = wave & 3;
    const int kb_per_chunk = chunk_rows / GK;
    const int64_t nchunks = (N + chunk_rows - 1) / chunk_rows;
    const int64_t cstride = (int64_t)8 * S;
    const bool vloader = DIAG && tid < GK;             // threads 0..GK-1 of a diagonal tile also stage [valid, y] of the panel
    const bool corner = DIAG && I == 0 && tid < GK;    // ... and, for tile (0, 0), accumulate the 2 x 2 corner
    double c11 = 0.0, c1y = 0.0, cyy = 0.0;

    static_assert(GK == 16 && NL == 4, "the slot schedule below is written for 16-sample panels and 8 waves");
    // Pipeline: in phase p the MFMAs run from LDS buffer p & 1 while, slot by slot, the registers holding panel p + 1 (loaded in
    // phase p - 1) are stored to the other buffer and immediately refilled by the loads of panel p + 2.  One barrier per phase
    // separates the reads of a buffer from its refill and the stores from their reads.  Everything inside a phase is branch-free: a
    // panel that does not exist is "loaded" from a clamped valid address with its row predicate false (zeros reach the LDS).
    double va[NL], vb[DIAG ? 1 : NL], vy = 0.0;        // raw values of panel p + 1 (+ y for the diagonal tiles)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    struct Panel { const double *Xrow; const double *yrow; unsigned lane_off, rowoff; bool rv; };
    int64_t chunk = xg + (int64_t)8 * sl;
    int kb = 0;
    bool live = chunk < nchunks;                        // does the panel (chunk, kb) exist?
    auto panel_here = [&]() {                           // addresses + row predicate of panel (chunk, kb); clamped when !live
        Panel P;
        const int64_t ch = live ? chunk : 0;
        const int64_t r1 = ((ch + 1) * chunk_rows < N) ? (ch + 1) * chunk_rows : N;
        const int64_t row0 = live ? ch * chunk_rows + (int64_t)kb * GK : 0;   // wave-uniform; row0 < r1 whenever live
        const int left = live ? (int)(r1 - row0 < GK ? r1 - row0 : GK) : 0;
        P.rv = smp < left;
        P.rowoff = (unsigned)(P.rv ? smp : (left > 0 ? left - 1 : 0));
        P.lane_off = (unsigned)csub * (unsigned)ldX + P.rowoff;
        P.Xrow = X + row0; P.yrow = y + row0;
        return P;
    };
    auto advance = [&]() {                              // step (chunk, kb) to this workgroup's next panel
        if (++kb == kb_per_chunk) { kb = 0; chunk += cstride; }
        live = live && chunk < nchunks && (chunk * chunk_rows + (int64_t)kb * GK) < N;
    };
    auto load_slot = [&](const Panel &P, int g) {       // slots 0..3: A columns, 4..7: B columns (off-diagonal tiles only)
        const int i = g & (NL - 1);
        const int cu = i * (NWV * CPW) + wave_u * CPW;  // wave-uniform column inside the tile
        if (g < NL) va[i] = z_load<EA>(P.Xrow, ldX, M, I * GT + cu, csub, P.lane_off, P.rowoff);
        else if constexpr (!DIAG) vb[i] = z_load<EB>(P.Xrow, ldX, M, J * GT + cu, csub, P.lane_off, P.rowoff);
    };
    auto store_slot = [&](int buf, bool rv, int g) {
        const int i = g & (NL - 1);
        const int c = i * (NWV * CPW) + wave * CPW + csub;
        if (g < NL) (sA + buf * (GT * GLD))[c * GLD + smp] = z_value<EA>(va[i], rv, M, I * GT + c);
        else if constexpr (!DIAG) (sB + buf * (GT * GLD))[c * GLD + smp] = z_value<EB>(vb[i], rv, M, J * GT + c);
    };
    auto store_virtual = [&](int buf, bool rv) {        // diagonal tiles: [valid, y] of the panel, and the corner sums of tile (0, 0)
        if constexpr (DIAG) {
            if (vloader) {
                const double one = rv ? 1.0 : 0.0, yv = rv ? vy : 0.0;
                double *dV = sV + buf * (2 * GK);
                dV[smp] = one; dV[GK + smp] = yv;
                if (corner) { c11 += one; c1y += yv; cyy = fma(yv, yv, cyy); }
            }
        }
    };
    constexpr int NSLOT = DIAG ? NL : 2 * NL;

    // prologue: panel 0 -> LDS buffer 0, panel 1 -> registers
    bool cur = live;
    Panel P = panel_here();
#pragma unroll
    for (int g = 0; g < NSLOT; ++g) load_slot(P, g);
    if constexpr (DIAG) { if (vloader) vy = P.yrow[P.rowoff]; }
#pragma unroll
    for (int g = 0; g < NSLOT; ++g) store_slot(0, P.rv, g);
    store_virtual(0, P.rv);
    advance();
    P = panel_here();
    bool nxt = live, rv1 = P.rv;
#pragma unroll
    for (int g = 0; g < NSLOT; ++g) load_slot(P, g);
    if constexpr (DIAG) { if (vloader) vy = P.yrow[P.rowoff]; }
    __syncthreads();
    int buf = 0;
    while (cur) {
        advance();
        const Panel P2 = panel_here();                  // panel p + 2
        const bool nn = live;
        const double *pA = sA + buf * (GT * GLD), *pB = sB + buf * (GT * GLD), *pV = sV + buf * (2 * GK);
        auto slot = [&](int g) {
            __builtin_amdgcn_sched_barrier(0);         // keep the slot where it was written: the scheduler otherwise hoists the stores of
                                                       // all slots (and their vmcnt waits) to the top of the phase, i.e. right behind the loads
            if (DIAG && g == 0) { store_virtual(buf ^ 1, rv1); if (vloader) vy = P2.yrow[P2.rowoff]; }
            store_slot(buf ^ 1, rv1, g);
            load_slot(P2, g);
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (DIAG) {
            switch (wave) {                            // wave-uniform: each wave runs its own item list (see DR / DC)
                case 0: diag_panel<0>(pA, pV, fr, fk, acc, slot); break;
                case 1: diag_panel<1>(pA, pV, fr, fk, acc, slot); break;
                case 2: diag_panel<2>(pA, pV, fr, fk, acc, slot); break;
                case 3: diag_panel<3>(pA, pV, fr, fk, acc, slot); break;
                case 4: diag_panel<4>(pA, pV, fr, fk, acc, slot); break;
                case 5: diag_panel<5>(pA, pV, fr, fk, acc, slot); break;
                case 6: diag_panel<6>(pA, pV, fr, fk, acc, slot); break;
                default: diag_panel<7>(pA, pV, fr, fk, acc, slot); break;
            }
        } else {
            full_panel(pA, pB, wr, wc, fr, fk, acc, slot);
        }
        __syncthreads();
        cur = nxt; nxt = nn; rv1 = P2.rv; buf ^= 1;
    }
    // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    double *slice = slab + (size_t)(xg * S + sl) * slice_doubles(np, nt);
    double *out = slice + (size_t)pair * GT * GT;
    if constexpr (!DIAG) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int li = (wr * 4 + i) * 16 + fk + 4 * reg, lj = (wc * 2 + j) * 16 + fr;
                    out[li * GT + lj] = acc[i * 2 + j][reg];
                }
    } else {
        const int ldv = nt * GT + 2;
        double *virt = slice + (size_t)np * GT * GT;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            if (w != wave) continue;
#pragma unroll
            for (int t = 0; t < DCNT[w]; ++t) {
                const int r = DR[w][t], c = DC[w][t];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int li = r * 16 + fk + 4 * reg;
                    if (c < NB) out[li * GT + c * 16 + fr] = acc[t][reg];
                    else if (fr < 2) virt[fr * ldv + I * GT + li] = acc[t][reg];       // X_I' 1 and X_I' y
                }
            }
        }
        if (I == 0 && wave == 0) {                     // 2 x 2 corner: fixed-order butterfly over the GK staging lanes
#pragma unroll
            for (int off = GK / 2; off > 0; off >>= 1) {
                c11 += __shfl_xor(c11, off); c1y += __shfl_xor(c1y, off); cyy += __shfl_xor(cyy, off);
            }
            if (lane == 0) { virt[ldv - 2] = c11; virt[ldv + ldv - 2] = c1y; virt[ldv + ldv - 1] = cyy; }   // (M, M), (M, M+1), (M+1, M+1)
        }
    }
}

__global__ __launch_bounds__(64 * NWV, 4) void gram_kernel(      // 4 waves per SIMD (HIP: the second argument counts waves per EU): <= 128 VGPRs
const double *__restrict__ X, int64_t N, int M, int64_t ldX,
                                                   const double *__restrict__ y, double *__restrict__ slab,
                                                   int chunk_rows, int S, int np)
{
    __shared__ double sA[2 * GT * GLD];              // [2 buffers][128 columns][GLD]
    __shared__ double sB[2 * GT * GLD];
    __shared__ double sV[2 * 2 * GK];                // [2 buffers][ones, y][GK]
    const int nt = (M + GT - 1) / GT;                  // feature tiles only
    const int xg = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int pair = q % np, sl = q / np;
    int I = 0, rem = pair;
    while (rem >= nt - I) { rem -= nt - I; ++I; }
    const int J = I + rem;
    // the last feature tile is an "edge" tile when M is not a multiple of 128 (zero padding beyond column M - 1).  I <= J.
    const bool ragged = (M % GT) != 0;
    const bool ea = ragged && I == nt - 1, eb = ragged && J == nt - 1;
    if (I == J) {
        if (ea) gram_body<true, true, true>(X, N, M, ldX, y, slab, np, nt, pair, chunk_rows, S, I, J, xg, sl, sA, sB, sV);
        else gram_body<false, false, true>(X, N, M, ldX, y, slab, np, nt, pair, chunk_rows, S, I, J, xg, sl, sA, sB, sV);
    } else {
        if (eb) gram_body<false, true, false>(X, N, M, ldX, y, slab, np, nt, pair, chunk_rows, S, I, J, xg, sl, sA, sB, sV);
        else gram_body<false, false, false>(X, N, M, ldX, y, slab, np, nt, pair, chunk_rows, S, I, J, xg, sl, sA, sB, sV);
    }
}

// G[i][j] = sum over the slices of the computed entry: a feature pair lives in the tile of its upper 16 x 16 sub-tile pair
// (diagonal tiles hold their upper triangle of sub-tiles only), an entry with a virtual index (ones = M, y = M + 1) in the virtual
// columns of the slice.  Fixed summation order: bitwise reproducible, no float atomics.
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double *__restrict__ slab, int slices, int np, int nt, int ldg, int M, double *__restrict__ G)
{
    // Only the threads of STORED entries walk the slices (consecutive threads = consecutive columns of a tile row: coalesced) and write
    // the mirror entry too; in round 1/2a the thread of a mirrored entry walked the slices itself, 1 KB apart from its neighbours — one
    // cache line per lane and slice for half of the matrix (0.30 ms at C4, a quarter of the C3 build).  64 entries per workgroup: wave g
    // sums the slices c = g, g + 4, ... of its entries, wave 0 adds the four partial sums in order (fixed order; 4 x the loads in flight).
    __shared__ double part[4][64];
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + e;
    const bool in = idx < ldg * ldg;
    const int i = in ? idx / ldg : 0, j = in ? idx % ldg : 0;
    const int ldv = nt * GT + 2;
    size_t off = 0;
    bool stored = false, mirror = false, zero = false;
    if (in) {
        if (i > M + 1 || j > M + 1) zero = true;
        else if (i >= M || j >= M) {
            if (i <= j) {                                                   // (j, i) is written by the thread of (i, j)
                const size_t vbase = (size_t)np * GT * GT;
                if (i < M) off = vbase + (size_t)(j - M) * ldv + i;        // X' 1, X' y
                else off = vbase + (size_t)(j - M) * ldv + (ldv - 2) + (i - M);   // corner: (M, M), (M, M + 1), (M + 1, M + 1)
                stored = true; mirror = i != j;
            }
        } else if ((i >> 4) <= (j >> 4)) {
            const int I = i / GT, J = j / GT;                              // I <= J
            const int pair = I * nt - I * (I - 1) / 2 + (J - I);
            off = (size_t)pair * GT * GT + (size_t)(i % GT) * GT + (j % GT);
            stored = true; mirror = (i >> 4) < (j >> 4);                   // a diagonal 16 x 16 block is stored whole: both threads have their own entry
        }
    }
    double s = 0.0;
    if (stored) {
        const size_t stride = slice_doubles(np, nt);
        for (int c = g; c < slices; c += 4) s += slab[(size_t)c * stride + off];
    }
    part[g][e] = s;
    __syncthreads();
    if (g == 0) {
        if (zero) G[idx] = 0.0;
        else if (stored) {
            const double t = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
            G[idx] = t;
            if (mirror) G[(size_t)j * ldg + i] = t;
        }
    }
}

static void gram_plan(int64_t N, int64_t M, int S_env, int cr_env, int *ldg_out, int *S_out, int *chunk_rows_out, int *np_out, int *nt_out)
{
    const int n_aug = (int)M + 2;
    const int ldg = ((n_aug + 63) / 64) * 64;
    const int nt = ((int)M + GT - 1) / GT, np = nt * (nt + 1) / 2;        // tile pairs over the features; ones / y ride on the diagonal
    // chunk of all columns ~ 1 MiB so that the concurrent chunks of an XCD group stay L2 resident
    int64_t cr = ((int64_t)1 << 20) / ((int64_t)n_aug * 8);
    cr = (cr / GK) * GK;
    if (cr < GK) cr = GK;
    if (cr > 4096) cr = 4096;
    if (cr_env > 0) { cr = (cr_env / GK) * GK; if (cr < GK) cr = GK; }
    // Workgroups run in rounds over the 512 resident slots (2 per CU, 256 CUs): the slice count is chosen so that the grid is just
    // UNDER a whole number k of rounds (a grid slightly over it would pay a full extra round for a handful of workgroups), with k as
    // large as leaves every workgroup >= 64 panels to amortise its prologue and its 128 KB tile write (k <= 4)
    int S = 1;
    if (S_env > 0) S = S_env;
    else {
        const int64_t panels = (N + GK - 1) / GK;
        for (int k = 4; k >= 1; k >>= 1) {
            S = (k * 512) / (8 * np);
            if (S < 1) S = 1;
            if (k == 1 || panels / ((int64_t)8 * S) >= 64) break;
        }
        // small problems: a workgroup needs >= 8 panels to be worth its 128 KB tile write and its share of the reduction
        if (panels / ((int64_t)8 * S) < 8) { S = (int)(panels / 64); if (S < 1) S = 1; }
    }
    // no more slices than chunks: shrink the chunks before giving up slices
    if (cr_env <= 0 && (N + cr - 1) / cr < (int64_t)8 * S) {
        cr = (N / ((int64_t)8 * S) / GK) * GK;
        if (cr < GK) cr = GK;
    }
    const int64_t nchunks = (N + cr - 1) / cr;
    while (S > 1 && (int64_t)8 * S > nchunks) --S;
    *ldg_out = ldg; *S_out = S; *chunk_rows_out = (int)cr; *np_out = np; *nt_out = nt;
}

size_t gram_slab_doubles(int64_t N, int64_t M, int gram_S, int gram_cr, int *chunks_out, int *ldg_out)
{
    int ldg, S, cr, np, nt;
    gram_plan(N, M, gram_S, gram_cr, &ldg, &S, &cr, &np, &nt);
    *chunks_out = 8 * S;                                            // number of partial slices
    *ldg_out = ldg;
    return (size_t)8 * S * slice_doubles(np, nt);
}

hipError_t launch_gram(const double *X, int64_t N, int64_t M, int64_t ldX, const double *y, double *slab, int chunks,
                       int ldg, int gram_S, int gram_cr, double *G, hipStream_t s)
{
    int ldg2, S, cr, np, nt;
    gram_plan(N, M, gram_S, gram_cr, &ldg2, &S, &cr, &np, &nt);
    if (ldX >= ((int64_t)1 << 30)) return hipErrorInvalidValue;          // 32-bit per-lane element offsets (see z_load)
    hipLaunchKernelGGL(gram_kernel, dim3(8 * S * np), dim3(64 * NWV), 0, s, X, N, (int)M, ldX, y, slab, cr, S, np);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int tot = ldg * ldg;
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((tot + 63) / 64), dim3(256), 0, s, slab, chunks, np, nt, ldg, (int)M, G);
    return hipGetLastError();
}

}  // namespace partls
