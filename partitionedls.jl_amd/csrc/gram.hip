// gram.hip — the one dense contraction of the path: G_aug = Z'Z with Z = [X 1 y]  (N x (M+2)), fp64 MFMA.
//
// Replaces the data passes that every nonneg_lsq / norm call of the reference repeats per pattern
// (Opt.jl:88-90: bmatrix copy, NNLS workspace copy, Householder sweeps, objective GEMM): X is read ONCE here and all
// 2^K' subproblems then work on the (M+2)^2 Gram block.  G = Xo'Xo is G_aug[0..M, 0..M],
// c = Xo'y is column M+1, yy = y'y the corner; the ones column and y are virtual (never materialised).
//
// Kernel: 64x64 output macro-tile per workgroup (4 waves, each 16 rows x 64 cols = 4 v_mfma_f64_16x16x4_f64
// accumulators, + 1 on diagonal tiles for the ones / y columns), samples split into row chunks, each chunk staged through LDS
// in 32-sample panels ([col][34] layout: conflict-free for the ds_read_b64 fragment reads).  Only tile pairs I <= J (SYRK).
// Partial tiles go to a per-chunk slab and are summed in a fixed order by gram_reduce (bitwise reproducible;
// no float atomics).  Roofline: fp64 MFMA bound — 2*N*(M+2)^2/2 flops over 8*N*M bytes (intensity ~ M/8 flop/B).
#include "common.h"
#include <cstdlib>

namespace partls {

typedef double double4_t __attribute__((ext_vector_type(4)));

static constexpr int GT = 64;      // macro tile edge
static constexpr int GK = 32;      // samples per LDS panel
static constexpr int GLD = 34;     // padded panel stride (doubles): bank = (4c + 2s) % 64, distinct per 32-lane group

// Tiles cover the FEATURES only (nt = ceil(M / 64) tile columns, pairs I <= J).  The two virtual columns of Z = [X 1 y] — the
// ones column and y — would cost a whole 64-wide tile column for two columns (9 of 45 tiles at D = 512); instead they ride on
// the DIAGONAL tiles: tile (I, I) carries a fifth 16 x 16 accumulator whose B fragment is [1, y, 0, ...], which yields
// X_I' 1 and X_I' y for its 64 columns at a quarter of a tile's cost, and the workgroups of tile (0, 0) also accumulate the
// 2 x 2 corner (N, 1'y, y'y) as plain sums.
//
// Loading one feature element is split in two so that the global loads stay in flight under the MFMAs:
//   z_load  : unconditional load from a clamped address (issued one panel ahead, result untouched),
//   z_value : zero for the padding columns of the last tile (EDGE) and for the row tail, applied at the LDS store.
template <bool EDGE>
__device__ __forceinline__ double z_load(const double *__restrict__ X, int64_t ldX, int M, int col, int64_t rr)
{
    const int cc = EDGE ? (col < M ? col : M - 1) : col;
    return X[rr + (int64_t)cc * ldX];
}
template <bool EDGE>
__device__ __forceinline__ double z_value(double x, bool rv, int M, int col)
{
    double v = x;
    if constexpr (EDGE) v = col < M ? x : 0.0;
    return rv ? v : 0.0;
}

// Diagonal tiles compute the upper triangle only, at 16 x 16 granularity: 10 sub-tiles (r <= c) + the 4 virtual ones (r, V)
// = 14 MFMAs per k-step, dealt to the four waves so that none issues more than an off-diagonal tile's 4 — every workgroup then
// advances through the samples at the same pace, which is what keeps the panels of a chunk L2-resident for all the tile pairs
// that read them (with full diagonal tiles + the virtual accumulator the diagonal workgroups fell 25 % behind: L2 hit rate
// 22 %, 26 GB of L2 misses per C4 build for 4.1 GB of data).  Item t of wave W: row block DR[W][t], column block DC[W][t] (4 = V).
__device__ constexpr int DN[4] = {4, 4, 4, 2};
__device__ constexpr int DR[4][4] = {{0, 0, 0, 0}, {1, 1, 1, 0}, {2, 2, 1, 2}, {3, 3, 0, 0}};
__device__ constexpr int DC[4][4] = {{0, 1, 2, 3}, {1, 2, 3, 4}, {2, 3, 4, 4}, {3, 4, 0, 0}};

template <int W>
__device__ __forceinline__ void diag_panel(const double *sA, const double *sV, int fr, int fk, double4_t (&acc)[4])
{
    auto afrag = [&](int r, int ks) { return sA[(r * 16 + fr) * GLD + ks * 4 + fk]; };
    auto bfrag = [&](int c, int ks) { return c < 4 ? sA[(c * 16 + fr) * GLD + ks * 4 + fk] : (fr < 2 ? sV[fr * GK + ks * 4 + fk] : 0.0); };
    double a_n[DN[W]], b_n[DN[W]];
#pragma unroll
    for (int t = 0; t < DN[W]; ++t) { a_n[t] = afrag(DR[W][t], 0); b_n[t] = bfrag(DC[W][t], 0); }
#pragma unroll
    for (int ks = 0; ks < GK / 4; ++ks) {
        double a[DN[W]], b[DN[W]];
#pragma unroll
        for (int t = 0; t < DN[W]; ++t) { a[t] = a_n[t]; b[t] = b_n[t]; }
        if (ks + 1 < GK / 4) {
#pragma unroll
            for (int t = 0; t < DN[W]; ++t) { a_n[t] = afrag(DR[W][t], ks + 1); b_n[t] = bfrag(DC[W][t], ks + 1); }
        }
#pragma unroll
        for (int t = 0; t < DN[W]; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[t], acc[t], 0, 0, 0);
    }
}

// Work decomposition (XCD-aware): the samples are cut into row chunks of `chunk_rows`.  Workgroups are dealt round-robin
// over the 8 XCDs, so `blockIdx.x & 7` labels the XCD group; group x, slice s walks the chunks x + 8*(s + S*j), and the np
// tile-pair workgroups of one (x, s) walk the SAME chunk sequence in step, so panels are re-read from that XCD's L2 by the
// other tile pairs (speed only — correctness does not depend on placement).  Each workgroup accumulates all its chunks in
// registers and writes one partial tile to slab (x*S + s).  Inside, the next 32-sample panel is prefetched into registers
// (16 independent, unconditional global loads) while the MFMAs of the current one run from LDS, and the MFMA fragments of
// k-step ks+1 are read from LDS before the MFMAs of k-step ks are issued (register double buffer).
template <bool EA, bool EB, bool DIAG>
__device__ __forceinline__ void gram_body(const double *__restrict__ X, int64_t N, int M, int64_t ldX,
                                          const double *__restrict__ y, double *__restrict__ slab, int ldg,
                                          int chunk_rows, int S, int I, int J, int xg, int sl, double *sA, double *sB, double *sV)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double4_t acc[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc[jt] = (double4_t){0.0, 0.0, 0.0, 0.0};

    const int smp = lane & 31, csub = lane >> 5;       // loader: 32 lanes cover one column's 32 contiguous samples
    const int fr = lane & 15, fk = lane >> 4;          // MFMA fragment coordinates
    const double *pB = DIAG ? sA : sB;
    const int kb_per_chunk = chunk_rows / GK;
    const int64_t nchunks = (N + chunk_rows - 1) / chunk_rows;
    const int64_t cstride = (int64_t)8 * S;
    const bool vloader = DIAG && tid < GK;             // threads 0..31 of a diagonal tile also stage [valid, y] of the panel
    const bool corner = DIAG && I == 0 && tid < GK;    // ... and, for tile (0, 0), accumulate the 2 x 2 corner
    double c11 = 0.0, c1y = 0.0, cyy = 0.0;

    double va[8], vb[8], vy = 0.0;                     // raw prefetched values (+ y for the diagonal tiles) and the row predicate
    bool rv = false;
    auto fetch = [&](int64_t chunk, int kb) {
        const int64_t r1 = ((chunk + 1) * chunk_rows < N) ? (chunk + 1) * chunk_rows : N;
        const int64_t row = chunk * chunk_rows + (int64_t)kb * GK + smp;
        rv = row < r1;
        const int64_t rr = rv ? row : r1 - 1;
        if constexpr (DIAG) { if (vloader) vy = y[rr]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = i * 8 + wave * 2 + csub;
            va[i] = z_load<EA>(X, ldX, M, I * GT + c, rr);
            if constexpr (!DIAG) vb[i] = z_load<EB>(X, ldX, M, J * GT + c, rr);
        }
    };

    int64_t chunk = xg + (int64_t)8 * sl;
    int kb = 0;
    bool have = chunk < nchunks;
    if (have) fetch(chunk, 0);
    while (have) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = i * 8 + wave * 2 + csub;
            sA[c * GLD + smp] = z_value<EA>(va[i], rv, M, I * GT + c);
            if constexpr (!DIAG) sB[c * GLD + smp] = z_value<EB>(vb[i], rv, M, J * GT + c);
        }
        if constexpr (DIAG) {
            if (vloader) {
                const double one = rv ? 1.0 : 0.0, yv = rv ? vy : 0.0;
                sV[smp] = one; sV[GK + smp] = yv;
                if (corner) { c11 += one; c1y += yv; cyy = fma(yv, yv, cyy); }
            }
        }
        __syncthreads();
        // advance and prefetch the next panel (global loads stay in flight under the MFMAs below)
        if (++kb == kb_per_chunk) { kb = 0; chunk += cstride; }
        have = chunk < nchunks && (chunk * chunk_rows + (int64_t)kb * GK) < N;
        if (have) fetch(chunk, kb);
        if constexpr (DIAG) {
            switch (wave) {                            // wave-uniform: each wave runs its own item list (see DN / DR / DC)
                case 0: diag_panel<0>(sA, sV, fr, fk, acc); break;
                case 1: diag_panel<1>(sA, sV, fr, fk, acc); break;
                case 2: diag_panel<2>(sA, sV, fr, fk, acc); break;
                default: diag_panel<3>(sA, sV, fr, fk, acc); break;
            }
        } else {
            // fragments: a = A[row wave*16 + fr][k = 4 ks + fk], b[jt] = B[col jt*16 + fr][k]; k-step ks+1 is read before ks issues
            double a_n = sA[(wave * 16 + fr) * GLD + fk], b_n[4];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) b_n[jt] = pB[(jt * 16 + fr) * GLD + fk];
#pragma unroll
            for (int ks = 0; ks < GK / 4; ++ks) {
                const double a = a_n;
                double b[4];
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) b[jt] = b_n[jt];
                if (ks + 1 < GK / 4) {
                    a_n = sA[(wave * 16 + fr) * GLD + (ks + 1) * 4 + fk];
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) b_n[jt] = pB[(jt * 16 + fr) * GLD + (ks + 1) * 4 + fk];
                }
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[jt], acc[jt], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    double *out = slab + (size_t)(xg * S + sl) * (size_t)ldg * (size_t)ldg;
    if constexpr (!DIAG) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int gi = I * GT + wave * 16 + fk + 4 * reg;
                const int gj = J * GT + jt * 16 + fr;
                if (gi < M && gj < M) out[(size_t)gi * ldg + gj] = acc[jt][reg];
            }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t >= DN[wave]) break;
            const int r = DR[wave][t], c = DC[wave][t];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int gi = I * GT + r * 16 + fk + 4 * reg;
                if (c < 4) {
                    const int gj = I * GT + c * 16 + fr;
                    if (gi < M && gj < M) out[(size_t)gi * ldg + gj] = acc[t][reg];
                } else if (fr < 2 && gi < M) {
                    out[(size_t)gi * ldg + M + fr] = acc[t][reg];          // columns M (ones) and M + 1 (y)
                }
            }
        }
        if (I == 0 && wave == 0) {                     // 2 x 2 corner: fixed-order butterfly over the 32 staging lanes
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) {
                c11 += __shfl_xor(c11, off); c1y += __shfl_xor(c1y, off); cyy += __shfl_xor(cyy, off);
            }
            if (lane == 0) {
                out[(size_t)M * ldg + M] = c11; out[(size_t)M * ldg + M + 1] = c1y; out[(size_t)(M + 1) * ldg + M + 1] = cyy;
            }
        }
    }
}

__global__ __launch_bounds__(256, 4) void gram_kernel(const double *__restrict__ X, int64_t N, int M, int64_t ldX,
                                                   const double *__restrict__ y, double *__restrict__ slab, int ldg,
                                                   int chunk_rows, int S, int np)
{
    __shared__ double sA[GT * GLD];
    __shared__ double sB[GT * GLD];
    __shared__ double sV[2 * GK];
    const int nt = (M + GT - 1) / GT;                  // feature tiles only
    const int xg = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int pair = q % np, sl = q / np;
    int I = 0, rem = pair;
    while (rem >= nt - I) { rem -= nt - I; ++I; }
    const int J = I + rem;
    // the last feature tile is an "edge" tile when M is not a multiple of 64 (zero padding beyond column M - 1).  I <= J.
    const bool ragged = (M % GT) != 0;
    const bool ea = ragged && I == nt - 1, eb = ragged && J == nt - 1;
    if (I == J) {
        if (ea) gram_body<true, true, true>(X, N, M, ldX, y, slab, ldg, chunk_rows, S, I, J, xg, sl, sA, sB, sV);
        else gram_body<false, false, true>(X, N, M, ldX, y, slab, ldg, chunk_rows, S, I, J, xg, sl, sA, sB, sV);
    } else {
        if (eb) gram_body<false, true, false>(X, N, M, ldX, y, slab, ldg, chunk_rows, S, I, J, xg, sl, sA, sB, sV);
        else gram_body<false, false, false>(X, N, M, ldX, y, slab, ldg, chunk_rows, S, I, J, xg, sl, sA, sB, sV);
    }
}

// G[i][j] = sum over the slabs of the computed entry: features use the upper 16 x 16 sub-tile pair, an entry with a virtual index (ones = M,
// y = M + 1) lives at [min(i, j)][max(i, j)].  Fixed summation order: bitwise reproducible, no float atomics.
__global__ void gram_reduce_kernel(const double *__restrict__ slab, int chunks, int ldg, int M, double *__restrict__ G)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ldg * ldg) return;
    const int i = idx / ldg, j = idx % ldg;
    if (i > M + 1 || j > M + 1) { G[idx] = 0.0; return; }
    const bool virt = i >= M || j >= M;
    const bool upper = virt ? (i <= j) : ((i >> 4) <= (j >> 4));       // 16 x 16 sub-tiles: diagonal tiles hold their upper triangle only
    const size_t off = upper ? (size_t)i * ldg + j : (size_t)j * ldg + i;
    double s = 0.0;
    for (int c = 0; c < chunks; ++c) s += slab[(size_t)c * ldg * ldg + off];
    G[idx] = s;
}

static void gram_plan(int64_t N, int64_t M, int S_env, int cr_env, int *ldg_out, int *S_out, int *chunk_rows_out, int *np_out)
{
    const int n_aug = (int)M + 2;
    const int ldg = ((n_aug + GT - 1) / GT) * GT;
    const int nt = ((int)M + GT - 1) / GT, np = nt * (nt + 1) / 2;        // tile pairs over the features; ones / y ride on the diagonal
    // chunk of all columns ~ 1 MiB so that the S concurrent chunks of an XCD group stay L2 resident
    int64_t cr = ((int64_t)1 << 20) / ((int64_t)n_aug * 8);
    cr = (cr / GK) * GK;
    if (cr < GK) cr = GK;
    if (cr > 4096) cr = 4096;
    // many more workgroups than resident slots (512) so that the last partial wave of workgroups is a small tail
    if (cr_env > 0) { cr = (cr_env / GK) * GK; if (cr < GK) cr = GK; }
    int S = S_env > 0 ? S_env : (4096 + 8 * np - 1) / (8 * np);
    if (S < 1) S = 1;
    const int64_t nchunks = (N + cr - 1) / cr;
    while (S > 1 && (int64_t)8 * S > nchunks) --S;                 // no more slices than chunks
    *ldg_out = ldg; *S_out = S; *chunk_rows_out = (int)cr; *np_out = np;
}

size_t gram_slab_doubles(int64_t N, int64_t M, int gram_S, int gram_cr, int *chunks_out, int *ldg_out)
{
    int ldg, S, cr, np;
    gram_plan(N, M, gram_S, gram_cr, &ldg, &S, &cr, &np);
    *chunks_out = 8 * S;                                            // number of partial slabs
    *ldg_out = ldg;
    return (size_t)8 * S * ldg * ldg;
}

hipError_t launch_gram(const double *X, int64_t N, int64_t M, int64_t ldX, const double *y, double *slab, int chunks,
                       int ldg, int gram_S, int gram_cr, double *G, hipStream_t s)
{
    int ldg2, S, cr, np;
    gram_plan(N, M, gram_S, gram_cr, &ldg2, &S, &cr, &np);
    hipLaunchKernelGGL(gram_kernel, dim3(8 * S * np), dim3(256), 0, s, X, N, (int)M, ldX, y, slab, ldg, cr, S, np);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int tot = ldg * ldg;
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, slab, chunks, ldg, (int)M, G);
    return hipGetLastError();
}

}  // namespace partls
