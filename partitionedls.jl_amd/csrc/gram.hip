// gram.hip — the one dense contraction of the path: G_aug = Z'Z with Z = [X 1 y]  (N x (M+2)), fp64 MFMA.
//
// Replaces the data passes that every nonneg_lsq / norm call of the reference repeats per pattern
// (Opt.jl:88-90: bmatrix copy, NNLS workspace copy, Householder sweeps, objective GEMM): X is read ONCE here and all
// 2^K' subproblems then work on the (M+2)^2 Gram block.  G = Xo'Xo is G_aug[0..M, 0..M],
// c = Xo'y is column M+1, yy = y'y the corner; the ones column and y are virtual (never materialised).
//
// Kernel: 64x64 output macro-tile per workgroup (4 waves, each 16 rows x 64 cols = 4 v_mfma_f64_16x16x4_f64
// accumulators), samples split into row chunks across blockIdx.y, each chunk staged through LDS in 32-sample panels
// ([col][34] layout: conflict-free for the ds_read_b64 fragment reads).  Only tile pairs I <= J are computed (SYRK).
// Partial tiles go to a per-chunk slab and are summed in a fixed order by gram_reduce (bitwise reproducible;
// no float atomics).  Roofline: fp64 MFMA bound — 2*N*(M+2)^2/2 flops over 8*N*M bytes (intensity ~ M/8 flop/B).
#include "common.h"

namespace partls {

typedef double double4_t __attribute__((ext_vector_type(4)));

static constexpr int GT = 64;      // macro tile edge
static constexpr int GK = 32;      // samples per LDS panel
static constexpr int GLD = 34;     // padded panel stride (doubles): bank = (4c + 2s) % 64, distinct per 32-lane group

__device__ __forceinline__ double z_value(const double *__restrict__ X, const double *__restrict__ y, int64_t ldX, int M,
                                          int col, int64_t row, int64_t row_end)
{
    if (row >= row_end) return 0.0;
    if (col < M) return X[row + (int64_t)col * ldX];
    if (col == M) return 1.0;
    if (col == M + 1) return y[row];
    return 0.0;
}

__global__ __launch_bounds__(256) void gram_kernel(const double *__restrict__ X, int64_t N, int M, int64_t ldX,
                                                   const double *__restrict__ y, double *__restrict__ slab, int ldg,
                                                   int64_t rows_per_chunk)
{
    __shared__ double sA[GT * GLD];
    __shared__ double sB[GT * GLD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nt = ldg / GT;
    int I = 0, rem = blockIdx.x;
    while (rem >= nt - I) { rem -= nt - I; ++I; }
    const int J = I + rem;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = (r0 + rows_per_chunk < N) ? r0 + rows_per_chunk : N;

    double4_t acc[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc[jt] = (double4_t){0.0, 0.0, 0.0, 0.0};

    const int smp = lane & 31, csub = lane >> 5;       // loader: 32 lanes cover one column's 32 contiguous samples
    const int fr = lane & 15, fk = lane >> 4;          // MFMA fragment coordinates
    const double *pB = (I == J) ? sA : sB;

    for (int64_t k0 = r0; k0 < r1; k0 += GK) {
        double va[8], vb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = i * 8 + wave * 2 + csub;
            va[i] = z_value(X, y, ldX, M, I * GT + c, k0 + smp, r1);
            if (I != J) vb[i] = z_value(X, y, ldX, M, J * GT + c, k0 + smp, r1);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = i * 8 + wave * 2 + csub;
            sA[c * GLD + smp] = va[i];
            if (I != J) sB[c * GLD + smp] = vb[i];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < GK / 4; ++ks) {
            const double a = sA[(wave * 16 + fr) * GLD + ks * 4 + fk];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const double b = pB[(jt * 16 + fr) * GLD + ks * 4 + fk];
                acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[jt], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    double *out = slab + (size_t)blockIdx.y * (size_t)ldg * (size_t)ldg;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int gi = I * GT + wave * 16 + fk + 4 * reg;
            const int gj = J * GT + jt * 16 + fr;
            out[(size_t)gi * ldg + gj] = acc[jt][reg];
        }
}

__global__ void gram_reduce_kernel(const double *__restrict__ slab, int chunks, int ldg, double *__restrict__ G)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ldg * ldg) return;
    const int i = idx / ldg, j = idx % ldg;
    const bool upper = (i / GT) <= (j / GT);
    const size_t off = upper ? (size_t)i * ldg + j : (size_t)j * ldg + i;
    double s = 0.0;
    for (int c = 0; c < chunks; ++c) s += slab[(size_t)c * ldg * ldg + off];
    G[idx] = s;
}

size_t gram_slab_doubles(int64_t N, int64_t M, int *chunks_out, int *ldg_out)
{
    const int n_aug = (int)M + 2;
    const int ldg = ((n_aug + GT - 1) / GT) * GT;
    const int nt = ldg / GT, np = nt * (nt + 1) / 2;
    int64_t chunks = 2048 / np;
    if (chunks < 1) chunks = 1;
    int64_t maxc = (N + 255) / 256;                 // at least 256 rows per chunk
    if (chunks > maxc) chunks = maxc;
    if (chunks < 1) chunks = 1;
    while (chunks > 1 && (size_t)chunks * ldg * ldg * 8 > ((size_t)1 << 30)) --chunks;
    *chunks_out = (int)chunks;
    *ldg_out = ldg;
    return (size_t)chunks * ldg * ldg;
}

hipError_t launch_gram(const double *X, int64_t N, int64_t M, int64_t ldX, const double *y, double *slab, int chunks,
                       int ldg, double *G, hipStream_t s)
{
    const int nt = ldg / GT, np = nt * (nt + 1) / 2;
    int64_t rpc = (N + chunks - 1) / chunks;
    rpc = ((rpc + GK - 1) / GK) * GK;
    dim3 grid(np, chunks);
    hipLaunchKernelGGL(gram_kernel, grid, dim3(256), 0, s, X, N, (int)M, ldX, y, slab, ldg, rpc);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int tot = ldg * ldg;
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, slab, chunks, ldg, G);
    return hipGetLastError();
}

}  // namespace partls
