// mfma_f64_lds.hip — ceiling of the Gram kernel's inner structure: per wave 8 accumulators (64 x 32 sub-tile), per k-step 4 + 2 fragment
// reads from a [col][17] LDS image (ds_read_b64, conflict-free) feeding 8 v_mfma_f64_16x16x4_f64; 512-thread workgroups, two per CU
// (4 waves per SIMD); no global traffic, optional barrier per 4 k-steps.  Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_lds.hip -o bin/mfma_f64_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
static constexpr int GLD = 17;
template <bool BARRIER, bool PREFETCH>
__global__ __launch_bounds__(512, 4) void k(double *out, int phases)
{
    __shared__ double sA[2 * 128 * GLD], sB[2 * 128 * GLD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 15, fk = lane >> 4, wr = wave >> 2, wc = wave & 3;
    for (int i = tid; i < 2 * 128 * GLD; i += 512) { sA[i] = 1e-3 * (i % 97); sB[i] = 1.0 + 1e-4 * (i % 89); }
    __syncthreads();
    double4_t acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (double4_t){0, 0, 0, 0};
    int buf = 0;
    for (int p = 0; p < phases; ++p) {
        const double *pA = sA + buf * 128 * GLD, *pB = sB + buf * 128 * GLD;
        double a[4], b_n[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = pA[((wr * 4 + i) * 16 + fr) * GLD + fk];
#pragma unroll
        for (int j = 0; j < 2; ++j) b_n[j] = pB[((wc * 2 + j) * 16 + fr) * GLD + fk];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            double b[2] = {b_n[0], b_n[1]};
            if (PREFETCH && ks + 1 < 4) {
#pragma unroll
                for (int j = 0; j < 2; ++j) b_n[j] = pB[((wc * 2 + j) * 16 + fr) * GLD + (ks + 1) * 4 + fk];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
                if (ks + 1 < 4) a[i] = pA[((wr * 4 + i) * 16 + fr) * GLD + (ks + 1) * 4 + fk];
            }
            if (!PREFETCH && ks + 1 < 4) {
#pragma unroll
                for (int j = 0; j < 2; ++j) b_n[j] = pB[((wc * 2 + j) * 16 + fr) * GLD + (ks + 1) * 4 + fk];
            }
        }
        if (BARRIER) __syncthreads();
        buf ^= 1;
    }
    double s = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 512 + tid] = s;
}
template <bool BARRIER, bool PREFETCH>
static void run(int phases)
{
    double *out;
    const int grid = 512;
    (void)hipMalloc(&out, sizeof(double) * grid * 512);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BARRIER, PREFETCH>), dim3(grid), dim3(512), 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<BARRIER, PREFETCH>), dim3(grid), dim3(512), 0, 0, out, phases);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)phases * 32 * 8 * grid;
    printf("barrier %d prefetch %d: %.3f ms, %.1f TFLOP/s executed\n", (int)BARRIER, (int)PREFETCH, ms, mfmas * 2048.0 / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}
int main()
{
    run<false, true>(4000); run<true, true>(4000); run<false, false>(4000); run<true, false>(4000);
    return 0;
}
