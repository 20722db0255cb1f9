// mfma_f64_peak.hip — what one SIMD sustains on v_mfma_f64_16x16x4_f64: NACC independent accumulators per wave, W waves per SIMD,
// no memory traffic.  Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_peak.hip -o bin/mfma_f64_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC, int OCC>
__global__ __launch_bounds__(256, OCC) void k(double *out, int iters, unsigned long long *cyc)
{
    double4_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 1e-3 + i; b[i] = 1.0 + threadIdx.x * 1e-4 * (i + 1); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC, int OCC>
static void run(int iters)
{
    double *out; unsigned long long *cyc, h;
    const int wgs_per_cu = OCC; const int grid = 256 * OCC;
    hipMalloc(&out, sizeof(double) * grid * 256); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, OCC>), dim3(grid), dim3(256), 0, 0, out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, OCC>), dim3(grid), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double mf = (double)iters * NACC;                       // MFMAs per wave
    const double tf = mf * 2048.0 * 4 * grid / (ms * 1e-3) / 1e12;
    printf("NACC %2d  waves/SIMD %d  %.3f ms  %.1f TFLOP/s  %.1f s_memtime ticks per MFMA per wave (x waves/SIMD = per SIMD: %.1f)\n", NACC, wgs_per_cu, ms, tf,
           (double)h / mf, (double)h / mf / wgs_per_cu);
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<16, 1>(20000); run<16, 2>(20000); run<8, 1>(40000); run<8, 2>(40000); run<8, 3>(40000); run<8, 4>(40000); run<4, 4>(80000); run<4, 8>(40000); run<2, 8>(80000); run<1, 8>(80000);
    return 0;
}
