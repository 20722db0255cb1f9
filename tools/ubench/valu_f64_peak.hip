// valu_f64_peak.hip — what one SIMD sustains on v_fma_f64: NACC independent accumulators per wave, OCC waves per SIMD, no memory traffic.
// Build: hipcc -O3 --offload-arch=gfx950 valu_f64_peak.hip -o bin/valu_f64_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NACC, int OCC>
__global__ __launch_bounds__(256, OCC) void k(double *out, int iters, double x, double y)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    double a[4] = {x, x + 1e-9, x + 2e-9, x + 3e-9}, b[4] = {y, y + 1e-9, y + 2e-9, y + 3e-9};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(a[i & 3], acc[i], b[(i >> 2) & 3]);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int OCC>
static void run(int iters)
{
    double *out;
    const int grid = 256 * OCC;
    (void)hipMalloc(&out, sizeof(double) * grid * 256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, OCC>), dim3(grid), dim3(256), 0, 0, out, 10, 0.999999, 1e-7);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, OCC>), dim3(grid), dim3(256), 0, 0, out, iters, 0.999999, 1e-7);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double fmas = (double)iters * NACC * 64.0 * 4 * grid;       // lane-FMAs
    const double tf = 2.0 * fmas / (ms * 1e-3) / 1e12;
    // cycles per wave-instruction on one SIMD at 2.2 GHz: time * clock / (instructions issued on that SIMD)
    const double cyc = ms * 1e-3 * 2.2e9 / ((double)iters * NACC * OCC);
    printf("v_fma_f64  NACC %2d  waves/SIMD %d  %.3f ms  %.1f TFLOP/s  ~%.1f cycles per wave-instruction per SIMD (at 2.2 GHz)\n", NACC, OCC, ms, tf, cyc);
    (void)hipFree(out);
}
int main()
{
    run<1, 1>(400000); run<2, 1>(400000); run<4, 1>(400000); run<8, 1>(200000); run<16, 1>(200000); run<32, 1>(100000); run<64, 1>(50000);
    run<1, 2>(400000); run<4, 2>(400000); run<8, 2>(200000); run<16, 2>(200000); run<32, 2>(100000); run<64, 2>(50000); run<16, 4>(100000); run<4, 8>(200000);
    return 0;
}
