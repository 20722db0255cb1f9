// mfma_f64_4x4x4.hip — v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks per wave, ONE f64 per lane for A, B and C/D):
// (1) the lane maps, found with one-hot operands: for every (la, lb) which lane of D receives A[la] * B[lb];
// (2) what a SIMD sustains on it with 2 waves per SIMD and 64 independent accumulators per wave (the register tableau's update).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_4x4x4.hip -o bin/mfma_f64_4x4x4 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void layout(int *tab)
{
    const int l = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == la ? 1.0 : 0.0, l == lb ? 1.0 : 0.0, 0.0, 0, 0, 0);
            if (d != 0.0) tab[la * 64 + lb] = l;
        }
}
template <int NACC, int OCC>
__global__ __launch_bounds__(256, OCC) void peak(double *out, int iters, unsigned long long *cyc)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 1e-3 + i; b[i] = 1.0 + threadIdx.x * 1e-4 * (i + 1); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC, int OCC>
static void run(int iters)
{
    double *out; unsigned long long *cyc, h;
    const int grid = 256 * OCC;
    hipMalloc(&out, sizeof(double) * grid * 256); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((peak<NACC, OCC>), dim3(grid), dim3(256), 0, 0, out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((peak<NACC, OCC>), dim3(grid), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double mf = (double)iters * NACC;
    printf("NACC %2d  waves/SIMD %d  %.3f ms  %.1f TFLOP/s  (512 flop per instruction)  %.1f s_memtime ticks per MFMA per wave\n", NACC, OCC, ms,
           mf * 512.0 * 4 * grid / (ms * 1e-3) / 1e12, (double)h / mf);
    hipFree(out); hipFree(cyc);
}
int main()
{
    int *tab; hipMalloc(&tab, 4096 * 4); hipMemset(tab, 0xFF, 4096 * 4);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, tab); hipDeviceSynchronize();
    std::vector<int> h(4096); hipMemcpy(h.data(), tab, 4096 * 4, hipMemcpyDeviceToHost);
    // for D lane ld: which (la, lb) pairs feed it -> expect 4 pairs (k = 0..3)
    for (int ld = 0; ld < 64; ++ld) {
        printf("D lane %2d <-", ld);
        for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] == ld) printf(" (A%d,B%d)", la, lb);
        printf("\n");
    }
    run<64, 1>(20000); run<64, 2>(20000); run<16, 2>(40000); run<16, 4>(40000);
    return 0;
}
