// fmac_dpp.hip — v_fmac_f64_dpp row_newbcast: (a) semantics: acc += x * (y of lane N of the lane's own row of 16), per row;
// (b) rate against plain v_fma_f64 with the same accumulator count and occupancy.  The sweep's update loop uses it to take the
// column operand of a rank-1 term from ONE register per 16 tile columns instead of one LDS read per tile column.
// Build: hipcc -O3 --offload-arch=gfx950 fmac_dpp.hip -o bin/fmac_dpp ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int G>
__device__ __forceinline__ void fmac_bcast(double &acc, double y, double x)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(y), "v"(x), "n"(G));
}
__global__ void check(double *out, const double *yin, const double *xin)
{
    const int l = threadIdx.x;
    double acc[16];
    const double y = yin[l], x = xin[l];
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.0;
    fmac_bcast<0>(acc[0], y, x); fmac_bcast<1>(acc[1], y, x); fmac_bcast<2>(acc[2], y, x); fmac_bcast<3>(acc[3], y, x);
    fmac_bcast<4>(acc[4], y, x); fmac_bcast<5>(acc[5], y, x); fmac_bcast<6>(acc[6], y, x); fmac_bcast<7>(acc[7], y, x);
    fmac_bcast<8>(acc[8], y, x); fmac_bcast<9>(acc[9], y, x); fmac_bcast<10>(acc[10], y, x); fmac_bcast<11>(acc[11], y, x);
    fmac_bcast<12>(acc[12], y, x); fmac_bcast<13>(acc[13], y, x); fmac_bcast<14>(acc[14], y, x); fmac_bcast<15>(acc[15], y, x);
#pragma unroll
    for (int g = 0; g < 16; ++g) out[g * 64 + l] = acc[g];
}
template <int NACC, bool DPP>
__global__ __launch_bounds__(512, 2) void rate(double *out, int iters, double x0, double y0)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    double x[4] = {x0, x0 + 1e-9, x0 + 2e-9, x0 + 3e-9};
    double y = y0 + 1e-12 * (threadIdx.x & 15);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (DPP) {
                if ((i & 3) == 0) fmac_bcast<0>(acc[i], y, x[i & 3]);
                else if ((i & 3) == 1) fmac_bcast<5>(acc[i], y, x[i & 3]);
                else if ((i & 3) == 2) fmac_bcast<10>(acc[i], y, x[i & 3]);
                else fmac_bcast<15>(acc[i], y, x[i & 3]);
            } else {
                acc[i] = fma(x[i & 3], y, acc[i]);
            }
        }
        y = -y;
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, bool DPP>
static void run(int iters)
{
    double *out;
    const int grid = 256;
    (void)hipMalloc(&out, sizeof(double) * grid * 512);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<NACC, DPP>), dim3(grid), dim3(512), 0, 0, out, 10, 0.999999, 1e-7);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((rate<NACC, DPP>), dim3(grid), dim3(512), 0, 0, out, iters, 0.999999, 1e-7);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double tf = 2.0 * (double)iters * NACC * 512.0 * grid / (ms * 1e-3) / 1e12;
    printf("%s NACC %2d, 512-thread workgroups (2 waves/SIMD): %.3f ms  %.1f TFLOP/s\n", DPP ? "v_fmac_f64_dpp row_newbcast" : "v_fma_f64                  ", NACC, ms, tf);
    (void)hipFree(out);
}
int main()
{
    double *dy, *dx, *dout, hy[64], hx[64], ho[16 * 64];
    for (int l = 0; l < 64; ++l) { hy[l] = 100.0 + l; hx[l] = 1.0 + 0.001 * l; }
    (void)hipMalloc(&dy, sizeof(hy)); (void)hipMalloc(&dx, sizeof(hx)); (void)hipMalloc(&dout, sizeof(ho));
    (void)hipMemcpy(dy, hy, sizeof(hy), hipMemcpyHostToDevice); (void)hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, dout, dy, dx);
    (void)hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int g = 0; g < 16; ++g)
        for (int l = 0; l < 64; ++l) {
            const double want = hx[l] * hy[(l & ~15) + g];          // lane g of the lane's own row
            if (fabs(ho[g * 64 + l] - want) > 1e-12 * fabs(want)) { if (bad++ < 5) printf("MISMATCH g %d lane %d: got %.6f want %.6f\n", g, l, ho[g * 64 + l], want); }
        }
    printf("semantics: %s (acc += x[lane] * y[16 * (lane / 16) + N])\n", bad ? "WRONG" : "ok");
    run<64, false>(20000); run<64, true>(20000); run<32, false>(40000); run<32, true>(40000);
    return bad != 0;
}
