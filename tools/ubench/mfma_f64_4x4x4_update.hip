// mfma_f64_4x4x4_update.hip — the register tableau's rank-4 update in isolation: 512-thread workgroups (2 waves per SIMD, one workgroup per
// CU), 68 resident accumulators per thread, per group 16 x-reads + 16 y-reads from LDS and 68 v_mfma_f64_4x4x4 (MODE 1) or 4 x 68 v_fma_f64
// with the x / y of one pivot each (MODE 0).  Prints s_memtime ticks per group of 4 pivots as seen by wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int NS = 68, CW = 513, RS = 17;
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(double *out, int groups, unsigned long long *cyc)
{
    __shared__ double Z[8 * CW];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8 * CW; i += 512) Z[i] = 1e-3 * (i % 97);
    __syncthreads();
    double S[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) S[i] = 0.0;
    const int a = lane & 15, b = (tid >> 6 & 3) * 4 + (lane & 3), kk = lane >> 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int g = 0; g < groups; ++g) {
        if (MODE == 1) {
            const double *Zs = Z + ((g & 1) * 4 + kk) * CW;
            double x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = Zs[a * RS + r];
            int idx = 0;
#pragma unroll
            for (int gam = 5; gam < 16; ++gam) {
                const double y = -Zs[b * RS + gam] * 0.5;
#pragma unroll
                for (int r = 0; r <= gam && idx < NS; ++r, ++idx) S[idx] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[r], y, S[idx], 0, 0, 0);
            }
        } else {
#pragma unroll 1
            for (int s = 0; s < 4; ++s) {
                const double *Zs = Z + ((g & 1) * 4 + s) * CW;
                double x[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = Zs[a * RS + r];
                int idx = 0;
#pragma unroll
                for (int gam = 5; gam < 16; ++gam) {
                    const double y = -Zs[b * RS + gam] * 0.5;
#pragma unroll
                    for (int r = 0; r <= gam && idx < NS; ++r, ++idx) S[idx] = fma(x[r], y, S[idx]);
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NS; ++i) s += S[i];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> static void run(int groups)
{
    double *out; unsigned long long *cyc, h;
    (void)hipMalloc(&out, sizeof(double) * 256 * 512); (void)hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, 10, cyc); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, groups, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s: %.3f ms, %.0f ticks per group of 4 pivots (wave 0), %.1f TFLOP/s\n", MODE ? "mfma 4x4x4" : "v_fma_f64 ", ms, (double)h / groups,
           (double)groups * 4 * NS * 2 * 512 * 256 / (ms * 1e-3) / 1e12);
}
int main() { run<0>(20000); run<1>(20000); run<0>(20000); run<1>(20000); return 0; }
