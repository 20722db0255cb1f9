// panel_two_phase.hip — the block panel of sweep_blk.hip in its shipped form (one barrier per pivot step) against the two-phase forms of round 4
// (tools/experiments/README.md).  Build: hipcc -O3 --offload-arch=gfx950 panel_two_phase.hip -o bin/panel_two_phase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int CW = 513, RS = 17, MB = 8, US = MB + 64, NTILE = 17;

__device__ __forceinline__ double fast_rcp(double d)
{
    double y = __builtin_amdgcn_rcp(d);
    y = fma(fma(-d, y, 1.0), y, y);
    y = fma(fma(-d, y, 1.0), y, y);
    return y;
}


struct Lds {
    double Z[MB * CW];
    double U[2 * US];
    double Dinv[MB + 8];
    double P[MB * CW];
    int prog[4];
};
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane); hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
#define FB(G) case G: asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #G " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(y), "v"(x)); break;
__device__ __forceinline__ void fmac_bcast(double &acc, double y, double x, int g) { switch (g) { FB(0) FB(1) FB(2) FB(3) FB(4) FB(5) FB(6) FB(7) default: break; } }

// MODE 0: barrier per step (shipped).  1: phase 1 only (one wave, M dependent steps).  2: phase 1, barrier, phase 2.  3: phase 2 follows by progress word.
// 4: phase 2 only (table already there).
template <int MODE, int M>
__global__ __launch_bounds__(512, 2) void k(double *out, int blocks, unsigned long long *cyc)
{
    __shared__ Lds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < MB * CW; i += 512) {
        const int row = i % CW, col = i / CW;
        L.P[i] = (row % RS == 3 && row / RS == col) ? 4.0 : 1e-3 * ((i % 13) - 6);
        L.Z[i] = 0.0;
    }
    if (tid < 2 * US) L.U[tid] = 1e-3;
    if (tid < MB + 8) L.Dinv[tid] = 0.25;
    if (tid < 4) L.prog[tid] = 0;
    __syncthreads();
    const bool idle_wave = __builtin_amdgcn_readfirstlane(wave >= 5 ? 1 : 0) != 0;
    const int rowc = tid / RS, rowrho = tid - rowc * RS;
    const int myj = (rowrho == 3 && rowc < M) ? rowc : -1;
    const int uslot = myj >= 0 ? myj : MB + lane;
    double sink = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int blk = 0; blk < blocks; ++blk) {
        if constexpr (MODE == 0) {
            double pv[M];
            if (!idle_wave) {
#pragma unroll
                for (int j = 0; j < M; ++j) pv[j] = L.P[j * CW + tid];
            }
#pragma unroll
            for (int s = 0; s < M; ++s) {
                if (!idle_wave) {
                    L.Z[s * CW + tid] = pv[s];
                    L.U[(s & 1) * US + uslot] = pv[s];
                    if (myj == s) L.Dinv[s] = fast_rcp(pv[s]);
                }
                __syncthreads();
                if (!idle_wave) {
                    double inv = L.Dinv[s], u[M];
#pragma unroll
                    for (int j = 0; j < M; ++j) u[j] = L.U[(s & 1) * US + j];
                    if ((pv[s] * pv[s]) * (inv * 1e-11) >= 1.0) sink += 1.0;
                    if (__builtin_amdgcn_readfirstlane(__double2hiint(inv)) & 0x7ff00000) {
                        const double ainv = fabs(inv), fz = -pv[s] * inv;
                        if (myj == s) {
#pragma unroll
                            for (int j = 0; j < M; ++j) pv[j] = (j == s) ? -inv : u[j] * ainv;
                        } else {
#pragma unroll
                            for (int j = 0; j < M; ++j) pv[j] = (j == s) ? pv[s] * ainv : fma(fz, u[j], pv[j]);
                        }
                    }
                }
            }
            if (!idle_wave) {
#pragma unroll
                for (int j = 0; j < M; ++j) sink += pv[j] * 1e-30;
            }
            __syncthreads();
        } else {
            if ((MODE == 3 || MODE == 5 || MODE == 8) && tid == 0) L.prog[0] = 0;
            __syncthreads();
            if ((MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5 || MODE == 8) && wave == 7) {
                double pv[M];
                const int rowpos = (lane < M ? lane : 0) * RS + 3;
#pragma unroll
                for (int j = 0; j < M; ++j) pv[j] = L.P[j * CW + rowpos];
#pragma unroll
                for (int s = 0; s < M; ++s) {
                    double zi = pv[s];
                    if (lane < M) L.U[s * MB + lane] = zi;
                    const double d = readlane_f64(zi, s);
                    const bool pre = d > 1e-11;
                    const double inv = pre ? fast_rcp(d) : 0.0;
                    if (lane == 0) L.Dinv[s] = inv;
                    if (MODE == 3 || MODE == 5 || MODE == 8) { asm volatile("" ::: "memory"); if (lane == 0) __hip_atomic_store(&L.prog[0], s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                    if (pre) {
                        const bool piv = lane == s;
                        double mi = piv ? fabs(inv) : -zi * inv;
                        if (piv) {
#pragma unroll
                            for (int j = 0; j < M; ++j) pv[j] = 0.0;
                        }
                        asm volatile("s_nop 1" : "+v"(mi), "+v"(zi));
#pragma unroll
                        for (int j = 0; j < M; ++j) if (j != s) fmac_bcast(pv[j], zi, mi, j);
                        pv[s] = piv ? -inv : zi * fabs(inv);
                    }
                }
#pragma unroll
                for (int j = 0; j < M; ++j) sink += pv[j] * 1e-30;
            }
            if ((MODE == 5 || MODE == 6) && !idle_wave) {
                double pv[M], zs[M];
#pragma unroll
                for (int j = 0; j < M; ++j) pv[j] = L.P[j * CW + tid];
                const volatile int *prog = &L.prog[0];
#pragma unroll
                for (int s = 0; s < M; ++s) {
                    double inv, u[M];
                    for (;;) {
                        const int pg = MODE == 5 ? *prog : M;            // issued FIRST: the LDS serves a wave's reads in order
                        asm volatile("" ::: "memory");
                        inv = L.Dinv[s];
#pragma unroll
                        for (int j = 0; j < M; ++j) u[j] = L.U[s * MB + j];
                        if (__builtin_amdgcn_readfirstlane(pg) > s) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    zs[s] = pv[s];
                    if ((pv[s] * pv[s]) * (inv * 1e-11) >= 1.0) sink += 1.0;
                    if (__builtin_amdgcn_readfirstlane(__double2hiint(inv)) & 0x7ff00000) {
                        const double ainv = fabs(inv), fz = -pv[s] * inv;
                        if (myj == s) {
#pragma unroll
                            for (int j = 0; j < M; ++j) pv[j] = (j == s) ? -inv : u[j] * ainv;
                        } else {
#pragma unroll
                            for (int j = 0; j < M; ++j) pv[j] = (j == s) ? pv[s] * ainv : fma(fz, u[j], pv[j]);
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < M; ++s) L.Z[s * CW + tid] = zs[s];
#pragma unroll
                for (int j = 0; j < M; ++j) sink += pv[j] * 1e-30;
            }
            if ((MODE == 7 || MODE == 8) && !idle_wave) {
                // software-pipelined: the table row of step s + 1 (and the progress word, read FIRST) is requested before step s is computed
                double pv[M], zs[M];
#pragma unroll
                for (int j = 0; j < M; ++j) pv[j] = L.P[j * CW + tid];
                const volatile int *prog = &L.prog[0];
                double invn, un[M];
                int pgn = MODE == 8 ? *prog : M;
                asm volatile("" ::: "memory");
                invn = L.Dinv[0];
#pragma unroll
                for (int j = 0; j < M; ++j) un[j] = L.U[j];
#pragma unroll
                for (int s = 0; s < M; ++s) {
                    while (__builtin_amdgcn_readfirstlane(pgn) <= s) {       // the prefetched row was not complete yet: ask again
                        __builtin_amdgcn_s_sleep(1);
                        pgn = *prog;
                        asm volatile("" ::: "memory");
                        invn = L.Dinv[s];
#pragma unroll
                        for (int j = 0; j < M; ++j) un[j] = L.U[s * MB + j];
                    }
                    const double inv = invn;
                    double u[M];
#pragma unroll
                    for (int j = 0; j < M; ++j) u[j] = un[j];
                    if (s + 1 < M) {
                        pgn = MODE == 8 ? *prog : M;
                        asm volatile("" ::: "memory");
                        invn = L.Dinv[s + 1];
#pragma unroll
                        for (int j = 0; j < M; ++j) un[j] = L.U[(s + 1) * MB + j];
                    }
                    zs[s] = pv[s];
                    if ((pv[s] * pv[s]) * (inv * 1e-11) >= 1.0) sink += 1.0;
                    if (__builtin_amdgcn_readfirstlane(__double2hiint(inv)) & 0x7ff00000) {
                        const double ainv = fabs(inv), fz = -pv[s] * inv;
                        if (myj == s) {
#pragma unroll
                            for (int j = 0; j < M; ++j) pv[j] = (j == s) ? -inv : u[j] * ainv;
                        } else {
#pragma unroll
                            for (int j = 0; j < M; ++j) pv[j] = (j == s) ? pv[s] * ainv : fma(fz, u[j], pv[j]);
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < M; ++s) L.Z[s * CW + tid] = zs[s];
#pragma unroll
                for (int j = 0; j < M; ++j) sink += pv[j] * 1e-30;
            }
            if (MODE == 2) __syncthreads();
            if ((MODE == 2 || MODE == 3 || MODE == 4) && !idle_wave) {
                double pv[M];
#pragma unroll
                for (int j = 0; j < M; ++j) pv[j] = L.P[j * CW + tid];
#pragma unroll
                for (int s = 0; s < M; ++s) {
                    if (MODE == 3) {
                        int spins = 0;
                        while (__hip_atomic_load(&L.prog[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= s && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(2);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    }
                    double inv = L.Dinv[s], u[M];
#pragma unroll
                    for (int j = 0; j < M; ++j) u[j] = L.U[s * MB + j];
                    L.Z[s * CW + tid] = pv[s];
                    if ((pv[s] * pv[s]) * (inv * 1e-11) >= 1.0) sink += 1.0;
                    if (__builtin_amdgcn_readfirstlane(__double2hiint(inv)) & 0x7ff00000) {
                        const double ainv = fabs(inv), fz = -pv[s] * inv;
                        if (myj == s) {
#pragma unroll
                            for (int j = 0; j < M; ++j) pv[j] = (j == s) ? -inv : u[j] * ainv;
                        } else {
#pragma unroll
                            for (int j = 0; j < M; ++j) pv[j] = (j == s) ? pv[s] * ainv : fma(fz, u[j], pv[j]);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < M; ++j) sink += pv[j] * 1e-30;
            }
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + tid] = sink;
    if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE, int M>
static void run(int blocks)
{
    double *out; unsigned long long *cyc, h;
    (void)hipMalloc(&out, sizeof(double) * 256 * 512); (void)hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, M>), dim3(256), dim3(512), 0, 0, out, 50, cyc); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, M>), dim3(256), dim3(512), 0, 0, out, blocks, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const char *names[] = {"barrier per step (shipped)", "phase 1 only (one wave)", "phase 1 | barrier | phase 2", "phase 2 follows by progress word", "phase 2 only", "follows, batched reads, late Z", "phase 2 only, batched, late Z", "phase 2 only, prefetched rows", "follows, prefetched rows"};
    printf("M %d  %-34s %8.3f ms  %7.0f shader cycles/block  %6.0f per step\n", M, names[MODE], ms, (double)h / blocks, (double)h / blocks / M);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main()
{
    const int nb = 20000;
    run<0, 8>(nb); run<1, 8>(nb); run<2, 8>(nb); run<3, 8>(nb); run<4, 8>(nb); run<5, 8>(nb); run<6, 8>(nb); run<7, 8>(nb); run<8, 8>(nb);
    run<0, 6>(nb); run<1, 6>(nb); run<2, 6>(nb); run<3, 6>(nb); run<4, 6>(nb); run<5, 6>(nb); run<6, 6>(nb); run<7, 6>(nb); run<8, 6>(nb);
    run<0, 3>(nb); run<1, 3>(nb); run<2, 3>(nb); run<3, 3>(nb); run<4, 3>(nb); run<5, 3>(nb); run<6, 3>(nb); run<7, 3>(nb); run<8, 3>(nb);
    return 0;
}
