// panel_mfma_overlap.hip — the gate of round 4's sweep question (task 1): can the trailing update of a block run on the fp64
// MFMA pipe UNDER the next block's panel steps?  The production occupancy is reproduced: 512-thread workgroups, one per CU, 2 waves per
// SIMD, 17 whole 16x16 tiles per wave (68 accumulator doubles per lane, C/D map col = lane & 15, row = (lane >> 4) + 4 reg).
//   MODE 0  panel only : the shipped panel step of sweep_blk.hip (store pivot-column entry + pivot-row entry, the one pivot thread's
//                        reciprocal, barrier, broadcast reads, row update; waves 5..7 own no row and only count barriers), blocks of M steps
//   MODE 1  MFMA only  : per block every wave applies ceil(M / 4) k-steps to its 17 tiles: A / B fragments from the LDS image of Z
//                        (B scaled by -1/d), v_mfma_f64_16x16x4_f64
//   MODE 2  both, MFMAs of the PREVIOUS block dealt over the steps, issued right after the step's broadcast reads (while they fly)
//   MODE 3  both, MFMAs issued between the step's stores and its barrier (where the non-pivot waves wait for the reciprocal)
// Output: ms, cycles per step at the clock measured by s_memtime (100 MHz) against wall time, MFMA pipe utilisation = MFMAs x 64 cycles /
// (2 waves ... per SIMD) / elapsed cycles.  Build: hipcc -O3 --offload-arch=gfx950 panel_mfma_overlap.hip -o bin/panel_mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int CW = 513, RS = 17, MB = 8, US = MB + 64, NTILE = 17;

__device__ __forceinline__ double fast_rcp(double d)
{
    double y = __builtin_amdgcn_rcp(d);
    y = fma(fma(-d, y, 1.0), y, y);
    y = fma(fma(-d, y, 1.0), y, y);
    return y;
}

struct Lds {
    double Z[2][MB * CW];
    double U[2 * US];
    double Dinv[2][MB + 8];
    double P[MB * CW];
};

// one k-step (4 pending pivots) on tiles [T0, T1) of this wave
template <int T0, int T1>
__device__ __forceinline__ void mfma_tiles(double4_t (&acc)[NTILE], const double *Zs, const double *Dv, int ks, const int (&offA)[NTILE], const int (&offB)[NTILE], int kq)
{
    const double ninv = -Dv[4 * ks + kq];
    const double *Zk = Zs + 4 * ks * CW;
#pragma unroll
    for (int i = T0; i < T1; ++i) {
        const double af = Zk[offA[i]];
        const double bf = Zk[offB[i]] * ninv;
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[i], 0, 0, 0);
    }
}

template <int MODE, int M>
__global__ __launch_bounds__(512, 2) void k(double *out, int blocks, unsigned long long *cyc)
{
    __shared__ Lds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < MB * CW; i += 512) {
        const int row = i % CW, col = i / CW;
        const double v = (row % RS == 3 && row / RS == col) ? 4.0 : 1e-3 * ((i % 13) - 6);
        L.P[i] = v; L.Z[0][i] = v; L.Z[1][i] = v;
    }
    if (tid < 2 * US) L.U[tid] = 0.0;
    if (tid < 2 * (MB + 8)) (&L.Dinv[0][0])[tid] = 0.25;
    __syncthreads();
    double4_t acc[NTILE];
#pragma unroll
    for (int i = 0; i < NTILE; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    // tiles of this wave: slots 17 wave .. 17 wave + 16 of the column-major upper triangle (slot = tri(gamma) + rho), as a whole-tile layout would deal them
    int offA[NTILE], offB[NTILE];
    const int kq = lane >> 4;
#pragma unroll
    for (int i = 0; i < NTILE; ++i) {
        const int slot = 17 * wave + i;
        int gam = 0;
        while ((gam + 1) * (gam + 2) / 2 <= slot) ++gam;
        const int rho = slot - gam * (gam + 1) / 2;
        offA[i] = kq * CW + (lane & 15) * RS + rho;
        offB[i] = kq * CW + (lane & 15) * RS + gam;
    }
    const bool idle_wave = __builtin_amdgcn_readfirstlane(wave >= 5 ? 1 : 0) != 0;
    const int rowc = tid / RS, rowrho = tid - rowc * RS;
    const int myj = (rowrho == 3 && rowc < M) ? rowc : -1;
    const int uslot = myj >= 0 ? myj : MB + lane;
    constexpr int KS = (M + 3) / 4;
    double sink = 0.0;
    double ra = 1e-3 * lane, rb = 1.0 - 1e-4 * lane;
    asm volatile("" : "+v"(ra), "+v"(rb));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int blk = 0; blk < blocks; ++blk) {
        const int par = blk & 1;
        double *Z = L.Z[par];
        const double *Zp = L.Z[par ^ 1];                 // the pending block's columns
        const double *Dp = L.Dinv[par ^ 1];
        double *Dinv = L.Dinv[par];
        if constexpr (MODE == 1) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) mfma_tiles<0, NTILE>(acc, Zp, Dp, ks, offA, offB, kq);
            continue;
        }
        double pv[M];
        if (!idle_wave) {
#pragma unroll
            for (int j = 0; j < M; ++j) pv[j] = L.P[j * CW + tid];
        }
#pragma unroll
        for (int s = 0; s < M; ++s) {
            // MFMA work of the pending block dealt over the M steps: KS * 17 tile-steps in M portions
            constexpr int TOT = KS * NTILE;
            const int lo = s * TOT / M, hi = (s + 1) * TOT / M;
            if (!idle_wave) {
                Z[s * CW + tid] = pv[s];
                L.U[(s & 1) * US + uslot] = pv[s];
                if (myj == s) Dinv[s] = fast_rcp(pv[s]);
            }
            if constexpr (MODE == 8 || MODE == 9) {         // MFMAs only from the waves that own no panel row (8: waves 5..7; 9: wave 5 only)
                if (idle_wave && (MODE == 8 || wave == 5)) {
#pragma unroll
                    for (int w = lo; w < hi; ++w) {
                        const int i = w % NTILE;
                        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra, rb, acc[i], 0, 0, 0);
                    }
                }
            }
            if constexpr (MODE == 6) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int w = lo; w < hi; ++w) {
                    const int i = w % NTILE;
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra, rb, acc[i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (MODE == 3) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int w = lo; w < hi; ++w) {
                    const int ks = w / NTILE, i = w % NTILE;
                    const double ninv = -Dp[4 * ks + kq];
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(Zp[4 * ks * CW + offA[i]], Zp[4 * ks * CW + offB[i]] * ninv, acc[i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            double inv = 0.0, u[M];
            if (!idle_wave) {
                inv = Dinv[s];
#pragma unroll
                for (int j = 0; j < M; ++j) u[j] = L.U[(s & 1) * US + j];
                asm volatile("" : "+v"(inv));
#pragma unroll
                for (int j = 0; j < M; ++j) asm volatile("" : "+v"(u[j]));
            }
            if constexpr (MODE == 5 || MODE == 7) {         // operands in registers: the pure issue / pipe interaction, no LDS traffic
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int w = lo; w < hi; ++w) {
                    const int i = w % NTILE;
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra, rb, acc[i], 0, 0, 0);
                    if constexpr (MODE == 7) __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (MODE == 2) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int w = lo; w < hi; ++w) {
                    const int ks = w / NTILE, i = w % NTILE;
                    const double ninv = -Dp[4 * ks + kq];
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(Zp[4 * ks * CW + offA[i]], Zp[4 * ks * CW + offB[i]] * ninv, acc[i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!idle_wave) {
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
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = sink;
#pragma unroll
    for (int i = 0; i < NTILE; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + tid] = s;
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
    const double ticks = (double)h;                                 // 100 MHz
    const double ghz = 2.1;                                         // what the sweep holds; only for the cycle columns
    const double cyc_per_block = ms * 1e-3 * ghz * 1e9 / blocks;
    constexpr int KS = (M + 3) / 4;
    const double mfma_cycles_per_simd = (MODE == 0) ? 0.0 : (MODE == 8 || MODE == 9 ? 1.0 : 2.0) * KS * NTILE * 64.0;   // busiest SIMD   // 2 waves per SIMD
    printf("mode %d M %d: %8.3f ms  %7.0f cycles/block @2.1GHz  %6.0f cycles/step  (s_memtime %.1f ticks/block)  MFMA pipe %4.1f %%  %5.1f TFLOP/s on the pipe\n",
           MODE, M, ms, cyc_per_block, cyc_per_block / M, ticks / blocks, 100.0 * mfma_cycles_per_simd / cyc_per_block,
           MODE == 0 ? 0.0 : (double)blocks * KS * NTILE * 8 * 256 * 2048.0 / (ms * 1e-3) / 1e12);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    const int nb = 20000;
    run<0, 8>(nb); run<1, 8>(nb); run<2, 8>(nb); run<3, 8>(nb);
    run<0, 6>(nb); run<1, 6>(nb); run<2, 6>(nb); run<3, 6>(nb);
    run<0, 4>(nb); run<1, 4>(nb); run<2, 4>(nb); run<3, 4>(nb);
    run<5, 8>(nb); run<6, 8>(nb); run<7, 8>(nb); run<5, 4>(nb); run<6, 4>(nb);
    run<8, 8>(nb); run<9, 8>(nb); run<8, 4>(nb);
    return 0;
}
