// Microbenchmarks that calibrate the cost model of the sweep kernel (cycles per primitive, one 512-thread workgroup per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(double *out, unsigned long long *cyc, int iters)
{
    __shared__ double lds[2048];
    const int tid = threadIdx.x;
    double x = 1.0 + tid * 1e-3, y = 0.5, z = 0.25;
    lds[tid] = x; lds[tid + 512] = y;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { __syncthreads(); }                                   // bare barrier
        else if (MODE == 1) { lds[tid] = x; __syncthreads(); x = lds[(tid + 1) & 511] + 1e-9; }   // write -> barrier -> read
        else if (MODE == 2) { x = fma(x, 1.0000001, 1e-9); }                 // dependent fp64 fma chain
        else if (MODE == 3) { x = __builtin_amdgcn_rcp(x) + 1.5; }           // dependent rcp
        else if (MODE == 4) { x = lds[((int)x) & 511] + 1.0; }               // dependent LDS read
        else if (MODE == 5) { if (__builtin_amdgcn_readfirstlane(i) & 1) x += 1e-9; else y += 1e-9; }  // uniform branch
        else if (MODE == 6) {                                                  // 16 independent fmas
            double a0=x,a1=y,a2=z,a3=x+1,a4=y+1,a5=z+1,a6=x+2,a7=y+2;
            #pragma unroll
            for (int r = 0; r < 2; ++r) { a0=fma(a0,y,z);a1=fma(a1,y,z);a2=fma(a2,y,z);a3=fma(a3,y,z);a4=fma(a4,y,z);a5=fma(a5,y,z);a6=fma(a6,y,z);a7=fma(a7,y,z); }
            x = a0+a1+a2+a3+a4+a5+a6+a7;
        }
        else if (MODE == 7) { lds[tid] = x; lds[tid + 512] = y; lds[tid + 1024] = z; lds[tid+1536] = x; __syncthreads(); x = lds[(tid + 1) & 511]; } // 4 writes+barrier+read
        else if (MODE == 8) { x += __shfl(x, 5); }                           // ds_bpermute / readlane path
        else if (MODE == 9) { unsigned long long b = __ballot(x > 0.0); x += (double)(b & 1); }
        else if (MODE == 10) {                                                 // flat chain of 16 independent ifs, one hit
            const int kap = __builtin_amdgcn_readfirstlane(i & 15);
            #pragma unroll
            for (int c = 0; c < 16; ++c) if (kap == c) x = fma(x, 1.0 + c * 1e-9, y);
        }
        else if (MODE == 11) {                                                 // same, bodies marked unlikely (out of line)
            const int kap = __builtin_amdgcn_readfirstlane(i & 15);
            #pragma unroll
            for (int c = 0; c < 16; ++c) if (__builtin_expect(kap == c, 0)) { x = fma(x, 1.0 + c * 1e-9, y); lds[(tid + c) & 2047] = x; }
        }
        else if (MODE == 12) {                                                 // binary tree, 4 levels
            const int kap = __builtin_amdgcn_readfirstlane(i & 15);
            if (kap < 8) { if (kap < 4) { if (kap < 2) { if (kap < 1) x = fma(x, 1.0000001, y); else x = fma(x, 1.0000002, y); } else { if (kap < 3) x = fma(x, 1.0000003, y); else x = fma(x, 1.0000004, y); } }
                           else { if (kap < 6) { if (kap < 5) x = fma(x, 1.0000005, y); else x = fma(x, 1.0000006, y); } else { if (kap < 7) x = fma(x, 1.0000007, y); else x = fma(x, 1.0000008, y); } } }
            else { if (kap < 12) { if (kap < 10) { if (kap < 9) x = fma(x, 1.0000009, y); else x = fma(x, 1.000001, y); } else { if (kap < 11) x = fma(x, 1.0000011, y); else x = fma(x, 1.0000012, y); } }
                   else { if (kap < 14) { if (kap < 13) x = fma(x, 1.0000013, y); else x = fma(x, 1.0000014, y); } else { if (kap < 15) x = fma(x, 1.0000015, y); else x = fma(x, 1.0000016, y); } } }
        }
        else if (MODE == 13) {                                                 // switch with distinct bodies
            const int kap = __builtin_amdgcn_readfirstlane(i & 15);
            switch (kap) {
                case 0: x = fma(x, 1.0000001, y); lds[tid] = x; break; case 1: x = fma(x, 1.0000002, z); lds[tid+1] = x; break;
                case 2: x = fma(x, 1.0000003, y); lds[tid+2] = x; break; case 3: x = fma(x, 1.0000004, z); lds[tid+3] = x; break;
                case 4: x = fma(x, 1.0000005, y); lds[tid+4] = x; break; case 5: x = fma(x, 1.0000006, z); lds[tid+5] = x; break;
                case 6: x = fma(x, 1.0000007, y); lds[tid+6] = x; break; case 7: x = fma(x, 1.0000008, z); lds[tid+7] = x; break;
                case 8: x = fma(x, 1.0000009, y); lds[tid+8] = x; break; case 9: x = fma(x, 1.000001, z); lds[tid+9] = x; break;
                case 10: x = fma(x, 1.0000011, y); lds[tid+10] = x; break; case 11: x = fma(x, 1.0000012, z); lds[tid+11] = x; break;
                case 12: x = fma(x, 1.0000013, y); lds[tid+12] = x; break; case 13: x = fma(x, 1.0000014, z); lds[tid+13] = x; break;
                case 14: x = fma(x, 1.0000015, y); lds[tid+14] = x; break; default: x = fma(x, 1.0000016, z); lds[tid+15] = x; break;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + tid] = x + y + z;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    const int iters = 20000, blocks = 256;
    double *out; unsigned long long *cyc;
    CK(hipMalloc(&out, blocks * 512 * sizeof(double)));
    CK(hipMalloc(&cyc, blocks * sizeof(unsigned long long)));
    const char *names[] = {"bare __syncthreads (8 waves)", "lds write -> barrier -> read", "dependent v_fma_f64", "dependent v_rcp_f64 + add",
                           "dependent ds_read_b64", "uniform branch (readfirstlane + s_cbranch)", "16 independent v_fma_f64 (+7 adds)",
                           "4 lds writes -> barrier -> read", "__shfl (bpermute)", "ballot + cvt", "flat chain of 16 ifs (1 hit)", "flat chain, bodies unlikely", "binary tree 4 levels", "switch 16 cases"};
    for (int mode = 0; mode < 14; ++mode) {
        switch (mode) {
#define L(M) case M: hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(512), 0, 0, out, cyc, iters); break;
            L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13)
        }
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(blocks);
        CK(hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double s = 0; for (auto v : h) s += (double)v;
        printf("%-45s %8.1f cycles/iter\n", names[mode], s / blocks / iters);
    }
    return 0;
}
