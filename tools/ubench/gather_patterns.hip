// gather_patterns.hip — what a 64-lane 8-byte global load costs a CU as a function of its ADDRESS PATTERN (round 4: the gather and the
// replaced-row stores of csrc/sweep_lazy.hip; a duplicated gather doubled that phase, two re-layouts that cut its cache lines 2.6x - 4x
// made it slower — so lines are not the unit).  One 512-thread workgroup per CU, each on its own image of `ld` x `ld` doubles (0.9 MB at
// ld = 342: far beyond the CU's L1, beyond its share of L2); every wave issues batches of 12 loads (3 tiles x 4 registers, as the gather
// does) and waits for them; the base row / column moves from batch to batch.  Patterns (lane = 16 fk + fr):
//   contig   : 64 consecutive doubles of one row                                     (1 segment of 512 B)
//   colpart  : rows r0 + fk + 4 r (4 rows), columns k0 + fr (16 adjacent)            (4 segments of 128 B)   — gather above the diagonal, adjacent pivots
//   colsplit : the same with the 16 columns 3 apart                                  (64 segments of 8 B in 4 rows)
//   rowpart  : rows k0 + fr (16 rows), columns r0 + fk + 4 r (4 adjacent... per r 1) (16 segments of 8 B per instruction; over the 4 r: 32 B)
//   column   : 64 consecutive rows, one column                                       (64 segments of 8 B)  — the replaced-row stores, thread per row
//   subblock : tile-major 4 x 4 sub-blocks (the layout tried in round 4): 16 columns x 4 rows = 4 x 4 segments of 32 B
// Output: nanoseconds per load instruction and CU (all CUs running), and the same for stores.
// Build: hipcc -O3 --offload-arch=gfx950 gather_patterns.hip -o bin/gather_patterns ; run: bin/gather_patterns [ld]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int PAT, bool STORE>
__global__ __launch_bounds__(512) void kern(double *img, int ld, int batches, double *sink)
{
    double *T = img + (size_t)blockIdx.x * ld * ld;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, fr = lane & 15, fk = lane >> 4;
    double acc = 0.0;
    unsigned seed = 1234567u + 977u * blockIdx.x + 31u * wv;
    for (int b = 0; b < batches; ++b) {
        seed = seed * 1664525u + 1013904223u;
        const int r0 = (int)((seed >> 8) % (unsigned)(ld - 80)), k0 = (int)((seed >> 4) % (unsigned)(ld - 80));
        double v[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int u = i >> 2, r = i & 3;
            size_t a;
            if (PAT == 0) a = (size_t)(r0 + i) * ld + (k0 & ~15) + lane;                                     // contig
            else if (PAT == 1) a = (size_t)(r0 + 16 * u + fk + 4 * r) * ld + k0 + fr;                          // colpart, adjacent pivots
            else if (PAT == 2) a = (size_t)(r0 + 16 * u + fk + 4 * r) * ld + k0 + 3 * fr;                      // colpart, pivots 3 apart
            else if (PAT == 3) a = (size_t)(k0 + fr) * ld + r0 + 16 * u + fk + 4 * r;                          // rowpart
            else if (PAT == 4) a = (size_t)((r0 + 64 * u + lane) % ld) * ld + k0 + r;                          // column (thread per row)
            else {                                                                                            // 4 x 4 sub-blocks, tile-major
                const int row = (r0 & ~15) + 16 * u + fk + 4 * r, col = (k0 & ~15) + fr;                      //   (one tile, 16 columns x 4 rows)
                a = ((size_t)((row >> 4) * (ld >> 4) + (col >> 4)) << 8) + (size_t)((((row & 15) >> 2) << 6) + (((col & 15) >> 2) << 4) + ((row & 3) << 2) + (col & 3));
            }
            if (STORE) T[a] = (double)(b + i);
            else v[i] = __builtin_nontemporal_load(&T[a]);
        }
        if (!STORE) {
#pragma unroll
            for (int i = 0; i < 12; ++i) acc += v[i];
        } else __builtin_amdgcn_s_waitcnt(0);                                                                  // (stores: drained per batch like the loads)
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

template <int PAT, bool STORE>
static void run(const char *name, double *img, int ld, int ncu, double *sink)
{
    const int batches = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((kern<PAT, STORE>), dim3(ncu), dim3(512), 0, 0, img, ld, 50, sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((kern<PAT, STORE>), dim3(ncu), dim3(512), 0, 0, img, ld, batches, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (double)batches * 12 * 8;
    printf("%-9s %-6s %8.1f ns per instruction and CU  (%7.1f cycles at 2.4 GHz; %6.2f us per batch of 8 x 12)\n", name, STORE ? "store" : "load", ms * 1e6 / instr_per_cu,
           ms * 1e6 / instr_per_cu * 2.4, ms * 1e3 / batches);
}

int main(int argc, char **argv)
{
    const int ld = argc > 1 ? atoi(argv[1]) : 352;                   // (a multiple of 16 for the sub-block pattern)
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int ncu = pr.multiProcessorCount;
    double *img, *sink;
    CK(hipMalloc(&img, (size_t)ncu * ld * ld * 8)); CK(hipMemset(img, 0, (size_t)ncu * ld * ld * 8)); CK(hipMalloc(&sink, 64));
    printf("%d CUs, one 512-thread workgroup each on its own %d x %d image (%.2f MB)\n", ncu, ld, ld, ld * ld * 8 / 1e6);
    run<0, false>("contig", img, ld, ncu, sink);   run<1, false>("colpart", img, ld, ncu, sink);  run<2, false>("colsplit", img, ld, ncu, sink);
    run<3, false>("rowpart", img, ld, ncu, sink);  run<4, false>("column", img, ld, ncu, sink);   run<5, false>("subblock", img, ld, ncu, sink);
    run<0, true>("contig", img, ld, ncu, sink);    run<1, true>("colpart", img, ld, ncu, sink);   run<3, true>("rowpart", img, ld, ncu, sink);
    run<4, true>("column", img, ld, ncu, sink);    run<5, true>("subblock", img, ld, ncu, sink);
    return 0;
}
