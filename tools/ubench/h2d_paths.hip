// h2d_paths.hip — what the host-pointer entry of the library can expect from this box's PCIe path (round 4, VERDICT task 4):
//   (a) hipMemcpy2DAsync from PAGEABLE memory (what ctx_prepare did through round 3),
//   (b) hipHostRegister of the caller's array + async copy + hipHostUnregister,
//   (c) staging through two pinned buffers: T host threads copy row chunks into the pinned buffer, DMA of chunk i overlaps the memcpy of i + 1,
//   (d) DMA from memory that is already pinned (the ceiling of the link).
// Build: hipcc -O3 --offload-arch=gfx950 h2d_paths.hip -o bin/h2d_paths -lpthread ; run: bin/h2d_paths [N] [M]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static void par_copy(char *dst, const char *src, size_t bytes, int T)
{
    if (T <= 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t per = (bytes / T + 4095) & ~(size_t)4095;
    for (int t = 0; t < T; ++t) {
        const size_t o = (size_t)t * per;
        if (o >= bytes) break;
        const size_t n = o + per < bytes ? per : bytes - o;
        th.emplace_back([=] { memcpy(dst + o, src + o, n); });
    }
    for (auto &w : th) w.join();
}
int main(int argc, char **argv)
{
    const size_t N = argc > 1 ? atoll(argv[1]) : 1000000, M = argc > 2 ? atoll(argv[2]) : 512;
    const size_t bytes = N * M * 8;
    printf("N %zu M %zu: %.1f MB\n", N, M, bytes / 1e6);
    double *h = (double *)malloc(bytes);
    for (size_t i = 0; i < N * M; i += 512) h[i] = (double)i;          // touch every page
    double *d; CK(hipMalloc(&d, bytes));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CK(hipMemcpy2DAsync(d, N * 8, h, N * 8, N * 8, M, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
        double t1 = now();
        printf("(a) pageable hipMemcpy2DAsync            : %8.2f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
        double t2 = now();
        CK(hipHostUnregister(h));
        double t3 = now();
        printf("(b) register %.2f + copy %.2f + unregister %.2f = %8.2f ms  %6.1f GB/s overall, %6.1f GB/s the DMA\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3,
               (t3 - t0) * 1e3, bytes / (t3 - t0) / 1e9, bytes / (t2 - t1) / 1e9);
    }
    const size_t CH = (size_t)64 << 20;
    char *pin[2]; CK(hipHostMalloc((void **)&pin[0], CH, hipHostMallocDefault)); CK(hipHostMalloc((void **)&pin[1], CH, hipHostMallocDefault));
    hipEvent_t ev[2]; CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
    for (int T : {1, 2, 4, 8}) {
        double t0 = now();
        size_t off = 0; int b = 0; bool used[2] = {false, false};
        while (off < bytes) {
            const size_t n = off + CH < bytes ? CH : bytes - off;
            if (used[b]) CK(hipEventSynchronize(ev[b]));
            par_copy(pin[b], (const char *)h + off, n, T);
            CK(hipMemcpyAsync((char *)d + off, pin[b], n, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(ev[b], s)); used[b] = true;
            off += n; b ^= 1;
        }
        CK(hipStreamSynchronize(s));
        double t1 = now();
        printf("(c) pinned double buffer, %d memcpy thread(s): %8.2f ms  %6.1f GB/s\n", T, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    {
        double t0 = now();
        for (size_t off = 0; off < bytes; off += CH) { const size_t n = off + CH < bytes ? CH : bytes - off; CK(hipMemcpyAsync((char *)d + off, pin[0], n, hipMemcpyHostToDevice, s)); }
        CK(hipStreamSynchronize(s));
        double t1 = now();
        printf("(d) DMA from pinned memory                 : %8.2f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    return 0;
}
