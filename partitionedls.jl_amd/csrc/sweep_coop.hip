// sweep_coop.hip — ONE large subproblem solved by many workgroups (grid-wide barriers), BLOCKED pivots.
//
// Used for single "node" solves whose tableau does not fit the register-resident kernel (n > 320): the α-step of
// fit(Alt) at BASELINE config 4 (n = 513, Alt.jl:80-90) and the winner re-solve of fit(Opt) at such sizes.  Same
// algorithm and decisions as sweep_generic.hip; the (n+1)^2 tableau lives in global memory (2.1 MB at n = 513,
// L2 resident) and every workgroup owns a slice of its ROWS.  All control state (basis flags, rejections, the violator
// list) is replicated per workgroup and evolves identically everywhere because every workgroup reads the same data after
// each grid barrier — no cross-workgroup messages besides the tableau itself.
//
// The violators of one KKT scan are exchanged in blocks of up to mb <= 16 pivots (as in sweep_blk.hip):
//   1. panel   : every workgroup copies the block's pivot rows (= columns, by symmetry) from global memory to LDS, P[mb][ld];
//   2. factor  : every workgroup eliminates the panel REDUNDANTLY — mb sequential Gauss–Jordan steps with block-local
//                barriers only; column s as of its own step is kept in Z[s][.], 1/d_s in dinv[s] (0 = rejected as dependent);
//   3. update  : each workgroup applies the fused rank-m update T_ic -= sum_s Z_s[i] Z_s[c] / d_s to ITS rows, then overwrites
//                the rows / columns of the pivoted variables from the final panel;
//   4. ONE grid barrier.  The update reads image `cur` of the tableau and writes image `cur ^ 1` (every row has exactly one owner,
//      so the other image is complete after the barrier): nobody rewrites a row that a slower workgroup may still be loading into
//      its panel or scanning, which in the in-place form needed a second grid barrier per block (~10 us each at 64 workgroups).
// So the grid synchronises once per block instead of twice per pivot (the first α-step of Alt at C4 exchanges ~250 variables in 17 blocks).
//
// Grid barrier.  An ordinary launch with a hand-written barrier on a monotone global counter (release add / acquire poll at
// agent scope), not hipLaunchCooperativeKernel: the runtime's cooperative queue cost ~7 ms on its first use in a process and,
// under rocprofv3, crashed the process inside exit() after the tool had finalised (round-1 finding, reproduced in round 2 with
// every context closed and every buffer freed: profiles/README.md).  What the cooperative launch guaranteed — that all
// workgroups are resident at once, so that nobody spins on a workgroup that cannot start — is ensured by the host instead:
// the grid is capped at (compute units) x (resident workgroups per CU from the occupancy query), far above the <= 128 used.
// The query cannot see other work on the device: a barrier that is not completed within 2 s of wall clock aborts the kernel and
// the host repeats the solve on the one-workgroup kernel (see grid_barrier).
#include "gj_panel.h"

namespace partls {

// All threads of all workgroups call it the same number of times.  `epoch` counts arrivals expected so far (this workgroup's
// private copy); ctr[0] is the arrival counter, ctr[1] the abort word; both are zeroed by the host before the launch and never
// reset inside the kernel.  The spin is bounded by WALL CLOCK (cdna_hip_programming.md §1): s_memrealtime runs at a constant
// 100 MHz, a workgroup that has waited 2 s gives up, raises the abort word — every other workgroup, resident now or scheduled
// later, then leaves at its next poll instead of serving its own 2 s — and the kernel exits with n_unconverged poisoned.  That
// can only happen when some workgroups of the grid are not resident (another process or context filled the CUs the occupancy
// query counted as free): the host then repeats the solve on the one-workgroup kernel (solve_nodes, api.hip).
static constexpr unsigned long long GRID_BARRIER_TIMEOUT_TICKS = 200000000ull;   // 2 s of s_memrealtime
__device__ __forceinline__ bool grid_barrier(unsigned *ctr, unsigned nwg, unsigned &epoch, int *s_ok)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // every wave's stores have left the wave ...
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch += nwg;
        // One release before the arrival and ONE acquire after the poll (MI355X_MICROARCH.md: an agent release writes back this
        // XCD's dirty L2 lines, ~1.7-6.5 us; an agent acquire invalidates this CU's L1, ~1.7 us; a __threadfence() is both, and
        // polling with acquire loads is 2-3x slower per hop than a relaxed poll followed by one fence).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");         // ... and are visible at agent scope before the arrival;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the compiler may drop the wait behind the write-back (ROCm 7.2)
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        unsigned long long t0 = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            __builtin_amdgcn_s_sleep(4);
            if ((++spins & 1023u) == 0) {                          // every ~1000 polls: the abort word and the clock
                if (__hip_atomic_load(ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { *s_ok = 0; break; }
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (t0 == 0) t0 = now;
                else if (now - t0 > GRID_BARRIER_TIMEOUT_TICKS) {
                    __hip_atomic_store(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *s_ok = 0;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");         // drops this CU's stale L1 lines (and the other XCDs' lines in L2)
    }
    __syncthreads();
    return *s_ok != 0;
}

static constexpr int COOP_THREADS = 1024;  // 16 waves per CU (512 threads, spill-free, is slower: 1.63 vs 1.39 ms per alpha-step at C4)
static constexpr int COOP_MAXWORDS = 16;
static constexpr int COOP_MB = GJ_MB;


__global__ __launch_bounds__(COOP_THREADS) void sweep_coop_kernel(SweepParams p, int mb)
{
    unsigned epoch = 0;                                            // grid-barrier arrivals expected so far
    const int n = p.n, ld = n + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    extern __shared__ double smem[];
    double *Pn = smem;                                            // [mb][ld] panel: the block's pivot columns, all rows
    double *Zn = Pn + (size_t)mb * ld;                            // [mb][ld] column s as of its own step
    double *dinv = Zn + (size_t)mb * ld;                          // [COOP_MB] 1/d_s (0: rejected)
    double *uj = dinv + COOP_MB;                                  // [2][COOP_MB] pivot-row entries of the current / next step
    double *red = uj + 2 * COOP_MB;                                   // [COOP_THREADS / 64] reduction scratch
    uint8_t *s_basic = reinterpret_cast<uint8_t *>(red + COOP_THREADS / 64);
    uint8_t *s_blocked = s_basic + n;
    __shared__ unsigned long long s_inf[COOP_MAXWORDS];
    __shared__ int s_viol[COOP_MAXWORDS * 64];
    __shared__ int s_nv;
    __shared__ int s_ok;                                           // 0 after a grid-barrier timeout
    if (threadIdx.x == 0) s_ok = 1;

    double *Timg[2] = {p.scratch, p.scratch + (size_t)ld * ld};    // two tableau images (ping-pong, see the file header)
    const int nwg = gridDim.x, wg = blockIdx.x;
    const unsigned nexp = (unsigned)(nwg + p.coop_fault);           // arrivals a barrier waits for (coop_fault: test hook, see SweepParams)
    const int rows_per = (ld + nwg - 1) / nwg;
    const int row0 = wg * rows_per, row1 = (row0 + rows_per < ld) ? row0 + rows_per : ld;
    const int nwords = (n + 63) >> 6;

    uint8_t *flagbuf = reinterpret_cast<uint8_t *>(p.scratch + (size_t)2 * ld * ld);   // basis flags + current image, kept across launches
    int cur = 0;
    if (!p.resume) {
        for (int i = row0 + wave; i < row1; i += COOP_THREADS / 64)
            for (int j = lane; j < ld; j += 64) Timg[0][(size_t)i * ld + j] = p.T0[(size_t)i * ld + j];
        for (int i = tid; i < n; i += COOP_THREADS) { s_basic[i] = 0; s_blocked[i] = 0; }
    } else {
        for (int i = tid; i < n; i += COOP_THREADS) { s_basic[i] = flagbuf[i]; s_blocked[i] = 0; }
        cur = flagbuf[n] & 1;
    }
    const int8_t *code = p.node_code;                              // one node: per-variable constraint codes
    if (!grid_barrier(p.grid_ctr, nexp, epoch, &s_ok)) { if (tid == 0 && p.n_unconverged) atomicAdd(p.n_unconverged, 1ULL << 40); return; }

    unsigned long long npiv = 0, nunconv = 0, nblk = 0;
    int ninf_best = n + 1, patience = 3, rounds = 0;
    bool progress = false;
    for (;;) {
        if (progress) { for (int i = tid; i < n; i += COOP_THREADS) s_blocked[i] = 0; __syncthreads(); }
        progress = false;
        // ---- KKT scan of the rhs row (every workgroup, identically) ------------------------------------------------------
        for (int base = 0; base < nwords * 64; base += COOP_THREADS) {
            const int v = base + tid;
            bool bad = false;
            if (v < n) {
                const double q = __builtin_nontemporal_load(&Timg[cur][(size_t)n * ld + v]);
                const int cd = (int)code[v];
                const int f = cd == 2 ? 0 : cd;
                const double fq = (f > 0) ? q : ((f < 0) ? -q : 0.0);
                if (cd == 2) bad = !s_basic[v] && !s_blocked[v] && (fabs(q) > p.tol);
                else if (s_basic[v]) bad = (f == 0) || (fq < -p.tol);
                else bad = (fq > p.tol) && !s_blocked[v];
            }
            const unsigned long long b = __ballot(bad);
            if (lane == 0 && (v >> 6) < nwords) s_inf[v >> 6] = b;
        }
        __syncthreads();
        int count = 0;
        for (int w = 0; w < nwords; ++w) count += __popcll(s_inf[w]);
        if (count == 0) break;
        bool all;
        if (count < ninf_best) { ninf_best = count; patience = 3; all = true; }
        else if (patience > 0) { --patience; all = true; }
        else all = false;                                         // backup rule: only the largest violator
        if (++rounds > p.max_rounds) { ++nunconv; break; }
        // the violator list of this round (ascending; the backup rule keeps only the last one)
        if (tid == 0) {
            int nv = 0;
            for (int w = 0; w < nwords; ++w) {
                unsigned long long bits = s_inf[w];
                while (bits) { s_viol[nv++] = (w << 6) + __builtin_ctzll(bits); bits &= bits - 1; }
            }
            if (!all) { s_viol[0] = s_viol[nv - 1]; nv = 1; }
            s_nv = nv;
        }
        __syncthreads();
        const int nv = s_nv;

        for (int b0 = 0; b0 < nv; b0 += mb) {
            const int m = (nv - b0 < mb) ? nv - b0 : mb;
            const int *ks = s_viol + b0;
            // ---- 1. panel, 2. redundant elimination (block-local barriers only), 3. fused update of the owned rows ------------
            gj_panel_load<COOP_THREADS>(Timg[cur], ld, ks, m, Pn, tid);
            gj_panel_eliminate<COOP_THREADS>(Pn, Zn, dinv, uj, red, ks, m, ld, s_basic, p.piv_eps, tid);
            gj_apply<COOP_THREADS>(Timg[cur ^ 1], ld, row0, row1, Pn, Zn, dinv, ks, m, tid, Timg[cur]);
            if (tid == 0) {
                for (int j = 0; j < m; ++j) {
                    if (dinv[j] == 0.0) s_blocked[ks[j]] = 1;                // accepted pivots flipped s_basic in the panel
                }
            }
            for (int j = 0; j < m; ++j) if (dinv[j] != 0.0) { progress = true; ++npiv; }
            ++nblk;
            cur ^= 1;
            // the other image is complete before anybody reads it (agent-scope release / acquire inside: per-XCD L2s are not coherent)
            if (!grid_barrier(p.grid_ctr, nexp, epoch, &s_ok)) { if (tid == 0 && p.n_unconverged) atomicAdd(p.n_unconverged, 1ULL << 40); return; }
        }
    }
    if (wg == 0) {
        for (int i = tid; i < n; i += COOP_THREADS) flagbuf[i] = s_basic[i];
        for (int i = tid; i < n; i += COOP_THREADS)
            p.node_sol[i] = s_basic[i] ? __builtin_nontemporal_load(&Timg[cur][(size_t)n * ld + i]) : 0.0;
        if (tid == 0) {
            flagbuf[n] = (uint8_t)cur;
            p.node_obj2[0] = __builtin_nontemporal_load(&Timg[cur][(size_t)n * ld + n]);
            p.best_obj[0] = 0.0; p.best_pat[0] = 0;
            if (p.n_pivots && npiv) atomicAdd(p.n_pivots, npiv);
            if (p.n_unconverged && nunconv) atomicAdd(p.n_unconverged, nunconv);
            if (p.n_pivots && nblk) atomicAdd(p.n_pivots + 2, nblk);          // counters[3]: blocks (diagnostic, PARTLS_ALT_TRACE)
        }
    }
}

hipError_t launch_sweep_coop(const SweepParams &p, int nwg, hipStream_t s)
{
    const int ld = p.n + 1;
    // pivots per block: two [mb][ld] LDS images within ~136 KB
    int mb = gj_block_size(ld, (size_t)136 * 1024);
    const size_t shmem = (size_t)2 * mb * ld * sizeof(double) + (3 * COOP_MB + COOP_THREADS / 64) * sizeof(double) + 2 * (size_t)p.n + 16;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_coop_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    // the hand-written grid barrier needs every workgroup resident at the same time: cap the grid by what the device can hold
    int dev = 0, ncu = 0, per_cu = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&sweep_coop_kernel), COOP_THREADS, shmem)) != hipSuccess) return e;
    const long long resident = (long long)ncu * per_cu;
    if (resident < 1) return hipErrorLaunchOutOfResources;
    if (nwg > resident) nwg = (int)resident;
    if ((e = hipMemsetAsync(p.grid_ctr, 0, 16, s)) != hipSuccess) return e;      // the polled word, in a 16-byte block of its own
    hipLaunchKernelGGL(sweep_coop_kernel, dim3(nwg), dim3(COOP_THREADS), shmem, s, p, mb);
    return hipGetLastError();
}

}  // namespace partls
