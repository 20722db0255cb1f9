// solvers.hip — fit(Alt) and fit(BnB) on the Gram / tableau kernels (SURVEY.md §8f-1,2).
//
// Both reduce to batches of sign-constrained least-squares "nodes" on the shared Gram block, solved on the device by the
// sweep kernels in node mode (solve_nodes, api.hip):
//   * Alt α-step (Alt.jl:80-90):  nonneg_lsq(Xo .* f', y) with f_m = sum_k Po[m,k] β_k has the constraint set
//     {Xo w : sign(f_m) w_m >= 0} (f_m == 0: zero column, α_m stays 0); α_m = w_m / f_m.  Any 0/1 partition matrix, as in the
//     reference: for a feature in several groups f_m is not a function of the sign pattern of β, hence per-variable codes.
//   * Alt β-step (Alt.jl:109-110): (Xo (Po∘α)) \ yo  ==  solve (A' G A) β = A' c  with A = Po∘α — (K+1)^2, on the host
//     from the device-built Gram copy, like cleanupResult an O(K^3) epilogue.
//   * BnB node bound (BnB.jl:69-92): the [Xp Xm] doubling is NNLS's way of writing free variables; a node is the same
//     tableau problem with the not-yet-branched groups FREE.  Nodes are bounded in batches (best-first frontier), the
//     incumbent prunes (BnB.jl:102); the node count depends on the search order and is not a parity quantity.  A feature of
//     several groups collects one constraint per branched group (BnB.jl:120-121): opposite ones force it to 0.
#include "ctx.h"
#include "frontier.h"
#include <new>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <queue>

using namespace partls;

namespace {

// dense solve H x = g (n x n, row-major), Gaussian elimination with partial pivoting; false if singular
bool solve_dense(std::vector<double> &H, std::vector<double> &g, int n)
{
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int i = k + 1; i < n; ++i) if (std::fabs(H[(size_t)i * n + k]) > std::fabs(H[(size_t)p * n + k])) p = i;
        if (H[(size_t)p * n + k] == 0.0) return false;
        if (p != k) { for (int j = 0; j < n; ++j) std::swap(H[(size_t)p * n + j], H[(size_t)k * n + j]); std::swap(g[(size_t)p], g[(size_t)k]); }
        for (int i = k + 1; i < n; ++i) {
            const double f = H[(size_t)i * n + k] / H[(size_t)k * n + k];
            if (f == 0.0) continue;
            for (int j = k; j < n; ++j) H[(size_t)i * n + j] -= f * H[(size_t)k * n + j];
            g[(size_t)i] -= f * g[(size_t)k];
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = g[(size_t)i];
        for (int j = i + 1; j < n; ++j) s -= H[(size_t)i * n + j] * g[(size_t)j];
        g[(size_t)i] = s / H[(size_t)i * n + i];
    }
    return true;
}

// ---- BnB node batches on the device ---------------------------------------------------------------------------------------------
// Snapshot pool: the final state of a bounded node (tableau tiles, rhs column, corner, basis flags: ~0.3 MB at n = 257) stays in
// HBM while the node has children in the frontier — a child is its parent's problem plus one group's sign constraint
// (BnB.jl:120-124), so it starts from the parent's tableau and exchanges only that group's wrong-signed variables instead of
// ~n/2 variables from scratch.  Slots live in chunks that are allocated on demand and kept by the context (288 GB of HBM:
// the default cap of 16 GB holds ~50 000 open nodes at n = 257); when the pool is full a node simply leaves no snapshot and
// its children start from the fresh tableau.
struct SnapshotPool {                                              // a view of the pool state the context keeps (ctx.h: bnb*)
    partls_ctx *c;
    std::vector<int> &free_list, &refs;
    explicit SnapshotPool(partls_ctx *ctx) : c(ctx), free_list(ctx->bnbFree), refs(ctx->bnbRefs) {}
    int chunk() const { return c->bnbChunkSlots; }                 // slots per hipMalloc: ~160 MB (512 slots at n = 257, 170 at n = 341, 19 at n = 1023)

    hipError_t begin(size_t bytes)
    {
        const size_t slot_bytes = (bytes + 255) & ~(size_t)255;
        if (c->bnbSlotBytes != slot_bytes) {                       // another tableau size: the old chunks are useless
            for (void *q : c->bnbChunks) (void)hipFree(q);
            c->bnbChunks.clear();
            c->bnbSlotBytes = slot_bytes;
            c->bnbChunkSlots = (int)std::max<size_t>(8, std::min<size_t>(512, ((size_t)160 << 20) / slot_bytes));
        }
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        size_t cap = (size_t)c->knobs.bnb_pool_mb << 20;
        cap = std::min(cap, free_b / 2 + c->bnbChunks.size() * chunk() * slot_bytes);
        c->bnbMaxSlots = cap / slot_bytes;
        free_list.clear();
        refs.assign(c->bnbChunks.size() * chunk(), 0);
        for (int i = (int)refs.size() - 1; i >= 0; --i) free_list.push_back(i);
        return hipSuccess;
    }
    bool valid(int slot) const { return slot >= 0 && (size_t)slot < c->bnbChunks.size() * chunk(); }
    double *ptr(int slot) const { return reinterpret_cast<double *>(static_cast<char *>(c->bnbChunks[(size_t)slot / chunk()]) + (size_t)(slot % chunk()) * c->bnbSlotBytes); }
    int alloc()                                                    // -1: pool exhausted
    {
        if (free_list.empty()) {
            if ((c->bnbChunks.size() + 1) * chunk() > c->bnbMaxSlots) return -1;
            void *q = nullptr;
            if (hipMalloc(&q, (size_t)chunk() * c->bnbSlotBytes) != hipSuccess) { (void)hipGetLastError(); c->bnbMaxSlots = 0; return -1; }
            const int base = (int)c->bnbChunks.size() * chunk();
            c->bnbChunks.push_back(q);
            refs.resize((size_t)base + chunk(), 0);
            for (int i = chunk() - 1; i >= 0; --i) free_list.push_back(base + i);
        }
        const int sl = free_list.back();
        free_list.pop_back();
        refs[(size_t)sl] = 0;
        return sl;
    }
    void release(int slot) { if (slot >= 0 && --refs[(size_t)slot] <= 0) free_list.push_back(slot); }
    void drop(int slot) { if (slot >= 0) free_list.push_back(slot); }          // a slot nobody references
};

// register kernel: the tile-cyclic image (tiles, rhs column, corner) + 16 T basis flags; deferred-update kernel (n > 320, sweep_lazy.hip):
// [ld x ld] base image, [ld] rhs column + corner, [n] basis flags
bool snapshots_supported(const partls_ctx *c) { return c->use_reg || !c->knobs.eager_generic; }
size_t snapshot_bytes(const partls_ctx *c)
{
    if (c->use_reg) return sweep_reg_t0_doubles(c->T) * sizeof(double) + (size_t)16 * c->T;
    const size_t ld = (size_t)c->n + 1;
    return (ld * ld + ld) * sizeof(double) + (size_t)c->n;
}

// Bound `cnt` nodes (pat, free) on the register kernel: codes, node solves (warm-started from src[i] when given, final state
// stored to dst[i] when given) and (bound, branch) all on the device; one upload, one download, one synchronisation per batch.
partls_status bnb_bound_batch(partls_ctx *c, size_t cnt, const uint64_t *pat, const uint64_t *free_, const double *const *src,
                              double *const *dst, double *lb, int32_t *branch, unsigned long long *unconv,
                              const std::function<void()> &between = {})
{
    const int n = c->n, Kp = (int)c->K + 1;
    c->tab_valid = false;
    c->coop_state_valid = false;
    if (unconv) *unconv = 0;
    if (cnt == 0) return PARTLS_OK;
    const bool snaps = src != nullptr || dst != nullptr;
    // input block: [pat | free | src | dst] (8 B each per node); output block: [counters 4 x 8 B | lb (cnt) | branch (cnt x 4 B)]
    const size_t in_words = (snaps ? 4 : 2) * cnt, out_bytes = 32 + cnt * 8 + ((cnt * 4 + 7) & ~(size_t)7);
    PARTLS_HIP_CHECK(c->bnbIn.ensure(in_words * 8));
    PARTLS_HIP_CHECK(c->bnbOut.ensure(out_bytes));
    PARTLS_HIP_CHECK(c->nodeCode.ensure(cnt * (size_t)n));
    PARTLS_HIP_CHECK(c->nodeSol.ensure((4 + cnt + cnt * (size_t)n) * sizeof(double)));
    PARTLS_HIP_CHECK(c->bestObj.ensure(sizeof(double) * (4 + 4 * 4096)));
    PARTLS_HIP_CHECK(c->bestPat.ensure(sizeof(int64_t) * 4096));
    // workgroups of the batch's grid: the 512-thread kernel runs one per CU at a time and the dispatcher balances nodes of different cost
    // best with one node per workgroup (tools/experiments/README.md); the global-memory kernel needs a scratch tableau per workgroup
    int ncu = 256;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || ncu < 1) ncu = 256;
    const size_t wg_cap = c->use_reg ? (size_t)ncu * (size_t)std::max(1, sweep_reg_concurrency(c->T)) * (size_t)std::max(1, c->knobs.bnb_wg_per_cu)
                                     : (size_t)2 * ncu;
    const int grid = (int)std::min<size_t>(cnt, wg_cap);
    PARTLS_HIP_CHECK(c->scratch.ensure(c->use_reg ? 64 * sizeof(double) : (size_t)grid * (n + 1) * (n + 1) * sizeof(double)));
    PARTLS_HIP_CHECK(c->bnbHostIn.resize(in_words));
    PARTLS_HIP_CHECK(c->bnbHostOut.resize((out_bytes + 7) / 8));
    uint64_t *hin = reinterpret_cast<uint64_t *>(c->bnbHostIn.data());
    std::memcpy(hin, pat, cnt * 8);
    std::memcpy(hin + cnt, free_, cnt * 8);
    if (snaps) {
        for (size_t i = 0; i < cnt; ++i) {
            hin[2 * cnt + i] = (uint64_t)(uintptr_t)(src ? src[i] : nullptr);
            hin[3 * cnt + i] = (uint64_t)(uintptr_t)(dst ? dst[i] : nullptr);
        }
    }
    uint64_t *din = c->bnbIn.as<uint64_t>();
    char *dout = static_cast<char *>(c->bnbOut.p);
    PARTLS_HIP_CHECK(hipMemcpyAsync(din, hin, in_words * 8, hipMemcpyHostToDevice, c->stream));
    PARTLS_HIP_CHECK(hipMemsetAsync(dout, 0, 32, c->stream));
    PARTLS_HIP_CHECK(launch_bnb_codes(c->maskTabP, n, din, din + cnt, (int)cnt, c->nodeCode.as<int8_t>(), c->stream));
    SweepParams p{};
    p.n = n; p.kbits = c->kbits;
    p.mask = c->maskTabP;
    p.scratch = c->scratch.as<double>();
    p.g_begin = 0; p.g_end = (int64_t)cnt; p.chain_len = 1;
    p.tol = c->tol; p.piv_eps = 1e-11; p.max_rounds = 20 * (n + 1);
    p.best_obj = c->bestObj.as<double>(); p.best_pat = c->bestPat.as<int64_t>();
    p.n_unconverged = reinterpret_cast<unsigned long long *>(dout);
    p.n_pivots = p.n_unconverged + 1;
    p.n_vetoes = p.n_unconverged + 2;
    p.node_code = c->nodeCode.as<int8_t>();
    p.node_obj2 = c->nodeSol.as<double>() + 4; p.node_sol = c->nodeSol.as<double>() + 4 + cnt; p.node_ld = n;
    if (snaps) {
        p.node_src = reinterpret_cast<const double *const *>(din + 2 * cnt);
        p.node_dst = reinterpret_cast<double *const *>(din + 3 * cnt);
    }
    if (c->use_reg) {
        p.T0 = c->T0reg.as<double>();
        PARTLS_HIP_CHECK(launch_sweep_blk(p, c->T, grid, c->stream));
    } else {
        p.T0 = c->Tfull.as<double>();
        PARTLS_HIP_CHECK(launch_sweep_lazy(p, grid, c->stream));
    }
    double *dlb = reinterpret_cast<double *>(dout + 32);
    int *dbr = reinterpret_cast<int *>(dout + 32 + cnt * 8);
    PARTLS_HIP_CHECK(launch_bnb_nu(p.node_sol, p.node_obj2, n, c->scale.as<double>(), c->maskTabP, Kp, din + cnt, (int)cnt, dlb, dbr, c->stream));
    PARTLS_HIP_CHECK(hipMemcpyAsync(c->bnbHostOut.data(), dout, out_bytes, hipMemcpyDeviceToHost, c->stream));
    if (between) between();                                  // host work that overlaps the batch (the search pops its next round here)
    PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));
    const char *hout = reinterpret_cast<const char *>(c->bnbHostOut.data());
    unsigned long long counters[4];
    std::memcpy(counters, hout, 32);
    std::memcpy(lb, hout + 32, cnt * 8);
    std::memcpy(branch, hout + 32 + cnt * 8, cnt * 4);
    if (unconv) *unconv = counters[0];
    c->last_pivots = counters[1]; c->last_vetoes = counters[2];
    return PARTLS_OK;
}

}  // namespace

extern "C" {

partls_status partls_fit_alt(partls_ctx *c, const double *X, int64_t N, int64_t M, int64_t ldX, const double *y,
                             const int64_t *P, int64_t K, int64_t ldP, double eta, double eps, int64_t T,
                             const double *alpha0, const double *beta0,
                             double *alpha, double *beta, double *t, double *opt, int64_t *iters)
try {
    partls_status st = ctx_prepare(c, X, N, M, ldX, y, 0, P, K, ldP, eta, /*faithful=*/true, 0);
    if (st != PARTLS_OK) return st;
    return partls_alt_prepared(c, eps, T, alpha0, beta0, alpha, beta, t, opt, iters);
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_alt_prepared(partls_ctx *c, double eps, int64_t T, const double *alpha0, const double *beta0,
                                  double *alpha, double *beta, double *t, double *opt, int64_t *iters)
try {
    if (!c || !c->prepared || !c->faithful) { set_error("partls_alt_prepared: needs a context prepared with PARTLS_OPT_FAITHFUL_INTERCEPT"); return PARTLS_ERR_STATE; }
    if (!alpha0 || !beta0 || !alpha || !beta || !t || !opt) { set_error("partls_fit_alt: NULL argument"); return PARTLS_ERR_BAD_ARG; }
    if (!(eps > 0.0) || T < 1) { set_error("partls_fit_alt: need eps > 0 and T >= 1 (PartitionedLS.jl:294-295)"); return PARTLS_ERR_BAD_ARG; }
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
    partls_status st = PARTLS_OK;
    const int64_t M = c->M, K = c->K;
    const int Mp = (int)M + 1, Kp = (int)K + 1, Y = (int)M + 1;
    std::vector<double> a(alpha0, alpha0 + Mp), b(beta0, beta0 + Kp), w((size_t)Mp, 0.0), f((size_t)Mp, 0.0), sols, obj2;
    // Po as lists: groups of every variable (features, then the intercept in its own group K; PartitionedLS.jl:76-81)
    std::vector<std::vector<int>> groups_of((size_t)Mp);
    for (int m = 0; m < Mp; ++m)
        for (int k = 0; k < Kp; ++k) if ((c->mask_aug[(size_t)m] >> k) & 1ULL) groups_of[(size_t)m].push_back(k);
    auto feature_mul = [&]() {                            // f_m = sum_k Po[m,k] β_k   (Alt.jl:80)
        for (int m = 0; m < Mp; ++m) { double v = 0.0; for (int k : groups_of[(size_t)m]) v += b[(size_t)k]; f[(size_t)m] = v; }
    };
    auto w_from = [&]() { feature_mul(); for (int m = 0; m < Mp; ++m) w[(size_t)m] = a[(size_t)m] * f[(size_t)m]; };
    std::vector<int8_t> codes((size_t)c->n);
    // Gershgorin radii of the scaled Gram block (one tiny kernel now, read back with the first alpha-step's results): decides after the
    // loop whether the last iteration must be verified against the data (see there)
    std::vector<double> gersh((size_t)c->n, 0.0);
    PARTLS_HIP_CHECK(c->altGersh.ensure((size_t)c->n * sizeof(double)));
    PARTLS_HIP_CHECK(launch_gersh(c->Tfull.as<double>(), c->n, c->altGersh.as<double>(), c->stream));
    PARTLS_HIP_CHECK(hipMemcpyAsync(gersh.data(), c->altGersh.p, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    // what the data-space check after the loop needs of the LAST iteration: the alpha-step's raw solution w = f∘α (before checkalpha and
    // the renormalisation), its constraint codes over [features, intercept], and the diagonal of the beta-step system
    std::vector<double> wv, hdiag((size_t)Kp, 0.0);
    std::vector<int8_t> vcode_last((size_t)Mp, 0);

    double oldopt = 1e20, optval = 1e10;                 // Alt.jl:73-74
    int64_t i = 1;
    unsigned long long unconv_total = 0;
    while (i <= T && std::fabs(oldopt - optval) > eps * oldopt) {
        // ---- α-step: one sign-constrained solve on the device (Alt.jl:80-90) ----------------------------------------------
        feature_mul();
        for (int v = 0; v < c->n; ++v) { const double fv = f[(size_t)c->perm[(size_t)v]]; codes[(size_t)v] = (int8_t)((fv > 0.0) - (fv < 0.0)); }
        unsigned long long unconv = 0;
        const auto tt0 = std::chrono::steady_clock::now();
        st = solve_nodes(c, codes, 1, sols, obj2, &unconv, /*resume=*/i > 1);   // warm start after the first α-step
        const auto tt1 = std::chrono::steady_clock::now();
        if (st != PARTLS_OK) return st;
        const unsigned long long apiv = c->last_pivots, ablk = c->last_blocks;
        unconv_total += unconv;
        unscale_solution(c, sols.data(), wv);
        for (int v = 0; v < c->n; ++v) vcode_last[(size_t)c->perm[(size_t)v]] = codes[(size_t)v];
        for (int m = 0; m < Mp; ++m) {
            const double am = (f[(size_t)m] != 0.0) ? wv[(size_t)m] / f[(size_t)m] : 0.0;
            a[(size_t)m] = am > 0.0 ? am : 0.0;
        }
        // ---- checkalpha (Alt.jl:5-20) and renormalisation (Alt.jl:95-98) ----------------------------------------------------
        std::vector<double> suma((size_t)Kp, 0.0), poa((size_t)Mp, 0.0);
        std::vector<int> cntk((size_t)Kp, 0);
        for (int m = 0; m < Mp; ++m) for (int k : groups_of[(size_t)m]) { suma[(size_t)k] += a[(size_t)m]; ++cntk[(size_t)k]; }
        for (int k = 0; k < Kp; ++k)
            if (suma[(size_t)k] == 0.0)
                for (int m = 0; m < Mp; ++m) if ((c->mask_aug[(size_t)m] >> k) & 1ULL) a[(size_t)m] = 1.0 / (double)cntk[(size_t)k];
        std::fill(suma.begin(), suma.end(), 0.0);
        for (int m = 0; m < Mp; ++m) for (int k : groups_of[(size_t)m]) suma[(size_t)k] += a[(size_t)m];          // sumα
        for (int m = 0; m < Mp; ++m) for (int k : groups_of[(size_t)m]) poa[(size_t)m] += suma[(size_t)k];        // Po * sumα'
        // a feature that belongs to no group has multiplier 0 in every alpha-step and P row 0 in predict: its alpha is irrelevant.  The
        // reference divides 0 / 0 there (Alt.jl:98) and carries the NaN through every later iterate; here it stays 0
        for (int m = 0; m < Mp; ++m) a[(size_t)m] = groups_of[(size_t)m].empty() ? 0.0 : a[(size_t)m] / poa[(size_t)m];
        for (int k = 0; k < Kp; ++k) b[(size_t)k] *= suma[(size_t)k];
        // ---- β-step: (A' G A) β = A' c,  A = Po∘α  (Alt.jl:109-110 in Gram form); the system is assembled on the device ---------
        std::vector<double> Hg((size_t)Kp * (Kp + 1)), H((size_t)Kp * Kp), g((size_t)Kp);
        PARTLS_HIP_CHECK(c->altA.ensure((size_t)Mp * sizeof(double)));
        PARTLS_HIP_CHECK(c->altGA.ensure((size_t)Mp * Kp * sizeof(double)));
        PARTLS_HIP_CHECK(c->altHg.ensure(Hg.size() * sizeof(double)));
        PARTLS_HIP_CHECK(hipMemcpyAsync(c->altA.p, a.data(), (size_t)Mp * sizeof(double), hipMemcpyHostToDevice, c->stream));
        PARTLS_HIP_CHECK(launch_alt_beta_system(c->G.as<double>(), c->ldg, (int)M, c->eta, c->maskAugD.as<uint64_t>(), c->altA.as<double>(), Kp,
                                                c->altGA.as<double>(), c->altHg.as<double>(), c->stream));
        PARTLS_HIP_CHECK(hipMemcpyAsync(Hg.data(), c->altHg.p, Hg.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));
        for (int k = 0; k < Kp; ++k) {
            for (int k2 = 0; k2 < Kp; ++k2) H[(size_t)k * Kp + k2] = Hg[(size_t)k * (Kp + 1) + k2];
            g[(size_t)k] = Hg[(size_t)k * (Kp + 1) + Kp];
        }
        // groups without members (possible only for user groups with no feature) get a unit diagonal so H stays regular
        for (int k = 0; k < Kp; ++k) if (H[(size_t)k * Kp + k] == 0.0) H[(size_t)k * Kp + k] = 1.0;
        const std::vector<double> H0 = H, g0 = g;            // solve_dense eliminates in place
        if (!solve_dense(H, g, Kp)) { set_error("partls_fit_alt: singular beta-step system"); return PARTLS_ERR_NOT_CONVERGED; }
        b = g;
        // ---- loss (Alt.jl:112-113): ||Xo (Po∘α) β - y||^2 = β' H β - 2 g' β + y'y  (w = A β; K'^2 work instead of M^2) ----------------
        double o2 = h_reg(c, Y, Y);
        for (int k = 0; k < Kp; ++k) {
            double hb = 0.0;
            for (int k2 = 0; k2 < Kp; ++k2) hb += H0[(size_t)k * Kp + k2] * b[(size_t)k2];
            o2 += b[(size_t)k] * (hb - 2.0 * g0[(size_t)k]);
        }
        for (int k = 0; k < Kp; ++k) hdiag[(size_t)k] = H0[(size_t)k * Kp + k];
        oldopt = optval;
        optval = std::sqrt(o2 > 0.0 ? o2 : 0.0);
        if (c->knobs.alt_trace) fprintf(stderr, "[alt] iter %d: alpha-step %.3f ms (%llu pivots, %llu blocks), rest %.3f ms\n", (int)i, std::chrono::duration<double, std::milli>(tt1 - tt0).count(), apiv, ablk, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt1).count());
        ++i;
    }
    w_from();
    // Final objective and the data-space check of the last iteration (round 4).  Both solves of an iteration work on the Gram form — the
    // alpha-step on the tableau, the beta-step on the K' x K' normal equations A'GA beta = A'c, whose condition is the SQUARE of that of the
    // reference's QR solve (Alt.jl:110) — so the model is verified against the DATA before it is returned, as fit(Opt) / fit(BnB) verify
    // theirs: one pass over X gives the loss (Alt.jl:112-113: no Gram cancellation) and g = Xo'(yo - Xo w) at the final w, and
    //   * the beta-step is stationary iff A'g = 0: |sum_m A_mk g_m| in units of ||Xo A_k|| ||y|| per group;
    //   * the alpha-step's KKT conditions (sign-constrained LS with the multipliers of the previous beta, Alt.jl:80-90) hold at ITS
    //     solution w_alpha, which differs from the final w by the beta-step only: g_alpha = g + B (w - w_alpha) is carried over with the
    //     host Gram copy (error ~eps |B| |w - w_alpha|: exact at convergence, ~1e-13 when the beta-step still moves w by O(1)) instead
    //     of a second pass over X.
    // Above PARTLS_KKT_TOL the call returns PARTLS_ERR_ILL_CONDITIONED with the model in the outputs (INTEGRATION.md reroutes that
    // status to the stock Julia body).
    // WHEN the pass is needed.  It costs two reads of X (2.2 ms of a 6.9 ms fit at C4), and on well-conditioned data it can only confirm
    // what perturbation theory already guarantees: the solves work on G~ (unit diagonal), and normal-equation solutions carry a relative
    // error of at most ~n eps cond(G~).  Gershgorin gives a RIGOROUS bound from the matrix itself: with r = max_i sum_{j != i} |G~_ij|,
    // lambda_min >= 1 - r and lambda_max <= 1 + r.  For r <= 0.75 (cond <= 7: n eps cond < 1e-12 up to n = 1023, the threshold of the check
    // itself) the fit is certified without touching X again — Gaussian-like designs (C4: r = 0.41).  Anything else — correlated features,
    // uncentred columns against the intercept, eta-coupled groups — takes the pass.  PARTLS_ALT_ALWAYS_CHECK forces it (tests).
    double radius = 0.0;
    for (int v = 0; v < c->n; ++v) radius = std::max(radius, gersh[(size_t)v]);
    const bool certified = i > 1 && radius <= 0.75 && std::isfinite(radius) && !c->knobs.alt_always_check;
    double dopt = optval;
    std::vector<double> g;
    c->last_kkt = 0.0;
    c->last_min_loo = 0.0;
    int worst = -1;
    const char *which = "";
    if (certified) {
        // the loss of the last iteration, beta'H beta - 2 g'beta + y'y from the K' x K' system, carries an absolute error of ~eps * y'y
        // (cancellation against y'y): only a near-interpolating fit (the reference's toy: opt = 0) needs the objective from the data
        if (!(optval * optval > 1e-6 * h_reg(c, Y, Y))) { st = data_objective(c, w, &dopt); if (st != PARTLS_OK) return st; }
    } else {
        st = data_objective(c, w, &dopt, &g);
        if (st != PARTLS_OK) return st;
    }
    if (!certified && i > 1 && unconv_total == 0) {          // at least one iteration ran
        const double yy = h_reg(c, Y, Y), ynorm = std::sqrt(yy > 0.0 ? yy : 0.0);
        for (int k = 0; k < Kp; ++k) {
            double s = 0.0;
            for (int m = 0; m < Mp; ++m) if ((c->mask_aug[(size_t)m] >> k) & 1ULL) s += a[(size_t)m] * g[(size_t)m];
            const double nrm = std::sqrt(hdiag[(size_t)k] > 0.0 ? hdiag[(size_t)k] : 0.0) * (ynorm > 0.0 ? ynorm : 1.0);
            const double v = nrm > 0.0 ? std::fabs(s) / nrm : 0.0;
            if (v > c->last_kkt) { c->last_kkt = v; worst = k; which = "beta-step, group"; }
        }
        std::vector<double> ga(g);
        for (int j = 0; j < Mp; ++j) {
            const double dj = w[(size_t)j] - wv[(size_t)j];
            if (dj == 0.0) continue;
            for (int m = 0; m < Mp; ++m) ga[(size_t)m] += h_reg(c, m, j) * dj;
        }
        int wa = -1;
        const double va = kkt_violation_data(c, wv, ga, vcode_last, &wa);
        if (va > c->last_kkt) { c->last_kkt = va; worst = wa; which = "alpha-step, variable"; }
    }
    for (int64_t m = 0; m < M; ++m) alpha[m] = a[(size_t)m];
    for (int64_t k = 0; k < K; ++k) beta[k] = b[(size_t)k];
    *t = b[(size_t)K] * a[(size_t)M];                    // Alt.jl:119
    *opt = dopt;
    if (iters) *iters = i - 1;
    if (unconv_total) { set_error("partls_fit_alt: an alpha-step hit the pivot cap"); return PARTLS_ERR_NOT_CONVERGED; }
    if (kkt_says_ill_conditioned(c)) {
        set_error("partls_fit_alt: the last iteration's solves do not hold in data space (violation %.2e of ||x|| ||y|| at %s %d, tolerance %.1e): "
                  "X is too ill-conditioned for the fp64 Gram form; the outputs hold the Gram-form iterate", c->last_kkt, which, worst, c->knobs.kkt_tol);
        return PARTLS_ERR_ILL_CONDITIONED;
    }
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_fit_bnb(partls_ctx *c, const double *X, int64_t N, int64_t M, int64_t ldX, const double *y,
                             const int64_t *P, int64_t K, int64_t ldP, double eta,
                             double *alpha, double *beta, double *t, double *opt, int64_t *nopen)
try {
    partls_status st = ctx_prepare(c, X, N, M, ldX, y, 0, P, K, ldP, eta, /*faithful=*/true, 0);
    if (st != PARTLS_OK) return st;
    return partls_bnb_prepared(c, alpha, beta, t, opt, nopen);
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

// ---- BnB primitives (shared by the single-rank driver below and the rank-sharded search of partitionedls.jl_amd/dist.py) -------
// A node is (pat, free): group k is branched iff bit k of `free` is clear, and then constrained to alpha_pk >= 0 (bit k of pat
// set) or <= 0 (BnB.jl:120-121).  partls_bnb_bound evaluates BnB.jl:99-118 for a batch: lb[i] = the node's lower bound
// (BnB.jl:69-92 on the device) and branch[i] = argmax_k nu_k (BnB.jl:107,117), or -1 when all nu_k == 0, i.e. the relaxed
// solution is feasible for the original problem and lb[i] is its value (BnB.jl:109-115).
partls_status partls_bnb_bound(partls_ctx *c, int64_t count, const uint64_t *pat, const uint64_t *free_, double *lb, int32_t *branch)
try {
    if (!c || !c->prepared || !c->faithful) { set_error("partls_bnb_bound: needs a context prepared with PARTLS_OPT_FAITHFUL_INTERCEPT"); return PARTLS_ERR_STATE; }
    if (count < 0 || (count > 0 && (!pat || !free_ || !lb || !branch))) { set_error("partls_bnb_bound: bad argument"); return PARTLS_ERR_BAD_ARG; }
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
    if (snapshots_supported(c)) {                              // codes, solves and (bound, branch) on the device (misc.hip)
        unsigned long long unc = 0;
        partls_status st2 = bnb_bound_batch(c, (size_t)count, pat, free_, nullptr, nullptr, lb, branch, &unc);
        if (st2 != PARTLS_OK) return st2;
        if (unc) { set_error("partls_bnb_bound: a node bound hit the pivot cap"); return PARTLS_ERR_NOT_CONVERGED; }
        return PARTLS_OK;
    }
    const int Mp = (int)c->M + 1, Kp = (int)c->K + 1, n = c->n;
    // per-variable constraint of a node: the branched groups of the variable each contribute alpha >= 0 or alpha <= 0 (the Σ of
    // BnB.jl:120-121 accumulates them); none -> free, both kinds -> the variable is forced to 0 (BnB.jl:74-79)
    std::vector<int8_t> codes((size_t)count * (size_t)n);
    for (int64_t i = 0; i < count; ++i)
        for (int v = 0; v < n; ++v) {
            const uint64_t br = c->mask_tab[(size_t)v] & ~free_[i];
            const bool pos = (br & pat[i]) != 0, neg = (br & ~pat[i]) != 0;
            codes[(size_t)i * (size_t)n + v] = (int8_t)(!br ? 2 : (pos && neg ? 0 : (pos ? 1 : -1)));
        }
    std::vector<double> sols, obj2, w;
    unsigned long long unconv = 0;
    partls_status st = solve_nodes(c, codes, (size_t)count, sols, obj2, &unconv);
    if (st != PARTLS_OK) return st;
    // nu_k = sum_{i<j in group k} max(0, -w_i w_j) = (sum w+)(sum |w-|)  (BnB.jl:42-57); only free groups can mix signs.
    // One pass over the variables per node (the groups of a variable come from its mask bits), not one pass per group.
    std::vector<double> pos((size_t)Kp), neg((size_t)Kp);
    for (int64_t i = 0; i < count; ++i) {
        lb[i] = std::sqrt(obj2[(size_t)i] > 0.0 ? obj2[(size_t)i] : 0.0);
        unscale_solution(c, sols.data() + (size_t)i * (size_t)n, w);
        std::fill(pos.begin(), pos.end(), 0.0);
        std::fill(neg.begin(), neg.end(), 0.0);
        for (int m = 0; m < Mp; ++m) {
            const double wm = w[(size_t)m];
            if (wm == 0.0) continue;
            for (uint64_t bits = c->mask_aug[(size_t)m] & free_[i]; bits; bits &= bits - 1) {
                const int k = __builtin_ctzll(bits);
                if (wm > 0.0) pos[(size_t)k] += wm; else neg[(size_t)k] -= wm;
            }
        }
        int kbest = -1; double nubest = 0.0;
        for (int k = 0; k < Kp; ++k) {
            const double nu = pos[(size_t)k] * neg[(size_t)k];
            if (nu > nubest) { nubest = nu; kbest = k; }                   // argmax: first maximal index
        }
        branch[i] = kbest;
    }
    if (unconv) { set_error("partls_bnb_bound: a node bound hit the pivot cap"); return PARTLS_ERR_NOT_CONVERGED; }
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

// The same with tableau snapshots, for a host that runs the search itself (partitionedls.jl_amd/dist.py: the rank-sharded search deals
// every node to the rank that holds its parent's snapshot).  src_slot[i]: snapshot node i starts from (-1: the fresh tableau);
// dst_slot[i] (out): the slot that now holds node i's final tableau, -1 when the pool is full, the node has no free group left, or the
// tableau is not register-resident (n > 320: cold bounds only).  The host owns the reference counts and returns slots with
// partls_bnb_snap_release; partls_bnb_snap_begin empties the pool for a new search.
partls_status partls_bnb_snap_begin(partls_ctx *c)
try {
    if (!c || !c->prepared || !c->faithful) { set_error("partls_bnb_snap_begin: needs a context prepared with PARTLS_OPT_FAITHFUL_INTERCEPT"); return PARTLS_ERR_STATE; }
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
    if (!snapshots_supported(c)) { c->bnbFree.clear(); c->bnbRefs.clear(); c->bnbMaxSlots = 0; return PARTLS_OK; }
    SnapshotPool pool(c);
    PARTLS_HIP_CHECK(pool.begin(snapshot_bytes(c)));
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

static partls_status bnb_bound_snap_impl(partls_ctx *c, int64_t count, const uint64_t *pat, const uint64_t *free_, const int32_t *src_slot,
                                         int32_t *dst_slot, double *lb, int32_t *branch, const std::function<void()> &between);
partls_status partls_bnb_bound_snap(partls_ctx *c, int64_t count, const uint64_t *pat, const uint64_t *free_, const int32_t *src_slot,
                                    int32_t *dst_slot, double *lb, int32_t *branch)
try {
    return bnb_bound_snap_impl(c, count, pat, free_, src_slot, dst_slot, lb, branch, {});
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

static partls_status bnb_bound_snap_impl(partls_ctx *c, int64_t count, const uint64_t *pat, const uint64_t *free_, const int32_t *src_slot,
                                         int32_t *dst_slot, double *lb, int32_t *branch, const std::function<void()> &between)
{
    if (!c || !c->prepared || !c->faithful) { set_error("partls_bnb_bound_snap: needs a context prepared with PARTLS_OPT_FAITHFUL_INTERCEPT"); return PARTLS_ERR_STATE; }
    if (count < 0 || (count > 0 && (!pat || !free_ || !src_slot || !dst_slot || !lb || !branch))) { set_error("partls_bnb_bound_snap: bad argument"); return PARTLS_ERR_BAD_ARG; }
    if (!snapshots_supported(c) || c->knobs.bnb_cold) {        // no snapshots on the eager global-memory kernel (A/B reference)
        for (int64_t i = 0; i < count; ++i) dst_slot[i] = -1;
        const partls_status cs = partls_bnb_bound(c, count, pat, free_, lb, branch);
        if (between) between();
        return cs;
    }
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
    SnapshotPool pool(c);
    if (c->bnbSlotBytes != ((snapshot_bytes(c) + 255) & ~(size_t)255)) { set_error("partls_bnb_bound_snap: call partls_bnb_snap_begin after preparing the problem"); return PARTLS_ERR_STATE; }
    std::vector<const double *> srcp((size_t)count);
    std::vector<double *> dstp((size_t)count);
    const auto tr0 = std::chrono::steady_clock::now();
    for (int64_t i = 0; i < count; ++i) {
        if (src_slot[i] >= 0 && !pool.valid(src_slot[i])) { set_error("partls_bnb_bound_snap: src_slot[%lld] = %d is no slot of this context", (long long)i, src_slot[i]); return PARTLS_ERR_BAD_ARG; }
        dst_slot[i] = free_[i] ? pool.alloc() : -1;
        srcp[(size_t)i] = src_slot[i] >= 0 ? pool.ptr(src_slot[i]) : nullptr;
        dstp[(size_t)i] = dst_slot[i] >= 0 ? pool.ptr(dst_slot[i]) : nullptr;
    }
    unsigned long long unc = 0;
    const auto tr1 = std::chrono::steady_clock::now();
    partls_status st = bnb_bound_batch(c, (size_t)count, pat, free_, srcp.data(), dstp.data(), lb, branch, &unc, between);
    if (getenv("PARTLS_BNB_TRACE")) {
        const auto tr2 = std::chrono::steady_clock::now();
        const double a = std::chrono::duration<double, std::milli>(tr1 - tr0).count(), b = std::chrono::duration<double, std::milli>(tr2 - tr1).count();
        if (a + b > 0.5) fprintf(stderr, "[bnb] batch of %lld: slots %.3f ms, device %.3f ms, chunks %zu\n", (long long)count, a, b, c->bnbChunks.size());
    }
    if (st == PARTLS_OK && unc) { set_error("partls_bnb_bound_snap: a node bound hit the pivot cap"); st = PARTLS_ERR_NOT_CONVERGED; }
    if (st != PARTLS_OK) {
        // the host never learns these slots (it must not trust outputs of a failed call): back to the free list, or the pool would
        // shrink by a batch with every failure until the next partls_bnb_snap_begin
        for (int64_t i = 0; i < count; ++i) { if (dst_slot[i] >= 0) pool.drop(dst_slot[i]); dst_slot[i] = -1; }
        return st;
    }
    return PARTLS_OK;
}


partls_status partls_bnb_snap_release(partls_ctx *c, int64_t count, const int32_t *slots)
try {
    if (!c || count < 0 || (count > 0 && !slots)) { set_error("partls_bnb_snap_release: bad argument"); return PARTLS_ERR_BAD_ARG; }
    SnapshotPool pool(c);
    for (int64_t i = 0; i < count; ++i)
        if (slots[i] >= 0) {
            if (!pool.valid(slots[i])) { set_error("partls_bnb_snap_release: %d is no slot of this context", slots[i]); return PARTLS_ERR_BAD_ARG; }
            pool.drop(slots[i]);
        }
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

// The model of one (feasible) node: re-solve, data-space refinement, BnB.jl:36-39 normalisation, objective from the data.
partls_status partls_bnb_leaf(partls_ctx *c, uint64_t pat, uint64_t free_, double *alpha, double *beta, double *t, double *opt)
try {
    if (!c || !c->prepared || !c->faithful) { set_error("partls_bnb_leaf: needs a context prepared with PARTLS_OPT_FAITHFUL_INTERCEPT"); return PARTLS_ERR_STATE; }
    if (!alpha || !beta || !t || !opt) { set_error("partls_bnb_leaf: NULL argument"); return PARTLS_ERR_BAD_ARG; }
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
    const int64_t M = c->M, K = c->K;
    const int Mp = (int)M + 1, Kp = (int)K + 1, n = c->n;
    std::vector<int8_t> codes((size_t)n);
    for (int v = 0; v < n; ++v) {
        const uint64_t br = c->mask_tab[(size_t)v] & ~free_;
        const bool pos = (br & pat) != 0, neg = (br & ~pat) != 0;
        codes[(size_t)v] = (int8_t)(!br ? 2 : (pos && neg ? 0 : (pos ? 1 : -1)));
    }
    std::vector<double> sols, obj2, w;
    unsigned long long unconv = 0;
    partls_status st = solve_nodes(c, codes, 1, sols, obj2, &unconv, false, /*want_tab=*/true);
    if (st != PARTLS_OK) return st;
    unscale_solution(c, sols.data(), w);
    st = refine_solution(c, w, false);
    if (st != PARTLS_OK) return st;
    // BnB.jl:36-37: β = sum(Po .* α, dims = 1); α = sum(Po .* α ./ β, dims = 2); t = β[end]
    std::vector<double> bsum((size_t)Kp, 0.0);
    for (int m = 0; m < Mp; ++m) for (int k = 0; k < Kp; ++k) if ((c->mask_aug[(size_t)m] >> k) & 1ULL) bsum[(size_t)k] += w[(size_t)m];
    for (int64_t m = 0; m < M; ++m) {
        double s = 0.0;
        for (int k = 0; k < Kp; ++k) if ((c->mask_aug[(size_t)m] >> k) & 1ULL) s += w[(size_t)m] / bsum[(size_t)k];
        alpha[m] = s;
    }
    for (int64_t k = 0; k < K; ++k) beta[k] = bsum[(size_t)k];
    *t = bsum[(size_t)K];
    std::vector<double> g;
    st = data_objective(c, w, opt, &g);
    if (st != PARTLS_OK) return st;
    if (unconv) { set_error("partls_bnb_leaf: the node solve hit the pivot cap"); return PARTLS_ERR_NOT_CONVERGED; }
    // the leaf's KKT conditions against the data, as partls_opt_finish checks its winner (api.hip: kkt_violation_data)
    std::vector<int8_t> vcode((size_t)Mp, 0);
    for (int v = 0; v < n; ++v) vcode[(size_t)c->perm[(size_t)v]] = codes[(size_t)v];
    int worst = -1;
    c->last_kkt = kkt_violation_data(c, w, g, vcode, &worst);
    if (kkt_says_ill_conditioned(c)) {
        set_error("partls_bnb_leaf: the model's KKT conditions do not hold in data space (violation %.2e at variable %d): X is too "
                  "ill-conditioned for the fp64 Gram form", c->last_kkt, worst);
        return PARTLS_ERR_ILL_CONDITIONED;
    }
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

// fit_BnB (BnB.jl:94-132) as a best-first search: frontier ordered by the parent's bound, nodes bounded in device batches, the
// incumbent prunes (BnB.jl:102).  Same optimum as the reference's depth-first recursion; the node count is not.  On the register
// kernel every node starts from its parent's final tableau (SnapshotPool): the reference's recursion hands the child the parent's
// constraint set plus one group (BnB.jl:120-124), here it also inherits the parent's basis.
// max_nodes > 0 (measurement): stop after that many bounded nodes and report the incumbent so far (*mu = inf when there is none).
partls_status partls_bnb_search(partls_ctx *c, int64_t max_nodes, double *mu_out, uint64_t *pat_out, uint64_t *free_out, int64_t *nodes_out)
try {
    if (!c || !c->prepared || !c->faithful) { set_error("partls_bnb_search: needs a context prepared with PARTLS_OPT_FAITHFUL_INTERCEPT"); return PARTLS_ERR_STATE; }
    if (!mu_out || !pat_out || !free_out) { set_error("partls_bnb_search: NULL output"); return PARTLS_ERR_BAD_ARG; }
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
    const int Kp = (int)c->K + 1;
    partls_frontier f;
    f.batch = std::max(1, c->knobs.bnb_batch);
    f.refs.resize(1);
    f.best_free = ((uint64_t)1 << Kp) - 1;
    f.heap.push_node({0.0, 0ULL, ((uint64_t)1 << Kp) - 1, f.seq++, -1, -1});     // root: everything free (Σ = [], BnB.jl:33)
    partls_status st = partls_bnb_snap_begin(c);
    if (st != PARTLS_OK) return st;
    const size_t cap = (size_t)f.batch;
    std::vector<uint64_t> bp(cap), bf(cap);
    std::vector<int32_t> src(cap), dst(cap), br(cap);
    std::vector<double> lb(cap);
    int32_t per_rank = 0;
    const bool trace = getenv("PARTLS_BNB_TRACE") != nullptr;
    double t_next = 0.0, t_bound = 0.0, t_ingest = 0.0;
    int rounds = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    while (!(max_nodes > 0 && f.bounded >= max_nodes)) {
        int64_t mine = 0;
        const auto q0 = now();
        const int64_t cnt = f.next(&mine, bp.data(), bf.data(), src.data(), &per_rank);
        const auto q1 = now();
        if (cnt > 0) {
            // while the device bounds this round the host pops the next one out of the big heap (merged with the new children by next())
            st = bnb_bound_snap_impl(c, cnt, bp.data(), bf.data(), src.data(), dst.data(), lb.data(), br.data(), [&f]() { f.prefetch(); });
            if (st != PARTLS_OK) return st;
        }
        const auto q2 = now();
        if (cnt > 0) f.ingest(lb.data(), br.data(), dst.data());
        if (!f.dead.empty()) {
            st = partls_bnb_snap_release(c, (int64_t)f.dead.size(), f.dead.data());
            if (st != PARTLS_OK) return st;
            f.dead.clear();
        }
        t_next += ms(q0, q1); t_bound += ms(q1, q2); t_ingest += ms(q2, now()); ++rounds;
        if (cnt == 0) break;
    }
    if (trace) fprintf(stderr, "[bnb] search: %d rounds, %lld nodes: frontier pop / deal %.2f ms, device batches %.2f ms, ingest / release %.2f ms\n", rounds,
                       (long long)f.bounded, t_next, t_bound, t_ingest);
    *mu_out = f.mu; *pat_out = f.best_pat; *free_out = f.best_free;
    if (nodes_out) *nodes_out = f.bounded;
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

// ---- the frontier behind the C ABI: what a host that shards the search over processes drives (partitionedls.jl_amd/dist.py) ------------
partls_status partls_frontier_create(int n_groups, int rank, int world, int64_t batch, partls_frontier **out)
try {
    if (!out) { set_error("partls_frontier_create: out is NULL"); return PARTLS_ERR_BAD_ARG; }
    *out = nullptr;
    if (n_groups < 1 || n_groups > 62 || world < 1 || rank < 0 || rank >= world || batch < 1) { set_error("partls_frontier_create: bad argument"); return PARTLS_ERR_BAD_ARG; }
    partls_frontier *f = new partls_frontier();
    f->rank = rank; f->world = world; f->batch = batch;
    f->refs.resize((size_t)world);
    f->best_free = ((uint64_t)1 << n_groups) - 1;
    f->heap.push_node({0.0, 0ULL, ((uint64_t)1 << n_groups) - 1, f->seq++, -1, -1});
    *out = f;
    return PARTLS_OK;
}
catch (...) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }

void partls_frontier_destroy(partls_frontier *f) { delete f; }

partls_status partls_frontier_next(partls_frontier *f, int64_t *total, int64_t *mine, uint64_t *pat, uint64_t *free_groups, int32_t *src_slot,
                                   int32_t *per_rank)
try {
    if (!f || !total || !mine || !pat || !free_groups || !src_slot || !per_rank) { set_error("partls_frontier_next: bad argument"); return PARTLS_ERR_BAD_ARG; }
    if (!f->round.empty()) { set_error("partls_frontier_next: the previous round has not been ingested"); return PARTLS_ERR_STATE; }
    *total = f->next(mine, pat, free_groups, src_slot, per_rank);
    return PARTLS_OK;
}
catch (...) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_frontier_ingest(partls_frontier *f, const double *lb, const int32_t *branch, const int32_t *dst_slot, int32_t *dead,
                                     int64_t dead_capacity, int64_t *ndead)
try {
    if (!f || !ndead || (!f->round.empty() && (!lb || !branch || !dst_slot))) { set_error("partls_frontier_ingest: bad argument"); return PARTLS_ERR_BAD_ARG; }
    f->ingest(lb, branch, dst_slot);
    const int64_t nd = (int64_t)f->dead.size() < dead_capacity ? (int64_t)f->dead.size() : dead_capacity;
    for (int64_t i = 0; i < nd; ++i) dead[i] = f->dead[f->dead.size() - 1 - (size_t)i];
    f->dead.resize(f->dead.size() - (size_t)nd);               // what did not fit comes with the next call
    *ndead = nd;
    return PARTLS_OK;
}
catch (...) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_frontier_result(const partls_frontier *f, double *mu, uint64_t *pat, uint64_t *free_groups, int64_t *nodes)
{
    if (!f || !mu || !pat || !free_groups) { set_error("partls_frontier_result: bad argument"); return PARTLS_ERR_BAD_ARG; }
    *mu = f->mu; *pat = f->best_pat; *free_groups = f->best_free;
    if (nodes) *nodes = f->bounded;
    return PARTLS_OK;
}

partls_status partls_bnb_prepared(partls_ctx *c, double *alpha, double *beta, double *t, double *opt, int64_t *nopen)
try {
    if (!c || !c->prepared || !c->faithful) { set_error("partls_bnb_prepared: needs a context prepared with PARTLS_OPT_FAITHFUL_INTERCEPT"); return PARTLS_ERR_STATE; }
    if (!alpha || !beta || !t || !opt) { set_error("partls_fit_bnb: NULL argument"); return PARTLS_ERR_BAD_ARG; }
    double mu = INFINITY;
    uint64_t best_pat = 0, best_free = 0;
    int64_t bounded = 0;
    partls_status st = partls_bnb_search(c, 0, &mu, &best_pat, &best_free, &bounded);
    if (st != PARTLS_OK) return st;
    if (!(mu < INFINITY)) { set_error("partls_fit_bnb: no feasible leaf found"); return PARTLS_ERR_NOT_CONVERGED; }
    if (nopen) *nopen = bounded;
    return partls_bnb_leaf(c, best_pat, best_free, alpha, beta, t, opt);
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

}  // extern "C"
