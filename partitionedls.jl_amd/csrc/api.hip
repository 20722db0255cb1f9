// api.hip — the C ABI of include/partls.h: host orchestration of the HIP kernels.  No CPU fallback: every compute
// entry needs a HIP device and fails with PARTLS_ERR_NO_DEVICE / PARTLS_ERR_HIP otherwise.
#include "ctx.h"
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <algorithm>
#include <numeric>
#include <new>
#include <thread>

namespace partls {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void t_begin(partls_ctx *c, int w) { (void)hipEventRecord(c->ev0[w], c->stream); }
void t_end(partls_ctx *c, int w) { (void)hipEventRecord(c->ev1[w], c->stream); c->timed[w] = true; }
void t_collect(partls_ctx *c)
{
    for (int w = 0; w < PARTLS_T_COUNT; ++w)
        if (c->timed[w]) {
            float f = 0.f;
            if (hipEventElapsedTime(&f, c->ev0[w], c->ev1[w]) == hipSuccess) c->ms[w] = (double)f;
            c->timed[w] = false;
        }
}

partls_status check_common(partls_ctx *c, const void *X, int64_t N, int64_t M, int64_t ldX, const void *P, int64_t K, int64_t ldP)
{
    if (!c) { set_error("context is NULL"); return PARTLS_ERR_BAD_ARG; }
    if (!X || !P) { set_error("X or P is NULL"); return PARTLS_ERR_BAD_ARG; }
    if (N < 1 || M < 1 || K < 1) { set_error("need N, M, K >= 1 (got %lld, %lld, %lld)", (long long)N, (long long)M, (long long)K); return PARTLS_ERR_BAD_ARG; }
    if (ldX < N || ldP < M) { set_error("leading dimension smaller than the row count"); return PARTLS_ERR_BAD_ARG; }
    if (ldX >= ((int64_t)1 << 30)) { set_error("ldX = %lld: this build supports leading dimensions below 2^30 rows", (long long)ldX); return PARTLS_ERR_UNSUPPORTED; }
    // group masks are 64-bit words with the intercept's group on bit K; the ENUMERATION of Opt is limited further (see opt_range_ok)
    if (K > 61) { set_error("K = %lld groups: this build supports K <= 61 (Alt, BnB, predict) and K <= 39 for the enumeration of fit(Opt)", (long long)K); return PARTLS_ERR_UNSUPPORTED; }
    if (M + 1 > 1023) { set_error("M = %lld features: this build supports M <= 1022", (long long)M); return PARTLS_ERR_UNSUPPORTED; }
    return PARTLS_OK;
}

static partls_status load_partition(partls_ctx *c, const int64_t *P, int64_t M, int64_t K, int64_t ldP)
{
    c->P.assign((size_t)M * K, 0);
    c->mask_aug.assign((size_t)M + 2, 0);
    for (int64_t k = 0; k < K; ++k)
        for (int64_t m = 0; m < M; ++m) {
            const int64_t v = P[m + k * ldP];
            if (v != 0 && v != 1) { set_error("P[%lld,%lld] = %lld is not 0/1", (long long)m, (long long)k, (long long)v); return PARTLS_ERR_BAD_PARTITION; }
            c->P[(size_t)m + (size_t)k * M] = v;
            if (v) c->mask_aug[(size_t)m] |= (1ULL << k);
        }
    c->mask_aug[(size_t)M] = 1ULL << K;            // the intercept's own group (homogeneousCoords, PartitionedLS.jl:78)
    c->mask_aug[(size_t)M + 1] = 0;                // y
    return PARTLS_OK;
}

double h_reg(const partls_ctx *c, int a, int b)
{
    double v = c->hG[(size_t)a * c->ldg + b];
    if (c->eta != 0.0 && a <= c->M && b <= c->M) v += c->eta * (double)__builtin_popcountll(c->mask_aug[a] & c->mask_aug[b]);
    return v;
}

// Host -> device copy of a column-major matrix (N x M, leading dimension ldX) into a packed device image (leading dimension N).
// Measured on the MI355X box (tools/ubench/h2d_paths.hip, profiles/r04_h2d_paths.txt): the link gives 57 GB/s from page-locked memory;
// hipMemcpy2DAsync from PAGEABLE memory reaches that only when the runtime has pinned the very same pages before — a caller's fresh
// array goes at 8 GB/s (205 MB, C3) to 25 GB/s (4.1 GB, C4), the pinning itself costs as much as the transfer.  Staged through
// page-locked buffers by a few copier threads the same array goes at 42-55 GB/s whatever its history: UP_T threads, each with its own
// stream and two staging buffers, own a contiguous range of columns; a thread packs a batch of columns into one buffer (memcpy) while
// the DMA of its previous batch runs from the other.  Small matrices (< 8 MB) take the plain copy.
namespace {
constexpr int UP_T = 4;
constexpr size_t UP_BUF = (size_t)8 << 20;
}
static partls_status upload_matrix(partls_ctx *c, double *dst, const double *X, int64_t N, int64_t M, int64_t ldX)
{
    const size_t bytes = (size_t)N * M * sizeof(double);
    if (bytes < ((size_t)8 << 20) || c->knobs.no_staged_upload) {
        PARTLS_HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)N * sizeof(double), X, (size_t)ldX * sizeof(double), (size_t)N * sizeof(double), (size_t)M,
                                          hipMemcpyHostToDevice, c->stream));
        return PARTLS_OK;
    }
    if (!c->upPin[0]) {
        for (int i = 0; i < 2 * UP_T; ++i) PARTLS_HIP_CHECK(hipHostMalloc((void **)&c->upPin[i], UP_BUF, hipHostMallocDefault));
        for (int t = 0; t < UP_T; ++t) PARTLS_HIP_CHECK(hipStreamCreateWithFlags(&c->upStream[t], hipStreamNonBlocking));
        for (int i = 0; i < 2 * UP_T; ++i) PARTLS_HIP_CHECK(hipEventCreateWithFlags(&c->upEvent[i], hipEventDisableTiming));
    }
    // rows per piece of a column (a column longer than a staging buffer goes in pieces), columns per batch otherwise
    const size_t col_bytes = (size_t)N * sizeof(double);
    hipError_t err[UP_T];
    for (int t = 0; t < UP_T; ++t) err[t] = hipSuccess;
    const int device = c->device;
    auto worker = [&](int t) {
        hipError_t e = hipSetDevice(device);
        const int64_t c0 = M * t / UP_T, c1 = M * (t + 1) / UP_T;
        char *pin[2] = {c->upPin[2 * t], c->upPin[2 * t + 1]};
        bool used[2] = {false, false};
        int b = 0;
        auto flush = [&](char *d, size_t n) {           // DMA of the buffer just filled; the other buffer is filled meanwhile
            if (e == hipSuccess) e = hipMemcpyAsync(d, pin[b], n, hipMemcpyHostToDevice, c->upStream[t]);
            if (e == hipSuccess) e = hipEventRecord(c->upEvent[2 * t + b], c->upStream[t]);
            used[b] = true;
            b ^= 1;
            if (used[b] && e == hipSuccess) e = hipEventSynchronize(c->upEvent[2 * t + b]);      // the buffer about to be refilled is free again
        };
        if (col_bytes <= UP_BUF) {
            const int64_t per = (int64_t)(UP_BUF / col_bytes);
            for (int64_t j0 = c0; j0 < c1 && e == hipSuccess; j0 += per) {
                const int64_t j1 = j0 + per < c1 ? j0 + per : c1;
                for (int64_t j = j0; j < j1; ++j) std::memcpy(pin[b] + (size_t)(j - j0) * col_bytes, X + j * ldX, col_bytes);
                flush(reinterpret_cast<char *>(dst + j0 * N), (size_t)(j1 - j0) * col_bytes);
            }
        } else {
            const int64_t rows = (int64_t)(UP_BUF / sizeof(double));
            for (int64_t j = c0; j < c1 && e == hipSuccess; ++j)
                for (int64_t r0 = 0; r0 < N && e == hipSuccess; r0 += rows) {
                    const int64_t r1 = r0 + rows < N ? r0 + rows : N;
                    std::memcpy(pin[b], X + j * ldX + r0, (size_t)(r1 - r0) * sizeof(double));
                    flush(reinterpret_cast<char *>(dst + j * N + r0), (size_t)(r1 - r0) * sizeof(double));
                }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->upStream[t]);
        err[t] = e;
    };
    {
        std::vector<std::thread> th;
        th.reserve(UP_T);
        bool spawned = true;
        try { for (int t = 1; t < UP_T; ++t) th.emplace_back(worker, t); }
        catch (...) { spawned = false; }
        worker(0);
        for (std::thread &w : th) w.join();
        if (!spawned) for (int t = (int)th.size() + 1; t < UP_T; ++t) worker(t);      // out of threads: the caller's thread takes the rest
    }
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
    for (int t = 0; t < UP_T; ++t) if (err[t] != hipSuccess) { set_error("staged upload of X failed: %s", hipGetErrorString(err[t])); return PARTLS_ERR_HIP; }
    return PARTLS_OK;                                    // every copier has synchronised its stream: the image is complete for c->stream
}

partls_status ctx_prepare(partls_ctx *c, const double *X, int64_t N, int64_t M, int64_t ldX, const double *y, int x_on_device,
                          const int64_t *P, int64_t K, int64_t ldP, double eta, bool faithful, uint32_t flags)
{
    partls_status st = check_common(c, X, N, M, ldX, P, K, ldP);
    if (st != PARTLS_OK) return st;
    if (!y) { set_error("y is NULL"); return PARTLS_ERR_BAD_ARG; }
    if (!(eta >= 0.0)) { set_error("eta must be >= 0"); return PARTLS_ERR_BAD_ARG; }
    c->prepared = false;
    c->peers.clear();                              // a row-sharded fit sets them again after every rank has prepared its block
    c->near_for = -1; c->near_pat.clear(); c->cand.clear();
    c->last_upload_ms = 0.0; c->last_upload_bytes = 0.0;
    c->sweep_vetoes = 0;
    c->coop_state_valid = false;
    c->order_ready = false; c->order_identity = true; c->flip_cost.clear(); c->ms[PARTLS_T_CALIB] = 0.0;
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
    st = load_partition(c, P, M, K, ldP);
    if (st != PARTLS_OK) return st;
    c->N = N; c->M = M; c->K = K; c->eta = eta; c->flags = flags; c->faithful = faithful;

    if (x_on_device) {
        c->dX = X; c->dy = y; c->ldX = ldX;
    } else {
        PARTLS_HIP_CHECK(c->ownX.ensure((size_t)N * M * sizeof(double)));
        PARTLS_HIP_CHECK(c->ownY.ensure((size_t)N * sizeof(double)));
        PARTLS_HIP_CHECK(hipMemcpyAsync(c->ownY.p, y, (size_t)N * sizeof(double), hipMemcpyHostToDevice, c->stream));
        const auto u0 = std::chrono::steady_clock::now();
        st = upload_matrix(c, c->ownX.as<double>(), X, N, M, ldX);
        if (st != PARTLS_OK) return st;
        c->last_upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - u0).count();
        c->last_upload_bytes = (double)N * (double)M * sizeof(double);
        c->dX = c->ownX.as<double>(); c->dy = c->ownY.as<double>(); c->ldX = N;
    }

    // Gram products (fp64 MFMA)
    const size_t slabd = gram_slab_doubles(N, M, c->knobs.gram_S, c->knobs.gram_cr, &c->chunks, &c->ldg);
    PARTLS_HIP_CHECK(c->slab.ensure(slabd * sizeof(double)));
    PARTLS_HIP_CHECK(c->G.ensure((size_t)c->ldg * c->ldg * sizeof(double)));
    t_begin(c, PARTLS_T_GRAM);
    PARTLS_HIP_CHECK(launch_gram(c->dX, N, M, c->ldX, c->dy, c->slab.as<double>(), c->chunks, c->ldg, c->knobs.gram_S, c->knobs.gram_cr,
                                 c->G.as<double>(), c->stream));
    t_end(c, PARTLS_T_GRAM);
    // rows of X sharded over several devices: the Gram products of the blocks are summed here (partls_fit_opt_multi, multi.hip)
    if (c->gram_hook) { st = c->gram_hook(c); if (st != PARTLS_OK) return st; }

    // tableau variables, grouped by partition (stable sort on the lowest group a variable belongs to) so that the
    // variables one Gray-code flip touches sit in as few 16-wide tile columns as possible
    c->n = faithful ? (int)M + 1 : (int)M;
    c->kbits = faithful ? (int)K + 1 : (int)K;
    c->perm.resize((size_t)c->n);
    std::iota(c->perm.begin(), c->perm.end(), 0);
    auto key = [&](int v) { const uint64_t m = c->mask_aug[(size_t)v]; return m ? __builtin_ctzll(m) : 64; };
    std::stable_sort(c->perm.begin(), c->perm.end(), [&](int a, int b) { return key(a) < key(b); });
    c->mask_tab.resize((size_t)c->n);
    for (int i = 0; i < c->n; ++i) c->mask_tab[(size_t)i] = c->mask_aug[(size_t)c->perm[(size_t)i]];

    // one upload: [group masks of the augmented variables (M + 2) | group masks in tableau order (n) | permutation (n ints)]
    {
        const size_t words = (size_t)M + 2 + (size_t)c->n + ((size_t)c->n + 1) / 2;
        c->pack.assign(words, 0);
        std::memcpy(c->pack.data(), c->mask_aug.data(), ((size_t)M + 2) * sizeof(uint64_t));
        std::memcpy(c->pack.data() + M + 2, c->mask_tab.data(), (size_t)c->n * sizeof(uint64_t));
        std::memcpy(c->pack.data() + M + 2 + c->n, c->perm.data(), (size_t)c->n * sizeof(int));
        PARTLS_HIP_CHECK(c->maskAugD.ensure(words * sizeof(uint64_t)));
        PARTLS_HIP_CHECK(hipMemcpyAsync(c->maskAugD.p, c->pack.data(), words * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        c->maskTabP = c->maskAugD.as<uint64_t>() + M + 2;
        c->permP = reinterpret_cast<int *>(c->maskAugD.as<uint64_t>() + M + 2 + c->n);
    }
    PARTLS_HIP_CHECK(c->scale.ensure((size_t)c->n * sizeof(double)));
    PARTLS_HIP_CHECK(c->Tfull.ensure((size_t)(c->n + 1) * (c->n + 1) * sizeof(double)));
    t_begin(c, PARTLS_T_PREP);
    PARTLS_HIP_CHECK(launch_prep(c->G.as<double>(), c->ldg, (int)M, eta, c->maskAugD.as<uint64_t>(), faithful ? 0 : 1,
                                 c->permP, c->scale.as<double>(), c->Tfull.as<double>(), c->n, c->stream));
    c->use_reg = sweep_reg_supported(c->n) && c->n <= 16 * c->knobs.reg_maxt && !(flags & PARTLS_OPT_GENERIC_KERNEL);
    if (c->use_reg) {
        c->T = sweep_reg_tiles(c->n);
        PARTLS_HIP_CHECK(c->T0reg.ensure(sweep_reg_t0_doubles(c->T) * sizeof(double)));
        PARTLS_HIP_CHECK(launch_layout_reg(c->Tfull.as<double>(), c->n, c->T, c->T0reg.as<double>(), c->stream));
    }
    t_end(c, PARTLS_T_PREP);
    PARTLS_HIP_CHECK(c->hG.resize((size_t)c->ldg * c->ldg));
    PARTLS_HIP_CHECK(c->hScale.resize((size_t)c->n));
    PARTLS_HIP_CHECK(hipMemcpyAsync(c->hG.data(), c->G.p, c->hG.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PARTLS_HIP_CHECK(hipMemcpyAsync(c->hScale.data(), c->scale.p, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));
    t_collect(c);
    // NaN / Inf anywhere in column m of [X y] makes the diagonal Gram entry sum_i z_im^2 non-finite (so does a finite column whose
    // squares overflow — equally outside the Gram form): M + 2 host compares instead of a separate pass over X, which cost 0.87 ms
    // of the 4.1 GB read at C4 before the Gram kernel read the same bytes again
    for (int64_t i = 0; i < M + 2; ++i) {
        if (i == M) continue;                                    // the ones column
        if (!std::isfinite(c->hG[(size_t)i * c->ldg + i])) { set_error("X or y contains NaN/Inf (or overflows in X'X)"); return PARTLS_ERR_NONFINITE; }
    }
    const double yy = c->hG[(size_t)(M + 1) * c->ldg + (M + 1)];
    c->tol = c->knobs.tol_rel * std::sqrt(yy > 0.0 ? yy : 0.0);
    if (!(c->tol > 0.0)) c->tol = 1e-300;
    c->prepared = true;
    return PARTLS_OK;
}

// fit(Opt) enumerates 2^K' patterns: beyond K' = 40 that is out of range (and of the tables of the visiting order)
static bool opt_range_ok(const partls_ctx *c, const char *who)
{
    if (c->kbits <= 40) return true;
    set_error("%s: %d sign bits: the enumeration of 2^(K+1) patterns is out of range (K <= 39); fit(Alt) and fit(BnB) take up to 61 groups", who, c->kbits);
    return false;
}

static hipError_t launch_any_sweep(partls_ctx *c, SweepParams &p, int grid)
{
    if (c->use_reg) {
        p.T0 = c->T0reg.as<double>();
        return launch_sweep_blk(p, c->T, grid, c->stream);
    }
    p.T0 = c->Tfull.as<double>();
    return c->knobs.eager_generic ? launch_sweep_generic(p, grid, c->stream) : launch_sweep_lazy(p, grid, c->stream);
}

void opt_codes(const partls_ctx *c, uint64_t pattern, std::vector<int8_t> &codes)
{
    // multiplier of Opt.jl:28-29: f_m = sum_k P[m,k] s_k; only its sign matters for the constraint f_m w_m >= 0 (0: column is zero)
    codes.resize((size_t)c->n);
    for (int i = 0; i < c->n; ++i) {
        const uint64_t m = c->mask_tab[(size_t)i];
        const int f = 2 * __builtin_popcountll(m & pattern) - __builtin_popcountll(m);
        codes[(size_t)i] = (int8_t)((f > 0) - (f < 0));
    }
}

partls_status solve_nodes(partls_ctx *c, const std::vector<int8_t> &codes, size_t cnt, std::vector<double> &sols,
                          std::vector<double> &obj2, unsigned long long *unconv, bool resume, bool want_tab)
{
    c->tab_valid = false;
    const int n = c->n, ld = n + 1;
    sols.assign(cnt * (size_t)n, 0.0);
    obj2.assign(cnt, 0.0);
    if (unconv) *unconv = 0;
    if (cnt == 0) return PARTLS_OK;
    if (codes.size() != cnt * (size_t)n) { set_error("solve_nodes: code array has the wrong size"); return PARTLS_ERR_BAD_ARG; }
    const int grid = (int)std::min<size_t>(cnt, c->use_reg ? 2048 : 512);
    PARTLS_HIP_CHECK(c->nodeCode.ensure(cnt * (size_t)n));
    // one output block on the device, one copy back: [counters (4 x 8 B) | objective^2 (cnt) | solutions (cnt x n)]
    const size_t out_words = 4 + cnt + cnt * (size_t)n;
    PARTLS_HIP_CHECK(c->nodeSol.ensure(out_words * sizeof(double)));
    PARTLS_HIP_CHECK(c->bestObj.ensure(sizeof(double) * 4096));
    PARTLS_HIP_CHECK(c->bestPat.ensure(sizeof(int64_t) * 4096));
    // one large problem: many workgroups on a single global-memory tableau (sweep_coop.hip) — unless a previous attempt of this very
    // call found the device too crowded for its grid barrier (`coop_fallback`, set below)
    const bool coop = !c->use_reg && cnt == 1 && !c->knobs.no_coop && !c->coop_fallback;
    c->coop_fallback = false;
    if (coop) {
        const size_t need = ((size_t)2 * ld * ld + (size_t)n / 8 + 2) * sizeof(double);  // two tableau images + basis flags + current image
        if (c->scratch.bytes < need) c->coop_state_valid = false;
        PARTLS_HIP_CHECK(c->scratch.ensure(need));
    } else {
        c->coop_state_valid = false;
        if (!c->use_reg) PARTLS_HIP_CHECK(c->scratch.ensure((size_t)grid * ld * ld * sizeof(double)));
        else PARTLS_HIP_CHECK(c->scratch.ensure(64 * sizeof(double)));
    }
    PARTLS_HIP_CHECK(hipMemsetAsync(c->nodeSol.p, 0, 4 * sizeof(unsigned long long), c->stream));
    PARTLS_HIP_CHECK(hipMemcpyAsync(c->nodeCode.p, codes.data(), cnt * (size_t)n, hipMemcpyHostToDevice, c->stream));
    SweepParams p{};
    p.n = n; p.kbits = c->kbits;
    p.mask = c->maskTabP;
    p.scratch = c->scratch.as<double>();
    p.g_begin = 0; p.g_end = (int64_t)cnt; p.chain_len = 1;
    p.tol = c->tol; p.piv_eps = 1e-11; p.max_rounds = 20 * (n + 1);
    p.all_opt = nullptr;
    p.best_obj = c->bestObj.as<double>(); p.best_pat = c->bestPat.as<int64_t>();
    p.n_unconverged = c->nodeSol.as<unsigned long long>();
    p.n_pivots = c->nodeSol.as<unsigned long long>() + 1;
    p.n_vetoes = c->nodeSol.as<unsigned long long>() + 2;
    p.node_code = c->nodeCode.as<int8_t>();
    p.node_obj2 = c->nodeSol.as<double>() + 4; p.node_sol = c->nodeSol.as<double>() + 4 + cnt; p.node_ld = n;
    // the caller will refine this one solution: have the register kernel leave its final tableau (refine_solution's solver)
    // (the cooperative kernel's tableau already lives in global memory: the current image and its basis flags are copied below)
    const bool dump_reg = want_tab && cnt == 1 && c->use_reg, dump_coop = want_tab && coop;
    const bool dump = dump_reg || dump_coop;
    const size_t tabd = dump_reg ? sweep_reg_t0_doubles(c->T) : (dump_coop ? (size_t)ld * ld : 0);
    if (dump) {
        if (dump_reg) {
            PARTLS_HIP_CHECK(c->nodeTab.ensure(tabd * sizeof(double)));
            PARTLS_HIP_CHECK(c->nodeBasic.ensure((size_t)16 * c->T));
        }
        if (c->hTabDoubles < tabd) {                                   // pinned: the 0.3 MB copy then costs ~20 us instead of ~150
            if (c->hTab) (void)hipHostFree(c->hTab);
            if (c->hBasic) (void)hipHostFree(c->hBasic);
            c->hTab = nullptr; c->hBasic = nullptr; c->hTabDoubles = 0;
            PARTLS_HIP_CHECK(hipHostMalloc((void **)&c->hTab, tabd * sizeof(double), hipHostMallocDefault));
            PARTLS_HIP_CHECK(hipHostMalloc((void **)&c->hBasic, 1024 + 16 /* >= 16 x MAXT of sweep_blk.hip, >= n + 1 <= 1024 flags of the cooperative kernel */, hipHostMallocDefault));
            c->hTabDoubles = tabd;
        }
        if (dump_reg) { p.node_tab = c->nodeTab.as<double>(); p.node_basic = c->nodeBasic.as<int8_t>(); }
    }
    if (coop) {
        // one large problem: many workgroups cooperate on a single global-memory tableau (sweep_coop.hip)
        p.T0 = c->Tfull.as<double>();
        p.resume = (resume && c->coop_state_valid) ? 1 : 0;
        PARTLS_HIP_CHECK(c->gridCtr.ensure(64));
        p.grid_ctr = c->gridCtr.as<unsigned>();
        p.coop_fault = c->knobs.coop_fault;
        c->coop_state_valid = false;                                   // until this launch is known to have completed
        // 6 rows per workgroup (measured at n = 513: 0.87 / 0.81 / 0.79 / 0.85 ms per alpha-step with 16 / 8 / 6 / 4): its 16 waves take
        // half a row each in the fused update (gj_apply); more workgroups than that only lengthen the grid barrier
        const int rows_wg = c->knobs.coop_rows > 0 ? c->knobs.coop_rows : 6;
        int nwg = (ld + rows_wg - 1) / rows_wg;
        if (nwg > 128) nwg = 128;
        PARTLS_HIP_CHECK(launch_sweep_coop(p, nwg, c->stream));
    } else {
        PARTLS_HIP_CHECK(launch_any_sweep(c, p, grid));
    }
    unsigned long long counters[4] = {0, 0, 0, 0};                 // unconverged, pivots, vetoes, (cooperative kernel) blocks
    // page-locked up to 64 MB (a fit's single solves and node batches: KBs to a few MB); a caller that bounds a million nodes in one cold
    // batch gets a pageable buffer instead of gigabytes of pinned host memory
    std::vector<double> outw_big;
    double *outw_buf;
    if (out_words <= ((size_t)64 << 20) / sizeof(double)) { PARTLS_HIP_CHECK(c->nodeOut.resize(out_words)); outw_buf = c->nodeOut.data(); }
    else { outw_big.resize(out_words); outw_buf = outw_big.data(); }
    const double *outw = outw_buf;
    PARTLS_HIP_CHECK(hipMemcpyAsync(outw_buf, c->nodeSol.p, out_words * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (dump_reg) {
        PARTLS_HIP_CHECK(hipMemcpyAsync(c->hTab, c->nodeTab.p, tabd * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        PARTLS_HIP_CHECK(hipMemcpyAsync(c->hBasic, c->nodeBasic.p, (size_t)16 * c->T, hipMemcpyDeviceToHost, c->stream));
    }
    if (dump_coop) {                                               // flags [n] + index of the current tableau image, then that image
        const char *flagbuf = static_cast<const char *>(c->scratch.p) + (size_t)2 * ld * ld * sizeof(double);
        PARTLS_HIP_CHECK(hipMemcpyAsync(c->hBasic, flagbuf, (size_t)n + 1, hipMemcpyDeviceToHost, c->stream));
        PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));
        const size_t img = (size_t)(c->hBasic[n] & 1) * ld * ld;
        PARTLS_HIP_CHECK(hipMemcpyAsync(c->hTab, c->scratch.as<double>() + img, tabd * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));
    std::memcpy(counters, outw, sizeof(counters));
    if (coop && (counters[0] >> 40)) {
        // grid-barrier timeout: some workgroups of the cooperative grid were not resident (the device is shared with another
        // context or process).  Nothing of that attempt is used; the same node is solved again by ONE workgroup, which needs no
        // co-residency (slower, never hangs).
        c->coop_fallback = true;
        return solve_nodes(c, codes, cnt, sols, obj2, unconv, false, want_tab);
    }
    if (coop) c->coop_state_valid = counters[0] == 0;
    std::copy(outw + 4, outw + 4 + cnt, obj2.begin());
    std::copy(outw + 4 + cnt, outw + out_words, sols.begin());
    if (unconv) *unconv = counters[0];
    c->last_pivots = counters[1]; c->last_vetoes = counters[2]; c->last_blocks = counters[3];
    c->tab_valid = dump && counters[0] == 0;
    c->tab_full = dump_coop;                                       // layout of hTab: full (n+1)^2 matrix, or the register kernel's tiles
    return PARTLS_OK;
}

void unscale_solution(const partls_ctx *c, const double *sol, std::vector<double> &w)
{
    const int M = (int)c->M;
    w.assign((size_t)M + 1, 0.0);
    for (int i = 0; i < c->n; ++i) w[(size_t)c->perm[(size_t)i]] = sol[i] * c->hScale[(size_t)i];
    if (!c->faithful) {
        // intercept eliminated up front: t = (c_I - sum_f G_If w_f) / G_II   (row I of the normal equations)
        double s = h_reg(c, M, M + 1);
        for (int f = 0; f < M; ++f) s -= h_reg(c, M, f) * w[(size_t)f];
        w[(size_t)M] = s / h_reg(c, M, M);
    }
}

// One pass over the DATA of the prepared problem — all of it: this context's rows and, when the rows of X are sharded over several
// devices (partls_fit_opt_multi), those its peers hold:  *obj2 = sum_i (Xo w - y)_i^2  (without the eta rows) and, optionally,
// g = Xo'(y - Xo w) over [features, intercept].  The kernels of every device are queued first (one host thread drives them all), the
// caller's `overlap` work runs on the host meanwhile, then the partial sums are added in a fixed order (context, then peers; slices in
// order): run-to-run reproducible.
partls_status data_pass(partls_ctx *c, const std::vector<double> &w, bool want_obj, bool want_grad, double *obj2, std::vector<double> *g,
                        const std::function<void()> &overlap)
{
    const int64_t M = c->M;
    const int nb = 1024;
    std::vector<partls_ctx *> cs{c};
    cs.insert(cs.end(), c->peers.begin(), c->peers.end());
    for (partls_ctx *q : cs) {
        PARTLS_HIP_CHECK(hipSetDevice(q->device));
        const int64_t N = q->N;
        const int xr = xtr_slices(N);
        PARTLS_HIP_CHECK(q->wdev.ensure((size_t)(M + 1) * sizeof(double)));
        PARTLS_HIP_CHECK(hipMemcpyAsync(q->wdev.p, w.data(), (size_t)(M + 1) * sizeof(double), hipMemcpyHostToDevice, q->stream));
        double *yhat = nullptr;
        if (want_obj) { PARTLS_HIP_CHECK(q->partial.ensure(nb * sizeof(double))); PARTLS_HIP_CHECK(q->hPart.resize((size_t)nb)); }
        if (want_grad) {
            PARTLS_HIP_CHECK(q->yhatD.ensure((size_t)N * sizeof(double)));
            PARTLS_HIP_CHECK(q->gD.ensure((size_t)xr * (M + 1) * sizeof(double)));
            yhat = q->yhatD.as<double>();
            PARTLS_HIP_CHECK(q->hGpart.resize((size_t)xr * (M + 1)));
        }
        PARTLS_HIP_CHECK(launch_residual(q->dX, N, M, q->ldX, want_obj ? q->dy : nullptr, q->wdev.as<double>(), w[(size_t)M],
                                         want_obj ? q->partial.as<double>() : nullptr, nb, yhat, q->stream));
        if (want_grad) {
            PARTLS_HIP_CHECK(launch_xtr(q->dX, N, M, q->ldX, q->dy, yhat, q->gD.as<double>(), q->stream));
            PARTLS_HIP_CHECK(hipMemcpyAsync(q->hGpart.data(), q->gD.p, q->hGpart.size() * sizeof(double), hipMemcpyDeviceToHost, q->stream));
        }
        if (want_obj) PARTLS_HIP_CHECK(hipMemcpyAsync(q->hPart.data(), q->partial.p, nb * sizeof(double), hipMemcpyDeviceToHost, q->stream));
    }
    if (overlap) overlap();
    double s = 0.0;
    if (want_grad) g->assign((size_t)M + 1, 0.0);
    for (partls_ctx *q : cs) {
        PARTLS_HIP_CHECK(hipSetDevice(q->device));
        PARTLS_HIP_CHECK(hipStreamSynchronize(q->stream));
        if (want_obj) for (int b = 0; b < nb; ++b) s += q->hPart[(size_t)b];
        if (want_grad) {
            const int xr = xtr_slices(q->N);
            for (int64_t m = 0; m <= M; ++m) {
                double sg = 0.0;
                for (int r = 0; r < xr; ++r) sg += q->hGpart[(size_t)r * (M + 1) + m];
                (*g)[(size_t)m] += sg;
            }
        }
    }
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
    if (want_obj) *obj2 = s;
    return PARTLS_OK;
}

// the eta rows of regularizeProblem (PartitionedLS.jl:108-123): sqrt(eta) * sum_{m in group k} w_m  ->  their share of obj^2 and of g
static void eta_terms(const partls_ctx *c, const std::vector<double> &w, double *obj2, std::vector<double> *grad)
{
    if (c->eta == 0.0) return;
    const int64_t M = c->M;
    for (int64_t k = 0; k <= c->K; ++k) {
        double gs = 0.0;
        for (int64_t m = 0; m <= M; ++m) if (c->mask_aug[(size_t)m] & (1ULL << k)) gs += w[(size_t)m];
        if (obj2) *obj2 += c->eta * gs * gs;
        if (grad) for (int64_t m = 0; m <= M; ++m) if (c->mask_aug[(size_t)m] & (1ULL << k)) (*grad)[(size_t)m] -= c->eta * gs;
    }
}

partls_status data_objective(partls_ctx *c, const std::vector<double> &w, double *opt, std::vector<double> *grad)
{
    double s = 0.0;
    partls_status st = data_pass(c, w, true, grad != nullptr, &s, grad, {});
    if (st != PARTLS_OK) return st;
    eta_terms(c, w, &s, grad);
    *opt = std::sqrt(s);
    return PARTLS_OK;
}

// When is a data-space KKT violation evidence that the Gram form has lost the problem?  Measured on problems of cond(Xo) 7e2 .. 8e7
// (tools/illcond_check.py, three seeds, Opt and BnB): every fit that equals the oracle's has a violation <= 3e-15 (the rounding of the
// data passes: eps * sqrt(N) * ||r|| / ||y||), every fit that differs from it has one >= 1.2e-11 — the residual gradient along a nearly
// dependent column the tableau could not resolve (d ~ 1e-13: worth g^2 / d in the objective, i.e. the whole difference to the reference's
// model).  The threshold sits between the two bands.  Neither the pivots of the final basis nor the refusals of the sweep separate the
// cases (a basis that avoids the nearly dependent columns is perfectly conditioned), so the violation alone decides.  The cost of the
// tight threshold: a well-conditioned optimum with a variable at its bound whose gradient lies within (1e-12, 1e-11] * ||x|| ||y|| of zero
// — inside the sweep's own tolerance — is reported although it is fine; on continuous data that has probability ~1e-8 per variable.
bool kkt_says_ill_conditioned(const partls_ctx *c)
{
    return c->last_kkt > c->knobs.kkt_tol;
}

double kkt_violation_data(const partls_ctx *c, const std::vector<double> &w, const std::vector<double> &g, const std::vector<int8_t> &code,
                          int *worst)
{
    const int M = (int)c->M;
    const double yy = h_reg(c, M + 1, M + 1);
    const double ynorm = std::sqrt(yy > 0.0 ? yy : 0.0);
    double worstv = 0.0, wmax = 0.0;
    for (int m = 0; m <= M; ++m) wmax = std::max(wmax, std::fabs(w[(size_t)m]));
    if (worst) *worst = -1;
    for (int m = 0; m <= M; ++m) {
        const double d = h_reg(c, m, m);
        if (!(d > 0.0) || !(d > 1e-14 * std::fabs(c->hG[(size_t)m * c->ldg + m]))) continue;   // null column: never in any basis
        const double gs = g[(size_t)m] / (std::sqrt(d) * (ynorm > 0.0 ? ynorm : 1.0));
        const int f = code[(size_t)m];
        double v = 0.0;
        if (f == 2 || w[(size_t)m] != 0.0) v = std::fabs(gs);
        else if (f != 0) v = std::max(0.0, (double)f * gs);
        if (f == 1 || f == -1) v = std::max(v, wmax > 0.0 ? std::max(0.0, -(double)f * w[(size_t)m] / wmax) : 0.0);
        if (v > worstv) { worstv = v; if (worst) *worst = m; }
    }
    return worstv;
}

// Row-oriented Cholesky of a dense SPD matrix, in place (lower triangle, row-major, leading dimension p): L[i][j] = (B[i][j] - <L[i][:j],
// L[j][:j]>) / L[j][j].  The inner products run on 2 x 4 AVX2 lanes where the host has them (every host an MI355X ships in; checked at
// run time) — 358 k multiply-adds at p = 129 in ~25 us instead of ~100: short enough to hide behind the first data pass of the refinement.
// The sums are taken in a fixed order per build target (reproducible run to run; the correction they serve is ~1e-15 of the solution).
#if defined(__x86_64__)
#include <immintrin.h>
// 4 rows x 2 columns at a time: L[i][j] for i = i0..i0+3 and j = j0, j0+1 share the six row loads of a k-step (0.75 loads per FMA instead of
// 2: the plain dot-product form streams the whole factor from L2 once per row — 45 MB at p = 256, which is what bounded it at ~5 GFLOP/s).
__attribute__((target("avx2,fma"))) static inline double hsum4(__m256d v)
{
    double t[4];
    _mm256_storeu_pd(t, v);
    return (t[0] + t[1]) + (t[2] + t[3]);
}
__attribute__((target("avx2,fma"))) static bool chol_rows_avx2(double *L, int p)
{
    int i0 = 0;
    for (; i0 + 3 < p; i0 += 4) {
        double *R0 = L + (size_t)i0 * p, *R1 = R0 + p, *R2 = R1 + p, *R3 = R2 + p;
        // columns strictly before the diagonal block, two at a time
        int j = 0;
        for (; j + 1 < i0; j += 2) {
            const double *C0 = L + (size_t)j * p, *C1 = C0 + p;
            __m256d a00 = _mm256_setzero_pd(), a01 = a00, a10 = a00, a11 = a00, a20 = a00, a21 = a00, a30 = a00, a31 = a00;
            int k = 0;
            for (; k + 3 < j; k += 4) {
                const __m256d c0 = _mm256_loadu_pd(C0 + k), c1 = _mm256_loadu_pd(C1 + k);
                const __m256d r0 = _mm256_loadu_pd(R0 + k), r1 = _mm256_loadu_pd(R1 + k), r2 = _mm256_loadu_pd(R2 + k), r3 = _mm256_loadu_pd(R3 + k);
                a00 = _mm256_fmadd_pd(r0, c0, a00); a01 = _mm256_fmadd_pd(r0, c1, a01);
                a10 = _mm256_fmadd_pd(r1, c0, a10); a11 = _mm256_fmadd_pd(r1, c1, a11);
                a20 = _mm256_fmadd_pd(r2, c0, a20); a21 = _mm256_fmadd_pd(r2, c1, a21);
                a30 = _mm256_fmadd_pd(r3, c0, a30); a31 = _mm256_fmadd_pd(r3, c1, a31);
            }
            double d[4][2] = {{hsum4(a00), hsum4(a01)}, {hsum4(a10), hsum4(a11)}, {hsum4(a20), hsum4(a21)}, {hsum4(a30), hsum4(a31)}};
            double *R[4] = {R0, R1, R2, R3};
            for (int a = 0; a < 4; ++a) {
                for (int kk = k; kk < j; ++kk) { d[a][0] += R[a][kk] * C0[kk]; d[a][1] += R[a][kk] * C1[kk]; }
                const double l0 = (R[a][j] - d[a][0]) / C0[j];
                R[a][j] = l0;
                R[a][j + 1] = (R[a][j + 1] - (d[a][1] + l0 * C1[j])) / C1[j + 1];       // column j + 1 also needs the entry of column j just made
            }
        }
        // the odd column before the block, then the 4 x 4 diagonal block: plain
        for (int a = 0; a < 4; ++a) {
            double *Ri = L + (size_t)(i0 + a) * p;
            for (int jj = j; jj <= i0 + a; ++jj) {
                const double *Cj = L + (size_t)jj * p;
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                int k = 0;
                for (; k + 3 < jj; k += 4) { s0 += Ri[k] * Cj[k]; s1 += Ri[k + 1] * Cj[k + 1]; s2 += Ri[k + 2] * Cj[k + 2]; s3 += Ri[k + 3] * Cj[k + 3]; }
                for (; k < jj; ++k) s0 += Ri[k] * Cj[k];
                const double sv = Ri[jj] - ((s0 + s1) + (s2 + s3));
                if (jj == i0 + a) { if (!(sv > 0.0)) return false; Ri[jj] = std::sqrt(sv); }
                else Ri[jj] = sv / Cj[jj];
            }
        }
    }
    for (int i = i0; i < p; ++i) {                           // the last p mod 4 rows
        double *Li = L + (size_t)i * p;
        for (int j = 0; j <= i; ++j) {
            const double *Lj = L + (size_t)j * p;
            double s0 = 0.0, s1 = 0.0;
            int k = 0;
            for (; k + 1 < j; k += 2) { s0 += Li[k] * Lj[k]; s1 += Li[k + 1] * Lj[k + 1]; }
            for (; k < j; ++k) s0 += Li[k] * Lj[k];
            const double sv = Li[j] - (s0 + s1);
            if (i == j) { if (!(sv > 0.0)) return false; Li[i] = std::sqrt(sv); }
            else Li[j] = sv / Lj[j];
        }
    }
    return true;
}
#endif
static bool chol_rows_plain(double *L, int p)
{
    for (int i = 0; i < p; ++i) {
        double *Li = L + (size_t)i * p;
        for (int j = 0; j <= i; ++j) {
            const double *Lj = L + (size_t)j * p;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int k = 0;
            for (; k + 3 < j; k += 4) { s0 += Li[k] * Lj[k]; s1 += Li[k + 1] * Lj[k + 1]; s2 += Li[k + 2] * Lj[k + 2]; s3 += Li[k + 3] * Lj[k + 3]; }
            for (; k < j; ++k) s0 += Li[k] * Lj[k];
            const double s = Li[j] - ((s0 + s1) + (s2 + s3));
            if (i == j) { if (!(s > 0.0)) return false; Li[i] = std::sqrt(s); }
            else Li[j] = s / Lj[j];
        }
    }
    return true;
}
static bool chol_rows(double *L, int p)
{
#if defined(__x86_64__)
    static const bool fast = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    if (fast) return chol_rows_avx2(L, p);
#endif
    return chol_rows_plain(L, p);
}

partls_status refine_solution(partls_ctx *c, std::vector<double> &w, bool free_intercept, int steps, RefineOut *out)
{
    const int64_t M = c->M;
    if (out) out->have = false;
    std::vector<int> sup;
    for (int m = 0; m <= (int)M; ++m)
        if (w[(size_t)m] != 0.0 || (m == (int)M && free_intercept)) sup.push_back(m);
    const int p = (int)sup.size();
    if (p == 0) return PARTLS_OK;
    // Solver of the correction equations.  With the final tableau of the node solve at hand (register kernel), its basic x basic
    // block is -(D B_BB D)^-1 (D = unit-diagonal scaling, B = regularised Gram, with the free intercept already eliminated by its
    // Schur complement): delta_B = -D T_BB D rhs is one p x p matrix-vector product per step instead of a p^3/6 factorisation.
    // Valid when the basis IS the support (a basic variable that came out exactly 0 would not be in `sup`): otherwise Cholesky.
    std::vector<int> tabsup;                             // tableau indices of the support, when the tableau path applies
    // conditioning of the basis the node solve ended on, while its tableau is at hand: the diagonal of the basic block is -1 / (leave-one-out
    // pivot of that variable), so the smallest leave-one-out pivot is a lower bound of 1 / cond(G~_BB)  (0 = unknown: no tableau)
    c->last_min_loo = 0.0;
    if (c->tab_valid) {
        double tmax = 0.0;
        for (int i = 0; i < c->n; ++i) {
            if (!c->hBasic[i]) continue;
            const int ti = i >> 4, a = i & 15;
            const double d = c->tab_full ? c->hTab[(size_t)i * (c->n + 1) + i] : c->hTab[((size_t)(ti * (ti + 1) / 2 + ti)) * 256 + a + 16 * a];
            tmax = std::max(tmax, std::fabs(d));
        }
        c->last_min_loo = tmax > 0.0 ? 1.0 / tmax : 1.0;
    }
    bool use_tab = c->tab_valid && free_intercept == !c->faithful && !c->knobs.no_tab_refine;
    c->tab_valid = false;                                // one use: the next node solve overwrites the buffers
    if (use_tab) {
        std::vector<int> inv_perm((size_t)M + 1, -1);
        for (int i = 0; i < c->n; ++i) inv_perm[(size_t)c->perm[(size_t)i]] = i;
        int nb = 0;
        for (int i = 0; i < c->n; ++i) nb += c->hBasic[i] ? 1 : 0;
        for (int m : sup) {
            if (m == (int)M && !c->faithful) continue;   // free intercept: eliminated from the tableau, recovered below
            const int i = inv_perm[(size_t)m];
            if (i < 0 || !c->hBasic[i]) { use_tab = false; break; }
            tabsup.push_back(i);
        }
        if ((int)tabsup.size() != nb) use_tab = false;
    }
    const bool tab_full = c->tab_full;
    const int tld = c->n + 1;
    std::vector<double> g((size_t)M + 1), d((size_t)p);
    // `out`: the pass that finds the correction negligible has already computed the squared residual and Xo'(yo - Xo w) one tiny step
    // before the final w; both are carried over exactly (obj^2 -= 2 g'delta, g -= B delta with the host Gram copy) instead of being
    // recomputed by two more passes over X
    std::vector<double> delta;
    if (out) delta.assign((size_t)M + 1, 0.0);
    double obj2_pre = 0.0;
    auto finish_out = [&]() {                                 // w = w_pre + delta, delta tiny: first-order update of (obj^2, g)
        double o2 = obj2_pre;
        for (int64_t m = 0; m <= M; ++m) o2 -= 2.0 * g[(size_t)m] * delta[(size_t)m];
        out->g = g;
        for (int64_t j = 0; j <= M; ++j) {
            const double dj = delta[(size_t)j];
            if (dj == 0.0) continue;
            const double *row = c->hG.data() + (size_t)j * c->ldg;          // row j of the symmetric Gram copy: contiguous
            double *og = out->g.data();
            for (int64_t m = 0; m <= M; ++m) og[m] -= row[m] * dj;
            if (c->eta != 0.0) {
                const uint64_t mj = c->mask_aug[(size_t)j];
                for (int64_t m = 0; m <= M; ++m) og[m] -= c->eta * (double)__builtin_popcountll(c->mask_aug[(size_t)m] & mj) * dj;
            }
        }
        out->obj = std::sqrt(o2 > 0.0 ? o2 : 0.0);
        out->have = true;
    };
    std::vector<double> Lc;                                  // Cholesky factor: allocated only when that path runs
    // row-oriented Cholesky of the regularised Gram on the support (host copy); the inner products carry four independent
    // partial sums so the compiler can vectorise them (the support is all of [features, intercept] in the typical case:
    // p^3 / 6 multiply-adds).  It runs while the device computes the first residual and gradient.
    auto factorise = [&]() -> bool {
        Lc.assign((size_t)p * p, 0.0);
        for (int i = 0; i < p; ++i) {                        // the regularised Gram block of the support, lower triangle
            const double *row = c->hG.data() + (size_t)sup[(size_t)i] * c->ldg;
            double *Li = &Lc[(size_t)i * p];
            for (int j = 0; j <= i; ++j) Li[j] = row[sup[(size_t)j]];
            if (c->eta != 0.0) {
                const uint64_t mi = c->mask_aug[(size_t)sup[(size_t)i]];
                for (int j = 0; j <= i; ++j) Li[j] += c->eta * (double)__builtin_popcountll(mi & c->mask_aug[(size_t)sup[(size_t)j]]);
            }
        }
        return chol_rows(Lc.data(), p);
    };
    const auto r0 = std::chrono::steady_clock::now();
    for (int it = 0; it < steps; ++it) {
        bool spd = true;
        // residual (and squared residual) and gradient on the device(s); the Cholesky factorisation, when it is needed, overlaps with them
        // (round 4 tried to skip the factorisation when the data-space gradient on the support is rounding noise already — C2: 2.6e-16 of
        // ||x|| ||y|| — and took it back: a tiny gradient says nothing about the error along a weak direction of the Gram block (error =
        // gradient / lambda_min: at cond(Xo) = 8e4 a gradient of 1e-16 goes with an error of 1e-7, exactly the case the refinement exists
        // for; found by test_ill_conditioned_model_parity under a forced bit-order calibration).  The factorisation is made cheap instead.)
        partls_status dst = data_pass(c, w, out != nullptr, true, &obj2_pre, &g, [&]() { if (it == 0 && !use_tab) spd = factorise(); });
        if (dst != PARTLS_OK) return dst;
        if (out) {
            eta_terms(c, w, &obj2_pre, nullptr);
            std::fill(delta.begin(), delta.end(), 0.0);
        }
        eta_terms(c, w, nullptr, &g);                        // gradient of the η rows: -eta * sum_k 1_k (1_k' w)
        if (!spd) return PARTLS_OK;                          // not numerically SPD: give up quietly, w unchanged
        if (use_tab) {
            const int nb = (int)tabsup.size();
            const bool elim = !c->faithful;                  // free intercept: rhs and solution go through its Schur complement
            const double gII = elim ? h_reg(c, (int)M, (int)M) : 1.0, gI = g[(size_t)M];
            std::vector<double> rhs((size_t)nb), ds((size_t)nb);
            for (int a = 0; a < nb; ++a) {
                const int i = tabsup[(size_t)a], m = c->perm[(size_t)i];
                double r = g[(size_t)m];
                if (elim) r -= h_reg(c, m, (int)M) * gI / gII;
                rhs[(size_t)a] = r * c->hScale[(size_t)i];
            }
            // y = T x over ALL tableau indices with x = 0 outside the basis (only the basic entries of y are used): contiguous inner
            // loops over the stored layout instead of p^2 indexed look-ups (100 us at p = 256)
            const int nt_ = tab_full ? 0 : c->T, nx = tab_full ? c->n : 16 * c->T;
            std::vector<double> xt((size_t)nx, 0.0), yt((size_t)nx, 0.0);
            for (int a = 0; a < nb; ++a) xt[(size_t)tabsup[(size_t)a]] = rhs[(size_t)a];
            if (tab_full) {
                for (int a = 0; a < nb; ++a) {
                    const double *row = c->hTab + (size_t)tabsup[(size_t)a] * tld;
                    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                    int j = 0;
                    for (; j + 3 < nx; j += 4) { s0 += row[j] * xt[(size_t)j]; s1 += row[j + 1] * xt[(size_t)j + 1]; s2 += row[j + 2] * xt[(size_t)j + 2]; s3 += row[j + 3] * xt[(size_t)j + 3]; }
                    for (; j < nx; ++j) s0 += row[j] * xt[(size_t)j];
                    yt[(size_t)tabsup[(size_t)a]] = (s0 + s1) + (s2 + s3);
                }
            } else {
                for (int tj = 0; tj < nt_; ++tj)
                    for (int ti = 0; ti <= tj; ++ti) {                     // stored tiles: element (16 ti + a, 16 tj + b) at [a + 16 b]
                        const double *tile = c->hTab + ((size_t)(tj * (tj + 1) / 2 + ti)) * 256;
                        double *yi = &yt[(size_t)16 * ti], *yj = &yt[(size_t)16 * tj];
                        const double *xi = &xt[(size_t)16 * ti], *xj = &xt[(size_t)16 * tj];
                        for (int b = 0; b < 16; ++b) {
                            const double xb = xj[b];
                            double sj = 0.0;
                            for (int a = 0; a < 16; ++a) { const double v = tile[a + 16 * b]; yi[a] += v * xb; sj += v * xi[a]; }
                            if (ti != tj) yj[b] += sj;                     // the mirrored tile (a diagonal tile is stored whole)
                        }
                    }
            }
            for (int a = 0; a < nb; ++a) {
                ds[(size_t)a] = -yt[(size_t)tabsup[(size_t)a]];
            }
            double dn = 0.0, wn = 0.0, dI = gI;
            for (int a = 0; a < nb; ++a) {
                const int i = tabsup[(size_t)a], m = c->perm[(size_t)i];
                const double dl = ds[(size_t)a] * c->hScale[(size_t)i];
                if (elim) dI -= h_reg(c, (int)M, m) * dl;
                w[(size_t)m] += dl; dn += dl * dl; wn += w[(size_t)m] * w[(size_t)m];
                if (out) delta[(size_t)m] = dl;
            }
            if (elim) { dI /= gII; w[(size_t)M] += dI; dn += dI * dI; wn += w[(size_t)M] * w[(size_t)M]; if (out) delta[(size_t)M] = dI; }
            if (c->knobs.finish_trace) fprintf(stderr, "[refine] step %d: |delta|/|w| = %.3e\n", it, std::sqrt(dn / (wn > 0 ? wn : 1)));
            if (dn <= 1e-18 * wn) { if (out) finish_out(); break; }
            continue;
        }
        for (int i = 0; i < p; ++i) {                        // L z = g_P
            double s = g[(size_t)sup[(size_t)i]];
            for (int k = 0; k < i; ++k) s -= Lc[(size_t)i * p + k] * d[(size_t)k];
            d[(size_t)i] = s / Lc[(size_t)i * p + i];
        }
        for (int i = p - 1; i >= 0; --i) {                   // L' delta = z
            double s = d[(size_t)i];
            for (int k = i + 1; k < p; ++k) s -= Lc[(size_t)k * p + i] * d[(size_t)k];
            d[(size_t)i] = s / Lc[(size_t)i * p + i];
        }
        double dn = 0.0, wn = 0.0;
        for (int i = 0; i < p; ++i) { w[(size_t)sup[(size_t)i]] += d[(size_t)i]; dn += d[(size_t)i] * d[(size_t)i]; wn += w[(size_t)sup[(size_t)i]] * w[(size_t)sup[(size_t)i]]; if (out) delta[(size_t)sup[(size_t)i]] = d[(size_t)i]; }
        if (c->knobs.finish_trace) {
            double gmax = 0.0;
            const double yy = h_reg(c, (int)M + 1, (int)M + 1);
            for (int i = 0; i < p; ++i) { const int m = sup[(size_t)i]; const double dd = h_reg(c, m, m); if (dd > 0.0 && yy > 0.0) gmax = std::max(gmax, std::fabs(g[(size_t)m]) / std::sqrt(dd * yy)); }
            fprintf(stderr, "[refine] step %d (Cholesky): |delta|/|w| = %.3e, max |g_S| / (|x||y|) before it = %.3e\n", it, std::sqrt(dn / (wn > 0 ? wn : 1)), gmax);
        }
        // the iteration contracts by cond^2 eps per step: once a correction is below 1e-9 relative, the next one is below
        // round-off for every problem the Gram path can solve at all
        if (dn <= 1e-18 * wn) { if (out) finish_out(); break; }
    }
    if (c->knobs.finish_trace)
        fprintf(stderr, "[refine] support %d: %.3f ms (factorisation overlapped with the first residual / gradient pass)\n", p,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r0).count());
    return PARTLS_OK;
}

// cleanupResult (Opt.jl:34-44) from w = f∘α: raw α_m = w_m / f_m, raw β_k = s_k
static void cleanup_opt(const partls_ctx *c, const std::vector<double> &w, uint64_t pattern, double *alpha, double *beta, double *t)
{
    const int64_t M = c->M, K = c->K;
    std::vector<double> a((size_t)M, 0.0);
    for (int64_t m = 0; m < M; ++m) {
        const int f = 2 * __builtin_popcountll(c->mask_aug[(size_t)m] & pattern) - __builtin_popcountll(c->mask_aug[(size_t)m]);
        a[(size_t)m] = (f != 0) ? w[(size_t)m] / (double)f : 0.0;
        if (a[(size_t)m] < 0.0) a[(size_t)m] = 0.0;      // round-off guard: nonneg_lsq never returns negatives
    }
    std::vector<double> A((size_t)K, 0.0);
    for (int64_t k = 0; k < K; ++k) {
        double s = 0.0;
        for (int64_t m = 0; m < M; ++m) s += (double)c->P[(size_t)m + (size_t)k * M] * a[(size_t)m];
        const double sk = ((pattern >> k) & 1ULL) ? 1.0 : -1.0;
        beta[k] = sk * s;
        A[(size_t)k] = (s == 0.0) ? 1.0 : s;
    }
    for (int64_t m = 0; m < M; ++m) {
        double s = 0.0;
        for (int64_t k = 0; k < K; ++k) s += (double)c->P[(size_t)m + (size_t)k * M] * a[(size_t)m] / A[(size_t)k];
        alpha[m] = s;
    }
    *t = w[(size_t)M];                                   // t = β[end]*α[end] = f_I α_I = w_I (Opt.jl:92)
}

}  // namespace partls

using namespace partls;

extern "C" {

int partls_version(void) { return 100; }
const char *partls_last_error(void) { return g_err; }

int partls_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

partls_status partls_create(int device, partls_ctx **out)
try {
    if (!out) { set_error("partls_create: out is NULL"); return PARTLS_ERR_BAD_ARG; }
    *out = nullptr;
    int n = partls_device_count();
    if (n <= 0 || device < 0 || device >= n) {
        set_error("partls_create: no usable HIP device (count=%d, requested=%d); this library has no CPU fallback", n, device);
        return PARTLS_ERR_NO_DEVICE;
    }
    PARTLS_HIP_CHECK(hipSetDevice(device));
    partls_ctx *c = new (std::nothrow) partls_ctx();
    if (!c) { set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
    c->device = device;
    // environment knobs: read here, once (the compute entries never call getenv)
    if (const char *e = getenv("PARTLS_TOL_REL")) c->knobs.tol_rel = atof(e);
    if (const char *e = getenv("PARTLS_CHAIN_LEN")) c->knobs.chain_len = atoll(e);
    if (const char *e = getenv("PARTLS_GRID")) c->knobs.grid = atoll(e);
    if (const char *e = getenv("PARTLS_GRAM_S")) c->knobs.gram_S = atoi(e);
    if (const char *e = getenv("PARTLS_GRAM_CR")) c->knobs.gram_cr = atoi(e);
    if (const char *e = getenv("PARTLS_COOP_ROWS")) c->knobs.coop_rows = atoi(e);
    if (const char *e = getenv("PARTLS_BIT_ORDER")) c->knobs.bit_order = !strcmp(e, "identity") ? 1 : (!strcmp(e, "calibrate") ? 2 : 0);
    if (const char *e = getenv("PARTLS_KKT_TOL")) c->knobs.kkt_tol = atof(e);
    if (const char *e = getenv("PARTLS_NEAR_TIE_REL")) c->knobs.near_tie_rel = atof(e);
    if (const char *e = getenv("PARTLS_CAL_WB")) c->knobs.cal_wb = atof(e);
    if (const char *e = getenv("PARTLS_CAL_WS")) c->knobs.cal_ws = atof(e);
    c->knobs.no_coop = getenv("PARTLS_NO_COOP") != nullptr;
    c->knobs.no_export = getenv("PARTLS_NO_EXPORT") != nullptr;
    c->knobs.no_staged_upload = getenv("PARTLS_NO_STAGED_UPLOAD") != nullptr;
    if (const char *e = getenv("PARTLS_REG_MAXT")) { const int v = atoi(e); if (v >= 1 && v <= 20) c->knobs.reg_maxt = v; }
    c->knobs.eager_generic = getenv("PARTLS_EAGER_GENERIC") != nullptr;
    c->knobs.bnb_cold = getenv("PARTLS_BNB_COLD") != nullptr;
    if (const char *e = getenv("PARTLS_BNB_BATCH")) c->knobs.bnb_batch = atoi(e);
    if (const char *e = getenv("PARTLS_BNB_POOL_MB")) c->knobs.bnb_pool_mb = atoi(e);
    if (const char *e = getenv("PARTLS_BNB_WG_PER_CU")) c->knobs.bnb_wg_per_cu = atoi(e);
    if (const char *e = getenv("PARTLS_COOP_FAULT")) c->knobs.coop_fault = atoi(e);
    c->knobs.lz_fault = getenv("PARTLS_LZ_FAULT") != nullptr;
    c->knobs.no_tab_refine = getenv("PARTLS_NO_TAB_REFINE") != nullptr;
    c->knobs.finish_trace = getenv("PARTLS_FINISH_TRACE") != nullptr;
    c->knobs.alt_trace = getenv("PARTLS_ALT_TRACE") != nullptr;
    c->knobs.alt_always_check = getenv("PARTLS_ALT_ALWAYS_CHECK") != nullptr;
    c->knobs.print_stamps = getenv("PARTLS_PRINT_STAMPS") != nullptr;
    // a failure below must not leak the context (or the objects already created)
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int w = 0; w < PARTLS_T_COUNT && e == hipSuccess; ++w) {
        e = hipEventCreate(&c->ev0[w]);
        if (e == hipSuccess) e = hipEventCreate(&c->ev1[w]);
    }
    if (e != hipSuccess) {
        set_error("partls_create: stream / event creation failed: %s", hipGetErrorString(e));
        partls_destroy(c);
        return PARTLS_ERR_HIP;
    }
    *out = c;
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

void partls_destroy(partls_ctx *c)
{
    if (!c) return;
    // At process exit the HIP runtime (or a profiler layered on it) may already be torn down when a late destructor gets
    // here: touch the device only while the runtime still answers, otherwise just drop the host object.
    int ndev = 0;
    const bool alive = hipGetDeviceCount(&ndev) == hipSuccess && ndev > c->device && hipSetDevice(c->device) == hipSuccess;
    if (alive) {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        DevBuf *bufs[] = {&c->ownX, &c->ownY, &c->slab, &c->G, &c->maskAugD, &c->scale, &c->Tfull,
                          &c->T0reg, &c->scratch, &c->bestObj, &c->bestPat, &c->counters, &c->allOpt, &c->wdev, &c->partial,
                          &c->yhatD, &c->gD, &c->nodeCode, &c->nodeSol, &c->nodeObj, &c->gridCtr,
                          &c->predX, &c->predY, &c->nodeTab, &c->nodeBasic, &c->altA, &c->altGA, &c->altHg,
                          &c->nodePiv, &c->maskInt, &c->allOptRef, &c->bnbIn, &c->bnbOut, &c->altGersh};
        for (DevBuf *b : bufs) b->release();
        for (void *q : c->bnbChunks) (void)hipFree(q);
        c->bnbChunks.clear();
        c->hG.release();
        c->bnbHostIn.release(); c->bnbHostOut.release();
        c->hScale.release(); c->hPart.release(); c->hGpart.release(); c->sweepOut.release(); c->nodeOut.release(); c->exportSol.release();
        for (int i = 0; i < 8; ++i) { if (c->upPin[i]) (void)hipHostFree(c->upPin[i]); if (c->upEvent[i]) (void)hipEventDestroy(c->upEvent[i]); }
        for (int t = 0; t < 4; ++t) if (c->upStream[t]) (void)hipStreamDestroy(c->upStream[t]);
        if (c->hTab) (void)hipHostFree(c->hTab);
        if (c->hBasic) (void)hipHostFree(c->hBasic);
        for (int w = 0; w < PARTLS_T_COUNT; ++w) {
            if (c->ev0[w]) (void)hipEventDestroy(c->ev0[w]);
            if (c->ev1[w]) (void)hipEventDestroy(c->ev1[w]);
        }
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

partls_status partls_opt_prepare(partls_ctx *c, const double *X, int64_t N, int64_t M, int64_t ldX, const double *y,
                                 int x_on_device, const int64_t *P, int64_t K, int64_t ldP, double eta, uint32_t flags)
try {
    return ctx_prepare(c, X, N, M, ldX, y, x_on_device, P, K, ldP, eta, (flags & PARTLS_OPT_FAITHFUL_INTERCEPT) != 0, flags);
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

int64_t partls_opt_num_patterns(const partls_ctx *c) { return (c && c->prepared && c->kbits <= 40) ? ((int64_t)1 << c->kbits) : 0; }

// Which group sits on which bit of the Gray index.  Bit b flips in 2^-(b+1) of all transitions and a flip exchanges roughly the
// variables of its group that carry signal, so the cheap groups belong on the fast bits: on C3 the reference's order (group k on
// bit k) costs 16.9 M pivots / 74.9 ms, the measured-cost order 13.2 M / 51.0 ms for the same 2^20 subproblems.  The cost of a flip
// is MEASURED on the prepared problem: `ncu` chains of nodes on the kernel the sweep will use, chain c solving a pseudo-random pattern from
// scratch and then flipping the groups of its half of the bits one after the other (each node warm-started from its predecessor,
// exactly as in the sweep); pivots per flip are averaged per group.  Wall time = one chain = (8 + K'/2) patterns' worth, paid once
// per prepare and only when the sweep is long enough to repay it.  Deterministic (fixed walks, no atomics in the solves), so every
// rank of a sharded sweep derives the same order from the same data; dist.py cross-checks that before trusting the shards.
static partls_status calibrate_bit_order(partls_ctx *c)
{
    const int kb = c->kbits, n = c->n;
    c->order_ready = true;
    c->order_identity = true;
    c->flip_cost.clear();
    for (int k = 0; k < 40; ++k) c->order.gbit[k] = (uint8_t)k;
    if (kb < 2 || c->knobs.bit_order == 1) return PARTLS_OK;
    int ncu = 256;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || ncu < 1) ncu = 256;
    const int nseg = kb >= 8 ? 2 : 1;
    const int seg_len = (kb + nseg - 1) / nseg, L = seg_len + 1;
    // one calibration chain costs about (8 + seg_len) patterns (8: the solve from scratch); the sweep gives every CU 2^kb / ncu of them.
    // What it buys depends on the data (nothing when the groups cost the same, a third of the sweep on C3): run it when it costs <= 3 %
    if (c->knobs.bit_order != 2 && ((int64_t)1 << kb) < (int64_t)ncu * 32 * (8 + seg_len)) return PARTLS_OK;
    const int chains = std::max(ncu - ncu % nseg, 2 * nseg);
    const size_t steps = (size_t)chains * L;

    PARTLS_HIP_CHECK(c->nodeCode.ensure(steps * (size_t)n));
    PARTLS_HIP_CHECK(c->nodePiv.ensure((3 * steps + 8) * sizeof(unsigned)));
    PARTLS_HIP_CHECK(c->nodeSol.ensure((4 + (size_t)chains + (size_t)chains * n) * sizeof(double)));
    PARTLS_HIP_CHECK(c->bestObj.ensure(sizeof(double) * (4 + 2 * 4096)));
    PARTLS_HIP_CHECK(c->bestPat.ensure(sizeof(int64_t) * 4096));
    PARTLS_HIP_CHECK(c->scratch.ensure(c->use_reg ? 64 * sizeof(double) : (size_t)chains * (n + 1) * (n + 1) * sizeof(double)));
    PARTLS_HIP_CHECK(hipMemsetAsync(c->nodePiv.p, 0, 8 * sizeof(unsigned), c->stream));          // [unconverged (8 B) | ... | pivots per step]
    t_begin(c, PARTLS_T_CALIB);
    PARTLS_HIP_CHECK(launch_walk_codes(c->maskTabP, n, kb, chains, L, seg_len, nseg, c->nodeCode.as<int8_t>(), c->stream));
    SweepParams p{};
    p.n = n; p.kbits = kb;
    p.mask = c->maskTabP;
    p.scratch = c->scratch.as<double>();
    p.g_begin = 0; p.g_end = (int64_t)steps; p.chain_len = L;
    p.tol = c->tol; p.piv_eps = 1e-11; p.max_rounds = 20 * (n + 1);
    p.best_obj = c->bestObj.as<double>(); p.best_pat = c->bestPat.as<int64_t>();
    p.n_unconverged = c->nodePiv.as<unsigned long long>();
    p.node_code = c->nodeCode.as<int8_t>();
    p.node_obj2 = c->nodeSol.as<double>() + 4; p.node_sol = c->nodeSol.as<double>() + 4 + chains; p.node_ld = n;
    p.node_piv = c->nodePiv.as<unsigned>() + 8;                    // 3 counters per step
    PARTLS_HIP_CHECK(launch_any_sweep(c, p, chains));
    t_end(c, PARTLS_T_CALIB);
    std::vector<unsigned> piv(3 * steps + 8);
    PARTLS_HIP_CHECK(hipMemcpyAsync(piv.data(), c->nodePiv.p, (3 * steps + 8) * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));
    t_collect(c);
    c->coop_state_valid = false;
    c->tab_valid = false;
    unsigned long long unconv = 0;
    std::memcpy(&unconv, piv.data(), sizeof(unconv));
    if (unconv) return PARTLS_OK;                            // a walk hit the pivot cap: the sample says nothing, keep the plain order

    // Cost of a flip in pivot equivalents.  On the register kernel a pattern's cycles split (stamp build) into ~730 + 4.4 NS
    // per pivot (panel step + update; NS = tile slots), ~4 400 per block pivot (gather, scatter, the update's start, barrier waits) and
    // ~2 100 per KKT scan beyond the first, which every pattern pays: a group whose variables straddle tile columns so that a flip takes
    // three blocks instead of two costs as much more as four extra pivots would.  All three counts are exact (no timing), so the
    // order stays a deterministic function of the data.
    const double ns = c->use_reg ? 0.5 * c->T * (c->T + 1) : 0.0;
    const double per_pivot = 730.0 + 4.4 * ns;
    const double w_block = c->use_reg ? c->knobs.cal_wb * 4400.0 / per_pivot : 0.0, w_scan = c->use_reg ? c->knobs.cal_ws * 2100.0 / per_pivot : 0.0;
    std::vector<double> cost((size_t)kb, 0.0);
    std::vector<int> cnt((size_t)kb, 0);
    for (int ch = 0; ch < chains; ++ch)
        for (int i = 1; i < L; ++i) {
            const int k = walk_flipped_bit(ch, i, kb, seg_len, nseg);
            const unsigned *now = &piv[8 + 3 * ((size_t)ch * L + i)], *was = now - 3;
            const double scans = (double)(now[2] - was[2]);
            cost[(size_t)k] += (double)(now[0] - was[0]) + w_block * (double)(now[1] - was[1]) + w_scan * (scans > 1.0 ? scans - 1.0 : 0.0);
            ++cnt[(size_t)k];
        }
    for (int k = 0; k < kb; ++k) cost[(size_t)k] = cnt[(size_t)k] ? cost[(size_t)k] / cnt[(size_t)k] : 0.0;
    std::vector<int> by_cost((size_t)kb);
    std::iota(by_cost.begin(), by_cost.end(), 0);
    std::stable_sort(by_cost.begin(), by_cost.end(), [&](int a, int b) { return cost[(size_t)a] < cost[(size_t)b]; });
    c->flip_cost = cost;
    // pivots per pattern the additive model predicts: sum_b 2^-(b+1) cost(group on bit b).  Sorting noisy estimates of equal costs always
    // "predicts" a gain of about their standard error (~1 %): below 2 % the reference's order stays (measured on such problems: +-1 %)
    double pred_ref = 0.0, pred_sorted = 0.0, wgt = 0.5;
    for (int b = 0; b < kb; ++b, wgt *= 0.5) { pred_ref += wgt * cost[(size_t)b]; pred_sorted += wgt * cost[(size_t)by_cost[(size_t)b]]; }
    if (c->knobs.bit_order != 2 && !(pred_sorted < 0.98 * pred_ref)) return PARTLS_OK;
    bool ident = true;
    for (int b = 0; b < kb; ++b) { c->order.gbit[by_cost[(size_t)b]] = (uint8_t)b; ident = ident && by_cost[(size_t)b] == b; }
    if (ident) return PARTLS_OK;
    std::vector<uint64_t> mi((size_t)n);
    for (int i = 0; i < n; ++i) {
        uint64_t m = c->mask_tab[(size_t)i], q = 0;
        for (; m; m &= m - 1) q |= 1ULL << c->order.gbit[__builtin_ctzll(m)];
        mi[(size_t)i] = q;
    }
    PARTLS_HIP_CHECK(c->maskInt.ensure((size_t)n * sizeof(uint64_t)));
    PARTLS_HIP_CHECK(hipMemcpyAsync(c->maskInt.p, mi.data(), (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));     // `mi` is pageable and goes out of scope
    c->order_identity = false;
    return PARTLS_OK;
}

// internal pattern (group k on bit gbit[k]) -> the reference's pattern index (group k on bit k)
static int64_t reference_pattern(const partls_ctx *c, int64_t q)
{
    if (q < 0 || c->order_identity) return q;
    uint64_t r = 0;
    for (int k = 0; k < c->kbits; ++k) r |= (((uint64_t)q >> c->order.gbit[k]) & 1ULL) << k;
    return (int64_t)r;
}

partls_status partls_opt_sweep(partls_ctx *c, int64_t g_begin, int64_t g_end, double *best_obj, int64_t *best_pattern,
                               double *all_opt, int64_t *n_unconverged)
try {
    if (!c || !c->prepared) { set_error("partls_opt_sweep: context not prepared"); return PARTLS_ERR_STATE; }
    if (!opt_range_ok(c, "partls_opt_sweep")) return PARTLS_ERR_UNSUPPORTED;
    const int64_t npat = (int64_t)1 << c->kbits;
    if (g_end < 0) g_end = npat;
    if (g_begin < 0 || g_begin > g_end || g_end > npat) { set_error("bad Gray-index range [%lld,%lld)", (long long)g_begin, (long long)g_end); return PARTLS_ERR_BAD_ARG; }
    if (all_opt && !c->faithful) { set_error("all_opt needs PARTLS_OPT_FAITHFUL_INTERCEPT"); return PARTLS_ERR_BAD_ARG; }
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
    if (g_begin == g_end) {
        if (best_obj) *best_obj = INFINITY;
        if (best_pattern) *best_pattern = -1;
        if (n_unconverged) *n_unconverged = 0;
        return PARTLS_OK;
    }
    if (!c->order_ready) {
        partls_status st = calibrate_bit_order(c);
        if (st != PARTLS_OK) return st;
    }
    const int n = c->n, ld = n + 1;
    const int64_t total = g_end - g_begin;
    // Chain length.  A chain start costs ~8 patterns' pivots, so chains should be long (~1024 patterns), but the register kernel runs ONE
    // chain per CU at a time and the chains of a range take almost equally long: the sweep lasts ceil(chains / CUs) chain times, and a
    // chain count that is not a multiple of the CU count pays for the whole last round (measured on C3, 256 CUs: 1024 chains of 1024
    // patterns 74.9 ms, 768 of 1366 75.1, but 820 of 1280 90.6 and 1366 of 768 81.5).  So: a whole number k >= 2 of chains per CU.
    int64_t chain_len;
    if (c->knobs.chain_len > 0) chain_len = c->knobs.chain_len;
    else if (c->use_reg) {
        int ncu = 256;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || ncu < 1) ncu = 256;
        // ~1024 patterns per chain; 2048 once that still leaves every CU 8 or more chains to balance with (C5, 2^24 patterns: 690.6 ->
        // 686.7 ms; 4096: 686.6, 8192: 689.8)
        const int64_t per_chain = total >= (int64_t)ncu * 2048 * 8 ? 2048 : 1024;
        const int conc = sweep_reg_concurrency(c->T);           // chains a CU runs at once: 1, or the 256-thread kernel's occupancy
        if (conc > 1) {
            // small tableaus: `slots` chains run at the same time, so a short enumeration is cut into exactly that many chains — down to
            // 4 patterns each: a chain start costs about 8 patterns' pivots, but an idle slot costs a whole chain (BASELINE config 2, 4096
            // patterns on 256 x 3 slots: 683 chains of 6 instead of 256 of 16)
            const int64_t slots = (int64_t)ncu * conc;
            const int64_t k = std::max<int64_t>(1, (total + slots * per_chain - 1) / (slots * per_chain));
            chain_len = std::max<int64_t>(4, (total + k * slots - 1) / (k * slots));
        } else {
        int64_t k = (total + (int64_t)ncu * per_chain - 1) / ((int64_t)ncu * per_chain);
        if (k < 2) k = 2;
        chain_len = (total + k * ncu - 1) / (k * ncu);
        if (chain_len < 16) chain_len = 16;                     // tiny ranges: fewer chains than CUs rather than chains of a few patterns
        }
    } else {
        // global-memory kernels (n > 320): a chain start costs ~n/2 pivots (the first pattern is solved from the empty basis) against ~25 per
        // warm-started pattern, so chains are as long as still leaves every CU one (measured at D = 340, 2^16 / 2^18 patterns: 64 -> 3.79 /
        // 4.17 M solves/s, 128 -> 3.90 / 4.37, 256 -> 4.01 / 4.46)
        int ncu = 256;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || ncu < 1) ncu = 256;
        chain_len = 256;
        while (chain_len > 16 && (total + chain_len - 1) / chain_len < ncu) chain_len >>= 1;
    }
    if (chain_len < 1) chain_len = 1;
    const int64_t nchains = (total + chain_len - 1) / chain_len;
    if (nchains >= (1LL << 31) || chain_len >= (1LL << 31)) { set_error("partls_opt_sweep: more than 2^31 chains in one call; split the Gray-index range"); return PARTLS_ERR_UNSUPPORTED; }
    int grid = (int)std::min<int64_t>(nchains, c->knobs.grid > 0 ? c->knobs.grid : (c->use_reg ? 4096 : 1024));
    if (grid < 1) grid = 1;

    // one output block on the device, one copy back: [counters (4 x 8 B) | best objective (grid) | best pattern (grid) | runner-up
    // objective (grid) | runner-up pattern (grid)]
    const size_t sweep_words = 4 + 4 * (size_t)grid;
    PARTLS_HIP_CHECK(c->bestObj.ensure(sizeof(double) * (4 + 4 * (size_t)std::max(grid, 4096))));
    PARTLS_HIP_CHECK(hipMemsetAsync(c->bestObj.p, 0, 4 * sizeof(unsigned long long), c->stream));
    if (all_opt) {
        PARTLS_HIP_CHECK(c->allOpt.ensure((size_t)npat * sizeof(double)));
        if (total < npat) PARTLS_HIP_CHECK(hipMemsetAsync(c->allOpt.p, 0xFF, (size_t)npat * sizeof(double), c->stream));   // NaN outside the shard
    }
    if (!c->use_reg) PARTLS_HIP_CHECK(c->scratch.ensure((size_t)grid * ld * ld * sizeof(double)));
    else PARTLS_HIP_CHECK(c->scratch.ensure(64 * sizeof(double)));

    SweepParams p{};
    p.n = n; p.kbits = c->kbits;
    p.mask = c->order_identity ? c->maskTabP : c->maskInt.as<uint64_t>();
    p.scratch = c->scratch.as<double>();
    p.g_begin = g_begin; p.g_end = g_end; p.chain_len = chain_len;
    p.tol = c->tol; p.piv_eps = 1e-11; p.max_rounds = 20 * (n + 1);
    p.all_opt = all_opt ? c->allOpt.as<double>() : nullptr;
    p.best_obj = c->bestObj.as<double>() + 4; p.best_pat = reinterpret_cast<int64_t *>(c->bestObj.as<double>() + 4 + grid);
    p.second_obj = c->bestObj.as<double>() + 4 + 2 * (size_t)grid; p.second_pat = reinterpret_cast<int64_t *>(c->bestObj.as<double>() + 4 + 3 * (size_t)grid);
    p.n_unconverged = c->bestObj.as<unsigned long long>();
    p.n_pivots = c->bestObj.as<unsigned long long>() + 1;
    p.n_vetoes = c->bestObj.as<unsigned long long>() + 2;
    if (!c->use_reg && c->knobs.lz_fault) p.coop_fault = 77;   // test hook of the deferred-update kernel's panel (SweepParams::coop_fault)
    for (int k = 0; k < 40; ++k) p.rbit.gbit[k] = (uint8_t)k;
    if (!c->order_identity) for (int k = 0; k < c->kbits; ++k) p.rbit.gbit[c->order.gbit[k]] = (uint8_t)k;   // exact ties: first REFERENCE index
    // the register kernels leave the solution of every workgroup's best pattern behind: partls_opt_finish starts from the winner's
    // instead of solving that pattern again from the empty basis (C2: 89 us of a 0.58 ms fit)
    c->export_wg = -1;
    // (the 256-thread register kernel for small tableaus and the deferred-update kernel beyond n = 320; not the 512-thread kernel)
    if (((c->use_reg && (sweep_reg_small(c->T) || sweep_reg_exports(c->T))) || (!c->use_reg && !c->knobs.eager_generic)) && !c->knobs.no_export) {
        PARTLS_HIP_CHECK(c->bestSol.ensure((size_t)grid * n * sizeof(double)));
        p.best_sol = c->bestSol.as<double>();
        p.node_ld = n;
    }

    t_begin(c, PARTLS_T_SWEEP);
    PARTLS_HIP_CHECK(launch_any_sweep(c, p, grid));
    t_end(c, PARTLS_T_SWEEP);

    std::vector<double> bo((size_t)grid);
    std::vector<int64_t> bp((size_t)grid);
    unsigned long long cnt[3] = {0, 0, 0};
    PARTLS_HIP_CHECK(c->sweepOut.resize(sweep_words));
    const double *sweep_out = c->sweepOut.data();
    PARTLS_HIP_CHECK(hipMemcpyAsync(c->sweepOut.data(), c->bestObj.p, sweep_words * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (all_opt) {
        // only the entries of this shard are set, the others are NaN; the caller merges shards (entries are indexed by pattern)
        const void *src = c->allOpt.p;
        if (!c->order_identity) {                        // the kernel indexed it by the internal pattern
            PARTLS_HIP_CHECK(c->allOptRef.ensure((size_t)npat * sizeof(double)));
            PARTLS_HIP_CHECK(launch_pattern_gather(c->allOpt.as<double>(), npat, c->kbits, c->order, c->allOptRef.as<double>(), c->stream));
            src = c->allOptRef.p;
        }
        PARTLS_HIP_CHECK(hipMemcpyAsync(all_opt, src, (size_t)npat * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));
    t_collect(c);
    std::memcpy(cnt, sweep_out, sizeof(cnt));
    std::memcpy(bo.data(), sweep_out + 4, (size_t)grid * sizeof(double));
    std::memcpy(bp.data(), sweep_out + 4 + grid, (size_t)grid * sizeof(int64_t));
    c->last_pivots = cnt[1];
    c->last_vetoes = cnt[2];
    c->sweep_vetoes = cnt[2];
    if (c->knobs.print_stamps) {                         // diagnostic build (-DPARTLS_STAMPS): phase shares of workgroup 0
        double st[32] = {0};
        if (hipMemcpy(st, c->scratch.p, sizeof(st), hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[partls stamps] scan: pre %.0f barrier %.0f post %.0f | gather: work %.0f barrier %.0f | panel: work %.0f barrier %.0f | "
                            "update %.0f | scatter %.0f | chain-load %.0f/%.0f | pivots %llu || gather split: block setup %.0f chain %.0f rhs/myj %.0f"
                            " || workgroup 0: %.0f blocks, %.0f scans, %.0f pivots\n",
                    st[9], st[10], st[0], st[12] + st[13] + st[8], st[1], st[11], st[2], st[3], st[4], st[5], st[6], cnt[1], st[12], st[13], st[8],
                    st[24], st[25], st[26]);
    }
    double bobj = INFINITY;
    int64_t bpat = -1;
    int best_wg = -1;
    for (int i = 0; i < grid; ++i) {                     // argmin with first-index tie-break (Opt.jl:96)
        if (bp[(size_t)i] < 0) continue;
        bp[(size_t)i] = reference_pattern(c, bp[(size_t)i]);
        if (bpat < 0 || bo[(size_t)i] < bobj || (bo[(size_t)i] == bobj && bp[(size_t)i] < bpat)) { bobj = bo[(size_t)i]; bpat = bp[(size_t)i]; best_wg = i; }
    }
    if (p.best_sol) c->export_wg = best_wg;                 // row of bestSol that holds the winner's solution (valid while near_for == winner)
    // Near ties.  The tracked objective^2 carries the Gram form's absolute error (~eps * y'y times the pivots of the chain), so two
    // patterns closer than that can come out in the wrong order relative to the reference, which computes every objective from the
    // data (Opt.jl:90).  Candidates within that error of the winner — each workgroup reports its minimum and its runner-up — are
    // remembered (at most 3, best first); partls_opt_finish re-ranks them with the objective from the data.
    c->near_pat.clear();
    c->cand.clear();
    c->near_for = bpat;
    if (bpat >= 0) {
        const double yy = h_reg(c, (int)c->M + 1, (int)c->M + 1);
        const double lim2 = bobj * bobj + c->knobs.near_tie_rel * (yy > 0.0 ? yy : 0.0);
        std::vector<std::pair<double, int64_t>> cand;
        const double *so = sweep_out + 4 + 2 * (size_t)grid;
        const int64_t *sp = reinterpret_cast<const int64_t *>(sweep_out + 4 + 3 * (size_t)grid);
        for (int i = 0; i < grid; ++i) {
            if (bp[(size_t)i] >= 0 && bp[(size_t)i] != bpat && bo[(size_t)i] * bo[(size_t)i] <= lim2) cand.emplace_back(bo[(size_t)i], bp[(size_t)i]);
            if (sp[i] >= 0 && so[i] * so[i] <= lim2) { const int64_t r = reference_pattern(c, sp[i]); if (r != bpat) cand.emplace_back(so[i], r); }
        }
        std::sort(cand.begin(), cand.end());
        cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
        c->cand.emplace_back(bobj, bpat);
        for (size_t i = 0; i < cand.size() && c->near_pat.size() < 3; ++i) { c->near_pat.push_back(cand[i].second); c->cand.push_back(cand[i]); }
    }
    if (best_obj) *best_obj = bobj;
    if (best_pattern) *best_pattern = bpat;
    if (n_unconverged) *n_unconverged = (int64_t)cnt[0];
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_opt_finish(partls_ctx *c, int64_t pattern, double *alpha, double *beta, double *t, double *opt,
                                int64_t *best_index)
try {
    if (!c || !c->prepared) { set_error("partls_opt_finish: context not prepared"); return PARTLS_ERR_STATE; }
    if (!alpha || !beta || !t || !opt) { set_error("partls_opt_finish: NULL output"); return PARTLS_ERR_BAD_ARG; }
    if (pattern < 0 || pattern >= ((int64_t)1 << (c->K + 1))) { set_error("pattern out of range"); return PARTLS_ERR_BAD_ARG; }
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
    const uint64_t kmask = ((uint64_t)1 << c->kbits) - 1;
    // A group without any feature leaves the subproblem unchanged: the reference then sees bitwise equal objectives for the two
    // patterns and argmin keeps the first, i.e. the one with that group's bit clear (Opt.jl:96)
    // — and so does a group whose every feature is a null column (scale 0: never in a basis; the reference's X .* f' has +-0 columns)
    uint64_t used = 1ULL << c->K;
    for (int i = 0; i < c->n; ++i) if (c->hScale[(size_t)i] != 0.0) used |= c->mask_tab[(size_t)i];

    // candidates: the given pattern and, when it is the winner of this context's last sweep, the near ties that sweep recorded —
    // distinct subproblems only (patterns that differ in the bits of unused groups are the same subproblem)
    std::vector<uint64_t> cands{(uint64_t)pattern & kmask & used};
    if (pattern == c->near_for)
        for (int64_t q : c->near_pat) {
            const uint64_t v = (uint64_t)q & kmask & used;
            if (std::find(cands.begin(), cands.end(), v) == cands.end()) cands.push_back(v);
        }
    const int export_wg = (pattern == c->near_for) ? c->export_wg : -1;   // the sweep's winner: its solution was left behind by the kernel
    c->export_wg = -1;
    c->near_for = -1;
    c->near_pat.clear();
    c->last_near_evaluated = (int64_t)cands.size();

    t_begin(c, PARTLS_T_FINISH);
    const auto f0 = std::chrono::steady_clock::now();
    std::vector<double> sols, obj2, w, wbest, g, gbest;
    std::vector<int8_t> codes;
    unsigned long long unconv = 0, unconv_best = 0;
    double obest = INFINITY, loo_best = 0.0;
    uint64_t pbest = cands[0];
    for (size_t ci = 0; ci < cands.size(); ++ci) {
        partls_status st = PARTLS_OK;
        bool taken = false;
        if (ci == 0 && export_wg >= 0) {
            // the winner's solution as the sweep left it (scaled, 0 for nonbasic variables — the format of a node solve); accepted when
            // it carries the winning pattern's signs (on an exact objective tie the kernel keeps the FIRST pattern's solution, which
            // may belong to the other pattern of the tie), refined and KKT-checked below like any other
            PARTLS_HIP_CHECK(c->exportSol.resize((size_t)c->n));
            PARTLS_HIP_CHECK(hipMemcpyAsync(c->exportSol.data(), c->bestSol.as<double>() + (size_t)export_wg * c->n, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));
            sols.assign(c->exportSol.data(), c->exportSol.data() + c->n);
            opt_codes(c, cands[ci], codes);
            taken = true;
            double smax = 0.0;
            for (int i = 0; i < c->n; ++i) smax = std::max(smax, std::fabs(sols[(size_t)i]));
            for (int i = 0; i < c->n && taken; ++i) {
                const double v = sols[(size_t)i];
                if (!std::isfinite(v)) taken = false;
                else if (v != 0.0 && (codes[(size_t)i] == 0 || (codes[(size_t)i] == 1 && v < -1e-9 * smax) || (codes[(size_t)i] == -1 && v > 1e-9 * smax))) taken = false;
            }
            if (taken) { c->tab_valid = false; unconv = 0; }
            if (c->knobs.finish_trace) fprintf(stderr, "[finish] the sweep's solution of its winner (workgroup %d): %s\n", export_wg, taken ? "taken" : "refused (signs), solving again");
        }
        if (!taken) {
            opt_codes(c, cands[ci], codes);
            st = solve_nodes(c, codes, 1, sols, obj2, &unconv, false, /*want_tab=*/true);
            if (st != PARTLS_OK) return st;
        }
        unscale_solution(c, sols.data(), w);
        RefineOut ro;
        st = refine_solution(c, w, !c->faithful, 2, &ro); // QR-level accuracy of the winner on ill-conditioned data
        if (st != PARTLS_OK) return st;
        double o = 0.0;
        // Opt.jl:90 from the data, and Xo'(yo - Xo w) for the KKT check below: left by the refinement's last pass when it converged
        if (ro.have) { o = ro.obj; g.swap(ro.g); }
        else { st = data_objective(c, w, &o, &g); if (st != PARTLS_OK) return st; }
        if (c->knobs.finish_trace && cands.size() > 1) fprintf(stderr, "[finish] near tie: pattern %llu data objective %.17g\n", (unsigned long long)cands[ci], o);
        // argmin over the data objectives, first reference index on exact ties (Opt.jl:96)
        if (ci == 0 || o < obest || (o == obest && cands[ci] < pbest)) { obest = o; pbest = cands[ci]; wbest = w; gbest = g; unconv_best = unconv; loo_best = c->last_min_loo; }
    }
    const auto f1 = std::chrono::steady_clock::now();
    uint64_t full = pbest;
    if (!c->faithful) { if (wbest[(size_t)c->M] > 0.0) full |= (1ULL << c->K); }     // first-index tie-break when t == 0
    full &= used;
    // data-space KKT conditions of the winner, every variable — including those the leave-one-out rule kept out of the basis
    std::vector<int8_t> vcode((size_t)c->M + 1, 0);
    for (int64_t m = 0; m <= c->M; ++m) {
        if (m == c->M && !c->faithful) { vcode[(size_t)m] = 2; continue; }          // free intercept
        const int f = 2 * __builtin_popcountll(c->mask_aug[(size_t)m] & full) - __builtin_popcountll(c->mask_aug[(size_t)m]);
        vcode[(size_t)m] = (int8_t)((f > 0) - (f < 0));
    }
    int worst = -1;
    c->last_kkt = kkt_violation_data(c, wbest, gbest, vcode, &worst);
    *opt = obest;
    if (c->knobs.finish_trace) {
        const auto f2 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[finish] %zu candidate(s): solve + refine + data objective / gradient %.3f ms, KKT check %.3f ms; data-space KKT violation %.3e (variable %d)\n",
                cands.size(), ms(f0, f1), ms(f1, f2), c->last_kkt, worst);
    }
    t_end(c, PARTLS_T_FINISH);
    PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));
    t_collect(c);
    cleanup_opt(c, wbest, full, alpha, beta, t);
    if (best_index) *best_index = (int64_t)full;
    if (unconv_best) { set_error("winner re-solve hit the pivot cap"); return PARTLS_ERR_NOT_CONVERGED; }
    c->last_min_loo = loo_best;
    if (kkt_says_ill_conditioned(c)) {
        set_error("the winner's KKT conditions do not hold in data space (violation %.2e of ||x|| ||y|| at variable %d, tolerance %.1e; %llu columns "
                  "refused as dependent in the sweep): X is too ill-conditioned for the fp64 Gram form (cond(X) >~ 1e6); the outputs hold the best "
                  "Gram-form model", c->last_kkt, worst, c->knobs.kkt_tol, c->sweep_vetoes);
        return PARTLS_ERR_ILL_CONDITIONED;
    }
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_opt_pattern(partls_ctx *c, int64_t pattern, double *raw_alpha, double *optval)
try {
    if (!c || !c->prepared) { set_error("partls_opt_pattern: context not prepared"); return PARTLS_ERR_STATE; }
    if (!c->faithful) { set_error("partls_opt_pattern needs a context prepared with PARTLS_OPT_FAITHFUL_INTERCEPT"); return PARTLS_ERR_STATE; }
    if (pattern < 0 || pattern >= ((int64_t)1 << c->kbits)) { set_error("pattern out of range"); return PARTLS_ERR_BAD_ARG; }
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
    std::vector<double> sols, obj2, w;
    unsigned long long unconv = 0;
    std::vector<int8_t> codes;
    opt_codes(c, (uint64_t)pattern, codes);
    partls_status st = solve_nodes(c, codes, 1, sols, obj2, &unconv, false, /*want_tab=*/true);
    if (st != PARTLS_OK) return st;
    unscale_solution(c, sols.data(), w);
    st = refine_solution(c, w, false);
    if (st != PARTLS_OK) return st;
    if (optval) { st = data_objective(c, w, optval); if (st != PARTLS_OK) return st; }
    if (raw_alpha)
        for (int64_t m = 0; m <= c->M; ++m) {
            const int f = 2 * __builtin_popcountll(c->mask_aug[(size_t)m] & (uint64_t)pattern) - __builtin_popcountll(c->mask_aug[(size_t)m]);
            const double a = (f != 0) ? w[(size_t)m] / (double)f : 0.0;
            raw_alpha[m] = a > 0.0 ? a : 0.0;
        }
    if (unconv) { set_error("pattern solve hit the pivot cap"); return PARTLS_ERR_NOT_CONVERGED; }
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_fit_opt(partls_ctx *c, const double *X, int64_t N, int64_t M, int64_t ldX, const double *y,
                             const int64_t *P, int64_t K, int64_t ldP, double eta, uint32_t flags,
                             double *alpha, double *beta, double *t, double *opt, int64_t *best_index, double *all_opt)
try {
    if (all_opt) flags |= PARTLS_OPT_FAITHFUL_INTERCEPT;
    partls_status st = partls_opt_prepare(c, X, N, M, ldX, y, 0, P, K, ldP, eta, flags);
    if (st != PARTLS_OK) return st;
    double bobj; int64_t bpat, unconv;
    st = partls_opt_sweep(c, 0, -1, &bobj, &bpat, all_opt, &unconv);
    if (st != PARTLS_OK) return st;
    if (bpat < 0) { set_error("sweep produced no candidate"); return PARTLS_ERR_NOT_CONVERGED; }
    st = partls_opt_finish(c, bpat, alpha, beta, t, opt, best_index);
    if (st != PARTLS_OK) return st;
    if (unconv) { set_error("%lld subproblems hit the pivot cap", (long long)unconv); return PARTLS_ERR_NOT_CONVERGED; }
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

// predict: w_m = sum_k P[m,k] alpha_m beta_k on the host (M*K flops), yhat = X w + t on the device (one pass over X)
static partls_status predict_common(partls_ctx *c, const double *X, int64_t N, int64_t M, int64_t ldX, int x_on_device,
                                    const int64_t *P, int64_t K, int64_t ldP, const double *alpha, const double *beta, double t,
                                    double *yhat)
{
    partls_status st = check_common(c, X, N, M, ldX, P, K, ldP);
    if (st != PARTLS_OK) return st;
    if (!alpha || !beta || !yhat) { set_error("partls_predict: NULL argument"); return PARTLS_ERR_BAD_ARG; }
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
    std::vector<double> w((size_t)M, 0.0);
    for (int64_t m = 0; m < M; ++m) {
        double s = 0.0;
        for (int64_t k = 0; k < K; ++k) {
            const int64_t v = P[m + k * ldP];
            if (v != 0 && v != 1) { set_error("P has an entry outside {0,1}"); return PARTLS_ERR_BAD_PARTITION; }
            s += (double)v * alpha[m] * beta[k];
        }
        w[(size_t)m] = s;
    }
    PARTLS_HIP_CHECK(c->wdev.ensure((size_t)(M + 1) * sizeof(double)));
    PARTLS_HIP_CHECK(hipMemcpyAsync(c->wdev.p, w.data(), (size_t)M * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const double *dX = X;
    int64_t ld = ldX;
    double *dyh = yhat;
    if (!x_on_device) {
        PARTLS_HIP_CHECK(c->predX.ensure((size_t)N * M * sizeof(double)));
        PARTLS_HIP_CHECK(c->predY.ensure((size_t)N * sizeof(double)));
        const partls_status us = upload_matrix(c, c->predX.as<double>(), X, N, M, ldX);
        if (us != PARTLS_OK) return us;
        dX = c->predX.as<double>(); ld = N; dyh = c->predY.as<double>();
    }
    PARTLS_HIP_CHECK(launch_residual(dX, N, M, ld, nullptr, c->wdev.as<double>(), t, nullptr, 1024, dyh, c->stream));
    if (!x_on_device)
        PARTLS_HIP_CHECK(hipMemcpyAsync(yhat, c->predY.p, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (!x_on_device) { c->predX.release(); c->predY.release(); }    // a prediction set is not retained
    return PARTLS_OK;
}

partls_status partls_predict(partls_ctx *c, const double *X, int64_t N, int64_t M, int64_t ldX, const int64_t *P, int64_t K,
                             int64_t ldP, const double *alpha, const double *beta, double t, double *yhat)
try {
    return predict_common(c, X, N, M, ldX, 0, P, K, ldP, alpha, beta, t, yhat);
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_predict_device(partls_ctx *c, const double *dX, int64_t N, int64_t M, int64_t ldX, const int64_t *P,
                                    int64_t K, int64_t ldP, const double *alpha, const double *beta, double t, double *dyhat)
try {
    return predict_common(c, dX, N, M, ldX, 1, P, K, ldP, alpha, beta, t, dyhat);
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_synth_truth(uint64_t seed, int64_t D, int64_t K, int64_t *P, double *wstar)
try {
    if (D < 1 || K < 1 || K > D) { set_error("partls_synth_truth: bad D/K"); return PARTLS_ERR_BAD_ARG; }
    std::vector<int64_t> grp((size_t)D);
    int64_t j = 0;
    for (int64_t k = 0; k < K; ++k) {
        const int64_t sz = D / K + ((k < D % K) ? 1 : 0);
        for (int64_t q = 0; q < sz; ++q) grp[(size_t)j++] = k;
    }
    if (P) {
        memset(P, 0, (size_t)D * K * sizeof(int64_t));
        for (int64_t m = 0; m < D; ++m) P[m + grp[(size_t)m] * D] = 1;
    }
    auto uni = [&](uint64_t stream, uint64_t idx) { return (double)(rnd64(seed, stream, idx) >> 11) * 0x1.0p-53; };
    if (wstar)
        for (int64_t k = 0; k < K; ++k) {
            double sum = 0.0;
            for (int64_t m = 0; m < D; ++m) if (grp[(size_t)m] == k) sum += uni(2, (uint64_t)m);
            const double bk = (uni(3, (uint64_t)k) - 0.5) * 10.0;
            for (int64_t m = 0; m < D; ++m) if (grp[(size_t)m] == k) wstar[m] = (uni(2, (uint64_t)m) / sum) * bk;
        }
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_synth_device(partls_ctx *c, uint64_t seed, int64_t N, int64_t D, const double *wstar, double *dX, double *dy)
try {
    if (!c || !wstar || !dX || !dy || N < 1 || D < 1) { set_error("partls_synth_device: bad argument"); return PARTLS_ERR_BAD_ARG; }
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
    PARTLS_HIP_CHECK(c->wdev.ensure((size_t)D * sizeof(double)));
    PARTLS_HIP_CHECK(hipMemcpyAsync(c->wdev.p, wstar, (size_t)D * sizeof(double), hipMemcpyHostToDevice, c->stream));
    PARTLS_HIP_CHECK(launch_synth(seed, N, D, c->wdev.as<double>(), dX, dy, c->stream));
    PARTLS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_get_timing(const partls_ctx *c, partls_timer which, double *ms)
try {
    if (!c || !ms || (int)which < 0 || (int)which >= PARTLS_T_COUNT) { set_error("partls_get_timing: bad argument"); return PARTLS_ERR_BAD_ARG; }
    *ms = c->ms[(int)which];
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_get_upload(const partls_ctx *c, double *ms, double *bytes)
try {
    if (!c || !ms || !bytes) { set_error("partls_get_upload: bad argument"); return PARTLS_ERR_BAD_ARG; }
    *ms = c->last_upload_ms; *bytes = c->last_upload_bytes;
    return PARTLS_OK;
}
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_get_pivots(const partls_ctx *c, int64_t *pivots)
try {
    if (!c || !pivots) { set_error("partls_get_pivots: bad argument"); return PARTLS_ERR_BAD_ARG; }
    *pivots = (int64_t)c->last_pivots;
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_get_kkt_violation(const partls_ctx *c, double *violation, double *min_pivot)
try {
    if (!c || !violation) { set_error("partls_get_kkt_violation: bad argument"); return PARTLS_ERR_BAD_ARG; }
    *violation = c->last_kkt;
    if (min_pivot) *min_pivot = c->last_min_loo;
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_get_vetoes(const partls_ctx *c, int64_t *vetoes)
try {
    if (!c || !vetoes) { set_error("partls_get_vetoes: bad argument"); return PARTLS_ERR_BAD_ARG; }
    *vetoes = (int64_t)c->last_vetoes;
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_opt_bit_order(partls_ctx *c, int64_t *gbit, double *flip_cost)
try {
    if (!c || !c->prepared) { set_error("partls_opt_bit_order: context not prepared"); return PARTLS_ERR_STATE; }
    if (!gbit) { set_error("partls_opt_bit_order: gbit is NULL"); return PARTLS_ERR_BAD_ARG; }
    if (!opt_range_ok(c, "partls_opt_bit_order")) return PARTLS_ERR_UNSUPPORTED;
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
    if (!c->order_ready) {
        partls_status st = calibrate_bit_order(c);
        if (st != PARTLS_OK) return st;
    }
    for (int k = 0; k < c->kbits; ++k) {
        gbit[k] = c->order.gbit[k];
        if (flip_cost) flip_cost[k] = c->flip_cost.empty() ? -1.0 : c->flip_cost[(size_t)k];
    }
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

// ---- near ties across shards -------------------------------------------------------------------------------------------------------
// A sharded enumeration (partls_fit_opt_multi's rank threads, dist.py's processes) must re-rank the SAME candidate set a single context
// would (Opt.jl:90,96: every objective from the data, first index on ties): each shard hands out its winner and its near ties with
// their tracked objectives, the lists of all shards are concatenated in rank order, and every rank installs the merged set.
partls_status partls_opt_candidates(const partls_ctx *c, int64_t capacity, double *obj, int64_t *pattern, int64_t *count)
try {
    if (!c || !c->prepared || !count || capacity < 0 || (capacity > 0 && (!obj || !pattern))) { set_error("partls_opt_candidates: bad argument"); return PARTLS_ERR_BAD_ARG; }
    const int64_t n = std::min<int64_t>(capacity, (int64_t)c->cand.size());
    for (int64_t i = 0; i < n; ++i) { obj[i] = c->cand[(size_t)i].first; pattern[i] = c->cand[(size_t)i].second; }
    *count = n;
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_opt_merge_candidates(partls_ctx *c, int64_t count, const double *obj, const int64_t *pattern, double *win_obj, int64_t *win_pattern)
try {
    if (!c || !c->prepared || count < 0 || (count > 0 && (!obj || !pattern))) { set_error("partls_opt_merge_candidates: bad argument"); return PARTLS_ERR_BAD_ARG; }
    std::vector<std::pair<double, int64_t>> all;
    for (int64_t i = 0; i < count; ++i) if (pattern[i] >= 0 && obj[i] == obj[i]) all.emplace_back(obj[i], pattern[i]);
    std::sort(all.begin(), all.end());                        // lexicographic (objective, reference index): argmin's first-index rule
    all.erase(std::unique(all.begin(), all.end()), all.end());
    const int64_t local_winner = c->cand.empty() ? -1 : c->cand[0].second;
    c->near_pat.clear();
    c->cand.clear();
    if (all.empty()) { c->near_for = -1; c->export_wg = -1; if (win_obj) *win_obj = INFINITY; if (win_pattern) *win_pattern = -1; return PARTLS_OK; }
    const double bobj = all[0].first;
    const int64_t bpat = all[0].second;
    const double yy = h_reg(c, (int)c->M + 1, (int)c->M + 1);
    const double lim2 = bobj * bobj + c->knobs.near_tie_rel * (yy > 0.0 ? yy : 0.0);
    c->cand.push_back(all[0]);
    for (size_t i = 1; i < all.size() && c->near_pat.size() < 3; ++i)
        if (all[i].second != bpat && all[i].first * all[i].first <= lim2) { c->near_pat.push_back(all[i].second); c->cand.push_back(all[i]); }
    c->near_for = bpat;
    if (local_winner != bpat) c->export_wg = -1;             // the solution this rank's sweep left behind belongs to another pattern
    if (win_obj) *win_obj = bobj;
    if (win_pattern) *win_pattern = bpat;
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_get_near_ties(const partls_ctx *c, int64_t *evaluated)
try {
    if (!c || !evaluated) { set_error("partls_get_near_ties: bad argument"); return PARTLS_ERR_BAD_ARG; }
    *evaluated = c->last_near_evaluated;
    return PARTLS_OK;
}
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

partls_status partls_get_gram(const partls_ctx *c, double *G_aug)
try {
    if (!c || !c->prepared || !G_aug) { set_error("partls_get_gram: context not prepared"); return PARTLS_ERR_STATE; }
    const int na = (int)c->M + 2;
    for (int j = 0; j < na; ++j)
        for (int i = 0; i < na; ++i) G_aug[(size_t)i + (size_t)j * na] = h_reg(c, i, j);
    return PARTLS_OK;
}
catch (const std::bad_alloc &) { partls::set_error("out of host memory"); return PARTLS_ERR_BAD_ARG; }
catch (...) { partls::set_error("internal error: an exception reached the C ABI"); return PARTLS_ERR_BAD_ARG; }

}  // extern "C"
