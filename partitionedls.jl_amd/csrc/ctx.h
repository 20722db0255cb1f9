// ctx.h — the context behind the opaque partls_ctx handle and the host helpers shared by api.hip and solvers.hip.
#pragma once
#include "common.h"
#include <functional>
#include <vector>

namespace partls {

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t b)
    {
        if (b <= bytes && p) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; bytes = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipMalloc(&p, b ? b : 8);
        if (e == hipSuccess) bytes = b;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// page-locked host array of doubles (device -> host copies at full PCIe rate instead of through a staging buffer)
struct PinnedDoubles {
    double *p = nullptr;
    size_t cap = 0, n = 0, cap_prev = 0;
    hipError_t resize(size_t count)
    {
        if (count > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr; cap = 0;
            // (page-locking costs ~0.4 ms a call: grow geometrically, at least a page — the node batches of a BnB search double from round to round)
            size_t want = count > 512 ? count : 512;
            if (want < 2 * cap_prev) want = 2 * cap_prev;
            hipError_t e = hipHostMalloc((void **)&p, want * sizeof(double), hipHostMallocDefault);
            if (e != hipSuccess) return e;
            cap = cap_prev = want;
        }
        n = count;
        return hipSuccess;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = n = 0; }
    double *data() const { return p; }
    size_t size() const { return n; }
    double &operator[](size_t i) const { return p[i]; }
};

}  // namespace partls

// Tuning / diagnostic knobs, read from the environment ONCE at partls_create (never on the per-call path).
struct partls_knobs {
    double tol_rel = 1e-11;      // PARTLS_TOL_REL: KKT tolerance relative to ||y||
    long long chain_len = 0;     // PARTLS_CHAIN_LEN: patterns per Gray chain (0 = automatic)
    long long grid = 0;          // PARTLS_GRID: workgroups of the sweep (0 = automatic)
    int gram_S = 0, gram_cr = 0; // PARTLS_GRAM_S / PARTLS_GRAM_CR: Gram work decomposition overrides
    int coop_rows = 0;           // PARTLS_COOP_ROWS: tableau rows per workgroup of the cooperative kernel (0 = automatic)
    bool no_coop = false;        // PARTLS_NO_COOP: single large solves on the one-workgroup kernel
    int reg_maxt = 18;           // PARTLS_REG_MAXT (1..20): tile columns up to which the register kernel runs (n <= 16 x this).  Its T = 19 / 20
                                 // instantiations spill 120-600 VGPRs and lose to the deferred-update kernel since round 4 (65 536 patterns: D = 304
                                 // 5.5 against 6.3 M solves/s, D = 320 2.3 against 6.1 M; BnB nodes 1.03 / 1.01 against 1.12 / 1.01 M per second)
    bool no_staged_upload = false; // PARTLS_NO_STAGED_UPLOAD: host X goes up with one pageable hipMemcpy2DAsync, as through round 3 (A/B tests)
    bool no_export = false;      // PARTLS_NO_EXPORT: the winner is always solved again from the empty basis (A/B tests)
    bool eager_generic = false;  // PARTLS_EAGER_GENERIC: n > 320 on sweep_generic.hip (every block applied to the whole tableau) instead of sweep_lazy.hip (A/B tests)
    int bnb_batch = 1024;        // PARTLS_BNB_BATCH: nodes bounded per device batch of the BnB search
    int bnb_pool_mb = 65536;     // PARTLS_BNB_POOL_MB: cap of the tableau-snapshot pool of the BnB search (warm-started node bounds); chunks are allocated on
                                 // demand, never beyond half of the free HBM.  16 GB (round 3) ran out in the 173 717-node search of bench.py's bnb_hard: the
                                 // late rounds then bounded a sixth of their nodes from the fresh tableau (47 pivots per node instead of 5.7) and took 3x longer
    int bnb_wg_per_cu = 8;       // PARTLS_BNB_WG_PER_CU: workgroups per resident slot in a node batch's grid (1: persistent; 8: one node per workgroup as in round 3)
    bool bnb_cold = false;       // PARTLS_BNB_COLD: every BnB node from the fresh tableau (A/B tests)
    bool lz_fault = false;       // PARTLS_LZ_FAULT (tests): the deferred-update kernel's first panel of workgroup 0 loses its progress word — the followers' bounded
                                 // wait must end it and the sweep must report PARTLS_ERR_NOT_CONVERGED instead of a result
    int coop_fault = 0;          // PARTLS_COOP_FAULT (tests): make the cooperative kernel's grid barrier time out (see SweepParams)
    bool no_tab_refine = false;  // PARTLS_NO_TAB_REFINE: refinement by host Cholesky even when the node solve left its tableau (A/B tests)
    bool finish_trace = false;   // PARTLS_FINISH_TRACE
    bool alt_trace = false;      // PARTLS_ALT_TRACE
    bool alt_always_check = false; // PARTLS_ALT_ALWAYS_CHECK: fit(Alt) verifies its last iteration against the data even when Gershgorin certifies the Gram form (tests)
    bool print_stamps = false;   // PARTLS_PRINT_STAMPS (diagnostic build only)
    double kkt_tol = 1e-12;      // PARTLS_KKT_TOL: data-space KKT violation of the winner (units of ||x_m|| ||y||) above which fit(Opt) / fit(BnB) report
                                 // PARTLS_ERR_ILL_CONDITIONED instead of PARTLS_OK (see kkt_says_ill_conditioned, api.hip)
    double near_tie_rel = 1e-13; // PARTLS_NEAR_TIE_REL (tests): width of the near-tie window of the sweep, in units of y'y on the objective^2
    double cal_wb = 1.0, cal_ws = 1.0;  // PARTLS_CAL_WB / PARTLS_CAL_WS: multipliers of the block / scan weights of the bit-order cost model (experiments)
    int bit_order = 0;           // PARTLS_BIT_ORDER: 0 automatic (calibrate when the sweep is long enough to repay it), "identity" = 1
                                 // (group k on Gray bit k), "calibrate" = 2 (always measure; small problems in the tests)
};

struct partls_ctx {
    int device = 0;
    partls_knobs knobs;
    hipStream_t stream = nullptr;
    hipEvent_t ev0[PARTLS_T_COUNT] = {}, ev1[PARTLS_T_COUNT] = {};
    bool timed[PARTLS_T_COUNT] = {};
    double ms[PARTLS_T_COUNT] = {};

    // problem
    bool prepared = false;
    int64_t N = 0, M = 0, K = 0, ldX = 0;
    double eta = 0.0;
    uint32_t flags = 0;
    bool faithful = false;                         // intercept is a sign-constrained tableau variable (2^(K+1) patterns)
    const double *dX = nullptr, *dy = nullptr;     // device views (owned copies below, or the caller's)
    partls::DevBuf ownX, ownY;
    std::vector<int64_t> P;                        // M x K compact
    std::vector<uint64_t> mask_aug;                // M + 2: features, intercept (bit K), y (0)
    std::vector<uint64_t> pack;                    // staging of the one upload of masks and permutation
    const uint64_t *maskTabP = nullptr;            // views into maskAugD: group masks in tableau order, permutation
    const int *permP = nullptr;
    // gram
    int ldg = 0, chunks = 0;
    partls::DevBuf slab, G, maskAugD /* + maskTabP, permP: one upload */, scale, Tfull, T0reg, scratch, bestObj, bestSol /* [workgroups][n]: solution of every workgroup's best pattern (register kernels) */, bestPat, counters, allOpt,
        wdev, partial, yhatD, gD, nodeCode, nodeSol, nodeObj, predX, predY, gridCtr, nodeTab, nodeBasic, altA, altGA, altHg,
        nodePiv, maskInt, allOptRef, bnbIn, bnbOut, altGersh;
    // BnB: tableau snapshots of open nodes (solvers.hip: SnapshotPool), kept across fits; host staging of a node batch
    std::vector<void *> bnbChunks;
    size_t bnbSlotBytes = 0, bnbMaxSlots = 0;
    int bnbChunkSlots = 512;
    std::vector<int> bnbFree, bnbRefs;             // free slots; reference counts of the in-library search (the ABI's host keeps its own)
    // rows of X sharded over several devices (partls_fit_opt_multi): the contexts that hold the OTHER row blocks of the problem this
    // context is prepared for; every pass over the data (data_pass, api.hip) then covers them too.  Cleared by every prepare.
    std::vector<partls_ctx *> peers;
    partls::PinnedDoubles hPart, hGpart;           // host staging of a data pass (page-locked like every device -> host destination of a fit: a pageable
                                                   // one costs 17-30 us of runtime staging per copy and blocks the caller — a third of a C2-sized fit's host time)
    partls::PinnedDoubles sweepOut, nodeOut, exportSol;   // ... of a sweep's per-workgroup results, of a node batch, of the sweep's solution of its winner
    // called between the Gram build and the tableau preparation (partls_fit_opt_multi: the Gram products of the row blocks are summed)
    std::function<partls_status(partls_ctx *)> gram_hook;
    partls::PinnedDoubles bnbHostIn, bnbHostOut;   // page-locked staging of a node batch (8-byte words): the two copies of a round cost ~10 us each instead of ~25 pageable
    // staged upload of a host X (api.hip: upload_matrix): 4 copier threads x 2 page-locked buffers, one stream each; wall time and bytes of
    // the last one (0 when the inputs were device-resident)
    char *upPin[8] = {};
    hipStream_t upStream[4] = {};
    hipEvent_t upEvent[8] = {};
    double last_upload_ms = 0.0, last_upload_bytes = 0.0;
    partls::PinnedDoubles hG;                      // host copy of the augmented Gram (pinned: 0.8 MB per prepare at C3)
    partls::PinnedDoubles hScale;
    // tableau: variable i of the tableau is augmented-Gram index perm[i] (features grouped by partition)
    int n = 0, kbits = 0, T = 0;
    std::vector<int> perm;
    std::vector<uint64_t> mask_tab;
    bool use_reg = false;
    // Opt sweep: which group sits on which bit of the Gray index (calibrate_bit_order, once per prepare, on the first sweep).
    // The boundary speaks the reference's pattern index (group k = bit k, Opt.jl:4-12) everywhere; only the sweep kernel and the
    // Gray-index ranges of partls_opt_sweep live in the internal order.
    partls::BitOrder order{};
    bool order_ready = false, order_identity = true;
    std::vector<double> flip_cost;                 // measured pivots per flip of group k (empty when not calibrated)
    double tol = 0.0;
    unsigned long long last_pivots = 0, last_vetoes = 0, last_blocks = 0;
    // near ties of the last sweep (reference pattern indices whose tracked objective^2 lies within the Gram form's own error of the
    // winner's): partls_opt_finish re-ranks them by the objective computed from the data before it fixes the winner
    std::vector<int64_t> near_pat;
    int64_t near_for = -1;
    // the same with the tracked objectives, winner first: what the ranks of a sharded enumeration exchange (partls_opt_candidates /
    // partls_opt_merge_candidates) so that every rank re-ranks the set a single context would
    std::vector<std::pair<double, int64_t>> cand;
    int64_t last_near_evaluated = 0;               // distinct subproblems the last partls_opt_finish solved (1: no near tie)
    int export_wg = -1;                            // row of bestSol with the solution of the last sweep's winner (-1: none)
    double last_kkt = 0.0;                         // data-space KKT violation of the last finished winner
    double last_min_loo = 0.0;                     // smallest leave-one-out pivot of the basis of the last refined node solve (0: unknown)
    unsigned long long sweep_vetoes = 0;           // leave-one-out refusals of the last sweep (node solves overwrite last_vetoes)
    bool coop_state_valid = false;                 // scratch holds the tableau/basis of the previous cooperative solve
    bool coop_fallback = false;                    // the cooperative attempt of the current solve timed out at its grid barrier
    // final tableau of the last single-node solve on the register kernel (pinned host copies; see solve_nodes `want_tab`)
    double *hTab = nullptr;
    int8_t *hBasic = nullptr;
    size_t hTabDoubles = 0;
    bool tab_valid = false, tab_full = false;
};

namespace partls {

void t_begin(partls_ctx *c, int w);
void t_end(partls_ctx *c, int w);
void t_collect(partls_ctx *c);

// Upload (or adopt) X, y; build the Gram products; lay the tableau out.  faithful = intercept is a regular variable.
partls_status ctx_prepare(partls_ctx *c, const double *X, int64_t N, int64_t M, int64_t ldX, const double *y, int x_on_device,
                          const int64_t *P, int64_t K, int64_t ldP, double eta, bool faithful, uint32_t flags);

// Solve a batch of `cnt` independent subproblems ("nodes") from the fresh tableau.  codes[i * n + v] is the constraint on
// tableau variable v in node i: +1 (w >= 0), -1 (w <= 0), 0 (w = 0), 2 (free) — see SweepParams::node_code.  sols: cnt x n
// scaled solutions in tableau order (0 for nonbasic); obj2: objective^2 from the tableau corner.
partls_status solve_nodes(partls_ctx *c, const std::vector<int8_t> &codes, size_t cnt, std::vector<double> &sols,
                          std::vector<double> &obj2, unsigned long long *unconv, bool resume = false, bool want_tab = false);
// node codes of one Opt sign pattern (Opt.jl:28-29): sign of the multiplier sum_k P[m,k] s_k of every tableau variable
void opt_codes(const partls_ctx *c, uint64_t pattern, std::vector<int8_t> &codes);

// scaled tableau solution -> w over [features, intercept] (length M+1); a free intercept is recovered from the Gram copy
void unscale_solution(const partls_ctx *c, const double *sol, std::vector<double> &w);
// ||Xo w - yo||_2 from the data (+ the eta rows): Opt.jl:90
partls_status data_pass(partls_ctx *c, const std::vector<double> &w, bool want_obj, bool want_grad, double *obj2, std::vector<double> *g,
                        const std::function<void()> &overlap);
// grad (optional): Xo'(yo - Xo w) over [features, intercept], from the data (one more pass over X)
partls_status data_objective(partls_ctx *c, const std::vector<double> &w, double *opt, std::vector<double> *grad = nullptr);
bool kkt_says_ill_conditioned(const partls_ctx *c);
double kkt_violation_data(const partls_ctx *c, const std::vector<double> &w, const std::vector<double> &g, const std::vector<int8_t> &code,
                          int *worst);
// Iterative refinement of a solution w (over [features, intercept]) on its own support, in data space: residual and
// X'r on the device; the small SPD solve uses the inverse the pivoting left in the final tableau of the node solve (want_tab;
// register kernel) or, without one, a Cholesky factorisation of the host Gram copy.  Brings a Gram-based solution (error ~ cond^2 eps) to the
// accuracy of a QR-based one (the reference's NNLS) as long as cond^2 eps < 1.  `free_intercept`: the intercept is part
// of the support even when w[M] == 0.
// `out` (optional): objective ||Xo w - yo|| and gradient Xo'(yo - Xo w) at the returned w, when the refinement converged (have)
struct RefineOut { bool have = false; double obj = 0.0; std::vector<double> g; };
partls_status refine_solution(partls_ctx *c, std::vector<double> &w, bool free_intercept, int steps = 2, RefineOut *out = nullptr);
// regularised augmented Gram entry on the host copy
double h_reg(const partls_ctx *c, int a, int b);
partls_status check_common(partls_ctx *c, const void *X, int64_t N, int64_t M, int64_t ldX, const void *P, int64_t K, int64_t ldP);

}  // namespace partls
