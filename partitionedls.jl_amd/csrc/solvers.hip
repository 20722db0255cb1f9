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
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
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

}  // namespace

extern "C" {

partls_status partls_fit_alt(partls_ctx *c, const double *X, int64_t N, int64_t M, int64_t ldX, const double *y,
                             const int64_t *P, int64_t K, int64_t ldP, double eta, double eps, int64_t T,
                             const double *alpha0, const double *beta0,
                             double *alpha, double *beta, double *t, double *opt, int64_t *iters)
{
    partls_status st = ctx_prepare(c, X, N, M, ldX, y, 0, P, K, ldP, eta, /*faithful=*/true, 0);
    if (st != PARTLS_OK) return st;
    return partls_alt_prepared(c, eps, T, alpha0, beta0, alpha, beta, t, opt, iters);
}

partls_status partls_alt_prepared(partls_ctx *c, double eps, int64_t T, const double *alpha0, const double *beta0,
                                  double *alpha, double *beta, double *t, double *opt, int64_t *iters)
{
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
        std::vector<double> wv;
        unscale_solution(c, sols.data(), wv);
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
        oldopt = optval;
        optval = std::sqrt(o2 > 0.0 ? o2 : 0.0);
        if (c->knobs.alt_trace) fprintf(stderr, "[alt] iter %d: alpha-step %.3f ms (%llu pivots, %llu blocks), rest %.3f ms\n", (int)i, std::chrono::duration<double, std::milli>(tt1 - tt0).count(), apiv, ablk, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt1).count());
        ++i;
    }
    w_from();
    // Final objective.  The loss of the last iteration, beta'H beta - 2 g'beta + y'y from the K' x K' system, carries an absolute error
    // of ~eps * y'y (cancellation against y'y): relative to opt^2 that is below 1e-10 as long as opt^2 > 1e-6 * y'y, and then it IS
    // the result; only a near-interpolating fit (the reference's toy: opt = 0) pays the extra pass over X (0.87 ms of 4.1 GB at C4).
    double dopt = optval;
    if (!(optval * optval > 1e-6 * h_reg(c, Y, Y))) {
        st = data_objective(c, w, &dopt);                // from the data: no Gram cancellation
        if (st != PARTLS_OK) return st;
    }
    for (int64_t m = 0; m < M; ++m) alpha[m] = a[(size_t)m];
    for (int64_t k = 0; k < K; ++k) beta[k] = b[(size_t)k];
    *t = b[(size_t)K] * a[(size_t)M];                    // Alt.jl:119
    *opt = dopt;
    if (iters) *iters = i - 1;
    if (unconv_total) { set_error("partls_fit_alt: an alpha-step hit the pivot cap"); return PARTLS_ERR_NOT_CONVERGED; }
    return PARTLS_OK;
}

partls_status partls_fit_bnb(partls_ctx *c, const double *X, int64_t N, int64_t M, int64_t ldX, const double *y,
                             const int64_t *P, int64_t K, int64_t ldP, double eta,
                             double *alpha, double *beta, double *t, double *opt, int64_t *nopen)
{
    partls_status st = ctx_prepare(c, X, N, M, ldX, y, 0, P, K, ldP, eta, /*faithful=*/true, 0);
    if (st != PARTLS_OK) return st;
    return partls_bnb_prepared(c, alpha, beta, t, opt, nopen);
}

// ---- BnB primitives (shared by the single-rank driver below and the rank-sharded search of partitionedls.jl_amd/dist.py) -------
// A node is (pat, free): group k is branched iff bit k of `free` is clear, and then constrained to alpha_pk >= 0 (bit k of pat
// set) or <= 0 (BnB.jl:120-121).  partls_bnb_bound evaluates BnB.jl:99-118 for a batch: lb[i] = the node's lower bound
// (BnB.jl:69-92 on the device) and branch[i] = argmax_k nu_k (BnB.jl:107,117), or -1 when all nu_k == 0, i.e. the relaxed
// solution is feasible for the original problem and lb[i] is its value (BnB.jl:109-115).
partls_status partls_bnb_bound(partls_ctx *c, int64_t count, const uint64_t *pat, const uint64_t *free_, double *lb, int32_t *branch)
{
    if (!c || !c->prepared || !c->faithful) { set_error("partls_bnb_bound: needs a context prepared with PARTLS_OPT_FAITHFUL_INTERCEPT"); return PARTLS_ERR_STATE; }
    if (count < 0 || (count > 0 && (!pat || !free_ || !lb || !branch))) { set_error("partls_bnb_bound: bad argument"); return PARTLS_ERR_BAD_ARG; }
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
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

// The model of one (feasible) node: re-solve, data-space refinement, BnB.jl:36-39 normalisation, objective from the data.
partls_status partls_bnb_leaf(partls_ctx *c, uint64_t pat, uint64_t free_, double *alpha, double *beta, double *t, double *opt)
{
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

// fit_BnB (BnB.jl:94-132) as a best-first search: frontier ordered by the parent's bound, nodes bounded in device batches of 512,
// the incumbent prunes (BnB.jl:102).  Same optimum as the reference's depth-first recursion; the node count is not.
partls_status partls_bnb_prepared(partls_ctx *c, double *alpha, double *beta, double *t, double *opt, int64_t *nopen)
{
    if (!c || !c->prepared || !c->faithful) { set_error("partls_bnb_prepared: needs a context prepared with PARTLS_OPT_FAITHFUL_INTERCEPT"); return PARTLS_ERR_STATE; }
    if (!alpha || !beta || !t || !opt) { set_error("partls_fit_bnb: NULL argument"); return PARTLS_ERR_BAD_ARG; }
    const int Kp = (int)c->K + 1;
    struct Node { double key; uint64_t pat, free_; unsigned long long seq; };
    struct Cmp { bool operator()(const Node &a, const Node &b) const { return a.key > b.key || (a.key == b.key && a.seq > b.seq); } };
    std::priority_queue<Node, std::vector<Node>, Cmp> frontier;
    unsigned long long seq = 0;
    frontier.push({0.0, 0ULL, ((uint64_t)1 << Kp) - 1, seq++});            // root: everything free (Σ = [], BnB.jl:33)
    double mu = INFINITY;
    uint64_t best_pat = 0, best_free = 0;
    bool have = false;
    int64_t bounded = 0;
    const size_t BATCH = 512;
    std::vector<uint64_t> bp, bf;
    std::vector<double> lb;
    std::vector<int32_t> br;
    while (!frontier.empty()) {
        bp.clear(); bf.clear();
        while (!frontier.empty() && bp.size() < BATCH) {
            const Node nd = frontier.top();
            frontier.pop();
            if (nd.key >= mu) continue;                                    // its bound can only be >= the parent's
            bp.push_back(nd.pat); bf.push_back(nd.free_);
        }
        if (bp.empty()) break;
        lb.resize(bp.size()); br.resize(bp.size());
        partls_status st = partls_bnb_bound(c, (int64_t)bp.size(), bp.data(), bf.data(), lb.data(), br.data());
        if (st != PARTLS_OK) return st;
        for (size_t i = 0; i < bp.size(); ++i) {
            ++bounded;
            if (lb[i] >= mu) continue;                                     // BnB.jl:102
            if (br[i] < 0) { mu = lb[i]; best_pat = bp[i]; best_free = bf[i]; have = true; continue; }   // BnB.jl:109-115
            const uint64_t bit = 1ULL << br[i];
            frontier.push({lb[i], bp[i] | bit, bf[i] & ~bit, seq++});      // α_pk >= 0 first (BnB.jl:120,123)
            frontier.push({lb[i], bp[i] & ~bit, bf[i] & ~bit, seq++});     // α_pk <= 0
        }
    }
    if (!have) { set_error("partls_fit_bnb: no feasible leaf found"); return PARTLS_ERR_NOT_CONVERGED; }
    if (nopen) *nopen = bounded;
    return partls_bnb_leaf(c, best_pat, best_free, alpha, beta, t, opt);
}

}  // extern "C"
