// solvers.hip — fit(Alt) and fit(BnB) on the Gram / tableau kernels (SURVEY.md §8f-1,2).
//
// Both reduce to batches of sign-constrained least-squares "nodes" on the shared Gram block, solved on the device by the
// sweep kernels in node mode (solve_nodes, api.hip):
//   * Alt α-step (Alt.jl:80-90):  nonneg_lsq(Xo .* (Po β)', y) has the constraint set {Xo w : sign(β_g(m)) w_m >= 0}, i.e. it
//     is the Opt subproblem of the sign pattern of β (groups with β_k == 0 have a zero column: α stays 0); α_m = w_m / β_g(m).
//   * Alt β-step (Alt.jl:109-110): (Xo (Po∘α)) \ yo  ==  solve (A' G A) β = A' c  with A = Po∘α — (K+1)^2, on the host
//     from the device-built Gram copy, like cleanupResult an O(K^3) epilogue.
//   * BnB node bound (BnB.jl:69-92): the [Xp Xm] doubling is NNLS's way of writing free variables; a node is the same
//     tableau problem with the not-yet-branched groups FREE.  Nodes are bounded in batches (best-first frontier), the
//     incumbent prunes (BnB.jl:102); the node count depends on the search order and is not a parity quantity.
#include "ctx.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <queue>

using namespace partls;

namespace {

// every feature must belong to at most one group for the group-pattern encoding of multipliers to be exact
bool proper_partition(const partls_ctx *c)
{
    for (int64_t m = 0; m < c->M; ++m)
        if (__builtin_popcountll(c->mask_aug[(size_t)m]) > 1) return false;
    return true;
}

int group_of(const partls_ctx *c, int64_t m)          // -1: in no group
{
    const uint64_t mk = c->mask_aug[(size_t)m];
    return mk ? __builtin_ctzll(mk) : -1;
}

// w'Gw - 2 w'c + yy on the regularised host Gram copy (w over [features, intercept])
double gram_objective2(const partls_ctx *c, const std::vector<double> &w)
{
    const int Mp = (int)c->M + 1, Y = (int)c->M + 1;
    double s = h_reg(c, Y, Y);
    for (int i = 0; i < Mp; ++i) {
        if (w[(size_t)i] == 0.0) continue;
        double gi = 0.0;
        if (c->eta == 0.0) {
            const double *Grow = &c->hG[(size_t)i * c->ldg];
            for (int j = 0; j < Mp; ++j) gi += Grow[j] * w[(size_t)j];
        } else {
            for (int j = 0; j < Mp; ++j) gi += h_reg(c, i, j) * w[(size_t)j];
        }
        s += w[(size_t)i] * (gi - 2.0 * h_reg(c, i, Y));
    }
    return s;
}

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
    if (!proper_partition(c)) { set_error("partls_fit_alt: a feature belongs to more than one group (overlapping partitions are not supported on the device path)"); return PARTLS_ERR_UNSUPPORTED; }
    const int Mp = (int)M + 1, Kp = (int)K + 1, Y = (int)M + 1;
    std::vector<double> a(alpha0, alpha0 + Mp), b(beta0, beta0 + Kp), w((size_t)Mp, 0.0), sols, obj2;
    std::vector<int> grp((size_t)Mp);
    for (int m = 0; m < Mp; ++m) grp[(size_t)m] = group_of(c, m);
    auto w_from = [&]() { for (int m = 0; m < Mp; ++m) w[(size_t)m] = (grp[(size_t)m] >= 0) ? a[(size_t)m] * b[(size_t)grp[(size_t)m]] : 0.0; };

    double oldopt = 1e20, optval = 1e10;                 // Alt.jl:73-74
    int64_t i = 1;
    unsigned long long unconv_total = 0;
    while (i <= T && std::fabs(oldopt - optval) > eps * oldopt) {
        // ---- α-step: one sign-constrained solve on the device (Alt.jl:80-90) ----------------------------------------------
        uint64_t pat = 0, zero = 0;
        for (int k = 0; k < Kp; ++k) { if (b[(size_t)k] > 0.0) pat |= 1ULL << k; else if (b[(size_t)k] == 0.0) zero |= 1ULL << k; }
        unsigned long long unconv = 0;
        const auto tt0 = std::chrono::steady_clock::now();
        st = solve_nodes(c, {pat}, {0}, {zero}, sols, obj2, &unconv, /*resume=*/i > 1);   // warm start after the first α-step
        const auto tt1 = std::chrono::steady_clock::now();
        if (st != PARTLS_OK) return st;
        unconv_total += unconv;
        std::vector<double> wv;
        unscale_solution(c, sols.data(), wv);
        for (int m = 0; m < Mp; ++m) {
            const double f = (grp[(size_t)m] >= 0) ? b[(size_t)grp[(size_t)m]] : 0.0;
            const double am = (f != 0.0) ? wv[(size_t)m] / f : 0.0;
            a[(size_t)m] = am > 0.0 ? am : 0.0;
        }
        // ---- checkalpha (Alt.jl:5-20) and renormalisation (Alt.jl:95-98) ----------------------------------------------------
        std::vector<double> suma((size_t)Kp, 0.0);
        std::vector<int> cntk((size_t)Kp, 0);
        for (int m = 0; m < Mp; ++m) if (grp[(size_t)m] >= 0) { suma[(size_t)grp[(size_t)m]] += a[(size_t)m]; ++cntk[(size_t)grp[(size_t)m]]; }
        for (int m = 0; m < Mp; ++m) { const int g = grp[(size_t)m]; if (g >= 0 && suma[(size_t)g] == 0.0) a[(size_t)m] = 1.0 / (double)cntk[(size_t)g]; }
        std::fill(suma.begin(), suma.end(), 0.0);
        for (int m = 0; m < Mp; ++m) if (grp[(size_t)m] >= 0) suma[(size_t)grp[(size_t)m]] += a[(size_t)m];
        for (int m = 0; m < Mp; ++m) { const int g = grp[(size_t)m]; if (g >= 0) a[(size_t)m] /= suma[(size_t)g]; }
        for (int k = 0; k < Kp; ++k) b[(size_t)k] *= suma[(size_t)k];
        // ---- β-step: (A' G A) β = A' c,  A = Po∘α  (Alt.jl:109-110 in Gram form) ---------------------------------------------
        std::vector<double> H((size_t)Kp * Kp, 0.0), g((size_t)Kp, 0.0);
        for (int m = 0; m < Mp; ++m) {
            const int gm = grp[(size_t)m];
            if (gm < 0 || a[(size_t)m] == 0.0) continue;
            g[(size_t)gm] += a[(size_t)m] * h_reg(c, m, Y);
            if (c->eta == 0.0) {                                   // plain Gram row: no per-entry call (M^2 entries per iteration)
                const double *Grow = &c->hG[(size_t)m * c->ldg];
                double *Hrow = &H[(size_t)gm * Kp];
                const double am = a[(size_t)m];
                for (int m2 = 0; m2 < Mp; ++m2) {
                    const int g2 = grp[(size_t)m2];
                    if (g2 >= 0) Hrow[g2] += am * Grow[m2] * a[(size_t)m2];
                }
            } else {
                for (int m2 = 0; m2 < Mp; ++m2) {
                    const int g2 = grp[(size_t)m2];
                    if (g2 < 0 || a[(size_t)m2] == 0.0) continue;
                    H[(size_t)gm * Kp + g2] += a[(size_t)m] * h_reg(c, m, m2) * a[(size_t)m2];
                }
            }
        }
        // groups without members (possible only for user groups with no feature) get a unit diagonal so H stays regular
        for (int k = 0; k < Kp; ++k) if (H[(size_t)k * Kp + k] == 0.0) H[(size_t)k * Kp + k] = 1.0;
        if (!solve_dense(H, g, Kp)) { set_error("partls_fit_alt: singular beta-step system"); return PARTLS_ERR_NOT_CONVERGED; }
        b = g;
        // ---- loss (Alt.jl:112-113) -----------------------------------------------------------------------------------------
        w_from();
        const double o2 = gram_objective2(c, w);
        oldopt = optval;
        optval = std::sqrt(o2 > 0.0 ? o2 : 0.0);
        if (c->knobs.alt_trace) fprintf(stderr, "[alt] iter %d: alpha-step %.3f ms, rest %.3f ms\n", (int)i, std::chrono::duration<double, std::milli>(tt1 - tt0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt1).count());
        ++i;
    }
    w_from();
    double dopt = optval;
    st = data_objective(c, w, &dopt);                    // final objective from the data (no Gram cancellation)
    if (st != PARTLS_OK) return st;
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

partls_status partls_bnb_prepared(partls_ctx *c, double *alpha, double *beta, double *t, double *opt, int64_t *nopen)
{
    if (!c || !c->prepared || !c->faithful) { set_error("partls_bnb_prepared: needs a context prepared with PARTLS_OPT_FAITHFUL_INTERCEPT"); return PARTLS_ERR_STATE; }
    if (!alpha || !beta || !t || !opt) { set_error("partls_fit_bnb: NULL argument"); return PARTLS_ERR_BAD_ARG; }
    PARTLS_HIP_CHECK(hipSetDevice(c->device));
    partls_status st = PARTLS_OK;
    const int64_t M = c->M, K = c->K;
    if (!proper_partition(c)) { set_error("partls_fit_bnb: a feature belongs to more than one group (overlapping partitions are not supported on the device path)"); return PARTLS_ERR_UNSUPPORTED; }
    const int Mp = (int)M + 1, Kp = (int)K + 1;
    std::vector<int> grp((size_t)Mp);
    for (int m = 0; m < Mp; ++m) grp[(size_t)m] = group_of(c, m);

    struct Node { double key; uint64_t pat, free_; unsigned long long seq; };
    struct Cmp { bool operator()(const Node &a, const Node &b) const { return a.key > b.key || (a.key == b.key && a.seq > b.seq); } };
    std::priority_queue<Node, std::vector<Node>, Cmp> frontier;           // best-first on the parent's bound
    unsigned long long seq = 0;
    frontier.push({0.0, 0ULL, ((uint64_t)1 << Kp) - 1, seq++});            // root: everything free (Σ = [], BnB.jl:33)
    double mu = INFINITY;
    std::vector<double> best_w;
    int64_t bounded = 0;
    unsigned long long unconv_total = 0;
    const size_t BATCH = 512;
    std::vector<uint64_t> bp, bf, bz;
    std::vector<double> sols, obj2, w;
    while (!frontier.empty()) {
        bp.clear(); bf.clear();
        while (!frontier.empty() && bp.size() < BATCH) {
            const Node nd = frontier.top();
            frontier.pop();
            if (nd.key >= mu) continue;                                    // its bound can only be >= the parent's
            bp.push_back(nd.pat); bf.push_back(nd.free_);
        }
        if (bp.empty()) break;
        bz.assign(bp.size(), 0ULL);
        unsigned long long unconv = 0;
        st = solve_nodes(c, bp, bf, bz, sols, obj2, &unconv);
        if (st != PARTLS_OK) return st;
        unconv_total += unconv;
        for (size_t i = 0; i < bp.size(); ++i) {
            ++bounded;
            const double lb = std::sqrt(obj2[i] > 0.0 ? obj2[i] : 0.0);
            if (lb >= mu) continue;                                        // BnB.jl:102
            unscale_solution(c, sols.data() + i * (size_t)c->n, w);
            // ν_k = Σ_{i<j in group k} max(0, -w_i w_j)  (BnB.jl:42-57); only free groups can mix signs
            double wmax = 0.0;
            for (int m = 0; m < Mp; ++m) wmax = std::max(wmax, std::fabs(w[(size_t)m]));
            const double tiny = 1e-12 * wmax;
            int kbest = -1; double nubest = 0.0;
            for (int k = 0; k < Kp; ++k) {
                if (!((bf[i] >> k) & 1ULL)) continue;
                double pos = 0.0, neg = 0.0;                               // Σ_{i<j} max(0,-w_i w_j) = (Σ w+)(Σ |w-|)
                for (int m = 0; m < Mp; ++m)
                    if (grp[(size_t)m] == k && std::fabs(w[(size_t)m]) > tiny) { if (w[(size_t)m] > 0.0) pos += w[(size_t)m]; else neg -= w[(size_t)m]; }
                const double nu = pos * neg;
                if (nu > nubest) { nubest = nu; kbest = k; }               // argmax: first maximal index
            }
            if (kbest < 0) {                                               // feasible for the original problem (BnB.jl:109-115)
                if (lb < mu) { mu = lb; best_w = w; }
                continue;
            }
            const uint64_t bit = 1ULL << kbest;
            frontier.push({lb, bp[i] | bit, bf[i] & ~bit, seq++});         // α_pk >= 0 first (BnB.jl:120,123)
            frontier.push({lb, bp[i] & ~bit, bf[i] & ~bit, seq++});        // α_pk <= 0
        }
    }
    if (best_w.empty()) { set_error("partls_fit_bnb: no feasible leaf found"); return PARTLS_ERR_NOT_CONVERGED; }
    st = refine_solution(c, best_w, false);
    if (st != PARTLS_OK) return st;
    // BnB.jl:36-39: β_k = Σ_{m∈k} α_m (signed); α_m ← α_m / β_k; t = β[end]
    std::vector<double> bsum((size_t)Kp, 0.0);
    for (int m = 0; m < Mp; ++m) if (grp[(size_t)m] >= 0) bsum[(size_t)grp[(size_t)m]] += best_w[(size_t)m];
    for (int64_t m = 0; m < M; ++m) alpha[m] = (grp[(size_t)m] >= 0) ? best_w[(size_t)m] / bsum[(size_t)grp[(size_t)m]] : 0.0;
    for (int64_t k = 0; k < K; ++k) beta[k] = bsum[(size_t)k];
    *t = bsum[(size_t)K];
    st = data_objective(c, best_w, opt);
    if (st != PARTLS_OK) return st;
    if (nopen) *nopen = bounded;
    if (unconv_total) { set_error("partls_fit_bnb: a node bound hit the pivot cap"); return PARTLS_ERR_NOT_CONVERGED; }
    return PARTLS_OK;
}

}  // extern "C"
