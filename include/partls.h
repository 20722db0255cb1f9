/*
 * partls.h — C ABI of the MI355X-native Partitioned-LS hot path (libpartls_hip.so, gfx950).
 *
 * The reference (/root/reference, pure Julia) has no FFI layer; its boundary is Julia multiple dispatch on
 *   fit(::Type{Opt}, X, y, P; η, nnlsalg, returnAllSolutions)        src/PartitionedLSOpt.jl:73-74
 *   fit(::Type{Alt}, X, y, P; η, ϵ, T, nnlsalg, rng)                 src/PartitionedLSAlt.jl:50-51
 *   fit(::Type{BnB}, X, y, P; η, nnlsalg)                            src/PartitionedLSBnB.jl:30
 *   predict(α, β, t, P, X) / predict(model, X)                       src/PartitionedLS.jl:132,152
 * A Julia shim keeps those signatures and `ccall`s the entry points below (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - plain C symbols, plain pointers and sizes; no C++ exception crosses this boundary;
 *   - matrices are COLUMN-MAJOR (Julia native) with an explicit leading dimension;
 *   - P is int64 (Julia Int on x64) with entries in {0,1}, M x K;
 *   - every output buffer is caller-allocated; the library never retains or frees caller memory;
 *   - every call returns a partls_status; partls_last_error() gives the thread-local message;
 *   - calls are synchronous (they return after the device work has completed);
 *   - a context is not thread-safe; distinct contexts may be used from distinct threads.
 *   - there is NO CPU fallback: without a HIP device every compute entry returns PARTLS_ERR_NO_DEVICE.
 */
#ifndef PARTLS_H
#define PARTLS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PARTLS_OK = 0,
    PARTLS_ERR_BAD_ARG = 1,        /* null pointer, negative size, ld < rows ...                         */
    PARTLS_ERR_BAD_PARTITION = 2,  /* P has an entry outside {0,1} (PartitionedLS.jl:292 validity clause) */
    PARTLS_ERR_NONFINITE = 3,      /* NaN/Inf in X or y, or a column whose sum of squares overflows fp64    */
    PARTLS_ERR_NO_DEVICE = 4,      /* no HIP device / device index out of range                           */
    PARTLS_ERR_HIP = 5,            /* a HIP runtime call failed (message has the hipError string)         */
    PARTLS_ERR_NOT_CONVERGED = 6,  /* an active-set solve hit its pivot cap                               */
    PARTLS_ERR_UNSUPPORTED = 7,    /* shape outside what the kernels are built for: M > 1022 features, ldX >= 2^30, K > 61 groups,
                                      K > 39 for the 2^K enumeration of fit(Opt)                                              */
    PARTLS_ERR_STATE = 8,          /* staged calls issued out of order                                    */
    PARTLS_ERR_ILL_CONDITIONED = 9 /* the model's KKT conditions fail when checked against the DATA: X is too ill-conditioned for the
                                      fp64 Gram form (cond(X)^2 * eps >~ 1).  Outputs are filled with the best Gram-form model;
                                      a QR-based solver (the reference's, Opt.jl:89) is the right tool for this input          */
} partls_status;

/* flags for the Opt entry points */
#define PARTLS_OPT_FAITHFUL_INTERCEPT 1u /* enumerate all 2^(K+1) sign vectors incl. the intercept's, as Opt.jl:81,85 does.
                                            Without it the intercept is left free and 2^K subproblems are solved: the
                                            optimum (model and objective) is identical, min over the ± pair.             */
#define PARTLS_OPT_GENERIC_KERNEL     2u /* force the global-memory tableau kernel (testing / cross-check)               */

typedef struct partls_ctx partls_ctx;

int          partls_version(void);                 /* 10000*major + 100*minor + patch */
const char  *partls_last_error(void);              /* thread-local, never NULL        */
int          partls_device_count(void);            /* 0 when no HIP device is visible */

partls_status partls_create(int device, partls_ctx **out);
void          partls_destroy(partls_ctx *ctx);

/* ---- fit(Opt, X, y, P; η, returnAllSolutions)  — replaces Opt.jl:73-104 ----------------------------------------------
 * X: N x M (N <= ldX < 2^30), y: N, P: M x K (ldP >= M).  Outputs follow cleanupResult (Opt.jl:34-44):
 *   alpha[M] (sums to 1 per group), beta[K], *t, *opt = ||Xo w - yo||_2 of the winner (un-squared, incl. the η rows),
 *   *best_index = the reference's 0-based pattern index b (bit k = sign of group k+1, bit K = intercept sign; in
 *   free-intercept mode bit K is set from the sign of the fitted intercept).
 *   all_opt: optional (may be NULL); needs PARTLS_OPT_FAITHFUL_INTERCEPT; 2^(K+1) doubles, all_opt[b] = optval of pattern b
 *   (Opt.jl:90) computed from the Gram form.  Models of individual patterns: partls_opt_pattern(). */
partls_status partls_fit_opt(partls_ctx *ctx, const double *X, int64_t N, int64_t M, int64_t ldX, const double *y,
                             const int64_t *P, int64_t K, int64_t ldP, double eta, uint32_t flags,
                             double *alpha, double *beta, double *t, double *opt, int64_t *best_index, double *all_opt);

/* ---- fit(Opt) on several GPUs of one node, inside the library  — the loop Opt.jl:85-94 has no loop-carried state ---------
 * One process, one host thread and one context per device.  Device r uploads rows [r N / R, (r+1) N / R) of X and y (1 / R of the PCIe
 * traffic each) and builds the Gram products of its block; their sum — ncclAllReduce(ncclSum) on (M+2)^2 doubles over xGMI — is the
 * problem every device then prepares.  Device r sweeps the Gray-index range [r * 2^K' / R, (r+1) * 2^K' / R) of the pattern space,
 * and the global lexicographic minimum
 * (objective, reference pattern index) — argmin's first-index rule, Opt.jl:96 — is taken with two RCCL all-reduces over xGMI:
 * ncclMin on the objective, then ncclMin on the index masked to the minimisers (RCCL has no MINLOC).  The key of the
 * visiting order rides in the first all-reduce; ranks that disagree on it fail with PARTLS_ERR_STATE instead of combining
 * shards that do not partition the pattern space.  The winner is re-solved on the first device; the passes over the data it needs
 * (refinement, objective, KKT check) run on every device's row block and are summed in rank order.
 *   devices[ndev]: HIP device indices (devices == NULL: devices 0 .. ndev-1; ndev == 0: every visible device).
 *   A list that names one device more than once (rehearsal of the R-rank control flow on a one-GPU box) cannot form an RCCL
 *   communicator: its reduction runs through the host instead; everything else is the same code.
 *   librccl.so.1 is loaded when the first communicator is needed (PARTLS_RCCL_LIB overrides the name), never for partls_fit_opt.
 * partls_fit_opt_multi: arguments and outputs exactly as partls_fit_opt (all_opt: the shards' entries merged).
 * partls_multi_context: the context of rank r (rank 0 holds the winner's problem after a fit: partls_opt_finish /
 * partls_opt_pattern on it rebuild the models of returnAllSolutions, Opt.jl:99-101).  A partls_multi is not thread-safe. */
typedef struct partls_multi partls_multi;
partls_status partls_multi_create(const int *devices, int ndev, partls_multi **out);
void          partls_multi_destroy(partls_multi *mc);
int           partls_multi_size(const partls_multi *mc);                 /* number of ranks (0 for NULL)                  */
int           partls_multi_uses_rccl(const partls_multi *mc);            /* 1: reductions over RCCL, 0: host (see above)   */
partls_ctx   *partls_multi_context(partls_multi *mc, int rank);          /* NULL when out of range                         */
partls_status partls_fit_opt_multi(partls_multi *mc, const double *X, int64_t N, int64_t M, int64_t ldX, const double *y,
                                   const int64_t *P, int64_t K, int64_t ldP, double eta, uint32_t flags,
                                   double *alpha, double *beta, double *t, double *opt, int64_t *best_index, double *all_opt);
/* ---- fit(BnB) on several GPUs of one node, inside the library  — BnB.jl:94-132: subtrees are independent given the incumbent --------
 * Same handle, same row-sharded upload and Gram sum as partls_fit_opt_multi.  Every rank thread runs the same best-first frontier
 * (partls_frontier_*): per round the batch * R most promising nodes are dealt — a node goes to the rank that holds its parent's tableau
 * snapshot (warm start), the surplus and the cold nodes to the least loaded ranks —, every rank bounds its share on its own GPU, and ONE
 * all-gather of (bound, branch, snapshot slot) per round — ncclAllGather over xGMI; host memory for a device list with duplicates —
 * carries the incumbent, so that all ranks prune, branch and count snapshot references identically.  The incumbent's model is built on
 * the first device (partls_bnb_leaf, data passes over every rank's row block).  Outputs as partls_fit_bnb; *nopen = nodes bounded by
 * all ranks (not a parity quantity: the search order differs from the reference's depth-first recursion). */
partls_status partls_fit_bnb_multi(partls_multi *mc, const double *X, int64_t N, int64_t M, int64_t ldX, const double *y,
                                   const int64_t *P, int64_t K, int64_t ldP, double eta,
                                   double *alpha, double *beta, double *t, double *opt, int64_t *nopen);
/* Robustness of the rank threads (both fits): every phase ends in a rendezvous at which the ranks agree to go on or to leave together; a
 * rank that cannot keep the protocol, or does not reach a rendezvous within PARTLS_MULTI_TIMEOUT_S seconds (default 3600), fails the
 * fit with PARTLS_ERR_STATE instead of hanging it; a failed RCCL enqueue aborts the communicators (ncclCommAbort) and the handle
 * reduces through host memory from then on (partls_multi_uses_rccl turns 0). */
/* per-rank HIP-event time (ms) of stage `which` in the last partls_fit_opt_multi */
partls_status partls_multi_get_timing(const partls_multi *mc, int rank, int which, double *ms);

/* ---- staged form of the same path (multi-GPU sharding, device-resident inputs, benchmarking) ------------------------
 * prepare:  builds G = Xo'Xo, c = Xo'y, yy (fp64 MFMA), applies η, scales, lays the tableau out for the sweep.
 *           x_on_device != 0: X and y are DEVICE pointers (hipMalloc / torch) and stay owned by the caller.
 * sweep:    enumerates Gray indices [g_begin, g_end) of the 2^K' pattern space (K' = K+1 faithful, K free intercept);
 *           g_end = -1 means "to the end".  Returns the shard's lexicographic minimum (objective, pattern index) —
 *           the pair a rank feeds into the all-reduce(min).  all_opt as above (indexed by pattern, not by Gray index;
 *           entries of patterns outside [g_begin, g_end) are set to NaN).
 *           Gray indices are positions in the context's own visiting order: which group sits on which Gray bit is chosen
 *           per prepared problem (partls_opt_bit_order; measured on the first sweep of long enumerations, deterministic in
 *           the data, so ranks that prepared the same problem agree).  Any disjoint ranges covering [0, 2^K') visit every
 *           pattern exactly once; every pattern index that crosses this boundary is the reference's (group k = bit k).
 * finish:   solves the given pattern once more on its own, computes opt from the data, normalises (cleanupResult). */
partls_status partls_opt_prepare(partls_ctx *ctx, const double *X, int64_t N, int64_t M, int64_t ldX, const double *y,
                                 int x_on_device, const int64_t *P, int64_t K, int64_t ldP, double eta, uint32_t flags);
partls_status partls_opt_sweep(partls_ctx *ctx, int64_t g_begin, int64_t g_end,
                               double *best_obj, int64_t *best_pattern, double *all_opt, int64_t *n_unconverged);
partls_status partls_opt_finish(partls_ctx *ctx, int64_t pattern,
                                double *alpha, double *beta, double *t, double *opt, int64_t *best_index);
/* Near ties across shards.  The sweep ranks patterns on the Gram-form objective (absolute error ~eps * y'y); partls_opt_finish re-ranks
 * the winner and the (at most 3) patterns within that error of it by the objective computed from the DATA, first reference index on exact
 * ties (Opt.jl:90,96).  A host that shards the enumeration (one context per GPU) must give every rank the candidate set a single
 * context would have had:
 *   candidates: the shard's winner and its near ties with their tracked objectives, best first (count <= 4 <= capacity);
 *   merge:      the concatenation of ALL ranks' candidate lists (any order; the same list on every rank) -> the global lexicographic
 *               minimum (objective, reference index) in *win_obj / *win_pattern, installed together with its near ties for the next
 *               partls_opt_finish(ctx, *win_pattern, ...) on this context.  Every rank then returns the same model.
 * partls_fit_opt_multi does this between its rank threads; partitionedls.jl_amd/dist.py with one all-gather of 64 bytes per rank. */
partls_status partls_opt_candidates(const partls_ctx *ctx, int64_t capacity, double *obj, int64_t *pattern, int64_t *count);
partls_status partls_opt_merge_candidates(partls_ctx *ctx, int64_t count, const double *obj, const int64_t *pattern,
                                          double *win_obj, int64_t *win_pattern);
/* raw NNLS solution of one pattern b (reference indexing, K+1 bits): raw_alpha[M+1] >= 0 as nonneg_lsq returns it at
 * Opt.jl:89, and its optval (Opt.jl:90).  Needs a prepared context. */
partls_status partls_opt_pattern(partls_ctx *ctx, int64_t pattern, double *raw_alpha, double *optval);
/* visiting order of the sweep: gbit[k] = Gray-index bit that carries group k (K' entries; identity unless calibrated);
 * flip_cost (optional, K' doubles): measured cost of a flip of group k in pivot equivalents (pivots + weighted block pivots and
 * extra KKT scans; exact counts, no timing), -1 when the calibration did not run (sweeps too short
 * to repay it, PARTLS_BIT_ORDER=identity); gbit stays the identity when the measured costs promise less than 2 % fewer pivots.  Runs the calibration if no sweep has done so yet.  Ranks of a sharded
 * sweep must hold the same gbit (partitionedls.jl_amd/dist.py checks it inside its first all-reduce). */
partls_status partls_opt_bit_order(partls_ctx *ctx, int64_t *gbit, double *flip_cost);
/* number of subproblems one full sweep solves (2^K or 2^(K+1)) for the prepared problem */
int64_t       partls_opt_num_patterns(const partls_ctx *ctx);

/* ---- fit(Alt, X, y, P; η, ϵ, T)  — replaces Alt.jl:50-124 ------------------------------------------------------------
 * alpha0[M+1], beta0[K+1]: the random initial point the Julia shim draws exactly as Alt.jl:58-66 does.
 * Outputs as Alt.jl:119: alpha[M], beta[K], *t = β[end]*α[end], *opt, *iters = completed iterations. */
partls_status partls_fit_alt(partls_ctx *ctx, const double *X, int64_t N, int64_t M, int64_t ldX, const double *y,
                             const int64_t *P, int64_t K, int64_t ldP, double eta, double eps, int64_t T,
                             const double *alpha0, const double *beta0,
                             double *alpha, double *beta, double *t, double *opt, int64_t *iters);

/* The same on a context already prepared with partls_opt_prepare(..., flags | PARTLS_OPT_FAITHFUL_INTERCEPT) — e.g. with
 * device-resident inputs; partls_fit_alt / partls_fit_bnb are "prepare from host pointers" + these. */
partls_status partls_alt_prepared(partls_ctx *ctx, double eps, int64_t T, const double *alpha0, const double *beta0,
                                  double *alpha, double *beta, double *t, double *opt, int64_t *iters);
partls_status partls_bnb_prepared(partls_ctx *ctx, double *alpha, double *beta, double *t, double *opt, int64_t *nopen);

/* ---- fit(BnB, X, y, P; η)  — replaces BnB.jl:30-132 -------------------------------------------------------------------
 * Outputs as BnB.jl:36-40; *nopen = nodes bounded (search order differs from the reference's DFS, so it is not a
 * parity quantity). */
partls_status partls_fit_bnb(partls_ctx *ctx, const double *X, int64_t N, int64_t M, int64_t ldX, const double *y,
                             const int64_t *P, int64_t K, int64_t ldP, double eta,
                             double *alpha, double *beta, double *t, double *opt, int64_t *nopen);

/* The two primitives of the BnB search on a prepared (PARTLS_OPT_FAITHFUL_INTERCEPT) context — what a host that shards the
 * frontier across GPUs calls (partitionedls.jl_amd/dist.py: one process per GPU, batches dealt round-robin, one all-gather of
 * (bound, branch) per batch carries the incumbent).  A node is (pat, free): group k < K+1 is branched iff bit k of free is
 * clear, then α_pk >= 0 if bit k of pat is set, <= 0 otherwise (BnB.jl:120-121; group K is the intercept's).
 *   bound: lb[i] = lower bound of node i (BnB.jl:69-92); branch[i] = argmax_k ν_k (BnB.jl:107,117) or -1 when the relaxed
 *          solution is already feasible (BnB.jl:109-115: lb[i] is then an upper bound too).
 *   leaf:  model of one feasible node, normalised as BnB.jl:36-39, *opt from the data. */
partls_status partls_bnb_bound(partls_ctx *ctx, int64_t count, const uint64_t *pat, const uint64_t *free_groups,
                               double *lb, int32_t *branch);
partls_status partls_bnb_leaf(partls_ctx *ctx, uint64_t pat, uint64_t free_groups,
                              double *alpha, double *beta, double *t, double *opt);
/* Node bounds with tableau SNAPSHOTS, for a host that runs the search itself and shards it over GPUs (dist.py deals every node to the
 * rank that holds its parent's snapshot): node i starts from the final tableau stored in src_slot[i] (-1: the fresh tableau) and leaves
 * its own in dst_slot[i] (out; -1: pool full, no free group left, or the PARTLS_EAGER_GENERIC test kernel; both production kernels take
 * snapshots, at any n <= 1023).  The host keeps the reference counts and returns slots
 * with partls_bnb_snap_release; partls_bnb_snap_begin (after every prepare, before the first bound) empties the pool. */
partls_status partls_bnb_snap_begin(partls_ctx *ctx);
partls_status partls_bnb_bound_snap(partls_ctx *ctx, int64_t count, const uint64_t *pat, const uint64_t *free_groups,
                                    const int32_t *src_slot, int32_t *dst_slot, double *lb, int32_t *branch);
partls_status partls_bnb_snap_release(partls_ctx *ctx, int64_t count, const int32_t *slots);
/* The FRONTIER of the search as an object, for a host that shards the search over processes (one per GPU): every rank creates one with
 * its (rank, world) and drives rounds —
 *   next:    pops the batch * world most promising nodes (pruned against the incumbent, BnB.jl:102) and deals them: a node goes to the
 *            rank that holds its parent's snapshot, up to that rank's quota; the rest, cold, to the least loaded ranks.  *total = nodes
 *            of the round (0: the search is over), per_rank[world] = every rank's share, this rank's share in pat / free_groups /
 *            src_slot (capacity >= batch) ready for partls_bnb_bound_snap;
 *   (the ranks exchange lb / branch / dst_slot of their shares: one all-gather)
 *   ingest:  the results of ALL ranks in rank-major order (rank 0's nodes in the order `next` gave them, then rank 1's, ...): prunes,
 *            records feasible nodes, branches (BnB.jl:107-124), counts snapshot references; `dead` receives this rank's slots that lost
 *            their last reference (at most dead_capacity per call; call again with an empty round for the rest) for
 *            partls_bnb_snap_release.
 * All ranks make identical decisions (the frontier depends on shared data only).  partls_bnb_search is this loop with world = 1. */
typedef struct partls_frontier partls_frontier;
partls_status partls_frontier_create(int n_groups, int rank, int world, int64_t batch, partls_frontier **out);
void          partls_frontier_destroy(partls_frontier *f);
partls_status partls_frontier_next(partls_frontier *f, int64_t *total, int64_t *mine, uint64_t *pat, uint64_t *free_groups,
                                   int32_t *src_slot, int32_t *per_rank);
partls_status partls_frontier_ingest(partls_frontier *f, const double *lb, const int32_t *branch, const int32_t *dst_slot,
                                     int32_t *dead, int64_t dead_capacity, int64_t *ndead);
partls_status partls_frontier_result(const partls_frontier *f, double *mu, uint64_t *pat, uint64_t *free_groups, int64_t *nodes);
/* The search itself (what partls_fit_bnb / partls_bnb_prepared run before partls_bnb_leaf): best-first frontier, device batches,
 * every node warm-started from its parent's final tableau, which stays in HBM while the node has children in the frontier
 * (BnB.jl:120-124: a child is the parent's constraint set plus one group).  Returns the incumbent node (pat, free_groups), its
 * value *mu and the number of nodes bounded.  max_nodes > 0 stops the search there (measurement; *mu = +inf when no feasible
 * node was met yet); max_nodes <= 0 runs to optimality. */
partls_status partls_bnb_search(partls_ctx *ctx, int64_t max_nodes, double *mu, uint64_t *pat, uint64_t *free_groups, int64_t *nodes);

/* ---- predict(α, β, t, P, X)  — replaces PartitionedLS.jl:132-134: yhat = X*(P.*α)*β .+ t ------------------------------ */
partls_status partls_predict(partls_ctx *ctx, const double *X, int64_t N, int64_t M, int64_t ldX,
                             const int64_t *P, int64_t K, int64_t ldP, const double *alpha, const double *beta, double t,
                             double *yhat);
/* The same with X (N x M, ldX) and yhat (N) resident in HBM: DEVICE pointers that stay owned by the caller (the large-N
 * form of PartitionedLS.jl:132-134 — no PCIe copy of X; P, alpha, beta stay host pointers, they are M*K numbers). */
partls_status partls_predict_device(partls_ctx *ctx, const double *dX, int64_t N, int64_t M, int64_t ldX,
                                    const int64_t *P, int64_t K, int64_t ldP, const double *alpha, const double *beta,
                                    double t, double *dyhat);

/* ---- synthetic inputs of BASELINE.md §4, generated in HBM (bit-identical to oracle_synth on the host) -----------------
 * dX: device N x D (ld = N), dy: device N; wstar: HOST D doubles (alpha*_j beta*_g(j), from partls_synth_truth). */
partls_status partls_synth_truth(uint64_t seed, int64_t D, int64_t K, int64_t *P, double *wstar);
partls_status partls_synth_device(partls_ctx *ctx, uint64_t seed, int64_t N, int64_t D, const double *wstar,
                                  double *dX, double *dy);

/* ---- measurement: HIP-event timings (ms) of the kernels of the LAST staged/fit call on this context -------------------- */
typedef enum {
    PARTLS_T_GRAM = 0,      /* gram_build + slab reduction            */
    PARTLS_T_PREP = 1,      /* η / scaling / tableau layout           */
    PARTLS_T_SWEEP = 2,     /* the sign-pattern sweep kernel          */
    PARTLS_T_FINISH = 3,    /* winner re-solve + residual from data   */
    PARTLS_T_CALIB = 4,     /* bit-order calibration of the sweep (0 when it did not run) */
    PARTLS_T_COUNT = 5
} partls_timer;
partls_status partls_get_timing(const partls_ctx *ctx, partls_timer which, double *ms);
/* host -> device upload of X inside the last prepare / fit on this context: wall time (ms) and bytes (0 / 0 when the inputs were device
 * pointers).  A host X larger than 8 MB is staged through page-locked buffers by four copier threads (42-55 GB/s on a 57 GB/s link
 * whatever the array's history; a plain copy from pageable memory pays for the pinning first: 8-25 GB/s on a fresh array). */
partls_status partls_get_upload(const partls_ctx *ctx, double *ms, double *bytes);
/* principal pivots executed by the last partls_opt_sweep (fp64 flop accounting: each pivot updates the whole symmetric tableau) */
partls_status partls_get_pivots(const partls_ctx *ctx, int64_t *pivots);
/* entering pivots the last partls_opt_sweep refused under the leave-one-out dependence rule (0 on well-conditioned data; a
 * large count says the data are rank deficient on the unit-diagonal scale — see DESIGN.md §4, numerical notes) */
partls_status partls_get_vetoes(const partls_ctx *ctx, int64_t *vetoes);
/* data-space KKT violation of the model the last partls_opt_finish / partls_bnb_leaf (hence fit(Opt), fit(BnB)) returned: the largest
 * of |x_m'r| (passive or free variable), f_m x_m'r (variable at its bound) and -f_m w_m / max|w| over every variable, r = yo - Xo w
 * computed from the data, in units of ||x_m|| ||y||.  <= 3e-15 on data the Gram form resolves; above 1e-12 (PARTLS_KKT_TOL) the call
 * returns PARTLS_ERR_ILL_CONDITIONED (see there).  min_pivot (optional): the smallest leave-one-out pivot of that model's basis on the
 * unit-diagonal scale, a lower bound of 1 / cond(G_BB) (0 when unknown). */
partls_status partls_get_kkt_violation(const partls_ctx *ctx, double *violation, double *min_pivot);
/* distinct subproblems the last partls_opt_finish solved and compared on the data objective (1: the sweep recorded no near tie) */
partls_status partls_get_near_ties(const partls_ctx *ctx, int64_t *evaluated);
/* debugging / tests: copy the Gram products of the prepared problem to the host: G ((M+2) x (M+2), column-major,
 * variables ordered [features, intercept, y]), i.e. G, c = G[:, M+1], yy = G[M+1, M+1], after η has been applied. */
partls_status partls_get_gram(const partls_ctx *ctx, double *G_aug);

#ifdef __cplusplus
}
#endif
#endif
