// common.h — internal declarations shared by the HIP translation units of libpartls_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include <string>
#include "../../include/partls.h"

namespace partls {

// ---------------------------------------------------------------------------------------------------------------------
// error plumbing: no exception crosses the C ABI; every entry point returns a status and records a message.
// ---------------------------------------------------------------------------------------------------------------------
void set_error(const char *fmt, ...);
#define PARTLS_HIP_CHECK(expr)                                                                      \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            ::partls::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return PARTLS_ERR_HIP;                                                                  \
        }                                                                                           \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// sweep parameters (shared by the register-resident and the global-memory tableau kernels)
// ---------------------------------------------------------------------------------------------------------------------
struct BitOrder { uint8_t gbit[40]; };                  // group k of the reference sits on bit gbit[k] of the internal pattern

struct SweepParams {
    int n;                       // tableau variables (features [+ intercept when it is sign-constrained])
    int kbits;                   // bits of the pattern space (K' = number of groups that carry a sign)
    const uint64_t *mask;        // [n] group-membership bit mask of every variable (bit k = member of group k)
    const double *T0;            // initial tableau, kernel-specific layout
    double *scratch;             // generic kernel: per-workgroup tableau scratch
    int64_t g_begin, g_end;      // Gray-index range [g_begin, g_end)
    int64_t chain_len;           // patterns per chain (fresh tableau at the start of every chain)
    double tol;                  // feasibility tolerance on the rhs column (scaled units)
    double piv_eps;              // an entering pivot below this is treated as a dependent column and skipped
    int max_rounds;              // cap on exchange rounds per pattern
    double *all_opt;             // optional [2^kbits] objective per pattern index
    double *best_obj;            // [gridDim.x] per-workgroup minimum objective
    int64_t *best_pat;           // [gridDim.x] its pattern index (internal bit order; ties broken on the REFERENCE index, see rbit)
    double *best_sol;            // optional [gridDim.x][node_ld] (register kernels, chain mode): scaled solution of the workgroup's best pattern
                                 // (0 for nonbasic variables), written whenever the workgroup finds a new minimum
    double *second_obj;          // optional [gridDim.x]: the workgroup's runner-up (second smallest objective, a different pattern) and
    int64_t *second_pat;         // its pattern, -1 if none: candidates of the host's near-tie re-rank by the data objective (api.hip)
    BitOrder rbit;               // chain mode: rbit.gbit[b] = the reference's bit (group) that internal pattern bit b carries.  Only
                                 // exact objective ties read it: argmin keeps the first REFERENCE index (Opt.jl:96)
    unsigned long long *n_unconverged;   // patterns that hit max_rounds
    unsigned long long *n_pivots;        // total pivots on the (n+1)^2 tableau (diagnostics / flop accounting)
    unsigned long long *n_vetoes;        // entering pivots refused by the leave-one-out rule (diagnostics)
    // Node mode (Alt alpha-steps, BnB node bounds): when node_code != nullptr chain c is ONE subproblem whose constraint on
    // tableau variable v is node_code[c * node_ld + v]:  +1 (w_v >= 0), -1 (w_v <= 0), 0 (w_v = 0: zero multiplier, or both
    // sign constraints of an overlapping partition), 2 (free: BnB's not yet branched groups, BnB.jl:70-79).  Per VARIABLE, not
    // per group: Alt's multipliers sum_k P[m,k] beta_k (Alt.jl:80-81) and BnB's accumulated constraints (BnB.jl:120-121) are
    // not functions of a group sign pattern once a feature sits in two groups.  Outputs per node:
    // node_sol[c * node_ld + v] = scaled solution (0 for nonbasic), node_obj2[c] = objective^2.
    // chain_len is 1 everywhere except in the bit-order calibration (api.hip: calibrate_bit_order; register and one-workgroup
    // global-memory kernel): there chain c
    // solves the nodes c * chain_len + 0, 1, ... one after the other, each warm-started from its predecessor's tableau, node_sol /
    // node_obj2 describe the LAST node of the chain, and node_piv[3 * (c * chain_len + i) + 0 / 1 / 2] = pivots / block pivots / KKT
    // scans of the workgroup up to and including node i (the global-memory kernel reports pivots only).
    const int8_t *node_code;
    double *node_sol;
    double *node_obj2;
    int node_ld;
    // optional (register kernel, node mode): the final tableau of node c in the tile-cyclic layout of T0 (sweep_reg_t0_doubles per
    // node) and its basis flags [16 T per node] — the basic x basic block is -(B_BB)^-1 on the unit-diagonal scale, which the
    // winner's iterative refinement uses as its solver (api.hip: refine_solution)
    double *node_tab;
    int8_t *node_basic;
    unsigned *node_piv;          // optional (see above)
    // optional (register kernel, node mode, chain_len == 1): tableau SNAPSHOTS — warm starts of BnB node bounds (BnB.jl:120-124: a child
    // is its parent's problem plus one group's sign constraint, so it starts from the parent's final tableau and exchanges only that
    // group's wrong-signed variables).  node_src[c]: snapshot node c starts from (nullptr: the fresh tableau T0); node_dst[c]: where its
    // final state is stored (nullptr: nowhere).  A snapshot = sweep_reg_t0_doubles(T) doubles in the layout of T0 (tiles, rhs column,
    // corner) followed by 16 T basis flags (bytes).
    const double *const *node_src;
    double *const *node_dst;
    // cooperative single-node kernel only: continue from the tableau / basis left in `scratch` by the previous launch
    // (warm start of consecutive Alt alpha-steps) instead of reloading T0
    int resume;
    unsigned *grid_ctr;          // multi-workgroup kernel: arrival counter [0] and abort word [1] of its grid barrier (zeroed by the launcher)
    int coop_fault;              // test hook (PARTLS_COOP_FAULT): the grid barrier expects this many arrivals too many, i.e. it can
                                 // only time out — exercises the abort word and the host's one-workgroup fallback.  77 in a launch of the deferred-update
                                 // kernel (PARTLS_LZ_FAULT): workgroup 0's first two-phase panel never publishes its progress word
};

// launchers (each returns hipError_t of the launch)
// Exact objective ties (rare path): does internal pattern a come before internal pattern b in the reference's index order?  The
// reference index of a pattern is sum_b bit_b << rbit[b]; of two patterns the one with a 0 in the differing bit of highest reference
// weight is the smaller.
__device__ __forceinline__ bool ref_index_less(unsigned long long a, unsigned long long b, const unsigned char *rbit)
{
    unsigned long long d = a ^ b;
    int top = -1, at = 0;
    while (d) {
        const int i = __builtin_ctzll(d);
        d &= d - 1;
        const int r = rbit[i];
        if (r > top) { top = r; at = i; }
    }
    return top >= 0 && !((a >> at) & 1ULL);
}

hipError_t launch_sweep_generic(const SweepParams &p, int grid, hipStream_t s);   // tableau in global memory, every block applied at once
hipError_t launch_sweep_lazy(const SweepParams &p, int grid, hipStream_t s);      // ... updates deferred (sweep_lazy.hip): the n > 320 path
hipError_t launch_sweep_blk(const SweepParams &p, int T, int grid, hipStream_t s);
hipError_t launch_sweep_coop(const SweepParams &p, int nwg, hipStream_t s);   // one node, many workgroups (n > 320)
bool       sweep_reg_supported(int n);
int        sweep_reg_tiles(int n);
bool       sweep_reg_exports(int T);                     // the 512-thread kernel of this build leaves its winner's solution behind as well
bool       sweep_reg_small(int T);                       // T tile columns run on the 256-thread kernel (which also exports best_sol)
size_t     sweep_reg_t0_doubles(int T);                  // size of the tile-cyclic initial tableau
int        sweep_reg_concurrency(int T);                 // chains a CU runs at the same time (1: the 512-thread kernel, > 1: small tableaus)
hipError_t launch_layout_reg(const double *Tfull, int n, int T, double *T0reg, hipStream_t s);

// gram build: G_aug = Z'Z with Z = [X 1 y]  (n_aug = M + 2), full symmetric, ld = ldg (multiple of 64)
// (gram_S / gram_cr: tuning overrides read once at partls_create, 0 = automatic)
size_t     gram_slab_doubles(int64_t N, int64_t M, int gram_S, int gram_cr, int *chunks_out, int *ldg_out);
hipError_t launch_gram(const double *X, int64_t N, int64_t M, int64_t ldX, const double *y, double *slab, int chunks,
                       int ldg, int gram_S, int gram_cr, double *G, hipStream_t s);

// tableau prep: B = regularised (and, free-intercept mode, intercept-eliminated) Gram; Tfull = unit-diagonal scaled
// perm[i] = augmented-Gram index of tableau variable i (variables are grouped by partition so a flip touches few tiles)
hipError_t launch_prep(const double *G, int ldg, int M, double eta, const uint64_t *mask_aug, int free_intercept,
                       const int *perm, double *scale, double *Tfull, int n, hipStream_t s);

// bit-order calibration (misc.hip): node codes of the calibration walks, and all_opt from internal to reference pattern order
int        walk_flipped_bit(int chain, int step, int kbits, int seg_len, int nseg);
hipError_t launch_walk_codes(const uint64_t *mask, int n, int kbits, int chains, int L, int seg_len, int nseg, int8_t *codes, hipStream_t s);
hipError_t launch_pattern_gather(const double *in, int64_t npat, int kbits, const BitOrder &order, double *out, hipStream_t s);

// BnB node batches (misc.hip): per-variable constraint codes of the nodes (pat, free) and (bound, branch) from their solutions
hipError_t launch_bnb_codes(const uint64_t *mask_tab, int n, const uint64_t *pat, const uint64_t *free_, int cnt, int8_t *codes, hipStream_t s);
hipError_t launch_bnb_nu(const double *sol, const double *obj2, int n, const double *scale, const uint64_t *mask_tab, int Kp,
                         const uint64_t *free_, int cnt, double *lb, int *branch, hipStream_t s);

hipError_t launch_gersh(const double *Tfull, int n, double *out, hipStream_t s);   // Gershgorin radii of the scaled Gram block (n doubles)
// residual from the data: out[0] = sum_i (sum_m X[i,m] w[m] + t - y[i])^2   (y may be nullptr -> plain prediction into yhat)
// beta-step system of fit(Alt): Hg[k * (Kp + 1) + k2] = H[k][k2] (k2 < Kp), g[k] (k2 = Kp); GA is (M + 1) x Kp scratch
hipError_t launch_alt_beta_system(const double *G, int ldg, int M, double eta, const uint64_t *mask_aug, const double *a, int Kp,
                                  double *GA, double *Hg, hipStream_t s);
hipError_t launch_residual(const double *X, int64_t N, int64_t M, int64_t ldX, const double *y, const double *w, double t,
                           double *partial, int nblocks, double *yhat, hipStream_t s);
// g[0..M] = Xo' (y - yhat): the gradient pass of the data-space refinement
// gpart: xtr_slices(N) x (M + 1) partial sums; g[m] = sum over the slices in order
int        xtr_slices(int64_t N);
hipError_t launch_xtr(const double *X, int64_t N, int64_t M, int64_t ldX, const double *y, const double *yhat, double *gpart,
                      hipStream_t s);
hipError_t launch_synth(uint64_t seed, int64_t N, int64_t D, const double *wstar_dev, double *X, double *y, hipStream_t s);

// host-side mirror of the device generator (synth.hip); also used by partls_synth_truth
uint64_t rnd64(uint64_t seed, uint64_t stream, uint64_t idx);

}  // namespace partls
