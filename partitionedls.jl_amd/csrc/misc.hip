// misc.hip — the small kernels either side of the sweep: synthetic inputs, tableau preparation, residual from the data.
#include "common.h"

namespace partls {

// ---------------------------------------------------------------------------------------------------------------------
// Synthetic inputs: integer-exact counter-based generator — the device twin of oracle_synth().
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ static inline uint64_t sm64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__host__ __device__ static inline uint64_t rnd64_hd(uint64_t seed, uint64_t stream, uint64_t idx)
{
    return sm64(sm64(seed ^ (stream * 0xD6E8FEB86659FD93ULL)) + idx);
}
uint64_t rnd64(uint64_t seed, uint64_t stream, uint64_t idx) { return rnd64_hd(seed, stream, idx); }

__device__ static inline double gauss12(uint64_t seed, uint64_t stream, uint64_t idx)
{
    uint64_t s = 0;
#pragma unroll
    for (uint64_t r = 0; r < 3; ++r) {
        const uint64_t h = rnd64_hd(seed, stream, idx * 3 + r);
        s += (h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48);
    }
    return ((double)(int64_t)s - 393210.0) * 0x1.0p-16;
}

__global__ void synth_x_kernel(uint64_t seed, int64_t total, double *__restrict__ X)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        X[i] = gauss12(seed, 1, (uint64_t)i);
}

__global__ void synth_y_kernel(uint64_t seed, int64_t N, int64_t D, const double *__restrict__ X,
                               const double *__restrict__ wstar, double *__restrict__ y)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double acc = 0.0;
    for (int64_t j = 0; j < D; ++j) acc = fma(X[i + j * N], wstar[j], acc);      // same order as oracle_synth
    y[i] = fma(0.1, gauss12(seed, 4, (uint64_t)i), acc + 1.0);
}

hipError_t launch_synth(uint64_t seed, int64_t N, int64_t D, const double *wstar_dev, double *X, double *y, hipStream_t s)
{
    const int64_t total = N * D;
    int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(synth_x_kernel, dim3(blocks), dim3(256), 0, s, seed, total, X);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(synth_y_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, seed, N, D, X, wstar_dev, y);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Tableau preparation.  Input: G_aug ((M+2)^2, variables [features, intercept, y]).
//   1. regularisation, PartitionedLS.jl:108-123 in Gram form: G[a][b] += eta * #(groups containing both a and b)
//      (the K' extra rows sqrt(eta)*1_{group k}; y is padded with zeros so c and yy are unchanged);
//   2. free-intercept mode: eliminate the (always passive) intercept by a Schur complement;
//   3. scale to unit diagonal, s_i = 1/sqrt(B_ii) (sign constraints are invariant under positive scaling);
//   4. emit the full symmetric tableau  T = [[B~, c~], [c~', yy]]  ((n+1)^2, ld = n+1).
// ---------------------------------------------------------------------------------------------------------------------
__device__ static inline double reg_entry(const double *G, int ldg, int M, double eta, const uint64_t *mask_aug, int a, int b)
{
    double v = G[(size_t)a * ldg + b];
    if (eta != 0.0 && a <= M && b <= M) v += eta * (double)__popcll(mask_aug[a] & mask_aug[b]);
    return v;
}

__device__ static inline double base_entry(const double *G, int ldg, int M, double eta, const uint64_t *mask_aug,
                                           int free_intercept, const int *perm, int n, int i, int j)
{
    // tableau index -> augmented Gram index: variables perm[0..n-1] (grouped by partition) then the rhs (y) at index n
    const int a = (i < n) ? perm[i] : M + 1, b = (j < n) ? perm[j] : M + 1;
    double v = reg_entry(G, ldg, M, eta, mask_aug, a, b);
    if (free_intercept) {
        const double gII = reg_entry(G, ldg, M, eta, mask_aug, M, M);
        v -= reg_entry(G, ldg, M, eta, mask_aug, a, M) * reg_entry(G, ldg, M, eta, mask_aug, M, b) / gII;
    }
    return v;
}

__global__ void prep_scale_kernel(const double *__restrict__ G, int ldg, int M, double eta,
                                  const uint64_t *__restrict__ mask_aug, int free_intercept, const int *__restrict__ perm,
                                  int n, double *__restrict__ scale)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = base_entry(G, ldg, M, eta, mask_aug, free_intercept, perm, n, i, i);
    const double ref = G[(size_t)perm[i] * ldg + perm[i]];
    // a (numerically) null column carries no information: scale 0 keeps it out of every basis
    scale[i] = (d > 0.0 && d > 1e-14 * fabs(ref)) ? 1.0 / sqrt(d) : 0.0;
}

__global__ void prep_tableau_kernel(const double *__restrict__ G, int ldg, int M, double eta,
                                    const uint64_t *__restrict__ mask_aug, int free_intercept,
                                    const int *__restrict__ perm, int n,
                                    const double *__restrict__ scale, double *__restrict__ Tfull)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int ld = n + 1;
    if (idx >= ld * ld) return;
    const int i = idx / ld, j = idx % ld;
    const double si = (i < n) ? scale[i] : 1.0, sj = (j < n) ? scale[j] : 1.0;
    double v = base_entry(G, ldg, M, eta, mask_aug, free_intercept, perm, n, i, j) * si * sj;
    if (i == j && i < n && si == 0.0) v = 1.0;          // dead variable: harmless unit pivot, never selected
    Tfull[idx] = v;
}

hipError_t launch_prep(const double *G, int ldg, int M, double eta, const uint64_t *mask_aug, int free_intercept,
                       const int *perm, double *scale, double *Tfull, int n, hipStream_t s)
{
    hipLaunchKernelGGL(prep_scale_kernel, dim3((n + 255) / 256), dim3(256), 0, s, G, ldg, M, eta, mask_aug,
                       free_intercept, perm, n, scale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int tot = (n + 1) * (n + 1);
    hipLaunchKernelGGL(prep_tableau_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, G, ldg, M, eta, mask_aug,
                       free_intercept, perm, n, scale, Tfull);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Bit-order calibration of the Opt sweep (api.hip: calibrate_bit_order).  Bit b of the Gray index flips in 2^-(b+1) of all
// transitions, so which group sits on which bit decides how many variables the sweep exchanges: measured on C3, the same 2^20
// subproblems cost 16.9 M pivots / 74.9 ms in one order and 13.5 M / 52.2 ms in another.  The flip cost of a group is measured,
// not guessed: chain c starts from a pseudo-random pattern and flips the groups of its segment one after the other.
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ static inline uint64_t walk_base(int chain, int kbits)
{
    return sm64(0xB17C0DE5ULL + (uint64_t)chain) & ((kbits >= 64) ? ~0ULL : ((1ULL << kbits) - 1));
}
int walk_flipped_bit(int chain, int step, int kbits, int seg_len, int nseg)       // step >= 1 flips this group
{
    return ((chain % nseg) * seg_len + step - 1) % kbits;
}
// codes[(c * L + i) * n + v]: constraint of tableau variable v at step i of chain c (sign of its multiplier, Opt.jl:28-29)
__global__ void walk_codes_kernel(const uint64_t *__restrict__ mask, int n, int kbits, int L, int seg_len, int nseg,
                                  int8_t *__restrict__ codes)
{
    const int c = blockIdx.x / L, i = blockIdx.x - c * L;
    uint64_t pat = walk_base(c, kbits);
    for (int j = 1; j <= i; ++j) pat ^= 1ULL << (((c % nseg) * seg_len + j - 1) % kbits);
    for (int v = threadIdx.x; v < n; v += blockDim.x) {
        const uint64_t m = mask[v];
        const int f = 2 * __popcll(m & pat) - __popcll(m);
        codes[(size_t)blockIdx.x * n + v] = (int8_t)((f > 0) - (f < 0));
    }
}
hipError_t launch_walk_codes(const uint64_t *mask, int n, int kbits, int chains, int L, int seg_len, int nseg, int8_t *codes, hipStream_t s)
{
    hipLaunchKernelGGL(walk_codes_kernel, dim3(chains * L), dim3(256), 0, s, mask, n, kbits, L, seg_len, nseg, codes);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// BnB node batches (solvers.hip), both ends of a batch on the device so that only (pat, free) goes up and (bound, branch) comes back:
//   codes: constraint of tableau variable v in node i — its branched groups each contribute alpha >= 0 or <= 0 (the Sigma of
//          BnB.jl:120-121 accumulates them); none -> free (2), both kinds -> forced to 0 (BnB.jl:74-79);
//   nu:    lb = sqrt(objective^2) (BnB.jl:69-92) and branch = argmax_k nu_k, nu_k = sum_{i<j in group k} max(0, -w_i w_j) =
//          (sum of the positive w)(sum of the |negative w|) over the variables of group k (BnB.jl:42-57,107,117); only groups that
//          are still free can mix signs; -1 when every nu_k is 0: the relaxed solution is feasible (BnB.jl:109-115).
//          Lane k walks the variables in tableau order (fixed summation order: run-to-run reproducible).
// ---------------------------------------------------------------------------------------------------------------------
__global__ void bnb_codes_kernel(const uint64_t *__restrict__ mask_tab, int n, const uint64_t *__restrict__ pat,
                                 const uint64_t *__restrict__ free_, int8_t *__restrict__ codes)
{
    const uint64_t pt = pat[blockIdx.x], fr = free_[blockIdx.x];
    for (int v = threadIdx.x; v < n; v += blockDim.x) {
        const uint64_t br = mask_tab[v] & ~fr;
        const bool pos = (br & pt) != 0, neg = (br & ~pt) != 0;
        codes[(size_t)blockIdx.x * n + v] = (int8_t)(!br ? 2 : (pos && neg ? 0 : (pos ? 1 : -1)));
    }
}

// One workgroup of 256 threads per node: the unscaled solution and the group masks are staged in LDS by all threads (one coalesced
// pass), then lane k of wave 0 adds group k's positive and negative parts in VARIABLE ORDER from LDS — the summation order of round 3's
// kernel (bit-identical bounds and branching decisions), without its 257 dependent global-memory round trips per lane (43 us per
// batch whatever its size: a fifth of a 1024-node round once the snapshots stopped dominating it).
__global__ __launch_bounds__(256) void bnb_nu_kernel(const double *__restrict__ sol, const double *__restrict__ obj2, int n,
                                                     const double *__restrict__ scale, const uint64_t *__restrict__ mask_tab, int Kp,
                                                     const uint64_t *__restrict__ free_, double *__restrict__ lb, int *__restrict__ branch)
{
    __shared__ double sw[1024];
    __shared__ uint64_t sm[1024];
    __shared__ double nu[64];
    const int i = blockIdx.x, k = threadIdx.x;
    const uint64_t fr = free_[i];
    const double *w = sol + (size_t)i * n;
    for (int v = threadIdx.x; v < n; v += 256) { sw[v] = w[v] * scale[v]; sm[v] = mask_tab[v]; }
    __syncthreads();
    if (k < 64) {
        double pos = 0.0, neg = 0.0;
        if (k < Kp && ((fr >> k) & 1ULL)) {
            for (int v = 0; v < n; ++v) {
                if (!((sm[v] >> k) & 1ULL)) continue;
                const double wv = sw[v];
                if (wv > 0.0) pos += wv; else neg -= wv;
            }
        }
        nu[k] = pos * neg;
    }
    __syncthreads();
    if (k == 0) {
        int kbest = -1;
        double nubest = 0.0;
        for (int g = 0; g < Kp; ++g) if (nu[g] > nubest) { nubest = nu[g]; kbest = g; }     // argmax: first maximal index
        const double o2 = obj2[i];
        lb[i] = sqrt(o2 > 0.0 ? o2 : 0.0);
        branch[i] = kbest;
    }
}

// Gershgorin radii of the scaled Gram block (unit diagonal): out[i] = sum_{j != i, j < n} |T[i][j]|.  max_i out[i] = r proves
// lambda_min >= 1 - r and lambda_max <= 1 + r for the matrix every solve of the prepared problem works on (solvers.hip: fit(Alt) skips
// its extra pass over X when that bound alone guarantees the Gram form's accuracy).
__global__ __launch_bounds__(256) void gersh_kernel(const double *__restrict__ T, int n, int ld, double *__restrict__ out)
{
    __shared__ double red[256];
    const int i = blockIdx.x;
    double s = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) if (j != i) s += fabs(T[(size_t)i * ld + j]);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[i] = red[0];
}
hipError_t launch_gersh(const double *Tfull, int n, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(gersh_kernel, dim3(n), dim3(256), 0, s, Tfull, n, n + 1, out);
    return hipGetLastError();
}

hipError_t launch_bnb_codes(const uint64_t *mask_tab, int n, const uint64_t *pat, const uint64_t *free_, int cnt, int8_t *codes, hipStream_t s)
{
    hipLaunchKernelGGL(bnb_codes_kernel, dim3(cnt), dim3(256), 0, s, mask_tab, n, pat, free_, codes);
    return hipGetLastError();
}
hipError_t launch_bnb_nu(const double *sol, const double *obj2, int n, const double *scale, const uint64_t *mask_tab, int Kp,
                         const uint64_t *free_, int cnt, double *lb, int *branch, hipStream_t s)
{
    hipLaunchKernelGGL(bnb_nu_kernel, dim3(cnt), dim3(256), 0, s, sol, obj2, n, scale, mask_tab, Kp, free_, lb, branch);
    return hipGetLastError();
}

// all_opt leaves the sweep indexed by the INTERNAL pattern (group k on bit gbit[k]); the boundary indexes it by the reference's
__global__ void pattern_gather_kernel(const double *__restrict__ in, int64_t npat, int kbits, BitOrder order, double *__restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // reference pattern
    if (r >= npat) return;
    uint64_t q = 0;
    for (int k = 0; k < kbits; ++k) q |= (((uint64_t)r >> k) & 1ULL) << order.gbit[k];
    out[r] = in[q];
}
hipError_t launch_pattern_gather(const double *in, int64_t npat, int kbits, const BitOrder &order, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(pattern_gather_kernel, dim3((unsigned)((npat + 255) / 256)), dim3(256), 0, s, in, npat, kbits, order, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// beta-step system of fit(Alt) in Gram form (Alt.jl:109-110):  H = A' Greg A,  g = A' c  with  A = Po o alpha  ((M+1) x K').
// Two tiny kernels with fixed summation orders (run-to-run reproducible, no atomics):
//   1. GA[m][k2] = sum_{m2 in group k2} Greg[m][m2] alpha_m2          (one workgroup per row m, lane = k2)
//   2. H[k][k2]  = sum_{m in group k} alpha_m GA[m][k2],   g[k] = sum_{m in group k} alpha_m Greg[m][y]
// On the host this was an M^2 loop per Alt iteration (0.3 ms at D = 512, a quarter of an iteration).
// ---------------------------------------------------------------------------------------------------------------------
// both kernels: lane = group index (K' <= 64), the 16 waves of the workgroup take the variables m = wave, wave + 16, ...; the 16
// partial sums of a lane are added in wave order by wave 0 (fixed order).  (One thread walking all M + 1 variables: 75 us per
// kernel at D = 512 — latency of 513 dependent steps.)
static constexpr int ALT_WAVES = 16;

__global__ __launch_bounds__(64 * ALT_WAVES) void alt_ga_kernel(const double *__restrict__ G, int ldg, int M, double eta,
                                                               const uint64_t *__restrict__ mask_aug, const double *__restrict__ a, int Kp,
                                                               double *__restrict__ GA)
{
    __shared__ double part[ALT_WAVES][64];
    const int m = blockIdx.x, k2 = threadIdx.x & 63, wave = threadIdx.x >> 6, Mp = M + 1;
    // branch-free and unrolled: the loads of a step are wave-uniform (row m of G, alpha and the mask of variable m2), eight steps
    // are in flight together; a lane adds the product when variable m2 belongs to ITS group
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    double s = 0.0;
    int m2 = wu;
    for (; m2 + 7 * ALT_WAVES < Mp; m2 += 8 * ALT_WAVES) {
        double v[8];
        uint64_t mk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int mm = m2 + u * ALT_WAVES;
            mk[u] = mask_aug[mm];
            v[u] = reg_entry(G, ldg, M, eta, mask_aug, m, mm) * a[mm];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += ((mk[u] >> k2) & 1ULL) ? v[u] : 0.0;
    }
    for (; m2 < Mp; m2 += ALT_WAVES) s += ((mask_aug[m2] >> k2) & 1ULL) ? reg_entry(G, ldg, M, eta, mask_aug, m, m2) * a[m2] : 0.0;
    if (k2 >= Kp) s = 0.0;
    part[wave][k2] = s;
    __syncthreads();
    if (wave == 0 && k2 < Kp) {
        double t = 0.0;
        for (int w = 0; w < ALT_WAVES; ++w) t += part[w][k2];
        GA[(size_t)m * Kp + k2] = t;
    }
}

__global__ __launch_bounds__(64 * ALT_WAVES) void alt_h_kernel(const double *__restrict__ G, int ldg, int M, double eta,
                                                              const uint64_t *__restrict__ mask_aug, const double *__restrict__ a, int Kp,
                                                              const double *__restrict__ GA, double *__restrict__ Hg)
{
    __shared__ double part[ALT_WAVES][65];
    const int k = blockIdx.x, k2 = threadIdx.x & 63, wave = threadIdx.x >> 6, Mp = M + 1;     // column Kp of row k holds g[k] (Kp <= 64: lane Kp <= 64 ... handled by lane 63 when Kp == 64)
    // lanes 0..Kp-1: H[k][k2]; the g column is computed by lane Kp when Kp < 64, else by a second pass of lane 0
    auto term = [&](int m, int col) { return col < Kp ? GA[(size_t)m * Kp + col] : reg_entry(G, ldg, M, eta, mask_aug, m, M + 1); };
    double s = 0.0, sg = 0.0;
    for (int m = wave; m < Mp; m += ALT_WAVES) {
        if (!((mask_aug[m] >> k) & 1ULL)) continue;
        const double am = a[m];
        if (k2 < Kp) s = fma(am, term(m, k2), s);
        if (k2 == 0) sg = fma(am, term(m, Kp), sg);
    }
    part[wave][k2] = s;
    if (k2 == 0) part[wave][64] = sg;
    __syncthreads();
    if (wave == 0) {
        if (k2 < Kp) {
            double t = 0.0;
            for (int w = 0; w < ALT_WAVES; ++w) t += part[w][k2];
            Hg[(size_t)k * (Kp + 1) + k2] = t;
        }
        if (k2 == 0) {
            double t = 0.0;
            for (int w = 0; w < ALT_WAVES; ++w) t += part[w][64];
            Hg[(size_t)k * (Kp + 1) + Kp] = t;
        }
    }
}

hipError_t launch_alt_beta_system(const double *G, int ldg, int M, double eta, const uint64_t *mask_aug, const double *a, int Kp,
                                  double *GA, double *Hg, hipStream_t s)
{
    hipLaunchKernelGGL(alt_ga_kernel, dim3(M + 1), dim3(64 * ALT_WAVES), 0, s, G, ldg, M, eta, mask_aug, a, Kp, GA);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(alt_h_kernel, dim3(Kp), dim3(64 * ALT_WAVES), 0, s, G, ldg, M, eta, mask_aug, a, Kp, GA, Hg);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Residual from the data (Opt.jl:90 / predict PartitionedLS.jl:132-134): one coalesced pass over column-major X.
//   partial[b] = sum over the block's rows of (sum_m X[i,m] w[m] + t - y[i])^2 ;  yhat (optional) = X w + t
// Blocks are summed on the host in index order (reproducible).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void residual_kernel(const double *__restrict__ X, int64_t N, int64_t M, int64_t ldX,
                                                       const double *__restrict__ y, const double *__restrict__ w, double t,
                                                       double *__restrict__ partial, double *__restrict__ yhat)
{
    extern __shared__ double sw[];                       // w staged once per block
    for (int64_t m = threadIdx.x; m < M; m += blockDim.x) sw[m] = w[m];
    __syncthreads();
    double acc2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int64_t m = 0;
        for (; m + 3 < M; m += 4) {
            a0 = fma(X[i + m * ldX], sw[m], a0);
            a1 = fma(X[i + (m + 1) * ldX], sw[m + 1], a1);
            a2 = fma(X[i + (m + 2) * ldX], sw[m + 2], a2);
            a3 = fma(X[i + (m + 3) * ldX], sw[m + 3], a3);
        }
        for (; m < M; ++m) a0 = fma(X[i + m * ldX], sw[m], a0);
        const double p = ((a0 + a1) + (a2 + a3)) + t;
        if (yhat) yhat[i] = p;
        if (y) { const double r = p - y[i]; acc2 = fma(r, r, acc2); }
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && partial) partial[blockIdx.x] = red[0];
}

hipError_t launch_residual(const double *X, int64_t N, int64_t M, int64_t ldX, const double *y, const double *w, double t,
                           double *partial, int nblocks, double *yhat, hipStream_t s)
{
    hipLaunchKernelGGL(residual_kernel, dim3(nblocks), dim3(256), (size_t)M * sizeof(double), s, X, N, M, ldX, y, w, t,
                       partial, yhat);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// g = Xo' (y - yhat) for iterative refinement of one solution in data space (corrected semi-normal equations).
// Grid (column, row slice): workgroup (m, r) reduces rows [r0, r1) of column m (the last column is the ones column) with four
// loads of X in flight per thread, coalesced along the column, fixed-order block reduction; the XTR_R partial sums of a column are
// added in slice order by the caller (reproducible).  One workgroup per whole column (round 1) left each CU with a handful of loads
// in flight: 100 us for 205 MB at C3.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xtr_kernel(const double *__restrict__ X, int64_t N, int64_t M, int64_t ldX,
                                                  const double *__restrict__ y, const double *__restrict__ yhat,
                                                  double *__restrict__ gpart, int R)
{
    const int64_t m = blockIdx.x;
    const int r = blockIdx.y;
    const int64_t rows = (N + R - 1) / R, r0 = (int64_t)r * rows, r1 = (r0 + rows < N) ? r0 + rows : N;
    const double *col = X + m * ldX;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int64_t i = r0 + threadIdx.x;
    if (m < M) {
        for (; i + 768 < r1; i += 1024) {
            double x[4], d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { x[u] = col[i + 256 * u]; d[u] = y[i + 256 * u] - yhat[i + 256 * u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = fma(x[u], d[u], acc[u]);
        }
        for (; i < r1; i += 256) acc[0] = fma(col[i], y[i] - yhat[i], acc[0]);
    } else {
        for (; i < r1; i += 256) acc[0] += y[i] - yhat[i];
    }
    __shared__ double red[256];
    red[threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) gpart[(size_t)r * (M + 1) + m] = red[0];
}

int xtr_slices(int64_t N)
{
    int R = (int)((N + 16383) / 16384);
    if (R < 1) R = 1;
    if (R > 16) R = 16;
    return R;
}

hipError_t launch_xtr(const double *X, int64_t N, int64_t M, int64_t ldX, const double *y, const double *yhat, double *gpart,
                      hipStream_t s)
{
    const int R = xtr_slices(N);
    hipLaunchKernelGGL(xtr_kernel, dim3((unsigned)(M + 1), (unsigned)R), dim3(256), 0, s, X, N, M, ldX, y, yhat, gpart, R);
    return hipGetLastError();
}

}  // namespace partls
