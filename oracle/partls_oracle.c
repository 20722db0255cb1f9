/*
 * partls_oracle.c — CPU oracle (TEST INFRASTRUCTURE ONLY; see partls_oracle.h for scope, citations and pinning status).
 * Plain C99, no dependencies beyond libm.  Column-major everywhere.
 */
#include "partls_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define AT(A, ld, i, j) ((A)[(size_t)(i) + (size_t)(j) * (size_t)(ld)])

/* ------------------------------------------------------------------------------------------------
 * Lawson–Hanson NNLS (Lawson & Hanson 1974, ch. 23, algorithm NNLS) — what nonneg_lsq(...; alg=:nnls)
 * computes at Opt.jl:89 / Alt.jl:90 / BnB.jl:82.  A is triangularised in place by Householder
 * reflections as columns enter the passive set and re-triangularised by Givens rotations when they leave.
 * ---------------------------------------------------------------------------------------------- */

/* Build the reflection that maps v[p], v[p+1..m-1] onto (-sign(v[p])*norm, 0..0).  Returns "up"; v[p] is replaced. */
static double house_make(double *v, int64_t p, int64_t m)
{
    double cl = fabs(v[p]);
    for (int64_t i = p + 1; i < m; ++i) { double a = fabs(v[i]); if (a > cl) cl = a; }
    if (cl <= 0.0) return 0.0;
    double clinv = 1.0 / cl, sm = (v[p] * clinv) * (v[p] * clinv);
    for (int64_t i = p + 1; i < m; ++i) { double s = v[i] * clinv; sm += s * s; }
    cl *= sqrt(sm);
    if (v[p] > 0.0) cl = -cl;
    double up = v[p] - cl;
    v[p] = cl;
    return up;
}

/* Apply the reflection defined by (u[p] stored as "up", u[p+1..m-1] in place, pivot value piv = u_new[p]) to c. */
static void house_apply(const double *u, double up, double piv, int64_t p, int64_t m, double *c)
{
    double b = up * piv;
    if (b >= 0.0) return;
    b = 1.0 / b;
    double sm = c[p] * up;
    for (int64_t i = p + 1; i < m; ++i) sm += c[i] * u[i];
    if (sm == 0.0) return;
    sm *= b;
    c[p] += sm * up;
    for (int64_t i = p + 1; i < m; ++i) c[i] += sm * u[i];
}

static void givens_make(double a, double b, double *c, double *s, double *sig)
{
    if (fabs(a) > fabs(b)) {
        double xr = b / a, yr = sqrt(1.0 + xr * xr);
        *c = copysign(1.0 / yr, a); *s = (*c) * xr; *sig = fabs(a) * yr;
    } else if (b != 0.0) {
        double xr = a / b, yr = sqrt(1.0 + xr * xr);
        *s = copysign(1.0 / yr, b); *c = (*s) * xr; *sig = fabs(b) * yr;
    } else { *sig = 0.0; *c = 0.0; *s = 1.0; }
}

/* back substitution on the triangular passive block; zz holds the rhs on entry and the solution on exit */
static void solve_tri(const double *A, int64_t m, const int64_t *index, int64_t nsetp, double *zz)
{
    for (int64_t ip = nsetp - 1; ip >= 0; --ip) {
        if (ip != nsetp - 1) {
            int64_t jn = index[ip + 1];
            for (int64_t ii = 0; ii <= ip; ++ii) zz[ii] -= AT(A, m, ii, jn) * zz[ip + 1];
        }
        zz[ip] /= AT(A, m, ip, index[ip]);
    }
}

int oracle_nnls(double *A, int64_t m, int64_t n, double *b, double *x, double *rnorm,
                double *w, double *zz, int64_t *index, int64_t *nsetp_out)
{
    const double factor = 0.01;
    int mode = 0;
    int64_t iter = 0, itmax = 3 * n;
    for (int64_t i = 0; i < n; ++i) { x[i] = 0.0; index[i] = i; w[i] = 0.0; }
    int64_t iz1 = 0, iz2 = n - 1, nsetp = 0, npp1 = 0;

    while (iz1 <= iz2 && nsetp < m) {
        /* dual vector w = A'(b - Ax) restricted to the active (zero) set */
        for (int64_t iz = iz1; iz <= iz2; ++iz) {
            int64_t j = index[iz];
            double sm = 0.0;
            for (int64_t l = npp1; l < m; ++l) sm += AT(A, m, l, j) * b[l];
            w[j] = sm;
        }
        int64_t izmax = -1, j = -1;
        double up = 0.0;
        int found = 0;
        for (;;) {
            double wmax = 0.0;
            for (int64_t iz = iz1; iz <= iz2; ++iz) {
                int64_t jj = index[iz];
                if (w[jj] > wmax) { wmax = w[jj]; izmax = iz; }
            }
            if (wmax <= 0.0) break;                       /* KKT satisfied: terminate */
            j = index[izmax];
            /* candidate column: reflect rows npp1.. and test independence + sign of the new coefficient */
            double asave = AT(A, m, npp1, j);
            up = house_make(&AT(A, m, 0, j), npp1, m);
            double unorm = 0.0;
            for (int64_t l = 0; l < nsetp; ++l) unorm += AT(A, m, l, j) * AT(A, m, l, j);
            unorm = sqrt(unorm);
            /* Independence test.  The classic test (unorm + |a| * factor) - unorm > 0 admits a column that is dependent on
             * the passive set up to round-off; the triangular factor is then singular and the outcome depends on rounding
             * (it differs between this C code, the Fortran original and the Julia port the reference uses, and is not the
             * NNLS optimum).  The oracle therefore ALSO rejects a candidate whose component outside the passive space is
             * below sqrt(1e-11) of its norm — the same rule as the HIP path (entering pivot <= 1e-11 on the unit-diagonal
             * Gram tableau) — which makes it return the true optimum on rank-deficient data (pinned against
             * scipy.optimize.nnls in tests/test_oracle.py).  On full-rank passive sets both tests agree. */
            const double adiag = AT(A, m, npp1, j);
            if (adiag * adiag > 1e-11 * (unorm * unorm + adiag * adiag) && (unorm + fabs(adiag) * factor) - unorm > 0.0) {
                memcpy(zz, b, (size_t)m * sizeof(double));
                house_apply(&AT(A, m, 0, j), up, AT(A, m, npp1, j), npp1, m, zz);
                double ztest = zz[npp1] / AT(A, m, npp1, j);
                if (ztest > 0.0) { found = 1; break; }
            }
            AT(A, m, npp1, j) = asave;                    /* reject this column */
            w[j] = 0.0;
        }
        if (!found) break;

        /* move j from the zero set to the passive set */
        memcpy(b, zz, (size_t)m * sizeof(double));
        index[izmax] = index[iz1];
        index[iz1] = j;
        ++iz1;
        nsetp = npp1 + 1;
        ++npp1;
        for (int64_t jz = iz1; jz <= iz2; ++jz) {
            int64_t jj = index[jz];
            house_apply(&AT(A, m, 0, j), up, AT(A, m, nsetp - 1, j), nsetp - 1, m, &AT(A, m, 0, jj));
        }
        for (int64_t l = nsetp; l < m; ++l) AT(A, m, l, j) = 0.0;
        w[j] = 0.0;

        memcpy(zz, b, (size_t)m * sizeof(double));
        solve_tri(A, m, index, nsetp, zz);

        /* inner loop: step back along the segment until the passive solution is feasible */
        for (;;) {
            if (++iter > itmax) { mode = 3; goto done; }
            double alpha = 2.0;
            int64_t jj = -1;
            for (int64_t ip = 0; ip < nsetp; ++ip) {
                int64_t l = index[ip];
                if (zz[ip] <= 0.0) {
                    double t = -x[l] / (zz[ip] - x[l]);
                    if (alpha > t) { alpha = t; jj = ip; }
                }
            }
            if (alpha == 2.0) break;
            for (int64_t ip = 0; ip < nsetp; ++ip) {
                int64_t l = index[ip];
                x[l] += alpha * (zz[ip] - x[l]);
            }
            int64_t i = index[jj];
            for (;;) {
                x[i] = 0.0;
                if (jj != nsetp - 1) {
                    for (int64_t jn = jj + 1; jn < nsetp; ++jn) {
                        int64_t ii = index[jn];
                        index[jn - 1] = ii;
                        double cc, ss, sig;
                        givens_make(AT(A, m, jn - 1, ii), AT(A, m, jn, ii), &cc, &ss, &sig);
                        AT(A, m, jn - 1, ii) = sig;
                        AT(A, m, jn, ii) = 0.0;
                        for (int64_t l = 0; l < n; ++l) {
                            if (l == ii) continue;
                            double t1 = AT(A, m, jn - 1, l), t2 = AT(A, m, jn, l);
                            AT(A, m, jn - 1, l) = cc * t1 + ss * t2;
                            AT(A, m, jn, l) = -ss * t1 + cc * t2;
                        }
                        double t1 = b[jn - 1], t2 = b[jn];
                        b[jn - 1] = cc * t1 + ss * t2;
                        b[jn] = -ss * t1 + cc * t2;
                    }
                }
                npp1 = nsetp - 1;
                --nsetp;
                --iz1;
                index[iz1] = i;
                /* round-off may have left other passive coefficients non-positive: remove them too */
                int again = 0;
                for (jj = 0; jj < nsetp; ++jj) {
                    i = index[jj];
                    if (x[i] <= 0.0) { again = 1; break; }
                }
                if (!again) break;
            }
            memcpy(zz, b, (size_t)m * sizeof(double));
            solve_tri(A, m, index, nsetp, zz);
        }
        for (int64_t ip = 0; ip < nsetp; ++ip) x[index[ip]] = zz[ip];
    }
done:;
    double sm = 0.0;
    for (int64_t l = npp1; l < m; ++l) sm += b[l] * b[l];
    if (npp1 >= m) sm = 0.0;
    *rnorm = sqrt(sm);
    if (nsetp_out) *nsetp_out = nsetp;
    return mode;
}

/* ------------------------------------------------------------------------------------------------
 * Problem rewriting — PartitionedLS.jl:76-81, :108-123
 * ---------------------------------------------------------------------------------------------- */
void oracle_homogeneous(const double *X, int64_t N, int64_t M, const int64_t *P, int64_t K, double *Xo, int64_t *Po)
{
    int64_t Mp = M + 1, Kp = K + 1;
    memcpy(Xo, X, (size_t)N * (size_t)M * sizeof(double));
    for (int64_t i = 0; i < N; ++i) AT(Xo, N, i, M) = 1.0;
    for (int64_t k = 0; k < Kp; ++k)
        for (int64_t m = 0; m < Mp; ++m)
            AT(Po, Mp, m, k) = (m < M && k < K) ? AT(P, M, m, k) : ((m == M && k == K) ? 1 : 0);
}

int64_t oracle_regularize(const double *Xo, int64_t N, int64_t Mp, const double *y, const int64_t *Po, int64_t Kp,
                          double eta, double *Xn, double *yn)
{
    int64_t rows = (eta == 0.0) ? N : N + Kp;
    for (int64_t j = 0; j < Mp; ++j) {
        memcpy(&AT(Xn, rows, 0, j), &AT(Xo, N, 0, j), (size_t)N * sizeof(double));
        if (eta != 0.0)
            for (int64_t k = 0; k < Kp; ++k) AT(Xn, rows, N + k, j) = (AT(Po, Mp, j, k) == 1) ? sqrt(eta) : 0.0;
    }
    memcpy(yn, y, (size_t)N * sizeof(double));
    for (int64_t k = N; k < rows; ++k) yn[k] = 0.0;
    return rows;
}

/* Opt.jl:4-20 — beta_k = 2*bit_{k}(b) - 1, least significant bit first */
static void index_to_beta(int64_t b, int64_t Kp, double *beta)
{
    for (int64_t k = 0; k < Kp; ++k) { beta[k] = 2.0 * (double)(b % 2) - 1.0; b >>= 1; }
}

/* Opt.jl:22-31 — featuremul_m = sum_k P[m,k]*beta_k */
static void feature_mul(const int64_t *Po, int64_t Mp, int64_t Kp, const double *beta, double *f)
{
    for (int64_t m = 0; m < Mp; ++m) {
        double s = 0.0;
        for (int64_t k = 0; k < Kp; ++k) s += (double)AT(Po, Mp, m, k) * beta[k];
        f[m] = s;
    }
}

/* norm(Xo*(Po.*a)*b - yo) — Opt.jl:90, Alt.jl:67 */
static double loss(const double *Xo, int64_t rows, int64_t Mp, const double *yo, const int64_t *Po, int64_t Kp,
                   const double *a, const double *b, double *tmp_f, double *tmp_r)
{
    feature_mul(Po, Mp, Kp, b, tmp_f);
    for (int64_t i = 0; i < rows; ++i) tmp_r[i] = -yo[i];
    for (int64_t m = 0; m < Mp; ++m) {
        double wm = a[m] * tmp_f[m];
        if (wm == 0.0) continue;
        const double *col = &AT(Xo, rows, 0, m);
        for (int64_t i = 0; i < rows; ++i) tmp_r[i] += col[i] * wm;
    }
    double s = 0.0;
    for (int64_t i = 0; i < rows; ++i) s += tmp_r[i] * tmp_r[i];
    return sqrt(s);
}

/* Opt.jl:34-44 */
static void cleanup_opt(const int64_t *P, int64_t M, int64_t K, const double *a_raw, const double *b_raw,
                        double *alpha, double *beta)
{
    for (int64_t k = 0; k < K; ++k) {
        double A = 0.0;
        for (int64_t m = 0; m < M; ++m) A += (double)AT(P, M, m, k) * a_raw[m];
        beta[k] = b_raw[k] * A;
    }
    for (int64_t m = 0; m < M; ++m) {
        double s = 0.0;
        for (int64_t k = 0; k < K; ++k) {
            double A = 0.0;
            for (int64_t mm = 0; mm < M; ++mm) A += (double)AT(P, M, mm, k) * a_raw[mm];
            if (A == 0.0) A = 1.0;
            s += (double)AT(P, M, m, k) * a_raw[m] / A;
        }
        alpha[m] = s;
    }
}

typedef struct {
    double *Xb, *yb, *w, *zz, *f, *r, *x, *beta;
    int64_t *index;
} opt_ws;

static int ws_alloc(opt_ws *s, int64_t rows, int64_t Mp, int64_t Kp)
{
    s->Xb = malloc((size_t)rows * (size_t)Mp * sizeof(double));
    s->yb = malloc((size_t)rows * sizeof(double));
    s->w = malloc((size_t)Mp * sizeof(double));
    s->zz = malloc((size_t)rows * sizeof(double));
    s->f = malloc((size_t)Mp * sizeof(double));
    s->r = malloc((size_t)rows * sizeof(double));
    s->x = malloc((size_t)Mp * sizeof(double));
    s->beta = malloc((size_t)Kp * sizeof(double));
    s->index = malloc((size_t)Mp * sizeof(int64_t));
    return (s->Xb && s->yb && s->w && s->zz && s->f && s->r && s->x && s->beta && s->index) ? 0 : -1;
}
static void ws_free(opt_ws *s)
{
    free(s->Xb); free(s->yb); free(s->w); free(s->zz); free(s->f); free(s->r); free(s->x); free(s->beta); free(s->index);
}

/* one trip of the loop Opt.jl:87-90: returns optval, leaves raw alpha in s->x and beta in s->beta */
static double opt_one_pattern(const double *Xo, int64_t rows, int64_t Mp, const double *yo, const int64_t *Po, int64_t Kp,
                              int64_t b, opt_ws *s)
{
    index_to_beta(b, Kp, s->beta);
    feature_mul(Po, Mp, Kp, s->beta, s->f);
    for (int64_t m = 0; m < Mp; ++m) {                      /* bmatrix: X .* featuremul' */
        const double *src = &AT(Xo, rows, 0, m);
        double *dst = &AT(s->Xb, rows, 0, m), fm = s->f[m];
        for (int64_t i = 0; i < rows; ++i) dst[i] = src[i] * fm;
    }
    memcpy(s->yb, yo, (size_t)rows * sizeof(double));
    double rn;
    oracle_nnls(s->Xb, rows, Mp, s->yb, s->x, &rn, s->w, s->zz, s->index, NULL);
    return loss(Xo, rows, Mp, yo, Po, Kp, s->x, s->beta, s->f, s->r);
}

int oracle_opt_patterns(const double *Xo, int64_t rows, int64_t Mp, const double *yo, const int64_t *Po, int64_t Kp,
                        const int64_t *patterns, int64_t npat, double *objs, double *raw_alpha)
{
    opt_ws s;
    if (ws_alloc(&s, rows, Mp, Kp)) { ws_free(&s); return -1; }
    for (int64_t i = 0; i < npat; ++i) {
        objs[i] = opt_one_pattern(Xo, rows, Mp, yo, Po, Kp, patterns[i], &s);
        if (raw_alpha) memcpy(raw_alpha + (size_t)i * (size_t)Mp, s.x, (size_t)Mp * sizeof(double));
    }
    ws_free(&s);
    return 0;
}

int oracle_fit_opt(const double *X, int64_t N, int64_t M, const double *y, const int64_t *P, int64_t K, double eta,
                   double *alpha, double *beta, double *t, double *opt, int64_t *best_index,
                   double *all_opt, double *all_alpha, double *all_beta, double *all_t)
{
    int64_t Mp = M + 1, Kp = K + 1;
    if (Kp > 40) return -2;
    double *Xo = malloc((size_t)N * (size_t)Mp * sizeof(double));
    int64_t *Po = malloc((size_t)Mp * (size_t)Kp * sizeof(int64_t));
    double *Xn = malloc((size_t)(N + Kp) * (size_t)Mp * sizeof(double));
    double *yn = malloc((size_t)(N + Kp) * sizeof(double));
    double *best_a = malloc((size_t)Mp * sizeof(double)), *best_b = malloc((size_t)Kp * sizeof(double));
    opt_ws s;
    int rc = 0;
    if (!Xo || !Po || !Xn || !yn || !best_a || !best_b) { rc = -1; goto out0; }
    oracle_homogeneous(X, N, M, P, K, Xo, Po);
    int64_t rows = oracle_regularize(Xo, N, Mp, y, Po, Kp, eta, Xn, yn);
    if (ws_alloc(&s, rows, Mp, Kp)) { rc = -1; goto out1; }

    double best = INFINITY;
    int64_t bi = -1, npat = (int64_t)1 << Kp;
    for (int64_t b = 0; b < npat; ++b) {
        double ov = opt_one_pattern(Xn, rows, Mp, yn, Po, Kp, b, &s);
        if (all_opt) all_opt[b] = ov;
        if (all_alpha && all_beta && all_t) {
            cleanup_opt(P, M, K, s.x, s.beta, all_alpha + (size_t)b * (size_t)M, all_beta + (size_t)b * (size_t)K);
            all_t[b] = s.beta[K] * s.x[M];
        }
        if (bi < 0 || ov < best) {                          /* argmin: first minimal index (Opt.jl:96) */
            best = ov; bi = b;
            memcpy(best_a, s.x, (size_t)Mp * sizeof(double));
            memcpy(best_b, s.beta, (size_t)Kp * sizeof(double));
        }
    }
    cleanup_opt(P, M, K, best_a, best_b, alpha, beta);
    *t = best_b[K] * best_a[M];                             /* Opt.jl:92 */
    *opt = best;
    if (best_index) *best_index = bi;
out1:
    ws_free(&s);
out0:
    free(Xo); free(Po); free(Xn); free(yn); free(best_a); free(best_b);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Householder QR helpers (compress, and the beta-step  Xoα \ yo  of Alt.jl:110)
 * ---------------------------------------------------------------------------------------------- */
/* In-place QR of A (m x n, m >= n) applied to rhs b as well; afterwards R is the upper triangle, b = Q'b. */
static void qr_inplace(double *A, int64_t m, int64_t n, double *b)
{
    for (int64_t j = 0; j < n && j < m; ++j) {
        double up = house_make(&AT(A, m, 0, j), j, m);
        double piv = AT(A, m, j, j);
        for (int64_t l = j + 1; l < n; ++l) house_apply(&AT(A, m, 0, j), up, piv, j, m, &AT(A, m, 0, l));
        if (b) house_apply(&AT(A, m, 0, j), up, piv, j, m, b);
        for (int64_t i = j + 1; i < m; ++i) AT(A, m, i, j) = 0.0;
    }
}

int oracle_compress(const double *Xo, int64_t rows, int64_t Mp, const double *yo, double *R, double *z)
{
    if (rows < Mp + 1) return -2;
    double *A = malloc((size_t)rows * (size_t)Mp * sizeof(double)), *b = malloc((size_t)rows * sizeof(double));
    if (!A || !b) { free(A); free(b); return -1; }
    memcpy(A, Xo, (size_t)rows * (size_t)Mp * sizeof(double));
    memcpy(b, yo, (size_t)rows * sizeof(double));
    qr_inplace(A, rows, Mp, b);
    int64_t ldr = Mp + 1;
    for (int64_t j = 0; j < Mp; ++j)
        for (int64_t i = 0; i < ldr; ++i) AT(R, ldr, i, j) = (i <= j) ? AT(A, rows, i, j) : 0.0;
    double tail = 0.0;
    for (int64_t i = Mp; i < rows; ++i) tail += b[i] * b[i];
    for (int64_t i = 0; i < Mp; ++i) z[i] = b[i];
    z[Mp] = sqrt(tail);
    free(A); free(b);
    return 0;
}

/* least squares  A \ b  for full-column-rank A (m x n); A, b overwritten; x[n] */
static void lstsq(double *A, int64_t m, int64_t n, double *b, double *x)
{
    qr_inplace(A, m, n, b);
    for (int64_t i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int64_t j = i + 1; j < n; ++j) s -= AT(A, m, i, j) * x[j];
        x[i] = s / AT(A, m, i, i);
    }
}

/* ------------------------------------------------------------------------------------------------
 * fit(Alt) — Alt.jl:50-124
 * ---------------------------------------------------------------------------------------------- */
int oracle_fit_alt(const double *X, int64_t N, int64_t M, const double *y, const int64_t *P, int64_t K, double eta,
                   double eps, int64_t T, const double *alpha0, const double *beta0,
                   double *alpha, double *beta, double *t, double *opt, int64_t *iters)
{
    int64_t Mp = M + 1, Kp = K + 1;
    double *Xo = malloc((size_t)N * (size_t)Mp * sizeof(double));
    int64_t *Po = malloc((size_t)Mp * (size_t)Kp * sizeof(int64_t));
    double *Xn = malloc((size_t)(N + Kp) * (size_t)Mp * sizeof(double));
    double *yn = malloc((size_t)(N + Kp) * sizeof(double));
    double *a = malloc((size_t)Mp * sizeof(double)), *b = malloc((size_t)Kp * sizeof(double));
    double *Xa = malloc((size_t)(N + Kp) * (size_t)Kp * sizeof(double)), *yb2 = malloc((size_t)(N + Kp) * sizeof(double));
    opt_ws s;
    int rc = 0;
    if (!Xo || !Po || !Xn || !yn || !a || !b || !Xa || !yb2) { rc = -1; goto out0; }
    oracle_homogeneous(X, N, M, P, K, Xo, Po);
    int64_t rows = oracle_regularize(Xo, N, Mp, y, Po, Kp, eta, Xn, yn);
    if (ws_alloc(&s, rows, Mp, Kp)) { rc = -1; goto out1; }
    memcpy(a, alpha0, (size_t)Mp * sizeof(double));
    memcpy(b, beta0, (size_t)Kp * sizeof(double));

    double oldopt = 1e20, optval = 1e10;                   /* Alt.jl:73-74 */
    int64_t i = 1;
    while (i <= T && fabs(oldopt - optval) > eps * oldopt) {
        /* alpha-step: NNLS on Xo .* (Po*beta)'  (Alt.jl:80-90; the reference passes y, which equals yo when eta == 0) */
        feature_mul(Po, Mp, Kp, b, s.f);
        for (int64_t m = 0; m < Mp; ++m) {
            const double *src = &AT(Xn, rows, 0, m);
            double *dst = &AT(s.Xb, rows, 0, m), fm = s.f[m];
            for (int64_t r = 0; r < rows; ++r) dst[r] = src[r] * fm;
        }
        memcpy(s.yb, yn, (size_t)rows * sizeof(double));
        double rn;
        oracle_nnls(s.Xb, rows, Mp, s.yb, a, &rn, s.w, s.zz, s.index, NULL);
        /* checkalpha (Alt.jl:5-20): a group whose alphas sum to exactly 0 becomes uniform */
        /* suma and sumP are taken BEFORE any group is rewritten (Alt.jl:6-7): it matters when groups overlap */
        for (int64_t k = 0; k < Kp; ++k) {
            double suma = 0.0;
            for (int64_t m = 0; m < Mp; ++m) suma += (double)AT(Po, Mp, m, k) * a[m];
            s.beta[k] = suma;
        }
        for (int64_t k = 0; k < Kp; ++k) {
            int64_t sumP = 0;
            for (int64_t m = 0; m < Mp; ++m) sumP += AT(Po, Mp, m, k);
            if (s.beta[k] == 0.0)
                for (int64_t m = 0; m < Mp; ++m) if (AT(Po, Mp, m, k) == 1) a[m] = 1.0 / (double)sumP;
        }
        /* renormalise (Alt.jl:95-98) */
        for (int64_t k = 0; k < Kp; ++k) {
            double sa = 0.0;
            for (int64_t m = 0; m < Mp; ++m) sa += (double)AT(Po, Mp, m, k) * a[m];
            s.beta[k] = sa;                                /* sumα */
        }
        for (int64_t m = 0; m < Mp; ++m) {
            double pa = 0.0;
            for (int64_t k = 0; k < Kp; ++k) pa += (double)AT(Po, Mp, m, k) * s.beta[k];
            a[m] = a[m] / pa;
        }
        for (int64_t k = 0; k < Kp; ++k) b[k] *= s.beta[k];
        /* beta-step: (Xo*(Po.*alpha)) \ yo  (Alt.jl:109-110) */
        for (int64_t k = 0; k < Kp; ++k) {
            double *dst = &AT(Xa, rows, 0, k);
            for (int64_t r = 0; r < rows; ++r) dst[r] = 0.0;
            for (int64_t m = 0; m < Mp; ++m) {
                double c = (double)AT(Po, Mp, m, k) * a[m];
                if (c == 0.0) continue;
                const double *src = &AT(Xn, rows, 0, m);
                for (int64_t r = 0; r < rows; ++r) dst[r] += src[r] * c;
            }
        }
        memcpy(yb2, yn, (size_t)rows * sizeof(double));
        lstsq(Xa, rows, Kp, yb2, b);
        oldopt = optval;
        optval = loss(Xn, rows, Mp, yn, Po, Kp, a, b, s.f, s.r);
        ++i;
    }
    for (int64_t m = 0; m < M; ++m) alpha[m] = a[m];
    for (int64_t k = 0; k < K; ++k) beta[k] = b[k];
    *t = b[K] * a[M];                                      /* Alt.jl:119 */
    *opt = optval;
    if (iters) *iters = i - 1;
out1:
    ws_free(&s);
out0:
    free(Xo); free(Po); free(Xn); free(yn); free(a); free(b); free(Xa); free(yb2);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * fit(BnB) — BnB.jl:30-132
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const double *Xo, *yo;
    const int64_t *Po;
    int64_t rows, Mp, Kp;
    double *XX, *yb, *aa, *w, *zz, *r;
    int64_t *index;
} bnb_ctx;

/* BnB.jl:69-92.  sigma[m] encodes Σ as two flags: bit 0 = +m is in Σ (alpha_m >= 0), bit 1 = -m is in Σ (alpha_m <= 0).  Both
 * can be set when a feature belongs to two branched groups (overlapping partitions): the reference then zeroes both columns. */
static double bnb_lower_bound(bnb_ctx *c, const int8_t *sigma, double *alpha)
{
    int64_t rows = c->rows, Mp = c->Mp;
    for (int64_t m = 0; m < Mp; ++m) {
        const double *src = &AT(c->Xo, rows, 0, m);
        double *dp = &AT(c->XX, rows, 0, m), *dm = &AT(c->XX, rows, 0, Mp + m);
        for (int64_t i = 0; i < rows; ++i) {
            dp[i] = (sigma[m] & 2) ? 0.0 : src[i];          /* Xp[:, negConstr] .= 0 */
            dm[i] = (sigma[m] & 1) ? 0.0 : -src[i];         /* Xm[:, posConstr] .= 0 */
        }
    }
    memcpy(c->yb, c->yo, (size_t)rows * sizeof(double));
    double rn;
    oracle_nnls(c->XX, rows, 2 * Mp, c->yb, c->aa, &rn, c->w, c->zz, c->index, NULL);
    for (int64_t m = 0; m < Mp; ++m) {
        double ap = (sigma[m] & 2) ? 0.0 : c->aa[m];
        double an = (sigma[m] & 1) ? 0.0 : c->aa[Mp + m];
        alpha[m] = ap - an;
    }
    /* norm(XX*αα - y) with the zeroed columns == norm(Xo*(αp-αn) - y) evaluated on the original data */
    for (int64_t i = 0; i < rows; ++i) c->r[i] = -c->yo[i];
    for (int64_t m = 0; m < Mp; ++m) {
        double wp = (sigma[m] & 2) ? 0.0 : c->aa[m], wn = (sigma[m] & 1) ? 0.0 : c->aa[Mp + m];
        double wm = wp - wn;
        if (wm == 0.0) continue;
        const double *src = &AT(c->Xo, rows, 0, m);
        for (int64_t i = 0; i < rows; ++i) c->r[i] += src[i] * wm;
    }
    double s = 0.0;
    for (int64_t i = 0; i < rows; ++i) s += c->r[i] * c->r[i];
    return sqrt(s);
}

/* BnB.jl:94-132; returns value, writes the best alpha (signed) into out (valid only if value < inf), counts nodes */
static double bnb_node(bnb_ctx *c, double mu, int8_t *sigma, double *out, int64_t *nopen)
{
    int64_t Mp = c->Mp, Kp = c->Kp;
    double *alpha = malloc((size_t)Mp * sizeof(double));
    double lb = bnb_lower_bound(c, sigma, alpha);
    if (lb >= mu) { free(alpha); *nopen = 1; return INFINITY; }
    /* ν_k = Σ_{i<j in group k} max(0, -α_i α_j)  (BnB.jl:42-57) */
    int64_t kbest = -1; double nubest = 0.0; int allzero = 1;
    for (int64_t k = 0; k < Kp; ++k) {
        double nu = 0.0;
        for (int64_t i = 0; i < Mp; ++i) {
            if (AT(c->Po, Mp, i, k) == 0) continue;
            for (int64_t j = i + 1; j < Mp; ++j) {
                if (AT(c->Po, Mp, j, k) == 0) continue;
                double v = -alpha[i] * alpha[j];
                if (v > 0.0) nu += v;
            }
        }
        if (nu != 0.0) allzero = 0;
        if (kbest < 0 || nu > nubest) { kbest = k; nubest = nu; }   /* argmax: first maximal index */
    }
    if (allzero) {
        for (int64_t i = 0; i < c->rows; ++i) c->r[i] = -c->yo[i];
        for (int64_t m = 0; m < Mp; ++m) {
            const double *src = &AT(c->Xo, c->rows, 0, m);
            for (int64_t i = 0; i < c->rows; ++i) c->r[i] += src[i] * alpha[m];
        }
        double s = 0.0;
        for (int64_t i = 0; i < c->rows; ++i) s += c->r[i] * c->r[i];
        memcpy(out, alpha, (size_t)Mp * sizeof(double));
        free(alpha); *nopen = 1;
        return sqrt(s);
    }
    int8_t *saved = malloc((size_t)Mp);
    memcpy(saved, sigma, (size_t)Mp);
    double *ap = malloc((size_t)Mp * sizeof(double)), *am = malloc((size_t)Mp * sizeof(double));
    int64_t np_ = 0, nm_ = 0;
    for (int64_t m = 0; m < Mp; ++m) if (AT(c->Po, Mp, m, kbest) == 1) sigma[m] |= 1;      /* Σp = [Σ; pk] */
    double mup = bnb_node(c, mu, sigma, ap, &np_);
    memcpy(sigma, saved, (size_t)Mp);
    for (int64_t m = 0; m < Mp; ++m) if (AT(c->Po, Mp, m, kbest) == 1) sigma[m] |= 2;      /* Σm = [Σ; -pk] */
    double mum = bnb_node(c, mu < mup ? mu : mup, sigma, am, &nm_);
    memcpy(sigma, saved, (size_t)Mp);
    /* argmin([μ, μp, μm]) first index; index 0 returns this node's relaxed α with value μ (BnB.jl:126-128) */
    double val = mu; const double *src = alpha;
    if (mup < val) { val = mup; src = ap; }
    if (mum < val) { val = mum; src = am; }
    memcpy(out, src, (size_t)Mp * sizeof(double));
    *nopen = np_ + nm_ + 1;
    free(alpha); free(saved); free(ap); free(am);
    return val;
}

int oracle_fit_bnb(const double *X, int64_t N, int64_t M, const double *y, const int64_t *P, int64_t K, double eta,
                   double *alpha, double *beta, double *t, double *opt, int64_t *nopen)
{
    int64_t Mp = M + 1, Kp = K + 1;
    double *Xo = malloc((size_t)N * (size_t)Mp * sizeof(double));
    int64_t *Po = malloc((size_t)Mp * (size_t)Kp * sizeof(int64_t));
    double *Xn = malloc((size_t)(N + Kp) * (size_t)Mp * sizeof(double));
    double *yn = malloc((size_t)(N + Kp) * sizeof(double));
    if (!Xo || !Po || !Xn || !yn) { free(Xo); free(Po); free(Xn); free(yn); return -1; }
    oracle_homogeneous(X, N, M, P, K, Xo, Po);
    int64_t rows = oracle_regularize(Xo, N, Mp, y, Po, Kp, eta, Xn, yn);
    bnb_ctx c = { Xn, yn, Po, rows, Mp, Kp, NULL, NULL, NULL, NULL, NULL, NULL, NULL };
    c.XX = malloc((size_t)rows * (size_t)(2 * Mp) * sizeof(double));
    c.yb = malloc((size_t)rows * sizeof(double));
    c.aa = malloc((size_t)(2 * Mp) * sizeof(double));
    c.w = malloc((size_t)(2 * Mp) * sizeof(double));
    c.zz = malloc((size_t)rows * sizeof(double));
    c.r = malloc((size_t)rows * sizeof(double));
    c.index = malloc((size_t)(2 * Mp) * sizeof(int64_t));
    int8_t *sigma = calloc((size_t)Mp, 1);
    double *a = calloc((size_t)Mp, sizeof(double));
    int64_t no = 0;
    double val = bnb_node(&c, INFINITY, sigma, a, &no);
    /* BnB.jl:36-39: β_k = Σ_{m∈k} α_m (signed); α_m ← Σ_k Po[m,k] α_m / β_k; t = β[end] */
    double *bsum = malloc((size_t)Kp * sizeof(double));
    for (int64_t k = 0; k < Kp; ++k) {
        double s = 0.0;
        for (int64_t m = 0; m < Mp; ++m) s += (double)AT(Po, Mp, m, k) * a[m];
        bsum[k] = s;
    }
    for (int64_t m = 0; m < M; ++m) {
        double s = 0.0;
        for (int64_t k = 0; k < Kp; ++k) s += (double)AT(Po, Mp, m, k) * a[m] / bsum[k];
        alpha[m] = s;
    }
    for (int64_t k = 0; k < K; ++k) beta[k] = bsum[k];
    *t = bsum[K];
    *opt = val;
    if (nopen) *nopen = no;
    free(bsum); free(sigma); free(a);
    free(c.XX); free(c.yb); free(c.aa); free(c.w); free(c.zz); free(c.r); free(c.index);
    free(Xo); free(Po); free(Xn); free(yn);
    return 0;
}

void oracle_predict(const double *X, int64_t N, int64_t M, const int64_t *P, int64_t K,
                    const double *alpha, const double *beta, double t, double *yhat)
{
    for (int64_t i = 0; i < N; ++i) yhat[i] = t;
    for (int64_t m = 0; m < M; ++m) {
        double wm = 0.0;
        for (int64_t k = 0; k < K; ++k) wm += (double)AT(P, M, m, k) * alpha[m] * beta[k];
        const double *col = &AT(X, N, 0, m);
        for (int64_t i = 0; i < N; ++i) yhat[i] += col[i] * wm;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Synthetic inputs — integer-exact counter-based generator, identical on host and device
 * (the device twin is partitionedls.jl_amd/csrc/synth.hip; tests compare the two bit for bit).
 * ---------------------------------------------------------------------------------------------- */
static inline uint64_t sm64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint64_t rnd64(uint64_t seed, uint64_t stream, uint64_t idx)
{
    return sm64(sm64(seed ^ (stream * 0xD6E8FEB86659FD93ULL)) + idx);
}
static inline double uni01(uint64_t seed, uint64_t stream, uint64_t idx)
{
    return (double)(rnd64(seed, stream, idx) >> 11) * 0x1.0p-53;
}
/* sum of twelve 16-bit uniforms, centred and scaled: mean 0, variance 1 - 2^-32, exact in binary64 */
static inline double gauss12(uint64_t seed, uint64_t stream, uint64_t idx)
{
    uint64_t s = 0;
    for (uint64_t r = 0; r < 3; ++r) {
        uint64_t h = rnd64(seed, stream, idx * 3 + r);
        s += (h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48);
    }
    return ((double)(int64_t)s - 393210.0) * 0x1.0p-16;
}

void oracle_synth(uint64_t seed, int64_t N, int64_t D, int64_t K, double *X, double *y, int64_t *P, double *wstar)
{
    int64_t *grp = malloc((size_t)D * sizeof(int64_t));
    int64_t j = 0;
    for (int64_t k = 0; k < K; ++k) {
        int64_t sz = D / K + ((k < D % K) ? 1 : 0);
        for (int64_t q = 0; q < sz; ++q) grp[j++] = k;
    }
    if (P) {
        memset(P, 0, (size_t)D * (size_t)K * sizeof(int64_t));
        for (int64_t m = 0; m < D; ++m) AT(P, D, m, grp[m]) = 1;
    }
    double *ws = malloc((size_t)D * sizeof(double));
    for (int64_t k = 0; k < K; ++k) {
        double sum = 0.0;
        for (int64_t m = 0; m < D; ++m) if (grp[m] == k) sum += uni01(seed, 2, (uint64_t)m);
        double bk = (uni01(seed, 3, (uint64_t)k) - 0.5) * 10.0;
        for (int64_t m = 0; m < D; ++m) if (grp[m] == k) ws[m] = (uni01(seed, 2, (uint64_t)m) / sum) * bk;
    }
    if (wstar) memcpy(wstar, ws, (size_t)D * sizeof(double));
    if (X)
        for (int64_t jj = 0; jj < D; ++jj)
            for (int64_t i = 0; i < N; ++i)
                AT(X, N, i, jj) = gauss12(seed, 1, (uint64_t)i + (uint64_t)jj * (uint64_t)N);
    if (y)
        for (int64_t i = 0; i < N; ++i) {
            double acc = 0.0;
            for (int64_t jj = 0; jj < D; ++jj)
                acc = fma(gauss12(seed, 1, (uint64_t)i + (uint64_t)jj * (uint64_t)N), ws[jj], acc);
            y[i] = fma(0.1, gauss12(seed, 4, (uint64_t)i), acc + 1.0);
        }
    free(grp); free(ws);
}
