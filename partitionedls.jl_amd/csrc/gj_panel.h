// gj_panel.h — block principal pivots on a global-memory tableau (shared by sweep_generic.hip and sweep_coop.hip).
//
// A block is a list ks[0..m) of pivot variables.  The panel P[j][i] = T[i][ks[j]] (all rows i; read as row ks[j] by symmetry)
// is eliminated in LDS by m sequential Gauss–Jordan steps in the unified convention of the project
//        T_ij -= T_ik T_kj / d ,  T_ik = T_ik / |d| ,  T_kk = -1/d        (d = T_kk; the same formula enters and removes),
// keeping column s as of its own step in Z[s][.] and 1/d_s in dinv[s] (0: the entering column was rejected as dependent on the
// current basis, Lawson–Hanson).  The caller then applies  T_ic -= sum_s Z_s[i] Z_s[c] / d_s  to its rows in ONE pass and
// overwrites the rows / columns of the pivoted variables from the final panel.  Block-local barriers only.
#pragma once
#include "common.h"

namespace partls {

static constexpr int GJ_MB = 16;              // pivots per block at most

template <int NT>
__device__ __forceinline__ void gj_panel_load(const double *T, int ld, const int *ks, int m, double *Pn, int tid)
{
    for (int j = 0; j < m; ++j) {
        const double *src = T + (size_t)ks[j] * ld;
        for (int i = tid; i < ld; i += NT) Pn[(size_t)j * ld + i] = __builtin_nontemporal_load(&src[i]);
    }
    __syncthreads();
}

// returns the number of accepted pivots (uniform).  Acceptance of an ENTERING variable k follows the leave-one-out rule of
// sweep_blk.hip: d_k > piv_eps and d_k > piv_eps * T[j][k]^2 for every variable row j — the basis B + k must not be
// numerically dependent at the piv_eps level in ANY of its members (an exactly dependent column next to a nearly collinear
// pair passes the first test alone: its computed pivot carries an error of eps * |c|^2).  s_basic is kept live here: the flag of
// an accepted pivot flips at the end of its step (the callers only record the rejections).  red: NT / 64 doubles of scratch.
template <int NT>
__device__ __forceinline__ int gj_panel_eliminate(double *Pn, double *Zn, double *dinv, double *uj, double *red, const int *ks, int m,
                                                  int ld, uint8_t *s_basic, double piv_eps, int tid)
{
    int accepted = 0;
    for (int s = 0; s < m; ++s) {
        const int k = ks[s];
        // Two barriers per step.  Before the first: the entries of the pivot COLUMN s at the pivot rows are published (not of column j
        // at row k: equal only in exact arithmetic — with the factor taken from column s the panel receives exactly the symmetric
        // rank-1 term z_s z_s'/d_s of the fused update; mixing the two loses the solution on ill-conditioned data, see
        // sweep_blk.hip), and every wave leaves its part of max_i T_ik^2 over the variable rows (nonbasic rows satisfy
        // T_ik^2 <= d_k, so only a basic row can trip the leave-one-out test) — computed unconditionally: cheaper than a barrier.
        if (tid < m) uj[tid] = Pn[(size_t)s * ld + ks[tid]];
        double c2 = 0.0;
        for (int i = tid; i < ld - 1; i += NT)
            if (i != k) { const double z = Pn[(size_t)s * ld + i]; c2 = fmax(c2, z * z); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c2 = fmax(c2, __shfl_xor(c2, off));
        if ((tid & 63) == 0) red[tid >> 6] = c2;
        __syncthreads();
        const double d = uj[s];
        const bool bas = s_basic[k] != 0;
        bool ok = bas || (d > piv_eps);
        if (ok && !bas) {                                                // uniform
            double cmax2 = 0.0;
            for (int w = 0; w < NT / 64; ++w) cmax2 = fmax(cmax2, red[w]);
            ok = !(cmax2 * piv_eps >= d);
        }
        const double inv = ok ? 1.0 / d : 0.0, ainv = fabs(inv);
        if (ok) {
            ++accepted;
            for (int i = tid; i < ld; i += NT) {
                const double zi = Pn[(size_t)s * ld + i];
                Zn[(size_t)s * ld + i] = zi;
                const double mi = -zi * inv;
                for (int j = 0; j < m; ++j) {
                    if (j == s) continue;
                    const double pji = Pn[(size_t)j * ld + i];
                    Pn[(size_t)j * ld + i] = (i == k) ? uj[j] * ainv : fma(mi, uj[j], pji);
                }
                Pn[(size_t)s * ld + i] = (i == k) ? -inv : zi * ainv;
            }
        }
        if (tid == 0) dinv[s] = inv;
        __syncthreads();                                                 // panel, uj, red and the old basis flag are done with
        if (ok && tid == 0) s_basic[k] ^= 1;                             // the next read of THIS flag is behind a later barrier
    }
    __syncthreads();
    return accepted;
}

// fused rank-m update of rows [row0, row1) (one wave per row), then the pivoted rows / columns from the final panel
template <int NT>
__device__ __forceinline__ void gj_apply(double *T, int ld, int row0, int row1, const double *Pn, const double *Zn, const double *dinv,
                                         const int *ks, int m, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    // Each wave owns whole rows; the row is walked in chunks of 8 x 64 columns with all eight loads issued before the first
    // use: a single workgroup per CU has little memory-level parallelism otherwise (one dependent HBM latency per 64 elements).
    constexpr int U = 8;
    for (int i = row0 + wave; i < row1; i += NT / 64) {
        double *row = T + (size_t)i * ld;
        double fi[GJ_MB];
#pragma unroll
        for (int s = 0; s < GJ_MB; ++s) fi[s] = (s < m && dinv[s] != 0.0) ? -Zn[(size_t)s * ld + i] * dinv[s] : 0.0;   // rejected pivot: Zn[s] was never written
        for (int c0 = lane; c0 < ld; c0 += 64 * U) {
            double acc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + 64 * u;
                acc[u] = (c < ld) ? __builtin_nontemporal_load(&row[c]) : 0.0;
            }
#pragma unroll
            for (int s = 0; s < GJ_MB; ++s) {
                if (s < m && dinv[s] != 0.0) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int c = c0 + 64 * u;
                        acc[u] = fma(fi[s], Zn[(size_t)s * ld + (c < ld ? c : 0)], acc[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + 64 * u;
                if (c < ld) row[c] = acc[u];
            }
        }
    }
    __syncthreads();                                                       // all generic updates of this workgroup are issued
    for (int j = 0; j < m; ++j) {
        const int k = ks[j];
        for (int i = row0 + tid; i < row1; i += NT) T[(size_t)i * ld + k] = Pn[(size_t)j * ld + i];           // column k
        if (k >= row0 && k < row1)
            for (int c = tid; c < ld; c += NT) T[(size_t)k * ld + c] = Pn[(size_t)j * ld + c];               // row k
    }
    __syncthreads();
}

// pivots per block such that the two [mb][ld] LDS images fit `budget` bytes
inline int gj_block_size(int ld, size_t budget)
{
    int mb = (int)(budget / ((size_t)2 * ld * sizeof(double)));
    if (mb > GJ_MB) mb = GJ_MB;
    if (mb < 1) mb = 1;
    return mb;
}

}  // namespace partls
