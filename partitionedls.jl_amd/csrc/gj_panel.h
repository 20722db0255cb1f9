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
#include <type_traits>

namespace partls {

static constexpr int GJ_MB = 16;              // pivots per block at most

// TRI: only the upper triangle of the symmetric tableau is kept (entries (i, c) with c >= i; the rhs is column n): column k of the
// tableau is T[i][k] for i <= k (strided) and T[k][i] beyond (row k, contiguous).  Halves the bytes of the fused update, which is what
// bounds the one-workgroup kernel (256 tableaus of (n+1)^2 doubles stream through HBM once per block).
template <int NT, bool TRI = false>
__device__ __forceinline__ void gj_panel_load(const double *T, int ld, const int *ks, int m, double *Pn, int tid)
{
    if constexpr (TRI) {
        for (int i = tid; i < ld; i += NT) {
            double v[GJ_MB];
#pragma unroll
            for (int j = 0; j < GJ_MB; ++j) {
                if (j < m) { const int k = ks[j]; v[j] = __builtin_nontemporal_load(i <= k ? &T[(size_t)i * ld + k] : &T[(size_t)k * ld + i]); }
                else v[j] = 0.0;
            }
#pragma unroll
            for (int j = 0; j < GJ_MB; ++j) if (j < m) Pn[(size_t)j * ld + i] = v[j];
        }
        __syncthreads();
        return;
    }
    // all m row loads of a thread are issued before the first LDS store: in the cooperative kernel the rows were just rewritten by
    // other XCDs and come from memory (~2 us each) — one exposed latency per block instead of one per pivot row
    for (int i = tid; i < ld; i += NT) {
        double v[GJ_MB];
#pragma unroll
        for (int j = 0; j < GJ_MB; ++j) v[j] = (j < m) ? __builtin_nontemporal_load(&T[(size_t)ks[j] * ld + i]) : 0.0;
#pragma unroll
        for (int j = 0; j < GJ_MB; ++j) if (j < m) Pn[(size_t)j * ld + i] = v[j];
    }
    __syncthreads();
}

__device__ __forceinline__ double gj_rcp(double d)       // v_rcp_f64 + two Newton steps (as sweep_blk.hip's fast_rcp): ~1 ulp
{
    double y = __builtin_amdgcn_rcp(d);
    y = fma(fma(-d, y, 1.0), y, y);
    y = fma(fma(-d, y, 1.0), y, y);
    return y;
}

// returns the number of accepted pivots (uniform).  Acceptance of an ENTERING variable k follows the leave-one-out rule of
// sweep_blk.hip: d_k > piv_eps and d_k > piv_eps * T[j][k]^2 for every variable row j — the basis B + k must not be
// numerically dependent at the piv_eps level in ANY of its members (an exactly dependent column next to a nearly collinear
// pair passes the first test alone: its computed pivot carries an error of eps * |c|^2).  Every row tests its own entry (a
// nonbasic row satisfies T_jk^2 <= d_k, so only a basic row can trip it) and raises a flag in LDS — no reduction.  s_basic is
// kept live here: the flag of an accepted pivot flips at the end of its step (the callers only record the rejections).
// Layout, as in sweep_blk.hip: thread t keeps ROW t of the panel in registers for the whole block (NT >= ld: one row per
// thread); a step exchanges only the m pivot-row entries of its pivot column through LDS (uj, double buffered: the owners
// publish the entries of column s+1 at the end of step s) and stores column s as of its own step to Zn.  Two barriers per step.
// uj: 2 x GJ_MB doubles, red: >= 2 doubles (the veto flags of even / odd steps).
// Cost matters: the cooperative kernel runs this REDUNDANTLY in every workgroup, 16 sequential steps per block.  History at
// n = 513 (stamped, cycles per step): 6.8k with a max-reduction per step and a runtime-bounded column loop on the LDS panel
// (49 of the 82 us of a block), 3.9k with uniform control flow and grouped LDS reads, now ~1k.
template <int NT>
__device__ __forceinline__ int gj_panel_eliminate(double *Pn, double *Zn, double *dinv, double *uj, double *red, const int *ks, int m_,
                                                  int ld_, uint8_t *s_basic, double piv_eps, int tid)
{
    // everything that steers control flow is made provably wave-uniform (it comes from LDS words or arguments the compiler cannot
    // see through): scalar branches and scalar address arithmetic instead of exec masks and per-lane 64-bit index math
    const int m = __builtin_amdgcn_readfirstlane(m_), ld = __builtin_amdgcn_readfirstlane(ld_);
    const bool has_row = tid < ld;
    const bool var_row = tid < ld - 1;                                   // the last row is the rhs: no variable, never vetoes
    double pv[GJ_MB];
#pragma unroll
    for (int j = 0; j < GJ_MB; ++j) pv[j] = (j < m && has_row) ? Pn[j * ld + tid] : 0.0;
    int myj = -1;                                                        // this row is pivot row myj of the block
    for (int j = 0; j < m; ++j) if (ks[j] == tid) myj = j;
    if (myj >= 0) uj[myj] = pv[0];                                       // entries of pivot column 0 at the pivot rows
    if (tid < 2) red[tid] = 0.0;
    __syncthreads();
    int accepted = 0;
#pragma unroll
    for (int s = 0; s < GJ_MB; ++s) {
        if (s >= m) break;                                               // uniform
        const int k = __builtin_amdgcn_readfirstlane(ks[s]);
        const double *us = uj + (s & 1) * GJ_MB;
        // entries of the pivot COLUMN s at the pivot rows (not of column j at row k: equal only in exact arithmetic — with the
        // factor taken from column s the panel receives exactly the symmetric rank-1 term z_s z_s'/d_s of the fused update;
        // mixing the two loses the solution on ill-conditioned data, see sweep_blk.hip)
        const double d = us[s];
        const bool bas = __builtin_amdgcn_readfirstlane((int)s_basic[k]) != 0;
        if (!bas && var_row && tid != k && (pv[s] * pv[s]) * piv_eps >= d) red[s & 1] = 1.0;   // same value from every writer
        __syncthreads();
        const double flag = red[s & 1];
        const bool ok = bas || (__builtin_amdgcn_readfirstlane((int)(d > piv_eps && flag == 0.0)) != 0);
        if (tid == 0) red[(s + 1) & 1] = 0.0;                            // the other flag: last read before the previous closing barrier
        const double inv = ok ? gj_rcp(d) : 0.0, ainv = fabs(inv);
        if (ok) {
            ++accepted;
            const double zi = pv[s];
            if (has_row) Zn[s * ld + tid] = zi;
            const double mi = -zi * inv;
            const bool piv = tid == k;
#pragma unroll
            for (int j = 0; j < GJ_MB; ++j) {
                if (j < m && j != s) {                                   // uniform
                    const double u = us[j];
                    pv[j] = piv ? u * ainv : fma(mi, u, pv[j]);
                }
            }
            pv[s] = piv ? -inv : zi * ainv;
        }
        if (tid == 0) dinv[s] = inv;
        if (s + 1 < GJ_MB) { if (s + 1 < m && myj >= 0) uj[((s + 1) & 1) * GJ_MB + myj] = pv[s + 1]; }   // pivot column s+1 at the pivot rows
        __syncthreads();                                                 // uj, the flag and the old basis flag are done with
        if (ok && tid == 0) s_basic[k] ^= 1;                             // the next read of THIS flag is behind a later barrier
    }
#pragma unroll
    for (int j = 0; j < GJ_MB; ++j) if (j < m && has_row) Pn[j * ld + tid] = pv[j];     // the final panel: rows / columns of the pivoted variables
    __syncthreads();
    return accepted;
}

// fused rank-m update of rows [row0, row1), then the pivoted rows / columns from the final panel.  One wave per row, or two (each
// half of the columns) when the workgroup owns so few rows that half its waves would idle (the cooperative kernel at n = 513: 9
// rows for 16 waves).  A row is walked in spans of 8 x 64 columns — eight global loads in flight, then per accepted pivot as many LDS
// reads and FMAs — and one last span of the remaining 1..8 chunks (its own instantiation per length: straight-line code); which pivots of the block were accepted is a wave-uniform bit mask.
// Tsrc == nullptr: in place (sweep_generic.hip).  Otherwise the rows are read from Tsrc and written to T — the cooperative kernel
// ping-pongs between two tableau images so that nobody rewrites a row that another workgroup may still be loading.
template <int NT, bool TRI = false>
__device__ __forceinline__ void gj_apply(double *T, int ld_, int row0_, int row1_, const double *Pn, const double *Zn, const double *dinv,
                                         const int *ks, int m_, int tid, const double *Tsrc = nullptr)
{
    if (!Tsrc) Tsrc = T;
    constexpr int NWAVES = NT / 64;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = __builtin_amdgcn_readfirstlane(m_), ld = __builtin_amdgcn_readfirstlane(ld_);
    const int row0 = __builtin_amdgcn_readfirstlane(row0_), row1 = __builtin_amdgcn_readfirstlane(row1_);
    unsigned act = 0;                                                       // accepted pivots of the block (a rejected one never wrote Zn[s])
    for (int s = 0; s < m; ++s) if (dinv[s] != 0.0) act |= 1u << s;
    act = (unsigned)__builtin_amdgcn_readfirstlane((int)act);
    const int nrows = row1 - row0;
    const int split = (2 * nrows <= NWAVES) ? 2 : 1;
    const int hs = (((ld + 1) / 2 + 63) / 64) * 64;                         // first half [0, hs), second [hs, ld)
    for (int it = wave; it < nrows * split; it += NWAVES) {
        const int i = row0 + it / split, half = it % split;
        int cbeg = half ? hs : 0;
        const int cend = (split == 2 && !half) ? (hs < ld ? hs : ld) : ld;
        if constexpr (TRI) { const int lo = i & ~63; if (lo > cbeg) cbeg = lo; if (cbeg >= cend) continue; }   // columns >= i only (whole 64-chunks)
        double *row = T + (size_t)i * ld;
        const double *srow = Tsrc + (size_t)i * ld;
        double fi[GJ_MB];
#pragma unroll
        for (int s = 0; s < GJ_MB; ++s) fi[s] = ((act >> s) & 1u) ? -Zn[s * ld + i] * dinv[s] : 0.0;
        int c0 = cbeg;
        // a span of U x 64 columns with all its loads in flight together; LAST: its final chunk may be partial (lane predicate)
        auto span = [&](auto uc, auto lastc) {
            constexpr int U = decltype(uc)::value;
            constexpr bool LAST = decltype(lastc)::value;
            const int cl = c0 + 64 * (U - 1) + lane;
            const bool vl = !LAST || cl < cend;
            double acc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (LAST && u == U - 1) acc[u] = vl ? __builtin_nontemporal_load(&srow[cl]) : 0.0;
                else acc[u] = __builtin_nontemporal_load(&srow[c0 + 64 * u + lane]);
            }
#pragma unroll
            for (int s = 0; s < GJ_MB; ++s) {
                if ((act >> s) & 1u) {                                      // scalar branch
                    double z[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) z[u] = Zn[s * ld + ((LAST && u == U - 1) ? (vl ? cl : cbeg) : c0 + 64 * u + lane)];
#pragma unroll
                    for (int u = 0; u < U; ++u) acc[u] = fma(fi[s], z[u], acc[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (LAST && u == U - 1) { if (vl) row[cl] = acc[u]; }
                else row[c0 + 64 * u + lane] = acc[u];
            }
        };
        using std::integral_constant;
        for (; c0 + 512 < cend; c0 += 512) span(integral_constant<int, 8>{}, std::false_type{});
        switch ((cend - c0 + 63) >> 6) {                                    // the rest of the row (1..8 chunks) as ONE span: scalar jump
            case 1: span(integral_constant<int, 1>{}, std::true_type{}); break;
            case 2: span(integral_constant<int, 2>{}, std::true_type{}); break;
            case 3: span(integral_constant<int, 3>{}, std::true_type{}); break;
            case 4: span(integral_constant<int, 4>{}, std::true_type{}); break;
            case 5: span(integral_constant<int, 5>{}, std::true_type{}); break;
            case 6: span(integral_constant<int, 6>{}, std::true_type{}); break;
            case 7: span(integral_constant<int, 7>{}, std::true_type{}); break;
            case 8: span(integral_constant<int, 8>{}, std::true_type{}); break;
            default: break;
        }
    }
    __syncthreads();                                                       // all generic updates of this workgroup are issued
    for (int j = 0; j < m; ++j) {
        const int k = __builtin_amdgcn_readfirstlane(ks[j]);
        for (int i = row0 + tid; i < (TRI ? (k + 1 < row1 ? k + 1 : row1) : row1); i += NT) T[(size_t)i * ld + k] = Pn[j * ld + i];   // column k (TRI: rows <= k)
        if (k >= row0 && k < row1)
            for (int c = (TRI ? k : 0) + tid; c < ld; c += NT) T[(size_t)k * ld + c] = Pn[j * ld + c];     // row k (TRI: columns >= k)
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
