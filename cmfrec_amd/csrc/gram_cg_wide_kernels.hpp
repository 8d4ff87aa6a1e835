// gram_cg_wide_kernels.hpp -- explicit-model CG row update for 64 < k_t <= 145 through the row's own Gramian (round 4).
//
// The register tiles of cg_kernels.hpp end at 64 unknowns; beyond that every row used to run on cg_rows_generic_kernel
// (lane <-> unknown, one wave-wide reduction per gathered row and pass: config 3's shape under CG took 59 ms per iteration
// against 17 ms with the closed form).  The operator of the reference's CG for the explicit model,
//     factors_explicit_cg      (src/common.c:1098-1188):      Ap = diag(lam .. lam_last) p + sum_j (B_j.p) B_j
//     collective_block_cg      (src/collective.c:2134-2903):  ... + w C^T C p on the unknowns shared with the side information,
//                                                             right-hand side + w (U C)_row  (dense complete U, prefer_CtC)
// is linear in  G = sum_j B_j B_j^T,  and the first residual is  sum_j x_j B_j + w (U C)_row - (G + w C^T C + diag) a  -- the
// same G and the same right-hand side the closed form builds.  So the rank-k update is the Cholesky path's PRODUCER
// (chol_wave_kernel, WMODE 1: gathered rows straight into MFMA operands, one gather instead of max_cg_steps + 1), its raw
// tiles + right-hand-side partials go through HBM exactly as in the two-kernel Cholesky mode, and this kernel replaces the
// factorisation: one workgroup per row sums the row's partials (slice order: no atomics), lays the symmetric matrix out in LDS
// and runs the reference's CG steps on it with the same absolute thresholds (1e-12 / 1e-8).  Same arithmetic as the reference
// up to the order of the sums, as on the split rows of the k <= 64 path (gram_cg_kernels.hpp).
//
// The implicit model is not served here: its first residual weights the gathered rows by x + 1 and its products by x (quirk Q1,
// common.c:1939 against :1965), i.e. it needs B_j . a per gathered row inside the producer, which the Cholesky producer does
// not compute.
#pragma once
#include <hip/hip_runtime.h>

#include "cg_kernels.hpp"
#include "chol_wave_kernels.hpp"

namespace cmfhip {

constexpr int GCW_NF = 3;                      // unknowns per lane: k_t <= 192 by layout; the tile grid of the producer bounds it at 129
constexpr int GCW_NW = 8;                      // wavefronts per workgroup (one row per workgroup: the matrix fills most of a CU's LDS)

template <typename T>
struct WideCgParams {
    const T *part = nullptr;      // partials of the work items [part_base, ..), chol_wave_part_elems(NB) elements each
    int part_base = 0;
    int NB = 0;                   // 16-blocks of the producer's tile grid
    int border = 0;               // the last unknown is kept outside the tiles (k_t = 16 n + 1)
    const int *row_off = nullptr; // slices of the split rows (positions < n_heavy): items row_off[pos] .. row_off[pos + 1]
    int n_heavy = 0, n_slices = 0;
    int row_first = 0, row_last = 0;   // positions of the processing order handled by this launch
};

// LDS: M [kt][ldm] + x [64 NF] + partial products [NW][64 NF]
template <typename T>
__host__ __device__ constexpr size_t gcw_lds_elems(int kt) { return (size_t)kt * (size_t)(kt | 1) + (1 + GCW_NW) * 64 * (size_t)GCW_NF; }

template <typename T>
__global__ void __launch_bounds__(64 * GCW_NW)
gram_cg_wide_kernel(const CgParams<T> P, const WideCgParams<T> W)
{
    using Mf = CholMfma<T>;
    extern __shared__ __attribute__((aligned(16))) unsigned char gcw_smem[];
    const int kt = P.k, ldm = kt | 1;
    T *M = reinterpret_cast<T *>(gcw_smem);                 // [kt][ldm]
    T *xs = M + (size_t)kt * ldm;                           // [64 NF]
    T *yp = xs + 64 * GCW_NF;                               // [NW][64 NF]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int NB = W.NB, NT = NB * (NB + 1) / 2;
    const size_t PART = chol_wave_part_elems(NB);
    const int kq = W.border ? kt - 1 : kt;                  // unknowns inside the tiles
    const int kc = P.kc;

    for (int rix = W.row_first + blockIdx.x; rix < W.row_last; rix += gridDim.x) {
        const RowDesc d = P.desc[rix];
        const int row = d.row, nnz = d.nnz;
        const bool hv = rix < W.n_heavy;
        const int s0 = hv ? W.row_off[rix] : W.n_slices + (rix - W.n_heavy), s1 = hv ? W.row_off[rix + 1] : s0 + 1;
        const bool has_u = kc > 0 && row < P.rows_with_u;
        T lam = P.lam, lam_last = P.lam_last;
        if (has_u) {
            if (P.scale_lam || P.scale_lam_sideinfo) {      // collective.c:1285-1355
                T mult = (nnz > 0) ? row_lam_mult(P, row, nnz) : T(1);
                if (P.scale_lam_sideinfo) mult += (T)P.p_side;
                lam *= mult; lam_last *= mult;
            }
        } else if (P.scale_lam) {                           // common.c:679-723
            const T mult = row_lam_mult(P, row, nnz);
            lam *= mult;
            if (!P.scale_bias_const) lam_last *= mult;
        }
        __syncthreads();                                    // the previous row's CG is done with M
        // ---- M = sum of the partial tiles (slice order) + w C^T C + diag, both triangles ----
        constexpr int NTH = 64 * GCW_NW, EPR = 6;           // tile elements per thread and round: every slice's loads of a round in flight
        for (int e0 = 0; e0 < NT * 256; e0 += NTH * EPR) {
            T accq[EPR];
#pragma unroll
            for (int q = 0; q < EPR; q++) accq[q] = T(0);
            for (int sl = s0; sl < s1; sl++) {
                const T *pp = W.part + (size_t)(sl - W.part_base) * PART;
                T vq[EPR];
#pragma unroll
                for (int q = 0; q < EPR; q++) { const int e = e0 + NTH * q + tid; vq[q] = (e < NT * 256) ? pp[e] : T(0); }
#pragma unroll
                for (int q = 0; q < EPR; q++) accq[q] += vq[q];
            }
#pragma unroll
            for (int q = 0; q < EPR; q++) {
                const int e = e0 + NTH * q + tid;
                if (e >= NT * 256) continue;
                const int t = e >> 8, r = (e >> 6) & 3, l = e & 63;
                const int bi = tile_bi(t, NB), bj = tile_bj(t, NB);
                const int i = 16 * bi + Mf::row_of(l, r), j = 16 * bj + (l & 15);
                if (j >= i && j < kq) {                     // diagonal tiles: the upper half only, mirrored
                    T s = accq[q];
                    if (has_u && j < kc) s += P.w_side * P.CtC[(size_t)i * kc + j];
                    if (i == j) s += (i == kt - 1) ? lam_last : lam;
                    M[i * ldm + j] = s;
                    M[j * ldm + i] = s;
                }
            }
        }
        // ---- right-hand side; border column and border diagonal ----
        T v[GCW_NF];
#pragma unroll
        for (int c = 0; c < GCW_NF; c++) v[c] = T(0);
        for (int sl = s0; sl < s1; sl++) {
            const T *pv = W.part + (size_t)(sl - W.part_base) * PART + (size_t)NT * 256;
#pragma unroll
            for (int c = 0; c < GCW_NF; c++) {
                const int u = lane + 64 * c;
                if (u < kq) v[c] += pv[u];
                else if (W.border && u == kt - 1) v[c] += pv[32 * NB + 1];
            }
        }
        if (W.border) {
            if (wv == 0) {
                for (int u = lane; u < kq; u += 64) {
                    T g = T(0);
                    for (int sl = s0; sl < s1; sl++) g += W.part[(size_t)(sl - W.part_base) * PART + (size_t)NT * 256 + 16 * NB + u];
                    if (has_u && kt - 1 < kc) g += P.w_side * P.CtC[(size_t)u * kc + (kt - 1)];
                    M[u * ldm + (kt - 1)] = g;
                    M[(kt - 1) * ldm + u] = g;
                }
            } else if (wv == 1 && lane == 0) {
                T gam = T(0);
                for (int sl = s0; sl < s1; sl++) gam += W.part[(size_t)(sl - W.part_base) * PART + (size_t)NT * 256 + 32 * NB];
                if (has_u && kt - 1 < kc) gam += P.w_side * P.CtC[(size_t)(kt - 1) * kc + (kt - 1)];
                M[(kt - 1) * ldm + (kt - 1)] = gam + lam_last;
            }
        }
        T *arow = P.A + (size_t)row * P.lda;
        T a[GCW_NF], r[GCW_NF], p[GCW_NF];
#pragma unroll
        for (int c = 0; c < GCW_NF; c++) {
            const int u = lane + 64 * c;
            a[c] = (u < kt) ? arow[u] : T(0);
            if (has_u && u < kc) v[c] += P.w_side * P.UC[(size_t)row * kc + u];       // + w (U C)_row
        }
        // (M x)[u] for the unknowns of this lane; x is the same on the four waves: every wave takes the columns j = wv (mod NW),
        // the partial products meet in LDS and are summed in wave order
        auto mul = [&](const T (&x)[GCW_NF], T (&y)[GCW_NF]) {
            if (wv == 0) {
#pragma unroll
                for (int c = 0; c < GCW_NF; c++) xs[lane + 64 * c] = x[c];
            }
            __syncthreads();
            T part[GCW_NF];
#pragma unroll
            for (int c = 0; c < GCW_NF; c++) part[c] = T(0);
            for (int j = wv; j < kt; j += GCW_NW) {
                const T xj = xs[j];
#pragma unroll
                for (int c = 0; c < GCW_NF; c++) {
                    const int u = lane + 64 * c;
                    if (u < kt) part[c] += M[u * ldm + j] * xj;
                }
            }
#pragma unroll
            for (int c = 0; c < GCW_NF; c++) yp[wv * 64 * GCW_NF + lane + 64 * c] = part[c];
            __syncthreads();
#pragma unroll
            for (int c = 0; c < GCW_NF; c++) {
                const int u = lane + 64 * c;
                T sum = T(0);
#pragma unroll
                for (int w2 = 0; w2 < GCW_NW; w2++) sum += yp[w2 * 64 * GCW_NF + u];        // wave order
                y[c] = (u < kt) ? sum : T(0);
            }
        };
        auto dot = [&](const T (&x)[GCW_NF], const T (&y)[GCW_NF]) {
            T s = T(0);
#pragma unroll
            for (int c = 0; c < GCW_NF; c++) s += x[c] * y[c];
            return lanes::wave_sum(s);
        };
        __syncthreads();                                    // M is complete
        T Ma[GCW_NF];
        mul(a, Ma);
#pragma unroll
        for (int c = 0; c < GCW_NF; c++) { r[c] = (lane + 64 * c < kt) ? v[c] - Ma[c] : T(0); p[c] = r[c]; }   // common.c:1112-1139
        T r_old = dot(r, r);
        // (the waves carry identical copies of a, r, p and of every scalar: the exits below are uniform over the workgroup)
        if (r_old > (T)1e-12) {                             // :1147
            for (int step = 0; step < P.max_cg_steps; step++) {
                T Ap[GCW_NF];
                mul(p, Ap);
                const T alpha = cg_div(r_old, dot(Ap, p));
#pragma unroll
                for (int c = 0; c < GCW_NF; c++) { a[c] += alpha * p[c]; r[c] -= alpha * Ap[c]; }
                const T r_new = dot(r, r);
                if (r_new <= (T)1e-8) break;                // :1180
#pragma unroll
                for (int c = 0; c < GCW_NF; c++) p[c] = p[c] * cg_div(r_new, r_old) + r[c];
                r_old = r_new;
            }
        }
        if (wv == 0) {
#pragma unroll
            for (int c = 0; c < GCW_NF; c++) { const int u = lane + 64 * c; if (u < kt) arow[u] = a[c]; }
        }
    }
}

}  // namespace cmfhip
