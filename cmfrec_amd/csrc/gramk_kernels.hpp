// gramk_kernels.hpp -- the rank-k update of wide closed-form rows (k_t = 257 .. 272: 17 blocks of 16 unknowns, single precision)
// straight from the gather, without LDS and without barriers; the factorisation stays with chol_rows_kernel, which adds these
// partial matrices instead of running its own rank-k loop.
//
// Config 5's item step (collective_closed_form_block's sum of rank-1 terms, /root/reference/src/collective.c:1534-1846;
// common.c:1007-1012 for the plain model): per row  G = sum_j B_j B_j^T  over a few hundred to many thousand gathered rows of
// 257 numbers, v = sum_j x_j B_j.  In chol_rows_kernel that loop is staged through LDS by 16 wavefronts that own 10 scattered
// tiles each: two LDS reads, a multiplication and address arithmetic per MFMA, ~1 KB of scratch per lane, 143 of the step's
// 190 ms.  Here a row (or a slice of a long row) belongs to ONE workgroup of four wavefronts that do
// not talk to each other: each wave keeps a quarter of the 153 tiles (whole tile rows, so the quarter's A operands are five
// registers) in its accumulators, reads the 17 operand registers of a k-step -- entry 4 s + (lane >> 4), element 16 b +
// (lane & 15): the register is A operand of tile row b and B operand of tile column b at once, G being symmetric -- directly
// from the opposing matrix (the second to fourth wave of the workgroup find the lines in cache), and issues 38-39 MFMAs on
// them; three k-steps of operands are in flight.  The tiles leave in the accumulator layout of the matrix instruction, which is
// the layout chol_rows_kernel holds them in.
#pragma once
#include <hip/hip_runtime.h>

#include "chol_kernels.hpp"
#include "chol_wave_kernels.hpp"

namespace cmfhip {

constexpr int GK_NB = 17;                               // blocks of 16 unknowns
constexpr int GK_NT = GK_NB * (GK_NB + 1) / 2;          // 153 tiles of the upper triangle, packed as tile_bi / tile_bj do
constexpr int GK_PART = GK_NT * 256 + GK_NB * 16;       // elements of one work item's partial: the tiles, then the right-hand side
constexpr int GK_PD = 3;                                // k-steps of operands in flight (17 loads each; the counter tracks 63)

// tile rows of quarter Q: {Q, 7 - Q, 9 + Q, 16 - Q} -- 36 tiles each -- and two or three tiles of row 8
__host__ __device__ constexpr int gk_row_of(int Q, int s) { return s == 0 ? Q : s == 1 ? 7 - Q : s == 2 ? 9 + Q : 16 - Q; }
__host__ __device__ constexpr int gk_count(int Q) { return 36 + (Q == 0 ? 3 : 2); }
__host__ __device__ constexpr int gk_bi(int Q, int i)
{
    int rem = i;
    for (int s = 0; s < 4; s++) {
        const int r = gk_row_of(Q, s), len = GK_NB - r;
        if (rem < len) return r;
        rem -= len;
    }
    return 8;
}
__host__ __device__ constexpr int gk_bj(int Q, int i)
{
    int rem = i;
    for (int s = 0; s < 4; s++) {
        const int r = gk_row_of(Q, s), len = GK_NB - r;
        if (rem < len) return r + rem;
        rem -= len;
    }
    return 8 + (Q == 0 ? 0 : 1 + 2 * Q) + rem;
}
__host__ __device__ constexpr int gk_packed(int bi, int bj) { return bi * GK_NB - bi * (bi - 1) / 2 + (bj - bi); }
// is block b an A operand (a tile row) of quarter Q?
__host__ __device__ constexpr bool gk_is_row(int Q, int b) { return b == 8 || b == Q || b == 7 - Q || b == 9 + Q || b == 16 - Q; }

template <typename T, int Q>
__device__ __forceinline__ void gk_quarter(const CholParams<T> &P, size_t st, int nnz, T *__restrict__ out, int lane)
{
    using Mf = CholMfma<T>;
    using vec = typename Mf::vec;
    constexpr int NTQ = gk_count(Q);
    vec acc[NTQ];
#pragma unroll
    for (int i = 0; i < NTQ; i++) acc[i] = vec{0, 0, 0, 0};
    // right-hand side: this quarter sums the blocks b = Q, Q + 4, ... (block 16 goes to quarter 0)
    constexpr int NRB = (Q == 0) ? 5 : 4;
    T racc[NRB];
#pragma unroll
    for (int j = 0; j < NRB; j++) racc[j] = T(0);
    const int kc = lane >> 4, lm = lane & 15;
    const int kt = P.kt;
    const bool v16 = (256 + lm) < kt;                  // block 16: the live columns 256 .. kt - 1
    const int col16 = v16 ? 256 + lm : 256;
    const unsigned long long ldb_bytes = (unsigned long long)P.ldb * sizeof(T);
    const char *base = reinterpret_cast<const char *>(P.B + lm);
    const int nsteps = (nnz + 3) >> 2;

    T op[GK_PD][GK_NB], xw[GK_PD], okf[GK_PD];
    int idxn[GK_PD]; T xn[GK_PD], okn[GK_PD];
    auto load_entry = [&](int s, int step) {
        const int e = 4 * step + kc;
        const int ec = max(min(e, nnz - 1), 0);
        idxn[s] = P.indices[st + ec];
        xn[s] = P.values[st + ec];
        okn[s] = (e < nnz) ? T(1) : T(0);
    };
    auto issue_rows = [&](int s) {
        const T *rowp = reinterpret_cast<const T *>(base + (unsigned long long)(unsigned)idxn[s] * ldb_bytes);
#pragma unroll
        for (int b = 0; b < 16; b++) op[s][b] = rowp[16 * b];
        op[s][16] = rowp[col16 - lm];
        T x = xn[s];
        if (P.bias_sub != nullptr) x -= P.bias_sub[idxn[s]];
        xw[s] = x * okn[s];                              // common.c:991-996 (0 for the padding of the last k-step)
        okf[s] = okn[s];
    };
    if (nsteps > 0) {
#pragma unroll
        for (int s = 0; s < GK_PD; s++) load_entry(s, s);
#pragma unroll
        for (int s = 0; s < GK_PD; s++) { issue_rows(s); load_entry(s, GK_PD + s); }
    }
    const int niter = (nsteps + GK_PD - 1) / GK_PD;
    for (int it = 0; it < niter; it++) {
#pragma unroll
        for (int s = 0; s < GK_PD; s++) {
            // steps past the end repeat the last entry with weight zero on the A side (the B side may hold anything finite)
            const T ok = (it * GK_PD + s < nsteps) ? okf[s] : T(0);
            T o[GK_NB], a[GK_NB];
#pragma unroll
            for (int b = 0; b < GK_NB; b++) o[b] = op[s][b];
            if (!v16) o[16] = T(0);
            const T xws = (it * GK_PD + s < nsteps) ? xw[s] : T(0);
            static_for<0, GK_NB>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                if constexpr (gk_is_row(Q, b)) a[b] = o[b] * ok;
                else a[b] = T(0);
            });
            // the next use of this buffer: step (it + 1) GK_PD + s
            if (it + 1 < niter) { issue_rows(s); load_entry(s, (it + 2) * GK_PD + s); }
            static_for<0, NTQ>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int bi = gk_bi(Q, i), bj = gk_bj(Q, i);
                acc[i] = Mf::mma(a[bi], o[bj], acc[i]);
            });
            static_for<0, NRB>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int b = (j == 4) ? 16 : Q + 4 * j;
                racc[j] += xws * o[b];
            });
        }
    }
    static_for<0, NTQ>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int t = gk_packed(gk_bi(Q, i), gk_bj(Q, i));
#pragma unroll
        for (int r = 0; r < 4; r++) out[t * 256 + r * 64 + lane] = acc[i][r];
    });
    static_for<0, NRB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int b = (j == 4) ? 16 : Q + 4 * j;
        T v = lanes::tswap32_add(racc[j], racc[j]);
        v = lanes::tswap16_add(v, v);                     // the four entries of the k-steps (lane bits 4, 5)
        if (kc == 0) out[GK_NT * 256 + 16 * b + lm] = v;
    });
}

// One workgroup of four wavefronts per work item (CholSlices: slices of the split rows first, then whole rows in processing
// order), items handed out by a counter.  W.row_first / W.nrows = the range of work items, SL.part_base the item of slot 0.
template <typename T>
__global__ void __launch_bounds__(256, 1)
gramk_producer_kernel(const CholParams<T> P, const RowDesc *__restrict__ desc, const CholSlices<T> SL)
{
    __shared__ int s_item;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (;;) {
        if (tid == 0) s_item = P.row_first + atomicAdd(P.counter, 1);
        __syncthreads();
        const int item = s_item;
        __syncthreads();
        if (item >= P.nrows) break;
        int pos, sfirst = 0, scount = -1;
        if (item < SL.n_slices) { pos = SL.vrow[item]; sfirst = SL.first[item]; scount = SL.count[item]; }
        else pos = SL.n_heavy + (item - SL.n_slices);
        const RowDesc d = desc[pos];
        const int nnz = (scount >= 0) ? scount : d.nnz;
        if (nnz <= 0) continue;                          // rows without entries have no partial (the consumer knows)
        const size_t st = (size_t)d.st + (size_t)sfirst;
        T *out = SL.part + (size_t)(item - SL.part_base) * GK_PART;
        switch (wave) {
            case 0: gk_quarter<T, 0>(P, st, nnz, out, lane); break;
            case 1: gk_quarter<T, 1>(P, st, nnz, out, lane); break;
            case 2: gk_quarter<T, 2>(P, st, nnz, out, lane); break;
            default: gk_quarter<T, 3>(P, st, nnz, out, lane); break;
        }
    }
}

}  // namespace cmfhip
