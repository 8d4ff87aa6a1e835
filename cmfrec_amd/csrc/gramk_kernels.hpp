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
#ifndef GK_OPEN_SLOTS
#define GK_OPEN_SLOTS 32                                  // workgroup slots of the 2 x CUs a launch leaves to other streams: one per shader engine (session.hip; 24: no effect, 32-40: best, profiles/r04)
#endif
constexpr int GK_PD = 2;                                // k-steps of operands in flight (5 loads each; the counter tracks 63)

// ORDER OF THE UNKNOWNS INSIDE THE TILES.  The operand register of block b holds, for lane l (l & 15 = position, l >> 4 = entry
// of the k-step), one number of the gathered row.  Reading "position p of block b = column 16 b + p" costs one 4-byte load per
// block, lane and k-step: 17 load instructions per k-step and wave, each of them 64 scattered 4-byte reads -- the texture
// addresser, not the matrix pipe, then sets the pace, and the 63-entry load counter allows only three k-steps in flight against
// a gather that goes to HBM (the opposing matrix of config 5's item step is 1.6 GB).  The Gramian does not care which unknown
// sits where, so the blocks take the order that 16-byte loads deliver: lane position p reads columns 64 q + 4 p .. + 3
// (q = 0..3) and register r of that load is the operand of block 4 q + r.  Position t = 16 b + p of the tile order therefore
// holds unknown  64 (b >> 2) + 4 p + (b & 3)  for b < 16; block 16 (columns 256 .. k_t - 1, one 4-byte load) keeps 256 + p.
// Five loads per k-step, six k-steps in flight.  The consumer factorises in this order (a symmetric permutation of the system)
// and puts the solution back; the initial matrices are packed in it (tile_pack_lane_kernel).
__host__ __device__ constexpr int gk_unknown(int t) { return (t < 256) ? 64 * (t >> 6) + 4 * (t & 15) + ((t >> 4) & 3) : t; }

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
    const char *base = reinterpret_cast<const char *>(P.B);
    const int nsteps = (nnz + 3) >> 2;
    typedef T vec4u __attribute__((ext_vector_type(4), aligned(sizeof(T))));     // rows start at any multiple of 4 bytes

    vec op4[GK_PD][4]; T op16[GK_PD], xw[GK_PD], okf[GK_PD];
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
        for (int q = 0; q < 4; q++) op4[s][q] = *reinterpret_cast<const vec4u *>(rowp + 64 * q + 4 * lm);
        op16[s] = rowp[col16];
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
            for (int b = 0; b < 16; b++) o[b] = op4[s][b >> 2][b & 3];
            o[16] = v16 ? op16[s] : T(0);
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
        *reinterpret_cast<vec *>(out + t * 256 + lane * 4) = acc[i];       // [tile][lane][register]: one 16-byte store
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
__global__ void __launch_bounds__(256, 2)
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

// ---------------------------------------------------------------------------------------------------------------------------
// The consumer: initial matrix + the producer's partials, blocked Cholesky, both substitutions, for one row per workgroup of
// FOUR wavefronts, two workgroups per CU (collective_closed_form_block's tposv, /root/reference/src/collective.c:1819-1846;
// factors_closed_form, common.c:1060-1075).
//
// The 16-wavefront row kernel spent ~870 k cycles on a 257 x 257 system whose matrix-pipe time is ~30 k: 51 barriers of 16
// waves, two LDS reads per MFMA, every phase a latency chain with nothing else on the CU.  Here
//  * a wave owns whole tile COLUMNS of the upper triangle -- {Q, 7 - Q, 9 + Q, 16 - Q} plus two or three tiles of column 8,
//    38-39 tiles, all indices compile-time (one instantiation per wave, as in the producer) -- so the panel tiles of a block
//    row are spread over the four waves, and the 17 panel tiles X_b of a step, read once per wave (one ds_read_b128 each, the
//    register of k-step q is A operand of tile row b and B operand of tile column b), feed all of its trailing MFMAs;
//  * the accumulators hold the NEGATED matrix, so the trailing update is a plain  N += X_bi^T X_bj  without a sign flip per
//    operand (the diagonal block and inv(R) are negated where they are read: 8 instructions per step);
//  * the panel tiles are double-buffered and inv(R_kk) has a slot per block: two barriers per block step;
//  * the backward substitution runs by block ROWS (every wave multiplies its tiles of row i by the x_j it keeps for its own
//    columns, one cross-lane reduction and one barrier per step) instead of one wave per block column;
//  * the second workgroup of the CU fills the diagonal-block and barrier latencies of the first.
// LDS per workgroup: 2 x 17 KB of panel tiles, 21.3 KB of inv(R), right-hand side / solution / partial sums.
// ---------------------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int gc_col_of(int Q, int s) { return s == 0 ? Q : s == 1 ? 7 - Q : s == 2 ? 9 + Q : 16 - Q; }
__host__ __device__ constexpr int gc_count(int Q) { return 36 + (Q == 0 ? 3 : 2); }
__host__ __device__ constexpr int gc8_first(int Q) { return Q == 0 ? 0 : 1 + 2 * Q; }
__host__ __device__ constexpr int gc_bj(int Q, int i)
{
    int rem = i;
    for (int s = 0; s < 4; s++) {
        const int c = gc_col_of(Q, s);
        if (rem <= c) return c;
        rem -= c + 1;
    }
    return 8;
}
__host__ __device__ constexpr int gc_bi(int Q, int i)
{
    int rem = i;
    for (int s = 0; s < 4; s++) {
        const int c = gc_col_of(Q, s);
        if (rem <= c) return rem;
        rem -= c + 1;
    }
    return gc8_first(Q) + rem;
}
// owner of the diagonal tile (and of the solution block) b
__host__ __device__ constexpr int gc_diag_owner(int b) { return b < 4 ? b : b < 8 ? 7 - b : b == 8 ? 3 : b < 13 ? b - 9 : 16 - b; }
// register slot of x_b in the wave that keeps it for its column b (columns {Q, 7 - Q, 9 + Q, 16 - Q} -> 0..3, column 8 -> 4)
__host__ __device__ constexpr int gc_xslot(int Q, int b) { return b == 8 ? 4 : b == Q ? 0 : b == 7 - Q ? 1 : b == 9 + Q ? 2 : 3; }
__host__ __device__ constexpr bool gc_has_col(int Q, int b) { return b == 8 || b == Q || b == 7 - Q || b == 9 + Q || b == 16 - Q; }

template <typename T> struct GcShared {
    __attribute__((aligned(16))) T Xt[2][GK_NB * 256];                      // panel tiles of a block step, [b][lane][k-step]
    T rinv[GK_NB * 16 * CholMfma<T>::LDR];     // inv(R_kk) of every block
    T rhs[GK_NB * 16];                         // right-hand side -> y (in place)
    T xall[GK_NB * 16];                        // solution
    T psum[2][4][16];                          // backward substitution: the waves' partial sums of a block row
    int rix;
};

struct GcRow {                                 // what the workgroup's four instantiations share about the row
    int item0, item1;                          // the producer's work items of this row (item0 == item1: none)
    bool has_u;
    int kt;
};

template <typename T, int Q>
__device__ __forceinline__ void gc_row(const CholParams<T> &P, GcShared<T> &S, const GcRow &R, T lam, T lam_last, T *__restrict__ arow, int lane)
{
    using Mf = CholMfma<T>;
    using vec = typename Mf::vec;
    constexpr int NTQ = gc_count(Q);
    constexpr int LDR = Mf::LDR, RSZ = 16 * LDR;
    const int lm = lane & 15, kt = R.kt;
    const int tid = 64 * Q + lane;
    vec acc[NTQ];
#pragma unroll
    for (int i = 0; i < NTQ; i++) acc[i] = vec{0, 0, 0, 0};
    // ---- 1. N = -(initial matrix + partials + diagonal);  right-hand side ----
    T r0 = T(0), r1 = T(0);                    // unknowns tid and 256 + tid
    if (R.has_u || P.rhs_prefilled_all) {      // w U C prefilled (collective.c:5768-5773)
        r0 = arow[gk_unknown(tid)];                // (tile-order position tid < 256 <-> an unknown < 256 <= kt - 1)
        if (256 + tid < kt) r1 = arow[256 + tid];
    }
    auto add_tiles = [&](const T *__restrict__ pp) {           // thirteen 16-byte loads in flight
        static_for<0, 3>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            vec ld[13];
            static_for<0, 13>([&](auto jc) {
                constexpr int i = 13 * g + decltype(jc)::value;
                if constexpr (i < NTQ) ld[i - 13 * g] = *reinterpret_cast<const vec *>(pp + gk_packed(gc_bi(Q, i), gc_bj(Q, i)) * 256 + lane * 4);
            });
            static_for<0, 13>([&](auto jc) {
                constexpr int i = 13 * g + decltype(jc)::value;
                if constexpr (i < NTQ) acc[i] -= ld[i - 13 * g];
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    for (int item = R.item0; item < R.item1; item++) {
        const T *pp = P.gk_part + (size_t)(item - P.gk_base) * P.gk_stride;
        add_tiles(pp);
        r0 += pp[GK_NT * 256 + tid];
        if (tid < 16) r1 += pp[GK_NT * 256 + 256 + tid];
    }
    if (P.gk_init1 != nullptr) add_tiles(P.gk_init1);
    if (R.has_u && P.gk_init2 != nullptr) add_tiles(P.gk_init2);
    static_for<0, NTQ>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (gc_bi(Q, i) == gc_bj(Q, i)) {
            constexpr int b = gc_bi(Q, i);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int gi = 16 * b + Mf::row_of(lane, r), gj = 16 * b + lm;
                if (gi == gj) acc[i][r] -= (gi >= kt) ? T(1) : ((gi == kt - 1) ? lam_last : lam);   // add_to_diag2: common.c:1060-1062, collective.c:1819
            }
        }
    });
    S.rhs[tid] = r0;
    if (tid < 16) S.rhs[256 + tid] = r1;
    // ---- 2. blocked Cholesky  M = R^T R  of M = -N ----
    for (int kbk = 0; kbk < GK_NB; kbk++) {
        T *rslot = S.rinv + kbk * RSZ;
        T *Xw = S.Xt[kbk & 1];
        // a. diagonal block, by its owner
        {
            vec d = vec{0, 0, 0, 0};
            bool mine = false;
            static_for<0, NTQ>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (gc_bi(Q, i) == gc_bj(Q, i)) {
                    if (kbk == gc_bi(Q, i)) { d = -acc[i]; mine = true; }
                }
            });
            if (mine) chol_diag_block<T>(d, rslot, lane, min(16, kt - 16 * kbk));
        }
        __syncthreads();
        // b. panel tiles of block row kbk:  X = inv(R_kk)^T tile  (tile = -N: the A operand carries the sign)
        {
            T ainv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) ainv[r] = -rslot[Mf::row_of(lane, r) * LDR + lm];
            if (Q == ((kbk + 1) & 3)) {           // y_k = inv(R_kk)^T rhs_k, in place (a wave that is not the next diagonal's owner... any wave)
                const T vk = S.rhs[16 * kbk + lm];
                T yv = T(0);
                static_for<0, 16>([&](auto lc) {
                    constexpr int l = decltype(lc)::value;
                    yv += rslot[l * LDR + lm] * lanes::row_bcast16<l>(vk);
                });
                if (lane < 16) S.rhs[16 * kbk + lane] = yv;
            }
            static_for<0, NTQ>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int bi = gc_bi(Q, i), bj = gc_bj(Q, i);
                if constexpr (bi < bj) {
                    if (kbk == bi) {
                        vec x = Mf::mma(ainv[0], acc[i][0], vec{0, 0, 0, 0});          // two independent chains
                        vec x2 = Mf::mma(ainv[2], acc[i][2], vec{0, 0, 0, 0});
                        x = Mf::mma(ainv[1], acc[i][1], x);
                        x2 = Mf::mma(ainv[3], acc[i][3], x2);
                        x += x2;
                        acc[i] = x;                                                   // R(bi, bj), kept for the backward pass
                        *reinterpret_cast<vec *>(Xw + bj * 256 + lane * 4) = x;
                    }
                }
            });
        }
        __syncthreads();
        // c. trailing tiles  N(bi, bj) += X_bi^T X_bj  (bi > kbk), forward substitution of the later blocks
        if (kbk + 1 < GK_NB) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                typedef T vec2 __attribute__((ext_vector_type(2)));
                vec2 xo[GK_NB];
#pragma unroll
                for (int b = 1; b < GK_NB; b++) xo[b] = *reinterpret_cast<const vec2 *>(Xw + b * 256 + lane * 4 + 2 * h);
                static_for<0, NTQ>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int bi = gc_bi(Q, i), bj = gc_bj(Q, i);
                    if constexpr (bi > 0) {
                        if (kbk < bi) {
                            acc[i] = Mf::mma(xo[bi][0], xo[bj][0], acc[i]);
                            acc[i] = Mf::mma(xo[bi][1], xo[bj][1], acc[i]);
                        }
                    }
                });
            }
            // rhs_j -= X_j^T y_k  for the later blocks: thread <-> unknown (element (k2, c) of tile b sits at
            // [b][16 (k2 >> 2) + c][k2 & 3])
#pragma unroll
            for (int rep = 0; rep < 2; rep++) {
                const int jg = tid + 256 * rep;
                if (jg >= 16 * (kbk + 1) && jg < 16 * GK_NB) {
                    T sacc = S.rhs[jg];
                    const T *xt = Xw + (jg >> 4) * 256 + (jg & 15) * 4;
#pragma unroll
                    for (int k2 = 0; k2 < 16; k2++) sacc -= xt[(k2 >> 2) * 64 + (k2 & 3)] * S.rhs[16 * kbk + k2];
                    S.rhs[jg] = sacc;
                }
            }
        }
    }
    // ---- 3. backward substitution  R x = y  by block rows:  x_i = inv(R_ii) (y_i - sum_{j > i} R_ij x_j) ----
    T xs[5];
#pragma unroll
    for (int c = 0; c < 5; c++) xs[c] = T(0);
    static_for<0, GK_NB>([&](auto sc) {
        constexpr int bi = GK_NB - 1 - decltype(sc)::value;
        T p0 = T(0), p1 = T(0), p2 = T(0), p3 = T(0);
        static_for<0, NTQ>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (gc_bi(Q, i) == bi && gc_bj(Q, i) > bi) {
                const T xv = xs[gc_xslot(Q, gc_bj(Q, i))];
                p0 += acc[i][0] * xv; p1 += acc[i][1] * xv; p2 += acc[i][2] * xv; p3 += acc[i][3] * xv;
            }
        });
        // sum over the 16 lanes of a row for the four registers at once: after two select-and-exchange steps lane l carries
        // register (l & 3), then two plain butterflies
        const bool o1 = (lm & 1) != 0, o2 = (lm & 2) != 0;
        const T s01 = (o1 ? p1 : p0) + lanes::xor1(o1 ? p0 : p1);
        const T s23 = (o1 ? p3 : p2) + lanes::xor1(o1 ? p2 : p3);
        T sr = (o2 ? s23 : s01) + lanes::xor2(o2 ? s01 : s23);
        sr += lanes::xor4(sr);
        sr += lanes::xor8(sr);
        T *ps = &S.psum[bi & 1][0][0];
        if (lm < 4) ps[Q * 16 + Mf::row_of(lane, lm)] = sr;
        // Round 6 (measured first in chol_wg8_kernel, where this loop as 96 LDS reads per block row cost more than the whole
        // factorisation's matrix instructions): the row of inv(R_ii) this lane multiplies with is on its way while the partial
        // sums meet at the barrier; t = y_i - sum_j R_ij x_j sits with element lm in lane lm of every 16-lane row and is
        // broadcast inside the row by DPP instead of being re-read from LDS sixteen times.  Same sums in the same order.
        const T *rslot = S.rinv + bi * RSZ;
        T ri[16];
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++) ri[n2] = rslot[lm * LDR + n2];
        __syncthreads();
        const T tv = S.rhs[16 * bi + lm] - ((ps[lm] + ps[16 + lm]) + (ps[32 + lm] + ps[48 + lm]));
        T xm = T(0);                              // x[16 bi + lm], computed redundantly by every 16-lane group of every wave
        static_for<0, 16>([&](auto nc) {
            constexpr int n2 = decltype(nc)::value;
            xm += ri[n2] * lanes::row_bcast16<n2>(tv);
        });
        if constexpr (gc_has_col(Q, bi)) xs[gc_xslot(Q, bi)] = xm;
        if (Q == 0 && lane < 16) S.xall[16 * bi + lane] = xm;
    });
    if (Q == 0)
        for (int t = lane; t < GK_NB * 16; t += 64)
            if (gk_unknown(t) < kt) arow[gk_unknown(t)] = S.xall[t];
}

// One workgroup of four wavefronts per row; rows [P.row_first, P.nrows) of the processing order handed out by P.counter.
template <typename T>
__global__ void __launch_bounds__(256, 2)
gramk_consumer_kernel(const CholParams<T> P)
{
    __shared__ GcShared<T> S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kt = P.kt;
    for (;;) {
        if (tid == 0) S.rix = P.row_first + atomicAdd(P.counter, 1);
        __syncthreads();
        const int rix = S.rix;
        __syncthreads();                            // S.rix may be rewritten; the previous row's LDS readers are done
        if (rix >= P.nrows) break;
        const int row = (P.order != nullptr) ? P.order[rix] : rix;
        const int nnz1 = (int)(P.indptr[row + 1] - P.indptr[row]);
        T *arow = P.A + (size_t)row * P.lda;
        const bool coll = (P.mode == CHOL_COLLECTIVE);
        GcRow R;
        R.kt = kt;
        R.has_u = coll && row < P.rows_with_u;
        if (coll && nnz1 == 0 && !R.has_u) {                            // collective.c:1258-1268, :1876-1885
            for (int e = tid; e < kt; e += 256) arow[e] = T(0);
            continue;
        }
        T lam = P.lam, lam_last = P.lam_last;
        if (P.mode == CHOL_EXPLICIT) {
            if (P.scale_lam) {                                           // common.c:679-723
                const T mult = (P.wsum != nullptr) ? P.wsum[row] : (T)nnz1;
                lam *= mult;
                if (!P.scale_bias_const) lam_last *= mult;
            }
        } else if (P.scale_lam || P.scale_lam_sideinfo) {               // collective.c:1285-1355
            T mult = (P.wsum != nullptr) ? P.wsum[row] : ((nnz1 > 0) ? (T)nnz1 : T(1));
            if (P.scale_lam_sideinfo && R.has_u) mult += (T)P.p_side;    // :1338-1346
            lam *= mult;
            if (R.has_u || !P.scale_bias_const) lam_last *= mult;
        }
        if (nnz1 <= 0) { R.item0 = R.item1 = 0; }                       // rows without entries have no partial
        else if (rix < P.gk_n_heavy) { R.item0 = P.gk_row_off[rix]; R.item1 = P.gk_row_off[rix + 1]; }
        else { R.item0 = P.gk_n_slices + (rix - P.gk_n_heavy); R.item1 = R.item0 + 1; }
        switch (wave) {
            case 0: gc_row<T, 0>(P, S, R, lam, lam_last, arow, lane); break;
            case 1: gc_row<T, 1>(P, S, R, lam, lam_last, arow, lane); break;
            case 2: gc_row<T, 2>(P, S, R, lam, lam_last, arow, lane); break;
            default: gc_row<T, 3>(P, S, R, lam, lam_last, arow, lane); break;
        }
    }
}

// out[t][lane][r]: the [lane][register] form of tile_pack_kernel's output, in the producer's order of the unknowns (gk_unknown):
// the layout of its partials
template <typename T>
__global__ void tile_pack_lane_kernel(const T *__restrict__ M, int lim, int NB, T *__restrict__ out)
{
    const int NT = NB * (NB + 1) / 2;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NT * 256) return;
    const int t = e >> 8, lane = (e >> 2) & 63, r = e & 3;
    int bi = 0, rem = t;
    while (rem >= NB - bi) { rem -= NB - bi; bi++; }
    const int bj = bi + rem;
    const int gi = gk_unknown(16 * bi + CholMfma<T>::row_of(lane, r)), gj = gk_unknown(16 * bj + (lane & 15));
    const int lo = min(gi, gj), hi = max(gi, gj);
    out[e] = (hi < lim) ? M[(size_t)lo * lim + hi] : T(0);
}

}  // namespace cmfhip
