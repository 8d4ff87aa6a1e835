// gram_cg_kernels.hpp -- CG row update of the very heavy rows through the row's own Gramian (k_t <= 64).
//
// A row with thousands of non-zeros does not fit a team's registers, so the tiled kernels of
// cg_kernels.hpp would have to re-stream its gathered rows for every one of the max_cg_steps+1 passes
// (4.9x the algorithmic HBM bytes measured on the split-row path).  The operator of the reference's CG,
//     implicit (factors_implicit_cg, src/common.c:1914-1986):  Ap = (BtB + lam I) p + sum_j x_j (B_j.p) B_j
//     explicit (factors_explicit_cg, src/common.c:1098-1188):  Ap = diag(lam..lam_last) p + sum_j (B_j.p) B_j
// is linear in the k_t x k_t matrix  G = sum_j w_j B_j B_j^T  (w = x | 1), so for these rows the gathered rows
// are read ONCE:
//   gram_slice_kernel : a slice (<= GRAM_SLICE non-zeros) of a row per workgroup; G on the matrix cores
//                       (same staging ring / C-layout tiles as chol_kernels.hpp) and the residual's vector
//                       part  v = sum_j (x_j - B_j.a) B_j  (implicit, quirk Q1)  |  sum_j x_j B_j  (explicit)
//                       -> per-slice partials in HBM (no floating-point atomics);
//   gram_cg_kernel    : per row, partials summed in slice order, M = G + (BtB + lam I | diag) in LDS,
//                       r = v - M a, then the reference's CG steps with dense matrix-vector products,
//                       same absolute thresholds (1e-12 / 1e-8).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cg_kernels.hpp"
#include "chol_kernels.hpp"
#include "lanes.hpp"

namespace cmfhip {

constexpr int GRAM_NTT = 4;                              // k_t <= 64
constexpr int GRAM_NTILES = GRAM_NTT * (GRAM_NTT + 1) / 2;
constexpr int GRAM_PART = GRAM_NTILES * 256 + 64;        // elements of one slice partial: tiles (C layout) + v
constexpr int GRAM_CH = 32;                              // gathered rows per staging round
constexpr int GRAM_NW = 4;                               // wavefronts per workgroup

template <typename T>
struct GramParams {
    const int *sl_vrow;        // slice -> ordinal of its row in the very-heavy bin (= position in desc)
    const int *sl_first;       // slice -> first non-zero inside the row
    const int *sl_count;       // slice -> number of non-zeros
    const int *row_sl_off;     // [nvh + 1] slices of a row are contiguous
    T *part;                   // [n_slices][GRAM_PART]
    int n_slices, nvh;
#ifdef CMF_CG_DEBUG
    unsigned long long *ticks = nullptr;   // [4] timing experiment: shader-clock ticks of lane 0 of every wave summed, waves, slices, non-zeros
#endif
};

// sums of c[0..7] over the 64 lanes in 10 exchanges: afterwards lane l holds the total of c[l >> 3]
template <typename T>
__device__ __forceinline__ T treduce8_rows(const T (&c)[8], int lane)
{
    T n4[4], n2[2];
#pragma unroll
    for (int i = 0; i < 4; i++) n4[i] = lanes::tswap32_add(c[i], c[4 + i]);
#pragma unroll
    for (int i = 0; i < 2; i++) n2[i] = lanes::tswap16_add(n4[i], n4[2 + i]);
    T n1 = ((lane & 8) ? n2[1] : n2[0]) + lanes::recv_xor8(n2[0], n2[1]);
    n1 += lanes::xor4(n1);
    n1 += lanes::xor2(n1);
    n1 += lanes::xor1(n1);
    return n1;
}

template <typename T, bool IMPLICIT>
__global__ void __launch_bounds__(64 * GRAM_NW, 2)
gram_slice_kernel(const CgParams<T> P, const GramParams<T> Gp)
{
    using Mf = CholMfma<T>;
    using vec = typename Mf::vec;
    constexpr int NTT = GRAM_NTT, NW = GRAM_NW, CH = GRAM_CH;
    constexpr int ldc = 16 * NTT + 16;                        // 80 == 16 (mod 32)
    constexpr int NTALL = GRAM_NTILES;
    constexpr int TPW = (NTALL + NW - 1) / NW;
    constexpr int RPW = CH / NW;
    static_assert(RPW == 8, "treduce8_rows");
    __shared__ T ring[2 * CH * ldc];
    __shared__ T wsc[2 * CH];
    __shared__ T vpart[NW * 64];
    const int kt = P.k;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 15;
    int offa[TPW], offb[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; tt++) {
        const int t = min(wave + NW * tt, NTALL - 1);         // idle slots recompute the last tile, never stored
        offa[tt] = 16 * tile_bi(t, NTT);
        offb[tt] = 16 * tile_bj(t, NTT);
    }
    const int scol = (lane < kt) ? lane : 0;
    const bool svalid = lane < kt;
    const int myrow = lane >> 3;                              // lane 8i owns the weights of staged row i of this wave

    for (int sl = blockIdx.x; sl < Gp.n_slices; sl += gridDim.x) {
        const RowDesc d = P.desc[Gp.sl_vrow[sl]];
        const size_t st = d.st + (size_t)Gp.sl_first[sl];
        const int nnz = Gp.sl_count[sl];
        const T *arow = P.A + (size_t)d.row * P.lda;
        const T a_l = (IMPLICIT && svalid) ? arow[lane] : T(0);
        vec acc[TPW];
#pragma unroll
        for (int tt = 0; tt < TPW; tt++) acc[tt] = vec{0, 0, 0, 0};
        T racc = T(0);                                        // v[lane], this wave's rows only
        T pre[RPW];
        int idn[RPW];
        int idx_l = 0; T x_l = T(0), xw_l = T(0), g_l = T(1);   // g_l: observation weight of the entry (explicit model)
        T gw_l = T(1);
        int nr_rows = 0, nr_idx = 0;
        auto load_idx = [&](int c0) {
            nr_idx = min(CH, nnz - c0);
#pragma unroll
            for (int i = 0; i < RPW; i++) idn[i] = P.indices[st + c0 + min(RPW * wave + i, nr_idx - 1)];
            const size_t e = st + c0 + min(RPW * wave + myrow, nr_idx - 1);
            idx_l = P.indices[e]; x_l = P.values[e]; g_l = entry_weight<T, IMPLICIT>(P, e);
        };
        auto load_rows = [&]() {
            nr_rows = nr_idx;
#pragma unroll
            for (int i = 0; i < RPW; i++) pre[i] = P.B[(size_t)idn[i] * P.ldb + scol];
            xw_l = x_l;
            if (!IMPLICIT && P.bias_sub != nullptr) xw_l -= P.bias_sub[idx_l];
            gw_l = g_l;
            if (!IMPLICIT) xw_l *= g_l;                                   // right-hand side weight g x (tgemv_dense_sp_weighted)
        };
        if (nnz > 0) { load_idx(0); load_rows(); }
        if (nnz > CH) load_idx(CH);
        __syncthreads();                      // previous slice's readers are done
        for (int c0 = 0, slot = 0; c0 < nnz; c0 += CH, slot ^= 1) {
            T *Bs = ring + slot * CH * ldc;
            T bv[RPW], prod[RPW];
#pragma unroll
            for (int i = 0; i < RPW; i++) {
                bv[i] = (RPW * wave + i < nr_rows && svalid) ? pre[i] : T(0);
                Bs[(RPW * wave + i) * ldc + lane] = bv[i];
                if (lane < ldc - 64) Bs[(RPW * wave + i) * ldc + 64 + lane] = T(0);
                prod[i] = bv[i] * a_l;
            }
            // weights of this wave's rows (lane 8i: row i):  G weight x | 1,  v weight x - B_j.a | x
            const bool live_l = (RPW * wave + myrow < nr_rows);
            T wv = live_l ? xw_l : T(0);
            if (IMPLICIT) wv -= treduce8_rows(prod, lane);                // common.c:1936-1943 (0 for padded rows)
            if ((lane & 7) == 0) wsc[slot * CH + RPW * wave + myrow] = live_l ? (IMPLICIT ? xw_l : gw_l) : T(0);
#pragma unroll
            for (int i = 0; i < RPW; i++) racc += bcast_lane(wv, 8 * i) * bv[i];
            __syncthreads();
            if (c0 + CH < nnz) load_rows();
            if (c0 + 2 * CH < nnz) load_idx(c0 + 2 * CH);
#pragma unroll
            for (int q = 0; q < CH / 4; q++) {
                const int rr = 4 * q + (lane >> 4);
                const T *brow = Bs + rr * ldc + lm;
                const T w = wsc[slot * CH + rr];
                T opa[TPW], opb[TPW];
#pragma unroll
                for (int tt = 0; tt < TPW; tt++) { opa[tt] = brow[offa[tt]]; opb[tt] = brow[offb[tt]]; }
#pragma unroll
                for (int tt = 0; tt < TPW; tt++) acc[tt] = Mf::mma(opa[tt] * w, opb[tt], acc[tt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        T *out = Gp.part + (size_t)sl * GRAM_PART;
#pragma unroll
        for (int tt = 0; tt < TPW; tt++) {
            const int t = wave + NW * tt;
            if (t < NTALL) {
#pragma unroll
                for (int r = 0; r < 4; r++) out[t * 256 + r * 64 + lane] = acc[tt][r];
            }
        }
        __syncthreads();                      // vpart of the previous slice has been read
        vpart[wave * 64 + lane] = racc;
        __syncthreads();
        if (tid < 64) {
            T vs = T(0);
#pragma unroll
            for (int w2 = 0; w2 < NW; w2++) vs += vpart[w2 * 64 + tid];
            out[NTALL * 256 + tid] = (tid < kt) ? vs : T(0);
        }
    }
}

// The same slice partial from ONE WAVEFRONT per slice, without LDS and without barriers: the 4 x 16 operand of
// v_mfma_*_16x16x4 (lane l: row l / 16 of the slab, column l % 16 of the block) is exactly what a coalesced gather delivers
// -- 16 lanes read 16 consecutive columns of one gathered row, 4 rows per instruction -- so a slab of 4 gathered rows x 64
// columns lands in 4 VGPRs and feeds all 10 tiles of the upper triangle (A operand: the slab scaled by the row weights, B
// operand: the slab itself).  Per slab: 4 loads, ~25 VALU instructions (B_j . a for the residual's weights, the weighted
// row sum v, the scaling), 10 MFMAs -- the matrix pipe is the limit (single precision: 320 of its cycles per 4 non-zeros,
// 20 cycles per non-zero and CU), and the rows are read once instead of max_cg_steps + 1 times.  Groups of 8 slabs are
// double-buffered in registers.  Output layout = gram_slice_kernel's, so gram_cg_kernel consumes either.
#ifndef GW_SLABS_F32
#define GW_SLABS_F32 4      // 120 VGPRs: four wavefronts per SIMD (8: 168 VGPRs, three; 16: 202, two -- measured 1.30 / 1.39 / 1.95 ms on c4shard's item rows)
#endif
#ifndef GW_SLABS_F64
#define GW_SLABS_F64 4
#endif
template <typename T> constexpr int gw_slabs() { return sizeof(T) == 4 ? GW_SLABS_F32 : GW_SLABS_F64; }     // slabs per prefetch group

// REM > 0 (round 3, double precision): only the first NTF = NTT - 1 column blocks go through the matrix cores; the REM columns
// 16 NTF .. kt - 1 (at most GW_REM of them: k = 50 has two) are VALU products -- lane (kc, lm) takes the column's element of its
// slab row from lane (kc, column - 16 NTF) with one row_newbcast move and adds  w e_j B_j[16 cb + lm]  for its four blocks.
// In double precision an MFMA occupies the pipe as long as ~30 vector FMAs and the last column block of k = 50 holds 2 live
// columns of 16: 4 of the 10 tiles per slab are replaced by 12 vector instructions.  Their sums land in the partial at the
// positions gram_cg_kernel reads (tile (cb, NTT - 1), column j - 16 (NTT - 1)); the rest of those tiles stays unwritten and unread.
// FULLK (round 4): k_t = 16 NTT exactly (config 4: k = 64) -- the four loads of a slab row are one address plus immediate offsets
// (the clamped column offsets of the general build cost a 64-bit multiply-add per load, quarter rate: 16 of them per group of four
// slabs, 256 of the ~900 vector cycles beside 1280 matrix cycles).  Both builds keep the two slab buffers in place and swap their
// roles (the loop body twice per trip) instead of copying `nxt` into `cur`.
constexpr int GW_REM = 4;
template <typename T, bool IMPLICIT, int REM = 0, bool FULLK = false>
__global__ void __launch_bounds__(256, 2)
gram_wave_kernel(const CgParams<T> P, const GramParams<T> Gp)
{
    using Mf = CholMfma<T>;
    using vec = typename Mf::vec;
    constexpr int NTT = GRAM_NTT, NTALL = GRAM_NTILES, NS = gw_slabs<T>(), GRP = 4 * NS;
    constexpr bool REMV = REM > 0;
    constexpr int NTF = REMV ? NTT - 1 : NTT;                 // the host launches REM = kt - 16 (NTT - 1)
    static_assert(REM >= 0 && REM <= GW_REM, "remainder columns");
    const int kt = P.k;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kc = lane >> 4, lm = lane & 15;                // slab row of this lane, column inside a 16-column block
    // column 16 cb + lm; past kt: column kt - 1 again (finite values; gram_cg_kernel ignores those rows / columns of G and v)
    size_t coff[NTT];
    bool cok[NTT];
#pragma unroll
    for (int cb = 0; cb < NTT; cb++) { cok[cb] = (16 * cb + lm) < kt; coff[cb] = (size_t)min(16 * cb + lm, kt - 1); }
    const unsigned ldb_bytes = (unsigned)(P.ldb * sizeof(T));
#ifdef CMF_CG_DEBUG
    const unsigned long long tick0 = __builtin_readcyclecounter();
    unsigned long long nsl_ = 0, nnz_ = 0;
#endif

    for (int sl = blockIdx.x * 4 + wave; sl < Gp.n_slices; sl += gridDim.x * 4) {
        const RowDesc d = P.desc[Gp.sl_vrow[sl]];
        const size_t st = d.st + (size_t)Gp.sl_first[sl];
        const int nnz = Gp.sl_count[sl];
#ifdef CMF_CG_DEBUG
        nsl_++; nnz_ += nnz;
#endif
        T a_c[NTT];
#pragma unroll
        for (int cb = 0; cb < NTT; cb++) a_c[cb] = (IMPLICIT && cok[cb]) ? P.A[(size_t)d.row * P.lda + coff[cb]] : T(0);
        vec acc[NTALL];
#pragma unroll
        for (int t = 0; t < NTALL; t++) acc[t] = vec{0, 0, 0, 0};
        T racc[NTT];
#pragma unroll
        for (int cb = 0; cb < NTT; cb++) racc[cb] = T(0);
        T cacc[REMV ? REM : 1][NTT];                          // remainder columns: G[16 cb + lm][16 NTF + c], this lane's slab rows
#pragma unroll
        for (int c = 0; c < (REMV ? REM : 1); c++)
#pragma unroll
            for (int cb = 0; cb < NTT; cb++) cacc[c][cb] = T(0);

        // entry (lane & 31) of a group: index, value (explicit: minus the fused bias), 1 / 0 for entries past the slice
        // (explicit model with observation weights: okf carries the entry's weight g, so G = sum g B_j B_j^T and
        //  v = sum g x B_j -- common.c:1007-1012, tgemv_dense_sp_weighted)
        auto load_entries = [&](int c0, int &idx, T &x, T &okf) {
            const int e = c0 + (lane & (GRP - 1));
            const bool ok = e < nnz;
            const size_t pos = st + (size_t)max(min(e, nnz - 1), 0);
            idx = P.indices[pos];
            x = P.values[pos];
            if (!IMPLICIT && P.bias_sub != nullptr) x -= P.bias_sub[idx];
            okf = ok ? entry_weight<T, IMPLICIT>(P, pos) : T(0);
        };
        auto load_slabs = [&](int idx, T (&slab)[NS][NTT]) {
#pragma unroll
            for (int q = 0; q < NS; q++) {
                const unsigned it = (unsigned)__shfl(idx, 4 * q + kc);
                const T *rp = reinterpret_cast<const T *>(reinterpret_cast<const char *>(P.B) + (unsigned long long)it * ldb_bytes);
                if constexpr (FULLK) {
                    const T *rl = rp + lm;
#pragma unroll
                    for (int cb = 0; cb < NTT; cb++) slab[q][cb] = rl[16 * cb];
                } else if constexpr (REMV) {
                    // (round 5) 16 NTF <= kt - 1: the full blocks are one address plus immediate offsets like the FULLK build, only
                    // the remainder block needs its clamped column
                    const T *rl = rp + lm;
#pragma unroll
                    for (int cb = 0; cb < NTF; cb++) slab[q][cb] = rl[16 * cb];
                    slab[q][NTT - 1] = rp[coff[NTT - 1]];
                } else {
#pragma unroll
                    for (int cb = 0; cb < NTT; cb++) slab[q][cb] = rp[coff[cb]];
                }
            }
        };
        int idx_c, idx_n;
        T x_c, ok_c, x_n, ok_n;
        T bufA[NS][NTT], bufB[NS][NTT];
        // (every load below is unconditional -- entries past the slice re-read its last one with weight 0 -- so that the
        //  loop-carried registers are plain copies: conditional loads into live registers make the compiler keep both sets)
        load_entries(0, idx_c, x_c, ok_c);
        load_slabs(idx_c, bufA);
        load_entries(GRP, idx_n, x_n, ok_n);
        // one group of NS slabs: its products from S_, the next group's rows into N_ (they land behind this group's MFMAs)
        auto group = [&](int c0, T (&S_)[NS][NTT], T (&N_)[NS][NTT]) {
            load_slabs(idx_n, N_);
            int idx_nn; T x_nn, ok_nn;
            load_entries(c0 + 2 * GRP, idx_nn, x_nn, ok_nn);
#pragma unroll
            for (int q = 0; q < NS; q++) {
                const T x = __shfl(x_c, 4 * q + kc), okf = __shfl(ok_c, 4 * q + kc);
                T wG, wv;
                if (IMPLICIT) {
                    T dp = T(0);
#pragma unroll
                    for (int cb = 0; cb < NTT; cb++) dp += S_[q][cb] * a_c[cb];
                    dp += lanes::xor1(dp); dp += lanes::xor2(dp); dp += lanes::qxor4(dp); dp += lanes::xor8(dp);   // B_j . a, 16 lanes
                    wG = x * okf;                                              // common.c:1965
                    wv = (x - dp) * okf;                                       // common.c:1936-1943 (quirk Q1)
                } else {
                    wG = okf;
                    wv = x * okf;
                }
                T sa[NTT];
#pragma unroll
                for (int cb = 0; cb < NTT; cb++) {
                    racc[cb] += wv * S_[q][cb];
                    sa[cb] = wG * S_[q][cb];
                }
                static_for<0, NTALL>([&](auto tc) {
                    constexpr int t = decltype(tc)::value, bi = tile_bi(t, NTT), bj = tile_bj(t, NTT);
                    if constexpr (bj < NTF) acc[t] = Mf::mma(sa[bi], S_[q][bj], acc[t]);
                });
                if constexpr (REMV) {
                    static_for<0, REM>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        const T we = wG * lanes::row_bcast16<c>(S_[q][NTT - 1]);
#pragma unroll
                        for (int cb = 0; cb < NTT; cb++) cacc[c][cb] += we * S_[q][cb];
                    });
                }
            }
            idx_c = idx_n; x_c = x_n; ok_c = ok_n;
            idx_n = idx_nn; x_n = x_nn; ok_n = ok_nn;
        };
        for (int c0 = 0; c0 < nnz; c0 += 2 * GRP) {
            group(c0, bufA, bufB);
            if (c0 + GRP < nnz) group(c0 + GRP, bufB, bufA);
        }
        T *out = Gp.part + (size_t)sl * GRAM_PART;
        static_for<0, NTALL>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (tile_bj(t, NTT) < NTF) {
#pragma unroll
                for (int r = 0; r < 4; r++) out[t * 256 + r * 64 + lane] = acc[t][r];
            }
        });
        if constexpr (REMV) {
            // element (16 cb + lm, 16 NTF + c): summed over the four slab rows (16 lanes apart), stored where tile (cb, NTT - 1) keeps it
            static_for<0, REM>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                static_for<0, NTT>([&](auto cbc) {
                    constexpr int cb = decltype(cbc)::value;
                    // packed index of tile (cb, NTT - 1): sum_{b < cb} (NTT - b) + (NTT - 1 - cb)
                    constexpr int t = cb * NTT - cb * (cb - 1) / 2 + (NTT - 1 - cb);
                    T v = lanes::tswap32_add(cacc[c][cb], cacc[c][cb]);
                    v = lanes::tswap16_add(v, v);
                    if (kc == 0) out[t * 256 + Mf::cidx(lm, c)] = v;
                });
            });
        }
        // v: the four slab rows of a column sit 16 lanes apart
#pragma unroll
        for (int cb = 0; cb < NTT; cb++) {
            T v = lanes::tswap32_add(racc[cb], racc[cb]);
            v = lanes::tswap16_add(v, v);
            if (kc == 0) out[NTALL * 256 + 16 * cb + lm] = cok[cb] ? v : T(0);
        }
    }
#ifdef CMF_CG_DEBUG
    if (Gp.ticks != nullptr && lane == 0) {
        atomicAdd(&Gp.ticks[0], __builtin_readcyclecounter() - tick0);
        atomicAdd(&Gp.ticks[1], 1ull); atomicAdd(&Gp.ticks[2], nsl_); atomicAdd(&Gp.ticks[3], nnz_);
    }
#endif
}

// one workgroup (256 threads) per very heavy row; wave 0 runs the CG
template <typename T, bool IMPLICIT>
__global__ void __launch_bounds__(256)
gram_cg_kernel(const CgParams<T> P, const GramParams<T> Gp)
{
    using Mf = CholMfma<T>;
    constexpr int NTT = GRAM_NTT, NTALL = GRAM_NTILES, LDM = 65;
    __shared__ T M[64 * LDM];
    __shared__ T vec_s[64];                   // the vector a dense product is taken with
    const int kt = P.k;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int vr = blockIdx.x; vr < Gp.nvh; vr += gridDim.x) {
        const RowDesc d = P.desc[vr];
        T *arow = P.A + (size_t)d.row * P.lda;
        const int s0 = Gp.row_sl_off[vr], s1 = Gp.row_sl_off[vr + 1];
        T lam = P.lam, lam_last = P.lam_last;
        if (!IMPLICIT && P.scale_lam) {                       // common.c:679-723
            const T mult = row_lam_mult(P, d.row, d.nnz);
            lam *= mult;
            if (!P.scale_bias_const) lam_last *= mult;
        }
        __syncthreads();                      // previous row's CG is done with M
        // columns kt .. 63 of M stay zero: the dense products below run over all 64 columns, four partial sums at a time
        if (kt < 64)
            for (int e = tid; e < 64 * LDM; e += 256) M[e] = T(0);
        if (kt < 64) __syncthreads();
        // G = sum of the slice partials, in slice order;  M = G + (BtB + lam I | diag(lam .. lam_last)), both triangles.
        // Slices outermost, the thread's ten elements inside: 80 independent loads per round, so that the most popular rows
        // (hundreds of slices: a chain of hundreds of round trips per element otherwise) do not become the tail of the launch.
        constexpr int EPT = NTALL * 256 / 256, SB = 4;     // slices per round (40 loads in flight per thread; more would cost resident workgroups)
        T sacc[EPT];
#pragma unroll
        for (int q = 0; q < EPT; q++) sacc[q] = T(0);
        for (int sl = s0; sl < s1; sl += SB) {
            T t8[EPT][SB];
#pragma unroll
            for (int q = 0; q < EPT; q++)
#pragma unroll
                for (int u = 0; u < SB; u++) t8[q][u] = (sl + u < s1) ? Gp.part[(size_t)(sl + u) * GRAM_PART + tid + 256 * q] : T(0);
#pragma unroll
            for (int q = 0; q < EPT; q++)
#pragma unroll
                for (int u = 0; u < SB; u++) sacc[q] += t8[q][u];
        }
#pragma unroll
        for (int q = 0; q < EPT; q++) {
            const int e = tid + 256 * q;
            T s = sacc[q];
            const int t = e >> 8, r = (e >> 6) & 3, l = e & 63;
            const int i = 16 * tile_bi(t, NTT) + Mf::row_of(l, r), j = 16 * tile_bj(t, NTT) + (l & 15);
            if (j >= i && j < kt) {           // diagonal tiles: the upper half only, mirrored
                if (IMPLICIT) s += P.BtB[(size_t)i * kt + j];
                if (i == j) s += (!IMPLICIT && i == kt - 1) ? lam_last : lam;
                M[i * LDM + j] = s;
                M[j * LDM + i] = s;
            }
        }
        T v = T(0);
        if (tid < 64) {
            for (int sl = s0; sl < s1; sl += 8) {
                T t8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) t8[u] = (sl + u < s1) ? Gp.part[(size_t)(sl + u) * GRAM_PART + NTALL * 256 + tid] : T(0);
#pragma unroll
                for (int u = 0; u < 8; u++) v += t8[u];
            }
        }
        __syncthreads();
        if (tid < 64) {                       // one wavefront: lane e <-> unknown e
            const bool live = lane < kt;
            T a = live ? arow[lane] : T(0);
            auto mul = [&](T x) {             // (M x)[lane];  x is zero on the lanes >= kt
                vec_s[lane] = x;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                T y0 = T(0), y1 = T(0), y2 = T(0), y3 = T(0);
                const T *mrow = M + lane * LDM;
#pragma unroll
                for (int c = 0; c < 64; c += 4) {
                    y0 += mrow[c] * vec_s[c]; y1 += mrow[c + 1] * vec_s[c + 1];
                    y2 += mrow[c + 2] * vec_s[c + 2]; y3 += mrow[c + 3] * vec_s[c + 3];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                return (y0 + y1) + (y2 + y3);
            };
            // (every lane takes part in `mul`: the lanes >= kt hand in zeros, so that no element of vec_s is ever read before
            //  it was written -- 0 x whatever the LDS held is NaN when that happens to be a NaN pattern, and a NaN residual
            //  skips the CG below and leaves the row at its start value)
            const T Ma = mul(a);
            T r = live ? v - Ma : T(0);       // common.c:1932-1943 / :1128-1139
            T p = r;
            T r_old = lanes::wave_sum(r * r);
            if (r_old > (T)1e-12) {           // :1952 / :1147
                for (int step = 0; step < P.max_cg_steps; step++) {
                    const T Mp = mul(p);
                    const T Ap = live ? Mp : T(0);
                    const T alpha = cg_div(r_old, lanes::wave_sum(Ap * p));
                    a += alpha * p; r -= alpha * Ap;
                    const T r_new = lanes::wave_sum(r * r);
                    if (r_new <= (T)1e-8) break;              // :1979 / :1180
                    p = p * cg_div(r_new, r_old) + r;
                    r_old = r_new;
                }
            }
            if (live) arow[lane] = a;
        }
    }
}

}  // namespace cmfhip
