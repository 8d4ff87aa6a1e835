// chol_wg_kernels.hpp -- the factorisation of the eight-block closed-form rows (k_t = 97 .. 129, double precision: config 3) by
// ONE WORKGROUP OF TWO (or four) WAVEFRONTS per row (round 6).
//
// Same arithmetic as chol_wave_kernel's second build (WMODE 2: the row's rank-k update has been done by the producer kernels, the
// partials wait in HBM) -- collective_closed_form_block /root/reference/src/collective.c:1534-1846 (dposv at :1823),
// factors_closed_form /root/reference/src/common.c:978-1070 -- but another mapping.  That build keeps the 36 tiles of the upper
// triangle in ONE wavefront: 288 accumulator registers + the factorisation's temporaries = one wavefront per SIMD with 263
// spilled registers (0.94-1.12 ms per launch, 4.75 ms of config 3's 14.2 ms per iteration).  Here
//   * the tiles are dealt to the NW wavefronts of a workgroup -- tile (bi, bj) belongs to wavefront (bi + bj) mod NW -- so the
//     panel tiles of every block row AND the trailing tiles of every step are spread over the wavefronts;
//   * the accumulators hold the NEGATED matrix (the trailing update is a plain N += X_bi^T X_bj), the panel tiles of a step
//     travel through LDS (lane-linear: conflict-free 8-byte accesses), inv(R_kk) has a slot per block: two barriers per block
//     step, the scheme of gramk_consumer_kernel (single precision, 17 blocks);
//   * right-hand side and border column (k_t = 16 n + 1: the bias column stays outside the tiles, chol_wave_kernels.hpp) are
//     forward-substituted thread <-> unknown while the trailing MFMAs run; the backward substitution goes by block rows with
//     one cross-lane reduction and one barrier per row, the row of inv(R_ii) prefetched and the right-hand side broadcast by DPP.
// What the measurements said (profiles/r06/r06_f, r06_g, r06_j: phases left out one at a time): the 450 matrix instructions of a
// row are a small part of its time (0.35 of 6.5 ms per iteration in the first version); the CHAIN is -- eight diagonal blocks of
// sixteen dependent pivots each (2.3-2.7 ms) and, until it was rewritten, the backward substitution's 96 LDS reads per block row
// (3.1 -> 0.7 ms).  A chain is bought back with rows in flight: four wavefronts per row = two rows per CU (registers: 10 tiles
// per wavefront, 232 VGPRs) ran at 14.0 ms, TWO wavefronts per row = FOUR rows per CU (20 / 16 tiles per wavefront, 256 VGPRs,
// 37 KB of LDS per workgroup) at 13.0 ms against 14.2 for the one-wavefront kernel -- the default.  Three workgroups of four
// wavefronts (168 registers, spills) were slower (15.5).
// Input: the partials of chol_wave_kernel's producer build / chol_parts_producer_kernel (tiles in the accumulator layout,
// [t][r][lane]; right-hand side; border column; two scalars) and the launch's initial matrices in the same layout
// (tile_pack_kernel).  Output: the row of A.
#pragma once
#include "chol_wave_kernels.hpp"

namespace cmfhip {

constexpr int WG8_NB = 8, WG8_NT = 36;
#ifndef CMF_WG8_WGS
#define CMF_WG8_WGS 2          // workgroups (rows) per CU the register budget is set for: 2 -> 256 registers, 3 -> 168 (a few spills; LDS 3 x 53 KB)
#endif
// NW wavefronts per row (4, or 2: more rows per CU in flight, round 6): tile (bi, bj) belongs to wavefront (bi + bj) mod NW
__host__ __device__ constexpr int wg8_owner(int NW, int bi, int bj) { return (bi + bj) & (NW - 1); }
__host__ __device__ constexpr int wg8_count(int NW, int Q)
{
    int c = 0;
    for (int t = 0; t < WG8_NT; t++) c += (wg8_owner(NW, tile_bi(t, WG8_NB), tile_bj(t, WG8_NB)) == Q) ? 1 : 0;
    return c;
}
// packed index (tile_bi / tile_bj order) of the i-th tile of wavefront Q
__host__ __device__ constexpr int wg8_tile(int NW, int Q, int i)
{
    int c = 0;
    for (int t = 0; t < WG8_NT; t++)
        if (wg8_owner(NW, tile_bi(t, WG8_NB), tile_bj(t, WG8_NB)) == Q) {
            if (c == i) return t;
            c++;
        }
    return 0;
}

// chol_diag_block (chol_kernels.hpp) with the pivot row broadcast by DPP instead of the LDS crossbar.  The original fetches R[c][i]
// for every later row i with two ds_bpermute per double -- 240 of them per block, each pivot a round trip through LDS; measured in this
// kernel (CMFREC_HIP_WAVE_SKIP=4): the eight diagonal blocks of a row cost more than its 450 matrix instructions.  Here the pivot row's
// values are copied from the R half (lanes 0-15) into the inverse half (lanes 16-31) with one v_permlane16_swap per dword, and
// R[c][i] is a row_newbcast move (one v_mov_b64_dpp) in both halves.  Same operations in the same order: the same bits.
template <typename T>
__device__ __forceinline__ T rows_from_even(T v);
template <> __device__ __forceinline__ double rows_from_even(double v)      // rows 1 / 3 of the result carry rows 0 / 2 of v
{
    auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]);
}
template <> __device__ __forceinline__ float rows_from_even(float v)
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]);
}
template <typename T>
__device__ __forceinline__ void chol_diag_block_dpp(typename CholMfma<T>::vec d, T *slot, int lane, int nact)
{
    using Mf = CholMfma<T>;
#pragma unroll
    for (int r = 0; r < 4; r++) slot[r * 64 + lane] = d[r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int j = lane & 15;
    const bool inv_half = (lane & 16) != 0;
    T u[16];
    T one = T(1);
    asm volatile("" : "+v"(one));          // opaque: keeps the identity columns from being hoisted out of the row loop
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const T dv = slot[Mf::cidx(i, j)];
        u[i] = inv_half ? ((i == j) ? one : T(0)) : dv;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    T *wrow = slot + j * Mf::LDR;
    RsqChain<T> ch;
    ch.s0(bcast_lane(u[0], 0)); ch.s1(); ch.s2();
    T rs = ch.s3();
    static_for<0, 16>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if (c < nact) {
            const T v = u[c] * rs;                  // lanes 0-15: R[c][j] (j >= c);  lanes 16-31: inv(R)[j][c]
            if (lane >= 16 && lane < 32) wrow[c] = v;
            const bool more = (c + 1 < nact);
            const T vr = rows_from_even(v);         // R[c][j] in both halves
            static_for<c + 1, 16>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                u[i] -= lanes::row_bcast16<i>(vr) * v;        // R[c][i] from lane i of the row
                // 1/sqrt of the next pivot, one step per update so its latency hides behind them
                if (more) {
                    if (i == c + 1) ch.s0(bcast_lane(u[c + 1], c + 1));
                    if (i == c + 2) ch.s1();
                    if (i == c + 3) ch.s2();
                    if (i == c + 4) rs = ch.s3();
                }
            });
            if (more) {                             // late pivots: fewer than four updates to hide behind
                if (c + 1 >= 15) ch.s1();
                if (c + 1 >= 14) ch.s2();
                if (c + 1 >= 13) rs = ch.s3();
            }
        } else {
            if (lane >= 16 && lane < 32) wrow[c] = u[c];   // identity padding: R = inv(R) = I there
        }
    });
}

template <typename T> struct Wg8Shared {
    // panel tiles of a block step, [b][r][lane].  One buffer: a step's panel tiles are written behind the barrier that follows the
    // diagonal block, and every wavefront reaches that barrier only when it is done with the previous step's trailing updates
    __attribute__((aligned(16))) T Xt[WG8_NB * 256];
    T rinv[WG8_NB * 16 * CholMfma<T>::LDR];                  // inv(R_kk) of every block
    T rhs[16 * WG8_NB];                                 // right-hand side -> y -> z (in place)
    T bcol[16 * WG8_NB];                                // border column g -> R^-T g
    T xall[16 * WG8_NB];                                // solution
    T psum[2][4][16];                                        // backward substitution: the waves' partial sums of a block row (NW of the four slots)
    T gam, rbs;                                              // border diagonal and border right-hand side
    int rix;
};

struct Wg8Row {
    int s0, s1;            // the producer's work items of this row
    bool has_u, pre_rhs, add_lam;
    int kt;
};

template <typename T, int NW, int Q, bool BORDER>
__device__ __forceinline__ void wg8_row(const CholParams<T> &P, const CholSlices<T> &SL, Wg8Shared<T> &S, const Wg8Row &R, T lam, T lam_last,
                                        T *__restrict__ arow, int lane)
{
    using Mf = CholMfma<T>;
    using vec = typename Mf::vec;
    constexpr int NB = WG8_NB, NT = WG8_NT, NTQ = wg8_count(NW, Q), NTHR = 64 * NW;
    constexpr int LDR = Mf::LDR, RSZ = 16 * LDR;
    constexpr size_t PART = chol_wave_part_elems(NB);
    const int lm = lane & 15, kt = R.kt;
    const int kq = kt - (BORDER ? 1 : 0);
    const int tid = 64 * Q + lane;
    vec acc[NTQ];
#pragma unroll
    for (int i = 0; i < NTQ; i++) acc[i] = vec{0, 0, 0, 0};
    // ---- 1. N = -(initial matrices + partials + diagonal);  right-hand side, border column ----
    // (the lane offset is made opaque per call: otherwise the addresses of the launch's initial matrices -- invariant over the rows --
    //  are hoisted out of the row loop and pinned in registers; one address per tile, the four registers at immediate offsets)
    auto add_tiles = [&](const T *__restrict__ pp) __attribute__((always_inline)) {      // five tiles (20 loads) in flight
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const T *pl = pp + lane_o;
        static_for<0, (NTQ + 4) / 5>([&](auto gc) {
            constexpr int g5 = decltype(gc)::value;
            vec ld[5];
            static_for<0, 5>([&](auto jc) {
                constexpr int i = 5 * g5 + decltype(jc)::value;
                if constexpr (i < NTQ) {
                    constexpr int t = wg8_tile(NW, Q, i);
                    const T *pt = pl + t * 256;
#pragma unroll
                    for (int r = 0; r < 4; r++) ld[i - 5 * g5][r] = pt[r * 64];
                }
            });
            static_for<0, 5>([&](auto jc) {
                constexpr int i = 5 * g5 + decltype(jc)::value;
                if constexpr (i < NTQ) acc[i] -= ld[i - 5 * g5];
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    const bool full = (P.mode == CHOL_IMPLICIT);
    const T *M1 = full ? P.Minit : P.Mfull;                          // [kt, kt], every row
    const T *M2 = (!full && R.has_u) ? P.Minit : nullptr;            // [kc, kc], rows with side information
    const int kc = P.kc;
    // element (gi, kt - 1) of the initial matrices: the border column
    auto border_init = [&](int gi) -> T {
        T v = T(0);
        if (M1 != nullptr) v += M1[(size_t)min(gi, kt - 1) * kt + (kt - 1)];
        if (M2 != nullptr && kc > 0) { const T q = M2[(size_t)min(gi, kc - 1) * kc + (kc - 1)]; v += (kt - 1 < kc) ? q : T(0); }
        return v;
    };
    // element e = tid, tid + NTHR, .. of [right-hand side (128) | border column (128)]: four wavefronts take one each, two take two
    constexpr int NE = 256 / NTHR;
    T v0[NE];
#pragma unroll
    for (int q = 0; q < NE; q++) {
        const int e = tid + NTHR * q, u = e & 127;
        v0[q] = T(0);
        if (e < 128) { if (R.pre_rhs && u < kq) v0[q] = arow[u]; }        // w U C prefilled (collective.c:5768-5773)
        else if (BORDER && u < kq) v0[q] = border_init(u);
    }
    T gam = T(0), rbs = T(0);
    if (BORDER && tid == 0) {
        gam = (R.add_lam ? lam_last : T(0)) + border_init(kt - 1);
        rbs = R.pre_rhs ? arow[kt - 1] : T(0);
    }
    const int skip = SL.dbg_skip;          // timing experiments only (CMFREC_HIP_WAVE_SKIP): 1 tile loads, 2 factorisation, 4 diagonal blocks, 8 backward pass, 16 trailing MFMAs
    for (int sl = R.s0; sl < R.s1; sl++) {
        const T *pp = SL.part + (size_t)(sl - SL.part_base) * PART;
        if (!(skip & 1)) add_tiles(pp);
        const T *pv = pp + (size_t)NT * 256;
#pragma unroll
        for (int q = 0; q < NE; q++) {
            const int e = tid + NTHR * q;
            if (e < 128 || BORDER) v0[q] += pv[e];                        // [0, 128): right-hand side; [128, 256): border column
        }
        if (BORDER && tid == 0) { gam += pv[32 * NB]; rbs += pv[32 * NB + 1]; }
    }
    if (M1 != nullptr && SL.init1 != nullptr && !(skip & 1)) add_tiles(SL.init1);
    if (M2 != nullptr && kc > 0 && SL.init2 != nullptr && !(skip & 1)) add_tiles(SL.init2);
    static_for<0, NTQ>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int t = wg8_tile(NW, Q, i), bi = tile_bi(t, NB), bj = tile_bj(t, NB);
        if constexpr (bi == bj) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int gi = 16 * bi + Mf::row_of(lane, r), gj = 16 * bi + lm;
                if (gi == gj) acc[i][r] -= (gi >= kq) ? T(1) : (!R.add_lam ? T(0) : ((gi == kt - 1) ? lam_last : lam));   // common.c:1060-1062, collective.c:1819
            }
        }
    });
#pragma unroll
    for (int q = 0; q < NE; q++) {
        const int e = tid + NTHR * q, u = e & 127;
        if (e < 128) S.rhs[u] = v0[q];
        else S.bcol[u] = BORDER ? v0[q] : T(0);
    }
    if (BORDER && tid == 0) { S.gam = gam; S.rbs = rbs; }
    // ---- 2. blocked Cholesky  M = R^T R  of M = -N ----
    for (int kbk = 0; kbk < ((skip & 2) ? 0 : NB); kbk++) {
        T *rslot = S.rinv + kbk * RSZ;
        T *Xw = S.Xt;
        // a. diagonal block, by its owner
        {
            vec d = vec{0, 0, 0, 0};
            bool mine = false;
            static_for<0, NTQ>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int t = wg8_tile(NW, Q, i);
                if constexpr (tile_bi(t, NB) == tile_bj(t, NB)) {
                    if (kbk == tile_bi(t, NB)) { d = -acc[i]; mine = true; }
                }
            });
            if (mine && !(skip & 4)) chol_diag_block_dpp<T>(d, rslot, lane, max(0, min(16, kq - 16 * kbk)));
        }
        __syncthreads();
        // b. panel tiles of block row kbk:  X = inv(R_kk)^T tile  (tile = -N: the A operand carries the sign);
        //    y_k = inv(R_kk)^T v_k for the right-hand side (wavefront 1) and the border column (wavefront 3), in place
        {
            T ainv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) ainv[r] = -rslot[Mf::row_of(lane, r) * LDR + lm];
            // (the diagonal blocks belong to the even wavefronts: wavefront 1 -- and 3 of four -- are free at this point)
            static_for<0, 2>([&](auto wc) {
                constexpr int which = decltype(wc)::value;                  // 0: right-hand side, 1: border column
                constexpr int owner = (which == 0 || NW == 2) ? 1 : 3;
                if constexpr (Q == owner && (which == 0 || BORDER)) {
                    T *vec_k = (which == 0) ? S.rhs : S.bcol;
                    const T vk = vec_k[16 * kbk + lm];
                    T yv = T(0);
                    static_for<0, 16>([&](auto lc) {
                        constexpr int l = decltype(lc)::value;
                        yv += rslot[l * LDR + lm] * lanes::row_bcast16<l>(vk);
                    });
                    if (lane < 16) vec_k[16 * kbk + lane] = yv;
                }
            });
            static_for<0, NTQ>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int t = wg8_tile(NW, Q, i), bi = tile_bi(t, NB), bj = tile_bj(t, NB);
                if constexpr (bi < bj) {
                    if (kbk == bi) {
                        vec x = Mf::mma(ainv[0], acc[i][0], vec{0, 0, 0, 0});          // two independent chains
                        vec x2 = Mf::mma(ainv[2], acc[i][2], vec{0, 0, 0, 0});
                        x = Mf::mma(ainv[1], acc[i][1], x);
                        x2 = Mf::mma(ainv[3], acc[i][3], x2);
                        x += x2;
                        acc[i] = x;                                                   // R(bi, bj), kept for the backward pass
#pragma unroll
                        for (int r = 0; r < 4; r++) Xw[bj * 256 + r * 64 + lane] = x[r];
                    }
                }
            });
        }
        __syncthreads();
        // c. trailing tiles  N(bi, bj) += X_bi^T X_bj  (bi > kbk);  forward substitution of the later blocks
        if (kbk + 1 < NB) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                T xo[NB][2];
#pragma unroll
                for (int b = 1; b < NB; b++)
#pragma unroll
                    for (int r2 = 0; r2 < 2; r2++) xo[b][r2] = Xw[b * 256 + (2 * h + r2) * 64 + lane];
                static_for<0, NTQ>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int t = wg8_tile(NW, Q, i), bi = tile_bi(t, NB), bj = tile_bj(t, NB);
                    if constexpr (bi > 0) {
                        if (kbk < bi && !(skip & 16)) {
                            acc[i] = Mf::mma(xo[bi][0], xo[bj][0], acc[i]);
                            acc[i] = Mf::mma(xo[bi][1], xo[bj][1], acc[i]);
                        }
                    }
                });
            }
            // v_j -= X_j^T y_k  for the later blocks: thread <-> unknown (element (k2, c) of tile b sits at [b][k2 >> 2][16 (k2 & 3) + c])
#pragma unroll
            for (int q = 0; q < NE; q++) {
                const int e = tid + NTHR * q, u = e & 127;
                if (e < 128 || BORDER) {
                    T *vec_k = (e < 128) ? S.rhs : S.bcol;
                    if (u >= 16 * (kbk + 1)) {
                        T sacc = vec_k[u];
                        const T *xt = Xw + (u >> 4) * 256 + (u & 15);
#pragma unroll
                        for (int k2 = 0; k2 < 16; k2++) sacc -= xt[(k2 >> 2) * 64 + (k2 & 3) * 16] * vec_k[16 * kbk + k2];
                        vec_k[u] = sacc;
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- 3. the border unknown:  rho^2 = gamma - r.r ;  x_last = (rhs_last - r.y) / rho^2 ;  z = y - r x_last ----
    T xlast = T(0);
    if (BORDER) {
        T s1 = T(0), s2 = T(0);
#pragma unroll
        for (int q = 0; q < 2; q++) { const T rv = S.bcol[lane + 64 * q]; s1 += rv * rv; s2 += rv * S.rhs[lane + 64 * q]; }
        s1 = lanes::wave_sum(s1); s2 = lanes::wave_sum(s2);
        xlast = (S.rbs - s2) / (S.gam - s1);
        __syncthreads();
        for (int e = tid; e < 128; e += NTHR) S.rhs[e] -= S.bcol[e] * xlast;
        __syncthreads();
    }
    // ---- 4. backward substitution  R x = z  by block rows:  x_i = inv(R_ii) (z_i - sum_{j > i} R_ij x_j) ----
    T xs[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) xs[b] = T(0);
    static_for<0, NB>([&](auto sc) {
        constexpr int bi = NB - 1 - decltype(sc)::value;
        if (skip & 8) return;
        T p0 = T(0), p1 = T(0), p2 = T(0), p3 = T(0);
        static_for<0, NTQ>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int t = wg8_tile(NW, Q, i), tbi = tile_bi(t, NB), tbj = tile_bj(t, NB);
            if constexpr (tbi == bi && tbj > bi) {
                const T xv = xs[tbj];
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
        // (the row of inv(R_ii) this lane multiplies with is on its way while the partial sums meet at the barrier)
        const T *rslot = S.rinv + bi * RSZ;
        T ri[16];
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++) ri[n2] = rslot[lm * LDR + n2];
        __syncthreads();
        // t = z_i - sum_{j > i} R_ij x_j, element lm in lane lm of every 16-lane row; x[16 bi + lm] = inv(R_ii)[lm][.] . t with the
        // elements of t broadcast inside the row (DPP) -- computed redundantly by every 16-lane group of every wave
        const T tv = S.rhs[16 * bi + lm] - ((NW == 4) ? ((ps[lm] + ps[16 + lm]) + (ps[32 + lm] + ps[48 + lm])) : (ps[lm] + ps[16 + lm]));
        T xm = T(0);
        static_for<0, 16>([&](auto nc) {
            constexpr int n2 = decltype(nc)::value;
            xm += ri[n2] * lanes::row_bcast16<n2>(tv);
        });
        xs[bi] = xm;
        if (Q == 0 && lane < 16) S.xall[16 * bi + lane] = xm;
    });
    if (Q == 0) {
        for (int t = lane; t < kq; t += 64) arow[t] = S.xall[t];
        if (BORDER && lane == 0) arow[kt - 1] = xlast;
    }
}

// One workgroup of NW wavefronts per row; rows [P.row_first, P.nrows) of the processing order handed out by P.counter.
// Rows in flight per CU: NW = 4 -> two workgroups (the registers of 10 tiles per wavefront), NW = 2 -> four (20 tiles per wavefront,
// 37 KB of LDS each).  The chain of a row -- eight diagonal blocks of sixteen dependent pivots -- is what a launch waits for, the
// matrix instructions are a small part of it: more rows in flight is what buys throughput.
template <typename T, bool BORDER, int NW>
// (second argument: wavefronts per SIMD -- four wavefronts per row: one per SIMD and workgroup, i.e. workgroups per CU; two per row: 2 = four workgroups)
__global__ void __launch_bounds__(64 * NW, (NW == 4 ? CMF_WG8_WGS : 2))
chol_wg8_kernel(const CholParams<T> P, const RowDesc *__restrict__ desc, const CholSlices<T> SL)
{
    __shared__ Wg8Shared<T> S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kt = P.kt;
    const bool coll = (P.mode == CHOL_COLLECTIVE || P.mode == CHOL_COLLECTIVE_IMPLICIT);
    for (;;) {
        if (tid == 0) S.rix = P.row_first + atomicAdd(P.counter, 1);
        __syncthreads();
        const int rix = S.rix;
        __syncthreads();                            // S.rix may be rewritten; the previous row's LDS readers are done
        if (rix >= P.nrows) break;
        const RowDesc d = desc[rix];
        const int row = d.row, nnz_row = d.nnz;
        T *arow = P.A + (size_t)row * P.lda;
        Wg8Row R;
        R.kt = kt;
        R.has_u = coll && row < P.rows_with_u;
        R.pre_rhs = R.has_u || P.rhs_prefilled_all;
        R.add_lam = (P.mode == CHOL_EXPLICIT || P.mode == CHOL_COLLECTIVE);
        if (coll && nnz_row == 0 && !R.has_u) {                          // collective.c:1258-1268, :1876-1885
            for (int e = tid; e < kt; e += 64 * NW) arow[e] = T(0);
            continue;
        }
        T lam = P.lam, lam_last = P.lam_last;
        if (P.mode == CHOL_EXPLICIT) {
            if (P.scale_lam) {                                           // common.c:679-723
                lam *= (T)nnz_row;
                if (!P.scale_bias_const) lam_last *= (T)nnz_row;
            }
        } else if (P.mode == CHOL_COLLECTIVE) {
            if (P.scale_lam || P.scale_lam_sideinfo) {                   // collective.c:1285-1355
                T mult = (nnz_row > 0) ? (T)nnz_row : T(1);
                if (P.scale_lam_sideinfo && R.has_u) mult += (T)P.p_side;  // :1338-1346
                lam *= mult;
                if (R.has_u || !P.scale_bias_const) lam_last *= mult;
            }
        }
        const bool hv = rix < SL.n_heavy;
        R.s0 = hv ? SL.row_off[rix] : SL.n_slices + (rix - SL.n_heavy);
        R.s1 = hv ? SL.row_off[rix + 1] : R.s0 + 1;
        if constexpr (NW == 4) {
            switch (wave) {
                case 0: wg8_row<T, 4, 0, BORDER>(P, SL, S, R, lam, lam_last, arow, lane); break;
                case 1: wg8_row<T, 4, 1, BORDER>(P, SL, S, R, lam, lam_last, arow, lane); break;
                case 2: wg8_row<T, 4, 2, BORDER>(P, SL, S, R, lam, lam_last, arow, lane); break;
                default: wg8_row<T, 4, 3, BORDER>(P, SL, S, R, lam, lam_last, arow, lane); break;
            }
        } else {
            if (wave == 0) wg8_row<T, 2, 0, BORDER>(P, SL, S, R, lam, lam_last, arow, lane);
            else wg8_row<T, 2, 1, BORDER>(P, SL, S, R, lam, lam_last, arow, lane);
        }
    }
}

}  // namespace cmfhip
