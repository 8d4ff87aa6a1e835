// chol_wg_kernels.hpp -- the factorisation of the eight-block closed-form rows (k_t = 113 .. 129, double precision: config 3) by
// ONE WORKGROUP OF FOUR WAVEFRONTS per row (round 6).
//
// Same arithmetic as chol_wave_kernel's second build (WMODE 2: the row's rank-k update has been done by the producer kernels, the
// partials wait in HBM) -- collective_closed_form_block /root/reference/src/collective.c:1534-1846 (dposv at :1823),
// factors_closed_form /root/reference/src/common.c:978-1070 -- but another mapping.  That build keeps the 36 tiles of the upper
// triangle in ONE wavefront: 288 accumulator registers + the factorisation's temporaries = one wavefront per SIMD with 263
// spilled registers, a chain of matrix-instruction latencies with nothing to hide them (0.94-1.12 ms per launch, ~0.13 of the
// fp64 peak, 5.6 of config 3's 14.3 ms in round 5).  Here
//   * the tiles are dealt to the four wavefronts of a workgroup -- tile (bi, bj) belongs to wavefront (bi + bj) mod 4: 10 / 8 /
//     10 / 8 tiles = 80 registers, so the panel tiles of every block row AND the trailing tiles of every step are spread over
//     all four SIMDs, and two to three workgroups (rows) are resident per CU;
//   * the accumulators hold the NEGATED matrix (the trailing update is a plain N += X_bi^T X_bj), the panel tiles of a step
//     travel through LDS (double-buffered, lane-linear: conflict-free 8-byte accesses), inv(R_kk) has a slot per block: two
//     barriers per block step, as in gramk_consumer_kernel (single precision, 17 blocks), whose scheme this is;
//   * right-hand side and border column (k_t = 16 n + 1: the bias column stays outside the tiles, chol_wave_kernels.hpp) are
//     forward-substituted thread <-> unknown while the trailing MFMAs run; the backward substitution goes by block rows with
//     one cross-lane reduction and one barrier per row.
// Input: the partials of chol_wave_kernel's producer build / chol_parts_producer_kernel (tiles in the accumulator layout,
// [t][r][lane]; right-hand side; border column; two scalars) and the launch's initial matrices in the same layout
// (tile_pack_kernel).  Output: the row of A.
#pragma once
#include "chol_wave_kernels.hpp"

namespace cmfhip {

constexpr int WG8_NB = 8, WG8_NT = 36;
__host__ __device__ constexpr int wg8_owner(int bi, int bj) { return (bi + bj) & 3; }
__host__ __device__ constexpr int wg8_count(int Q)
{
    int c = 0;
    for (int t = 0; t < WG8_NT; t++) c += (wg8_owner(tile_bi(t, WG8_NB), tile_bj(t, WG8_NB)) == Q) ? 1 : 0;
    return c;
}
// packed index (tile_bi / tile_bj order) of the i-th tile of wavefront Q
__host__ __device__ constexpr int wg8_tile(int Q, int i)
{
    int c = 0;
    for (int t = 0; t < WG8_NT; t++)
        if (wg8_owner(tile_bi(t, WG8_NB), tile_bj(t, WG8_NB)) == Q) {
            if (c == i) return t;
            c++;
        }
    return 0;
}

template <typename T> struct Wg8Shared {
    __attribute__((aligned(16))) T Xt[2][WG8_NB * 256];     // panel tiles of a block step, [b][r][lane]
    T rinv[WG8_NB * 16 * CholMfma<T>::LDR];                  // inv(R_kk) of every block
    T rhs[16 * WG8_NB + 16];                                 // right-hand side -> y -> z (in place)
    T bcol[16 * WG8_NB + 16];                                // border column g -> R^-T g
    T xall[16 * WG8_NB + 16];                                // solution
    T psum[2][4][16];                                        // backward substitution: the waves' partial sums of a block row
    T gam, rbs;                                              // border diagonal and border right-hand side
    int rix;
};

struct Wg8Row {
    int s0, s1;            // the producer's work items of this row
    bool has_u, pre_rhs, add_lam;
    int kt;
};

template <typename T, int Q, bool BORDER>
__device__ __forceinline__ void wg8_row(const CholParams<T> &P, const CholSlices<T> &SL, Wg8Shared<T> &S, const Wg8Row &R, T lam, T lam_last,
                                        T *__restrict__ arow, int lane)
{
    using Mf = CholMfma<T>;
    using vec = typename Mf::vec;
    constexpr int NB = WG8_NB, NT = WG8_NT, NTQ = wg8_count(Q);
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
                    constexpr int t = wg8_tile(Q, i);
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
    // threads 0 .. 127: unknown tid of the right-hand side; threads 128 .. 255: unknown tid - 128 of the border column
    const int u = tid & 127;
    T v0 = T(0);
    if (tid < 128) { if (R.pre_rhs && u < kq) v0 = arow[u]; }         // w U C prefilled (collective.c:5768-5773)
    else if (BORDER && u < kq) v0 = border_init(u);
    T gam = T(0), rbs = T(0);
    if (BORDER && tid == 0) {
        gam = (R.add_lam ? lam_last : T(0)) + border_init(kt - 1);
        rbs = R.pre_rhs ? arow[kt - 1] : T(0);
    }
    for (int sl = R.s0; sl < R.s1; sl++) {
        const T *pp = SL.part + (size_t)(sl - SL.part_base) * PART;
        add_tiles(pp);
        const T *pv = pp + (size_t)NT * 256;
        if (tid < 128 || BORDER) v0 += pv[tid];                       // [0, 128): right-hand side; [128, 256): border column
        if (BORDER && tid == 0) { gam += pv[32 * NB]; rbs += pv[32 * NB + 1]; }
    }
    if (M1 != nullptr && SL.init1 != nullptr) add_tiles(SL.init1);
    if (M2 != nullptr && kc > 0 && SL.init2 != nullptr) add_tiles(SL.init2);
    static_for<0, NTQ>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int t = wg8_tile(Q, i), bi = tile_bi(t, NB), bj = tile_bj(t, NB);
        if constexpr (bi == bj) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int gi = 16 * bi + Mf::row_of(lane, r), gj = 16 * bi + lm;
                if (gi == gj) acc[i][r] -= (gi >= kq) ? T(1) : (!R.add_lam ? T(0) : ((gi == kt - 1) ? lam_last : lam));   // common.c:1060-1062, collective.c:1819
            }
        }
    });
    if (tid < 128) S.rhs[u] = v0;
    else S.bcol[u] = BORDER ? v0 : T(0);
    if (BORDER && tid == 0) { S.gam = gam; S.rbs = rbs; }
    // ---- 2. blocked Cholesky  M = R^T R  of M = -N ----
    for (int kbk = 0; kbk < NB; kbk++) {
        T *rslot = S.rinv + kbk * RSZ;
        T *Xw = S.Xt[kbk & 1];
        // a. diagonal block, by its owner
        {
            vec d = vec{0, 0, 0, 0};
            bool mine = false;
            static_for<0, NTQ>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int t = wg8_tile(Q, i);
                if constexpr (tile_bi(t, NB) == tile_bj(t, NB)) {
                    if (kbk == tile_bi(t, NB)) { d = -acc[i]; mine = true; }
                }
            });
            if (mine) chol_diag_block<T>(d, rslot, lane, max(0, min(16, kq - 16 * kbk)));
        }
        __syncthreads();
        // b. panel tiles of block row kbk:  X = inv(R_kk)^T tile  (tile = -N: the A operand carries the sign);
        //    y_k = inv(R_kk)^T v_k for the right-hand side (wavefront 1) and the border column (wavefront 3), in place
        {
            T ainv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) ainv[r] = -rslot[Mf::row_of(lane, r) * LDR + lm];
            if (Q == 1 || (BORDER && Q == 3)) {
                T *vec_k = (Q == 1) ? S.rhs : S.bcol;
                T yv = T(0);
#pragma unroll
                for (int l = 0; l < 16; l++) yv += rslot[l * LDR + lm] * vec_k[16 * kbk + l];
                if (lane < 16) vec_k[16 * kbk + lane] = yv;
            }
            static_for<0, NTQ>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int t = wg8_tile(Q, i), bi = tile_bi(t, NB), bj = tile_bj(t, NB);
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
                    constexpr int t = wg8_tile(Q, i), bi = tile_bi(t, NB), bj = tile_bj(t, NB);
                    if constexpr (bi > 0) {
                        if (kbk < bi) {
                            acc[i] = Mf::mma(xo[bi][0], xo[bj][0], acc[i]);
                            acc[i] = Mf::mma(xo[bi][1], xo[bj][1], acc[i]);
                        }
                    }
                });
            }
            // v_j -= X_j^T y_k  for the later blocks: thread <-> unknown (element (k2, c) of tile b sits at [b][k2 >> 2][16 (k2 & 3) + c])
            if (tid < 128 || BORDER) {
                T *vec_k = (tid < 128) ? S.rhs : S.bcol;
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
        if (tid < 128) S.rhs[tid] -= S.bcol[tid] * xlast;
        __syncthreads();
    }
    // ---- 4. backward substitution  R x = z  by block rows:  x_i = inv(R_ii) (z_i - sum_{j > i} R_ij x_j) ----
    T xs[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) xs[b] = T(0);
    static_for<0, NB>([&](auto sc) {
        constexpr int bi = NB - 1 - decltype(sc)::value;
        T p0 = T(0), p1 = T(0), p2 = T(0), p3 = T(0);
        static_for<0, NTQ>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int t = wg8_tile(Q, i), tbi = tile_bi(t, NB), tbj = tile_bj(t, NB);
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
        __syncthreads();
        const T *rslot = S.rinv + bi * RSZ;
        T xm = T(0);                              // x[16 bi + lm], computed redundantly by every 16-lane group of every wave
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++)
            xm += rslot[lm * LDR + n2] * (S.rhs[16 * bi + n2] - ((ps[n2] + ps[16 + n2]) + (ps[32 + n2] + ps[48 + n2])));
        xs[bi] = xm;
        if (Q == 0 && lane < 16) S.xall[16 * bi + lane] = xm;
    });
    if (Q == 0) {
        for (int t = lane; t < kq; t += 64) arow[t] = S.xall[t];
        if (BORDER && lane == 0) arow[kt - 1] = xlast;
    }
}

// One workgroup of four wavefronts per row; rows [P.row_first, P.nrows) of the processing order handed out by P.counter.
template <typename T, bool BORDER>
__global__ void __launch_bounds__(256, 2)
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
            for (int e = tid; e < kt; e += 256) arow[e] = T(0);
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
        switch (wave) {
            case 0: wg8_row<T, 0, BORDER>(P, SL, S, R, lam, lam_last, arow, lane); break;
            case 1: wg8_row<T, 1, BORDER>(P, SL, S, R, lam, lam_last, arow, lane); break;
            case 2: wg8_row<T, 2, BORDER>(P, SL, S, R, lam, lam_last, arow, lane); break;
            default: wg8_row<T, 3, BORDER>(P, SL, S, R, lam, lam_last, arow, lane); break;
        }
    }
}

}  // namespace cmfhip
