// cg_pair_kernels.hpp -- the CG row update for rows of at most 32 entries, TWO ROWS PER WAVEFRONT (round 5).
//
// Same arithmetic as cg_rows_kernel / cg_rows_tiny_kernel (factors_implicit_cg, /root/reference/src/common.c:1914-1986;
// factors_explicit_cg, :1098-1188; the row loops :3259-3299, :3349-3368).  Why another kernel: the bin of the short rows is bound by
// vector-instruction issue, not by its gather -- a row of 16 entries moves 7 KB and costs ~230 vector instructions
// per CG pass of which only 84 are FMAs; the rest (cross-lane reductions, the two wave-wide dot products and the two quotients of a
// CG step, the vector's round trip through LDS) is the same for a row of 4 entries and for one of 64.  Here one instruction stream
// serves two rows, so that fixed part is paid once per PAIR:
//
//   lane = h*32 + jj*8 + ll      h  = which row of the pair (lanes 0..31 / 32..63)
//                                jj = 0..3  entry group of the half:  entries jj*NE + t, t < NE  (NE = 4: <= 16 entries, 8: <= 32)
//                                ll = 0..7  owns the factor columns ll + 8 s, s < S
//   tile products     : tile_pass4 / tile_pass of cg_kernels.hpp unchanged -- their reductions and broadcasts stay inside the
//                       8-lane groups, which belong to one row
//   Gramian product   : ALL EIGHT 8-lane groups of the wavefront share the work of both rows: group gj = lane >> 3 covers the
//                       Gramian rows 8 gj + t for the vector of row 0 AND the vector of row 1 -- every element of B^T B read from
//                       LDS once (or kept in registers, as in the one-row tiny kernel) feeds two FMAs.  (cg_rows_tiny2_kernel of
//                       round 4 gave each half the whole product of its own row: twice the LDS reads per lane, bound by the LDS
//                       pipe, measured slower than one row per wavefront.)  v_permlane32_swap adds the two halves' partial sums
//                       and leaves row 0's total in lanes 0..31, row 1's in lanes 32..63 in the same instruction.
//   vectors (a, r, p) : 64 elements over the 32 lanes of a half, two registers: lane (jj, ll) holds the elements ll + 16 jj and
//                       ll + 16 jj + 8 -- exactly what the transposed butterfly over the four entry groups (lane bits 4 and 3)
//                       leaves; every pass they go through the wavefront's own 2 x 64 LDS buffer, from which the lane reads the
//                       S replicated elements of its own row and the 2 x 8 Gramian weights as broadcast reads
//   dot products      : over the 32 lanes of a half (five stages), both rows in the same instructions; the quotients likewise
//   exits             : per row (1e-12 before the first step, 1e-8 after a step); a finished half idles with alpha = 0
// Rows are paired in processing order (sorted by length) inside their length class, pairs are claimed dynamically like rows
// elsewhere.
#pragma once
#include "cg_kernels.hpp"

namespace cmfhip {

// value of lane 0 of the lane's own half (lanes 0..31 <- lane 0, lanes 32..63 <- lane 32)
__device__ __forceinline__ int half_first(int x, int lane)
{
    const int lo = __builtin_amdgcn_readlane(x, 0), hi = __builtin_amdgcn_readlane(x, 32);
    return (lane & 32) ? hi : lo;
}

// gather of a half's tile: NE entries per lane x S columns; entry jj*NE + t of the half's row sits (index / value) in the lane
// src_lane(t) of the same 8-lane group.  Branch-free like load_tile: slots past the row's end re-read the row of its first entry
// (weight zero), columns past k re-read column k - 1 (their vector elements are zero).
template <typename T, int S, int NE, typename TILE>
__device__ __forceinline__ void load_tile_half(TILE &tile, const T *__restrict__ Bm, size_t ldb, int k, int my_idx, int nnz, int lane)
{
    const int jj = (lane >> 3) & 3, ll = lane & 7;
    int its[NE];
    if constexpr (NE == 8) {
        its[0] = lanes::bcast8<0>(my_idx); its[1] = lanes::bcast8<1>(my_idx); its[2] = lanes::bcast8<2>(my_idx); its[3] = lanes::bcast8<3>(my_idx);
        its[4] = lanes::bcast8<4>(my_idx); its[5] = lanes::bcast8<5>(my_idx); its[6] = lanes::bcast8<6>(my_idx); its[7] = lanes::bcast8<7>(my_idx);
    } else {
        its[0] = lanes::bcast8<0>(my_idx); its[1] = lanes::bcast8<2>(my_idx); its[2] = lanes::bcast8<4>(my_idx); its[3] = lanes::bcast8<6>(my_idx);
    }
    const int first_idx = half_first(my_idx, lane);
    const int col_last = min(ll + 8 * (S - 1), k - 1) - ll;
    const char *base = reinterpret_cast<const char *>(Bm + ll);
    const unsigned ldb_bytes = (unsigned)(ldb * sizeof(T));
#pragma unroll
    for (int t = 0; t < NE; t++) {
        const unsigned it = (unsigned)(((jj * NE + t) < nnz) ? its[t] : first_idx);
        const T *rp = reinterpret_cast<const T *>(base + (unsigned long long)it * ldb_bytes);
#pragma unroll
        for (int s = 0; s < S; s++) tile.set(t, s, rp[(s < S - 1) ? 8 * s : col_last]);
    }
}

// The pair kernel's own LDS layout of the Gramian: what lane (gj, ll) multiplies -- rows 8 gj + t, columns ll + 8 s -- as ONE
// contiguous run per lane, sixteen bytes at a time across the wavefront: 16-byte word j of lane l at byte (j * 64 + l) * 16 (double:
// the elements e = 2 j, 2 j + 1 with t = e / S, s = e % S; single: the row pairs (2 q, 2 q + 1) the packed FMAs want, pairs
// i = q * S + s = 2 j and 2 j + 1).  Every read is a
// ds_read_b128 of 64 consecutive 16-byte words -- half the LDS cycles per element of the ds_read2_b64 pairs the padded row layout
// of the other kernels is read with (tools/microbench/valu_costs.hip: 16 against 32 ticks per instruction), and the pair kernel
// reads its Gramian from LDS on every pass.
template <typename T, int S> __host__ __device__ constexpr int pair_gram_elems() { return 64 * 8 * S; }
template <typename T, int S>
__device__ __forceinline__ void stage_gramian_lanes(T *__restrict__ GL, const T *__restrict__ BtB, int k, int tid, int nthreads)
{
    for (int x = tid; x < 64 * 8 * S; x += nthreads) {
        int r, col;
        if constexpr (sizeof(T) == 8) {                              // x = (j * 64 + l) * 2 + c: elements e = 2 j + c of lane l
            const int c = x & 1, l = (x >> 1) & 63, j = x >> 7, e = 2 * j + c;
            r = (l >> 3) * 8 + e / S; col = (l & 7) + 8 * (e % S);
        } else {                                                     // x = (j * 64 + l) * 4 + c4: the row pairs i = 2 j, 2 j + 1 of lane l
            const int c4 = x & 3, l = (x >> 2) & 63, j = x >> 8, i = 2 * j + (c4 >> 1);
            r = (l >> 3) * 8 + 2 * (i / S) + (c4 & 1); col = (l & 7) + 8 * (i % S);
        }
        GL[x] = (r < k && col < k) ? BtB[(size_t)r * k + col] : T(0);
    }
}

// accA -= / += G wA, accB -= / += G wB with every staged Gramian element read once (rows 8 gj + t of this lane's group)
template <bool NEG, typename T, int S>
__device__ __forceinline__ void gram_pass_two(const T *__restrict__ GL, const T (&wA)[8], const T (&wB)[8], PassAcc<T> &accA, PassAcc<T> &accB, int lane)
{
    if constexpr (std::is_same<T, float>::value) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 *g4 = reinterpret_cast<const f32x4 *>(GL) + lane;       // [i / 2][lane]: the pairs i, i + 1 of this lane
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x2 a2 = f32x2{wA[2 * q], wA[2 * q + 1]}, b2 = f32x2{wB[2 * q], wB[2 * q + 1]};
#pragma unroll
            for (int s = 0; s < S; s++) {
                const int i = q * S + s;
                const f32x4 gq = lds_g(g4 + 64 * (i >> 1));
                const f32x2 gv = (i & 1) ? f32x2{gq[2], gq[3]} : f32x2{gq[0], gq[1]};
                accA.v[s] = NEG ? accA.v[s] - a2 * gv : accA.v[s] + a2 * gv;
                accB.v[s] = NEG ? accB.v[s] - b2 * gv : accB.v[s] + b2 * gv;
            }
        }
    } else {
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const f64x2 *g2 = reinterpret_cast<const f64x2 *>(GL) + lane;       // [i][lane]
#pragma unroll
        for (int t = 0; t < 8; t++) {
#pragma unroll
            for (int s = 0; s < S; s++) {
                const int e = t * S + s;
                const f64x2 gq = lds_g(g2 + 64 * (e >> 1));
                const T gv = (e & 1) ? gq[1] : gq[0];
                accA.v[s] = NEG ? accA.v[s] - wA[t] * gv : accA.v[s] + wA[t] * gv;
                accB.v[s] = NEG ? accB.v[s] - wB[t] * gv : accB.v[s] + wB[t] * gv;
            }
        }
    }
}
// the same from the lane's register copy of its Gramian elements
template <bool NEG, typename T, int S>
__device__ __forceinline__ void gram_pass_two_regs(const GramRegs<T, S> &R, const T (&wA)[8], const T (&wB)[8], PassAcc<T> &accA, PassAcc<T> &accB)
{
    if constexpr (std::is_same<T, float>::value) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x2 a2 = f32x2{wA[2 * q], wA[2 * q + 1]}, b2 = f32x2{wB[2 * q], wB[2 * q + 1]};
#pragma unroll
            for (int s = 0; s < S; s++) {
                accA.v[s] = NEG ? accA.v[s] - a2 * R.v[q][s] : accA.v[s] + a2 * R.v[q][s];
                accB.v[s] = NEG ? accB.v[s] - b2 * R.v[q][s] : accB.v[s] + b2 * R.v[q][s];
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int s = 0; s < S; s++) {
                accA.v[s] = NEG ? accA.v[s] - wA[t] * R.v[t][s] : accA.v[s] + wA[t] * R.v[t][s];
                accB.v[s] = NEG ? accB.v[s] - wB[t] * R.v[t][s] : accB.v[s] + wB[t] * R.v[t][s];
            }
    }
}

// sum over the four entry groups of a half (lane bits 4 and 3) of eight per-lane values: lane (jj, ll) ends with the totals of
// v[2 jj] and v[2 jj + 1], i.e. the elements ll + 16 jj and ll + 16 jj + 8 of the row's vector
template <typename T>
__device__ __forceinline__ void treduce_half(const T (&v)[8], int lane, T &o0, T &o1)
{
    const bool b3 = (lane & 8) != 0;
    const T u0 = lanes::tswap16_add(v[0], v[4]), u1 = lanes::tswap16_add(v[1], v[5]);
    const T u2 = lanes::tswap16_add(v[2], v[6]), u3 = lanes::tswap16_add(v[3], v[7]);
    o0 = (b3 ? u2 : u0) + lanes::recv_xor8(u0, u2);
    o1 = (b3 ? u3 : u1) + lanes::recv_xor8(u1, u3);
}

// GREG_ (NE = 4 only): the lane's Gramian elements in registers for the whole launch (16 S registers in double precision, as in the
// one-row tiny kernel) instead of read from LDS in every pass
// register budget: two wavefronts per SIMD in double precision (the 32-slot tile alone is 16 S registers), three in single --
// two with the register copy of the Gramian
#ifndef CMF_PAIR_WAVES_F64
#define CMF_PAIR_WAVES_F64 2
#endif
#ifndef CMF_PAIR_WAVES_F32
#define CMF_PAIR_WAVES_F32 3
#endif
template <typename T, bool GREG> constexpr int pair_waves_per_simd() { return sizeof(T) == 8 ? CMF_PAIR_WAVES_F64 : (GREG ? 2 : CMF_PAIR_WAVES_F32); }

// NE: 4 = every row of the launch has <= 16 entries, 8 = <= 32 entries, 0 = by pair (the 16-slot tile where both rows have <= 16)
template <typename T, int S, bool IMPLICIT, bool GRAMX = false, int NE = 0, bool GREG_ = false>
__global__ void __launch_bounds__(256, (pair_waves_per_simd<T, GREG_>()))
cg_rows_pair_kernel(const CgParams<T> P)
{
    static_assert(NE == 0 || NE == 4 || NE == 8, "entries per lane");
    constexpr bool GRAM = IMPLICIT || GRAMX;
    static_assert(!GREG_ || NE == 4, "the register copy of the Gramian fits beside the 16-slot tile only");
    constexpr bool GREG = GRAM && GREG_;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *G = reinterpret_cast<T *>(smem_raw);
    __shared__ __attribute__((aligned(16))) T s_pv[4][2][64];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, q = lane & 31, jj = q >> 3, ll = lane & 7, gj = lane >> 3;
    const int k = P.k;
    if (GRAM) {
        if constexpr (GREG) stage_gramian<T, S>(G, P.BtB, k, tid, blockDim.x);        // (padded row layout: what GramRegs::load reads)
        else stage_gramian_lanes<T, S>(G, P.BtB, k, tid, blockDim.x);
        __syncthreads();
    }
    GramRegs<T, GREG ? S : 1> greg;
    if constexpr (GREG) greg.load(G, lane);
    T *pvw = &s_pv[wave][0][0];
    // this lane's two vector elements
    const int e0 = ll + 16 * jj, e1 = e0 + 8;
    const bool live0 = e0 < k, live1 = e1 < k;

    // Pairs are formed inside the two length classes of the launch -- rows of more than 16 entries (the first n_long positions of
    // the processing order, 32-slot tile) and rows of at most 16 (16-slot tile) -- so that the tile, and with it the order of every
    // sum of a row, depends on the row alone: the same row gives the same bits whatever it is paired with (a shard, a part of a
    // block and the whole matrix pair differently).  A class with an odd number of rows leaves one half idle in its last pair.
    const int n_long = (NE == 4) ? 0 : (NE == 8) ? P.nrows : min(P.pair_split, P.nrows);
    const int npairs_long = (n_long + 1) >> 1;
    const int npairs = npairs_long + ((P.nrows - n_long + 1) >> 1);
    const int nwaves = gridDim.x * 4;
    struct Pre { int row, nnz; unsigned long long st; int idx; T x, g, a0, a1; bool shrt; };
    struct Desc { int row, nnz; unsigned long long st; bool shrt; };
    auto load_desc = [&](int pair, auto &r) {
        r.row = 0; r.nnz = 0; r.st = 0;
        r.shrt = pair >= npairs_long;                             // uniform over the wavefront
        const int pos = r.shrt ? n_long + 2 * (pair - npairs_long) + h : 2 * pair + h;
        if (pair < npairs && pos < (r.shrt ? P.nrows : n_long)) {
            const RowDesc d = P.desc[pos];
            r.row = d.row; r.nnz = d.nnz; r.st = d.st;
        }
    };
    auto load_pre = [&](Pre &r) {
        r.idx = 0; r.x = T(0); r.g = T(1); r.a0 = T(0); r.a1 = T(0);
        const int ent = r.shrt ? (q >> 1) : q;
        if (ent < r.nnz) {
            const size_t pos = (size_t)r.st + (size_t)ent;
            r.idx = P.indices[pos];
            r.x = P.values[pos];
            r.g = entry_weight<T, IMPLICIT>(P, pos);
            if (!IMPLICIT && P.bias_sub != nullptr) r.x -= P.bias_sub[r.idx];
        }
        if (r.nnz > 0) {
            const T *arow = P.A + (size_t)r.row * P.lda;
            if (live0) r.a0 = arow[e0];
            if (live1) r.a1 = arow[e1];
        }
    };

    auto solve = [&](const Pre &pr, const auto &tile, auto short_tag) {
        constexpr bool T4 = decltype(short_tag)::value;          // 16-slot tile: lanes 2 t, 2 t + 1 of a group carry entry 4 jj + t
        const int nnz = pr.nnz;
        T lam = P.lam, lam_last = P.lam_last;
        if (GRAMX && P.kc > 0) {                              // rows of the block system: collective.c:1285-1355
            if (P.scale_lam || P.scale_lam_sideinfo) {
                T mult = (P.wsum != nullptr) ? P.wsum[pr.row] : (T)nnz;
                if (P.scale_lam_sideinfo) mult += (T)P.p_side;
                lam *= mult; lam_last *= mult;
            }
        } else if (!IMPLICIT && P.scale_lam) {                // common.c:679-723
            const T mult = (P.wsum != nullptr) ? P.wsum[pr.row] : (T)nnz;
            lam *= mult;
            if (!P.scale_bias_const) lam_last *= mult;
        }
        const T d0 = (!IMPLICIT && e0 == k - 1) ? lam_last : lam, d1 = (!IMPLICIT && e1 == k - 1) ? lam_last : lam;
        const bool valid = (T4 ? (q >> 1) : q) < nnz;
        T a0 = pr.a0, a1 = pr.a1;
        auto run_pass = [&](T v0, T v1, auto mode_tag, T &o0, T &o1) {
            constexpr int MODE = decltype(mode_tag)::value;
            asm volatile("" ::: "memory");
            // the two vectors through LDS: own row's replicated elements, both rows' Gramian weights
            pvw[h * 64 + e0] = v0;
            pvw[h * 64 + e1] = v1;
            __builtin_amdgcn_wave_barrier();
            T vrep[S];
#pragma unroll
            for (int s = 0; s < S; s++) vrep[s] = pvw[h * 64 + ll + 8 * s];
            T own[8];
#pragma unroll
            for (int s = 0; s < 8; s++) own[s] = T(0);
            if constexpr (GRAM) {
                T wA[8], wB[8];
#pragma unroll
                for (int t = 0; t < 8; t++) { wA[t] = pvw[gj * 8 + t]; wB[t] = pvw[64 + gj * 8 + t]; }
                PassAcc<T> accA, accB;
                accA.zero(); accB.zero();
                if constexpr (GREG) gram_pass_two_regs<MODE == 0, T, S>(greg, wA, wB, accA, accB);
                else gram_pass_two<MODE == 0, T, S>(G, wA, wB, accA, accB, lane);
                T oa[8], ob[8];
                accA.close(oa); accB.close(ob);
#pragma unroll
                for (int s = 0; s < S; s++) own[s] = lanes::tswap32_add(oa[s], ob[s]);      // lanes 0..31: row 0's total, 32..63: row 1's
            }
            __builtin_amdgcn_wave_barrier();
            PassAcc<T> tacc;
            tacc.zero();
            if constexpr (T4) tile_pass4<T, S, IMPLICIT, MODE>(tile, vrep, pr.x, valid, tacc, lane, pr.g);
            else tile_pass<T, S, IMPLICIT, MODE>(tile, vrep, pr.x, valid, tacc, lane, pr.g);
            T to[8];
            tacc.close(to);
#pragma unroll
            for (int s = 0; s < S; s++) own[s] += to[s];
            treduce_half<T>(own, lane, o0, o1);
        };
        // ---- residual (common.c:1932-1943 / :1112-1139) ----
        T r0, r1;
        run_pass(a0, a1, std::integral_constant<int, 0>{}, r0, r1);
        r0 -= d0 * a0; r1 -= d1 * a1;
        if (GRAMX && P.rconst != nullptr) {
            if (live0) r0 += P.rconst[(size_t)pr.row * P.ldr + e0];
            if (live1) r1 += P.rconst[(size_t)pr.row * P.ldr + e1];
        }
        if (!live0) r0 = T(0);
        if (!live1) r1 = T(0);
        T p0 = r0, p1 = r1;
        T r_old = half_sum(r0 * r0 + r1 * r1);
        bool done = (r_old <= (T)1e-12) || nnz <= 0;            // common.c:1952 / :1147; a half without a row idles
        for (int step = 0; step < P.max_cg_steps && __builtin_amdgcn_ballot_w64(!done) != 0ull; step++) {
            T Ap0, Ap1;
            run_pass(p0, p1, std::integral_constant<int, 1>{}, Ap0, Ap1);
            Ap0 += d0 * p0; Ap1 += d1 * p1;
            if (!live0) Ap0 = T(0);
            if (!live1) Ap1 = T(0);
            const T pAp = half_sum(Ap0 * p0 + Ap1 * p1);
            const T alpha = done ? T(0) : cg_div(r_old, pAp);
            a0 += alpha * p0; a1 += alpha * p1;
            r0 -= alpha * Ap0; r1 -= alpha * Ap1;
            const T r_new = half_sum(r0 * r0 + r1 * r1);
            if (!done) {
                if (r_new <= (T)1e-8) done = true;              // common.c:1979 / :1180
                else {
                    const T beta = cg_div(r_new, r_old);
                    p0 = p0 * beta + r0; p1 = p1 * beta + r1;
                    r_old = r_new;
                }
            }
        }
        if (nnz > 0) {
            T *arow = P.A + (size_t)pr.row * P.lda;
            if (live0) arow[e0] = a0;
            if (live1) arow[e1] = a1;
        }
    };

    // pairs are claimed dynamically, two ahead (descriptor -> indices / warm start -> gather), as cg_rows_kernel claims rows
    int pix = blockIdx.x * 4 + wave;
    const int cslot = pix % CG_NCOUNTERS;
    int *const my_counter = P.counter + cslot * CG_COUNTER_STRIDE;
    const int cbase = nwaves + cslot;
    auto issue_claim = [&]() -> int {
        int v = 0;
        if (lane == 0) v = atomicAdd(my_counter, 1);
        return v;
    };
    const int c1 = issue_claim(), c2 = issue_claim();
    int pnxt = cbase + CG_NCOUNTERS * __builtin_amdgcn_readfirstlane(c1);
    int pnn = cbase + CG_NCOUNTERS * __builtin_amdgcn_readfirstlane(c2);
    Pre cur, nxt;
    Desc nn;
    load_desc(pix, cur);
    load_desc(pnxt, nxt);
    load_pre(cur);
    int pend = issue_claim();
#ifdef CMF_CG_TICKS
    unsigned long long tk_wait = 0, tk_pass = 0, tk_rows = 0;
    const unsigned long long tk_begin = CMF_TICK();
#endif
    while (pix < npairs) {
        const bool shrt = cur.shrt;
        RegTile<T, S> tile;                     // (the 16-slot tile uses its first four entries per lane group)
        if constexpr (NE == 4) load_tile_half<T, S, 4>(tile, P.B, P.ldb, k, cur.idx, cur.nnz, lane);
        else if constexpr (NE == 8) load_tile_half<T, S, 8>(tile, P.B, P.ldb, k, cur.idx, cur.nnz, lane);
        else {
            if (shrt) load_tile_half<T, S, 4>(tile, P.B, P.ldb, k, cur.idx, cur.nnz, lane);
            else load_tile_half<T, S, 8>(tile, P.B, P.ldb, k, cur.idx, cur.nnz, lane);
        }
        load_desc(pnn, nn);
        load_pre(nxt);
#ifdef CMF_CG_TICKS
        const unsigned long long tk0 = CMF_TICK();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tk1 = CMF_TICK();
        tk_wait += tk1 - tk0;
#endif
        if constexpr (NE == 4) solve(cur, tile, std::true_type{});
        else if constexpr (NE == 8) solve(cur, tile, std::false_type{});
        else {
            if (shrt) solve(cur, tile, std::true_type{});
            else solve(cur, tile, std::false_type{});
        }
#ifdef CMF_CG_TICKS
        tk_pass += CMF_TICK() - tk1; tk_rows += 2;
#endif
        const int p3 = cbase + CG_NCOUNTERS * __builtin_amdgcn_readfirstlane(pend);
        cur = nxt; nxt.row = nn.row; nxt.nnz = nn.nnz; nxt.st = nn.st; nxt.shrt = nn.shrt;
        pix = pnxt; pnxt = pnn; pnn = p3;
        pend = issue_claim();
    }
#ifdef CMF_CG_TICKS
    cg_ticks_flush(P.ticks, 4, tk_wait, tk_pass, tk_rows, CMF_TICK() - tk_begin, lane);
#endif
}

}  // namespace cmfhip
