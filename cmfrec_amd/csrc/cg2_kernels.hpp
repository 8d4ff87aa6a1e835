// cg2_kernels.hpp -- second generation of the register-tiled CG row update (round 3).
//
// Same operator, same arithmetic per entry and the same persistent / dynamically scheduled row loop as cg_rows_kernel
// (cg_kernels.hpp; reference: factors_implicit_cg /root/reference/src/common.c:1914-1986, factors_explicit_cg :1098-1188,
// row loops :3349-3368 / :3259-3299), rebuilt around what the counters of round 2 showed: the row kernels keep the vector
// ALU ~75 % busy and only ~half of its time goes into the FMAs of the tile products -- the rest are cross-lane moves (a
// double costs two DPP instructions), selects, and products with padding.  Changes:
//   * entries are dealt ROUND-ROBIN over the eight lane groups (entry e of a tile -> group e % 8, slot e / 8), so a tile with
//     cnt entries uses the first ceil(cnt / 8) slots of EVERY lane: the gather and both tile products run over the slots in
//     use only (wave-uniform slot count, scalar branches) instead of over all 8 -- rows of 33..64 entries average 6 slots;
//   * the vector a pass works with goes through a 512-byte per-wave LDS buffer once per pass and comes back in both
//     replicated forms the pass needs (columns ll + 8 s for the tile products, rows jj + 8 t for the Gramian product): no
//     ds_bpermute, no DPP broadcasts;
//   * the Gramian product takes rows jj + 8 t (t < S) instead of 8 jj + t (t < 8): S x S = 49 FMAs and LDS reads per lane and
//     pass at k = 50 instead of 56, leading dimension 8 (S | 1) keeps the reads conflict-free without padding columns;
//   * weights are broadcast inside a lane group with row_newbcast DPP moves (v_mov_b64_dpp: one instruction per double and
//     half-row instead of four 32-bit moves per double);
//   * the transposed reduction of a tile that uses at most half of its slots skips the selects of its first exchange stage.
// NT = slots per lane of a register tile: 8 (64 entries, 2 wavefronts per SIMD in double precision) or 4 (32 entries, the
// tiny rows, 4 wavefronts per SIMD; lanes 2t and 2t+1 of a group both carry slot t).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "cg_kernels.hpp"
#include "lanes.hpp"

namespace cmfhip {

template <typename T, int S, int NT>
struct Tile2 {
    T v[NT][S];
};

// leading dimension of the staged Gramian: rows jj + 8 t of the four lane groups of a half-wave land on distinct banks when
// the row stride is an odd multiple of 8 elements
__host__ __device__ constexpr int gram2_ld(int S) { return 8 * (S | 1); }
__host__ __device__ constexpr int gram2_elems(int S) { return 8 * S * gram2_ld(S); }

template <typename T, int S>
__device__ __forceinline__ void stage_gramian2(T *__restrict__ G, const T *__restrict__ BtB, int k, int tid, int nthreads)
{
    constexpr int LD = gram2_ld(S);
    for (int e = tid; e < 8 * S * LD; e += nthreads) {
        const int r = e / LD, c = e % LD;
        G[e] = (r < k && c < k) ? BtB[(size_t)r * k + c] : T(0);
    }
}

// entry of a tile this lane loads the index / value of: NT = 8: lane (jj, t) <-> slot t of group jj, NT = 4: lanes (jj, 2t) and
// (jj, 2t + 1) <-> slot t; slot t of group jj is entry 8 t + jj
template <int NT>
__device__ __forceinline__ int entry_of_lane(int lane)
{
    if constexpr (NT == 4) return ((lane & 7) >> 1) * 8 + (lane >> 3);
    else return ((lane & 7) < NT) ? (lane & 7) * 8 + (lane >> 3) : 8 * NT;       // NT = 5 .. 7: the last lanes of a group own no slot
}
template <int NT, int t, typename T>
__device__ __forceinline__ T slot_bcast(T x)
{
    if constexpr (NT == 4) return lanes::bcast8<2 * t>(x);
    else return lanes::bcast8<t>(x);
}

// gather of the slots in use (nt of NT, wave-uniform: scalar branches): entries past the end of the tile re-read the row of the
// tile's first entry (their weight is forced to zero), factor columns past k re-read column k - 1 (their vector / Gramian entries
// are zero)
template <typename T, int S, int NT>
__device__ __forceinline__ void load_tile2(Tile2<T, S, NT> &tile, const T *__restrict__ Bm, size_t ldb, int k, int my_idx,
                                           int cnt, int nt, int lane, int t_first = 0)
{
    const int jj = lane >> 3, ll = lane & 7;
    const int first_idx = __builtin_amdgcn_readfirstlane(my_idx);
    const int col_last = min(ll + 8 * (S - 1), k - 1) - ll;
    const char *base = reinterpret_cast<const char *>(Bm + ll);
    const unsigned ldb_bytes = (unsigned)(ldb * sizeof(T));
    // (the slot count is made opaque at every use: otherwise the compiler keeps one lane mask per `t < nt` alive across the
    //  whole row -- SGPR pairs it then spills into VGPR lanes -- instead of one scalar compare in front of each branch)
    asm volatile("" : "+s"(nt));
    asm volatile("" : "+s"(t_first));
    static_for<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if (t >= t_first && t < nt) {
            const int its = slot_bcast<NT, t>(my_idx);
            const unsigned it = (unsigned)(((8 * t + jj) < cnt) ? its : first_idx);
            const T *rp = reinterpret_cast<const T *>(base + (unsigned long long)it * ldb_bytes);
#pragma unroll
            for (int s = 0; s < S; s++) tile.v[t][s] = rp[(s < S - 1) ? 8 * s : col_last];
        }
    });
}

// transposed sum over the 8 lanes of a group of c[0 .. NT): the lane that owns slot t ends with the total of c[t]
// (NT = 8: lane t; NT = 4: lanes 2t and 2t + 1).  LOW: the upper half of the slots is unused (zero) -- its exchange stage only
// adds the partner lane.
template <typename T, int NT, bool LOW>
__device__ __forceinline__ T treduce_slots(const T (&c)[NT], int lane)
{
    if constexpr (NT > 4 && NT < 8) {
        T c8[8];
#pragma unroll
        for (int t = 0; t < 8; t++) c8[t] = (t < NT) ? c[t] : T(0);
        return treduce_slots<T, 8, LOW>(c8, lane);
    } else if constexpr (NT == 8) {
        T u[4], q[2];
        bool h = (lane & 4) != 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if constexpr (!LOW) {
                T keep = h ? c[i + 4] : c[i];
                u[i] = keep + lanes::recv_xor4(c[i], c[i + 4]);
            } else {
                u[i] = c[i] + lanes::half_mirror(c[i]);
            }
        }
        h = (lane & 2) != 0;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            T keep = h ? u[i + 2] : u[i];
            T send = h ? u[i] : u[i + 2];
            q[i] = keep + lanes::xor2(send);
        }
        h = (lane & 1) != 0;
        T keep = h ? q[1] : q[0];
        T send = h ? q[0] : q[1];
        return keep + lanes::xor1(send);
    } else {
        static_assert(NT == 4, "slots per lane");
        T u[2];
        bool h = (lane & 4) != 0;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            if constexpr (!LOW) {
                T keep = h ? c[i + 2] : c[i];
                u[i] = keep + lanes::recv_xor4(c[i], c[i + 2]);
            } else {
                u[i] = c[i] + lanes::half_mirror(c[i]);
            }
        }
        h = (lane & 2) != 0;
        T keep = h ? u[1] : u[0];
        T send = h ? u[0] : u[1];
        T q = keep + lanes::xor2(send);
        return q + lanes::xor1(q);
    }
}

// One tile contribution over the nt slots in use:  c_j = B_j . vrep ; w_j = f(c_j, x_j) ; out[s] += sum_t w_j B_j[s]
template <typename T, int S, int NT, bool IMPLICIT, int MODE>
__device__ __forceinline__ void tile_pass2(const Tile2<T, S, NT> &tile, const T (&vrep)[S], T x, bool valid, int nt, T (&out)[8], int lane)
{
    T c[NT];
    asm volatile("" : "+s"(nt));
    // two slots per scalar branch: two independent FMA chains
    static_for<0, (NT + 1) / 2>([&](auto qc) {
        constexpr int t0 = 2 * decltype(qc)::value, t1 = (t0 + 1 < NT) ? t0 + 1 : t0;
        c[t0] = T(0); c[t1] = T(0);
        if (t1 > t0 && t1 < nt) {
            T a0 = tile.v[t0][0] * vrep[0], a1 = tile.v[t1][0] * vrep[0];
#pragma unroll
            for (int s = 1; s < S; s++) { a0 += tile.v[t0][s] * vrep[s]; a1 += tile.v[t1][s] * vrep[s]; }
            c[t0] = a0; c[t1] = a1;
        } else if (t0 < nt) {
            T a0 = tile.v[t0][0] * vrep[0];
#pragma unroll
            for (int s = 1; s < S; s++) a0 += tile.v[t0][s] * vrep[s];
            c[t0] = a0;
        }
    });
    T coef;
    if (nt <= (NT == 4 ? 2 : 4)) coef = treduce_slots<T, NT, true>(c, lane);
    else coef = treduce_slots<T, NT, false>(c, lane);
    const T w = pass_weight<T, IMPLICIT, MODE>(coef, x, valid);
    asm volatile("" : "+s"(nt));
    static_for<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if (t < nt) {
            const T wt = slot_bcast<NT, t>(w);
#pragma unroll
            for (int s = 0; s < S; s++) out[s] += wt * tile.v[t][s];
        }
    });
}

// out[s] += sum_t (+-) v[jj + 8 t] G[jj + 8 t][ll + 8 s]: the rows jj + 8 t of the staged Gramian, dealt over the W waves of a
// team; the vector comes from the wave's LDS copy (one broadcast read per row)
template <typename T, int S, int W, bool NEG>
__device__ __forceinline__ void gram_pass2(const T *__restrict__ G, const T *__restrict__ vb, T (&out)[8], int lane, int wr)
{
    constexpr int LD = gram2_ld(S);
    const int jj = lane >> 3, ll = lane & 7;
    lds_cv<T> *g0 = (lds_cv<T> *)(G + jj * LD + ll);          // single ds_read_b64 / b32, never paired or hoisted (cg_kernels.hpp)
#pragma unroll
    for (int t = 0; t < S; t++) {
        if (W > 1 && (t % W) != wr) continue;
        const T vj = NEG ? -vb[jj + 8 * t] : vb[jj + 8 * t];
#pragma unroll
        for (int s = 0; s < S; s++) out[s] += vj * g0[8 * t * LD + 8 * s];
    }
}

// ---- the next row's tile by LDS-DMA (PF builds, one wavefront per row) -----------------------------------------------------
// The register-tiled bins are bound by what they keep in flight: halving the resident teams of the 33..64 bin costs 1.57x, the
// fit T(w) = 0.16 + 0.22 / w ms says a third more bytes in flight would be worth ~20 % (profiles/r03_h) -- and the registers of
// a wavefront hold ONE tile.  So the gathered rows of the NEXT row of a wavefront travel into LDS while the current row is being
// solved: global_load_lds_dwordx4 moves 64 x 16 bytes per instruction from per-lane addresses to M0 + 16 lane -- chunk c of a
// tile is bytes 16 (c % cpr) .. of entry c / cpr (cpr = k sizeof / 16 chunks per gathered row), so the LDS image is the dense
// [entry][k] array -- no VGPRs, no waiting.  The DMAs are issued after the row's first pass (by then every load the compiler
// counts has been consumed: vmcnt is in order, and a wait for an older load would otherwise wait for the DMAs too) and land
// during the remaining passes; at the next row an s_waitcnt vmcnt(0), 7 ds_read_b64 per slot and lane, and the row starts
// without its gather.  Entries past the prefetched ones (P.pf_entries per tile, what fits the CU's LDS) are gathered as before.
typedef __attribute__((address_space(3))) void cg2_lds_void;
typedef __attribute__((address_space(1))) const void cg2_glb_void;

// issue the DMAs of the first `pfe` entries (whole slots) of a tile: idx = the tile's indices (lane <-> entry as entry_of_lane),
// cnt = entries of the tile (slots past them re-read the first entry's row), dst = this wave's LDS buffer
template <typename T, int NT>
__device__ __forceinline__ void prefetch_tile_lds(const CgParams<T> &P, unsigned char *dst, int idx, int cnt, int pfe, int lane)
{
    const int cpr = P.pf_cpr;
    int nchunks = pfe * cpr;
    const int first_idx = __builtin_amdgcn_readfirstlane(idx);
    const char *base = reinterpret_cast<const char *>(P.B);
    const unsigned ldb_bytes = (unsigned)(P.ldb * sizeof(T));
    // groups of 8 DMA instructions: the 8 index look-ups (ds_bpermute) of a group are issued together -- one LDS round trip per
    // group instead of one per instruction (a rolled loop with a look-up, its wait and the address arithmetic per trip cost
    // ~200 cycles per instruction, 16 instructions per row: 20 % of the bin's time, profiles/r03_i)
    constexpr int GQ = 8;
    for (int g0 = 0; g0 < nchunks; g0 += 64 * GQ) {
        int it[GQ], part[GQ];
#pragma unroll
        for (int u = 0; u < GQ; u++) {
            const int c = g0 + 64 * u + lane;
            const int e = (c * P.pf_magic) >> 16;                          // entry and 16-byte chunk of its row
            part[u] = c - e * cpr;
            // entry e = 8 t + jj lives in lane (jj, t) of the tile's index register (NT = 4: lanes (jj, 2t), (jj, 2t + 1))
            const int src = ((e & 7) << 3) | (NT == 4 ? ((e >> 3) << 1) : (e >> 3));
            const int v = __builtin_amdgcn_ds_bpermute(src << 2, idx);
            it[u] = (e < cnt) ? v : first_idx;
        }
#pragma unroll
        for (int u = 0; u < GQ; u++) {
            const int c0 = g0 + 64 * u;                                    // wave-uniform
            if (c0 < nchunks) {
                const char *g = base + (unsigned long long)(unsigned)it[u] * ldb_bytes + (part[u] << 4);
                if (c0 + lane < nchunks)
                    __builtin_amdgcn_global_load_lds((cg2_glb_void *)g, (cg2_lds_void *)(dst + (size_t)c0 * 16), 16, 0, 0);
            }
        }
    }
}

// the prefetched slots [0, nslots) of a tile from the wave's LDS buffer into the register tile
template <typename T, int S, int NT>
__device__ __forceinline__ void tile_from_lds(Tile2<T, S, NT> &tile, const unsigned char *src, int k, int nslots, int lane)
{
    const int jj = lane >> 3, ll = lane & 7;
    const int col_last = min(ll + 8 * (S - 1), k - 1);
    const T *row0 = reinterpret_cast<const T *>(src) + (size_t)jj * k;
    asm volatile("" : "+s"(nslots));
    static_for<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if (t < nslots) {
            const T *rp = row0 + (size_t)(8 * t) * k;
#pragma unroll
            for (int s = 0; s < S; s++) tile.v[t][s] = rp[(s < S - 1) ? ll + 8 * s : col_last];
        }
    });
}

#ifndef CMF_CG2_WAVES_NT8
#define CMF_CG2_WAVES_NT8 2
#endif
#ifndef CMF_CG2_WAVES_NT7
#define CMF_CG2_WAVES_NT7 3     // 5 .. 7 slots per lane: the tile leaves room for a third wavefront per SIMD in double precision
#endif
#ifndef CMF_CG2_WAVES_NT4
#define CMF_CG2_WAVES_NT4 4
#endif

// Persistent kernel, W wavefronts per row, RPB rows per workgroup (W == 1 only); the row loop, the dynamic claiming of rows
// and the software pipeline over rows are those of cg_rows_kernel.
template <typename T, int S, int NT, bool IMPLICIT, int W, int RPB, bool GRAMX = false, bool PF = false>
__global__ void __launch_bounds__(64 * W * RPB, (NT == 4 ? CMF_CG2_WAVES_NT4 : (NT < 8 ? CMF_CG2_WAVES_NT7 : CMF_CG2_WAVES_NT8)))
cg2_rows_kernel(const CgParams<T> P)
{
    constexpr bool GRAM = IMPLICIT || GRAMX;
    constexpr int TE = 8 * NT;                                               // entries per tile
    static_assert(!PF || W == 1, "the LDS prefetch is built for one wavefront per row");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *G = reinterpret_cast<T *>(smem_raw);                                  // [8 S][LD] (implicit / block systems)
    T *vbuf = G + (GRAM ? gram2_elems(S) : 0);                               // [W RPB][64] the vector of the current pass, per wave
    T *red = vbuf + W * RPB * 64;                                            // [RPB][2][W][64] (W > 1)
    // PF: per wave a buffer of P.pf_entries gathered rows (k elements each), behind the vectors
    unsigned char *pfbuf = reinterpret_cast<unsigned char *>(W > 1 ? red + RPB * 2 * W * 64 : red);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave / W;      // which concurrent row of this workgroup
    const int wr = wave % W;       // wave index inside the row team
    const int jj = lane >> 3, ll = lane & 7;
    const int k = P.k;
    T *vb = vbuf + wave * 64;
    unsigned char *pfb = pfbuf + (size_t)wave * (size_t)P.pf_entries * (size_t)k * sizeof(T);
    int pf_slots = 0;              // slots of the CURRENT row's tile that wait in this wave's LDS buffer (PF)

    const int nteams = gridDim.x * RPB;
    __shared__ int s_claim[4];
    const int cslot = (blockIdx.x * RPB + grp) % CG_NCOUNTERS;
    int *const my_counter = P.counter + cslot * CG_COUNTER_STRIDE;
    const int cbase = nteams + cslot;
    auto issue_claim = [&]() -> int {
        int v = 0;
        if (wr == 0 && lane == 0) v = atomicAdd(my_counter, 1);
        return v;
    };
    int rnxt, rnn;
    if (W == 1) {
        const int c1 = issue_claim(), c2 = issue_claim();
        rnxt = cbase + CG_NCOUNTERS * __builtin_amdgcn_readfirstlane(c1);
        rnn = cbase + CG_NCOUNTERS * __builtin_amdgcn_readfirstlane(c2);
    } else if (wr == 0 && lane == 0) {
        s_claim[2] = cbase + CG_NCOUNTERS * atomicAdd(my_counter, 1);
        s_claim[3] = cbase + CG_NCOUNTERS * atomicAdd(my_counter, 1);
    }
    if (GRAM) stage_gramian2<T, S>(G, P.BtB, k, tid, blockDim.x);
    if (GRAM || W > 1) __syncthreads();
    if (W > 1) { rnxt = s_claim[2]; rnn = s_claim[3]; }
    T *myred = red + (size_t)grp * 2 * W * 64;

    int buf = 0;   // cross-wave exchange buffer parity; persists across rows
    static_assert(W == 1 || RPB == 1, "multi-wave teams own their workgroup (barriers are per row)");

    struct Pre { int idx; T x; T a; };
    auto load_desc = [&](int rix_) -> RowDesc {
        RowDesc d; d.row = 0; d.nnz = 0; d.st = 0;
        if (rix_ < P.nrows) d = P.desc[rix_];
        d.row = __builtin_amdgcn_readfirstlane(d.row);
        d.nnz = __builtin_amdgcn_readfirstlane(d.nnz);
        d.st = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(d.st >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)(d.st & 0xffffffffu));
        return d;
    };
    const int my_e = entry_of_lane<NT>(lane);
    auto load_pre = [&](const RowDesc &d) -> Pre {
        Pre q; q.idx = 0; q.x = T(0); q.a = T(0);
        const int cnt = min(TE, d.nnz - wr * TE);
        if (my_e < cnt) {
            const size_t pos = d.st + (size_t)wr * TE + my_e;
            q.idx = P.indices[pos];
            q.x = P.values[pos];
            if (!IMPLICIT && P.bias_sub != nullptr) q.x -= P.bias_sub[q.idx];
        }
        if (d.nnz > 0 && lane < k) q.a = P.A[(size_t)d.row * P.lda + lane];
        return q;
    };
    int rix = blockIdx.x * RPB + grp;
    RowDesc dcur = load_desc(rix);
    RowDesc dnxt = load_desc(rnxt);
    Pre pcur = load_pre(dcur);
    int pend = issue_claim();      // the position after rnn; lands while this row is solved
    for (int it = 0; rix < P.nrows; it++) {
        const int row = dcur.row;
        const size_t st = dcur.st;
        const int nnz = dcur.nnz;
        const int ntiles = (nnz + TE - 1) / TE;
        const int my_ntiles = (ntiles > wr) ? (ntiles - wr + W - 1) / W : 0;
        const bool resident = my_ntiles <= 1;

        T lam = P.lam, lam_last = P.lam_last;
        if (GRAMX && P.kc > 0) {                              // rows of the block system: collective.c:1285-1355
            if (P.scale_lam || P.scale_lam_sideinfo) {
                T mult = (T)nnz;
                if (P.scale_lam_sideinfo) mult += (T)P.p_side;
                lam *= mult; lam_last *= mult;
            }
        } else if (!IMPLICIT && P.scale_lam) {                // common.c:679-723
            lam *= (T)nnz;
            if (!P.scale_bias_const) lam_last *= (T)nnz;
        }
        T *arow = P.A + (size_t)row * P.lda;
        T a_d = pcur.a;

        // first tile of this wave: gather now (critical path), then start the next row's loads
        Tile2<T, S, NT> tile;
        const int cnt0 = min(TE, nnz - wr * TE);
        const int nt0 = (cnt0 + 7) >> 3;                       // slots in use (wave-uniform)
        const T x_res = pcur.x;
        const bool valid_res = my_e < cnt0;
        if constexpr (PF) {
            if (pf_slots > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMAs (and everything older) have landed
            // the slots the buffer had no room for: gathered as before, in flight while the others move from LDS to registers
            if (cnt0 > 0 && nt0 > pf_slots) load_tile2<T, S, NT>(tile, P.B, P.ldb, k, pcur.idx, cnt0, nt0, lane, pf_slots);
            if (pf_slots > 0) tile_from_lds<T, S, NT>(tile, pfb, k, pf_slots, lane);
        } else {
            if (cnt0 > 0 && !CMF_DBG(P, 1)) load_tile2<T, S, NT>(tile, P.B, P.ldb, k, pcur.idx, cnt0, nt0, lane);
        }
        const RowDesc dnn = load_desc(rnn);
        const Pre pnxt = load_pre(dnxt);

        auto run_pass = [&](T vdist, auto mode_tag, bool first) -> T {
            constexpr int MODE = decltype(mode_tag)::value;
            // keep the staged Gramian in LDS (see cg_rows_kernel)
            asm volatile("" ::: "memory");
            // the pass' vector, once through LDS: columns ll + 8 s for the tile products, rows jj + 8 t for the Gramian product
            vb[lane] = vdist;
            __builtin_amdgcn_wave_barrier();
            T vrep[S];
#pragma unroll
            for (int s = 0; s < S; s++) vrep[s] = vb[ll + 8 * s];
            __builtin_amdgcn_wave_barrier();
            T out[8];
#pragma unroll
            for (int s = 0; s < 8; s++) out[s] = T(0);
            if constexpr (PF) {
                // one resident tile per row by construction (the bin holds rows of at most 8 NT entries): no gather code inside
                // the passes -- a load the compiler counts anywhere in them would put an s_waitcnt vmcnt(0) in front of its use,
                // and with it the DMAs of the next row, issued after the first pass, back on the critical path
                if (cnt0 > 0) tile_pass2<T, S, NT, IMPLICIT, MODE>(tile, vrep, x_res, valid_res, nt0, out, lane);
            } else
            for (int tl = wr; tl < ntiles; tl += W) {
                T x; bool valid; int nt;
                const bool have = (tl == wr) && (resident || first);   // still in registers
                if (!have) {
                    const int cnt = min(TE, nnz - tl * TE);
                    nt = (cnt + 7) >> 3;
                    valid = my_e < cnt;
                    const size_t pos = st + (size_t)tl * TE + my_e;
                    int my_idx = valid ? P.indices[pos] : 0;
                    x = valid ? P.values[pos] : T(0);
                    if (!IMPLICIT && P.bias_sub != nullptr && valid) x -= P.bias_sub[my_idx];
                    if (!CMF_DBG(P, 1)) load_tile2<T, S, NT>(tile, P.B, P.ldb, k, my_idx, cnt, nt, lane);
                } else {
                    x = x_res; valid = valid_res; nt = nt0;
                }
                if (!CMF_DBG(P, 4)) tile_pass2<T, S, NT, IMPLICIT, MODE>(tile, vrep, x, valid, nt, out, lane);
            }
            if (GRAM && !CMF_DBG(P, 2)) gram_pass2<T, S, W, MODE == 0>(G, vb, out, lane, wr);   // common.c:1932 / :1958; collective.c:2609-2643
            T tot = treduce8_high<T>(out, lane);                              // lane f <- element f
            if (W > 1) {
                T *rb = myred + (size_t)buf * W * 64;
                rb[wr * 64 + lane] = tot;
                if (first && wr == 0 && lane == 0) s_claim[it & 1] = cbase + CG_NCOUNTERS * pend;
                __syncthreads();
                tot = T(0);
#pragma unroll
                for (int w = 0; w < W; w++) tot += rb[w * 64 + lane];
                buf ^= 1;
            }
            return tot;
        };

        // ---- residual (common.c:1932-1943 / :1112-1139) ----
        T r_d = run_pass(a_d, std::integral_constant<int, 0>{}, true);
        if constexpr (PF) {
            // the next row's first slots set out for LDS now: the tile of this row is in registers, the buffer is free, and every
            // load the compiler counts has been consumed (see prefetch_tile_lds)
            const int cnt_n = min(TE, dnxt.nnz);
            const int nt_n = (cnt_n + 7) >> 3;
            pf_slots = (rnxt < P.nrows) ? min(nt_n, P.pf_entries >> 3) : 0;
            if (pf_slots > 0) prefetch_tile_lds<T, NT>(P, pfb, pnxt.idx, cnt_n, 8 * pf_slots, lane);
        }
        r_d -= lam * a_d;
        if (!IMPLICIT && lam != lam_last && lane == k - 1) r_d -= (lam_last - lam) * a_d;
        if (GRAMX && P.rconst != nullptr && lane < k) r_d += P.rconst[(size_t)row * P.ldr + lane];
        if (lane >= k) r_d = T(0);
        T p_d = r_d;
        T r_old = CMF_DBG(P, 8) ? T(1) : wave_sum(r_d * r_d);
        bool done = (r_old <= (T)1e-12);            // common.c:1952 / :1147
        for (int step = 0; step < P.max_cg_steps && !done; step++) {
            T Ap_d = run_pass(p_d, std::integral_constant<int, 1>{}, false);
            Ap_d += lam * p_d;
            if (!IMPLICIT && lam != lam_last && lane == k - 1) Ap_d += (lam_last - lam) * p_d;
            if (lane >= k) Ap_d = T(0);
            T alpha = CMF_DBG(P, 8) ? T(0.001) : r_old / wave_sum(Ap_d * p_d);
            a_d += alpha * p_d;
            r_d -= alpha * Ap_d;
            T r_new = CMF_DBG(P, 8) ? T(0.5) : wave_sum(r_d * r_d);
            if (r_new <= (T)1e-8) done = true;      // common.c:1979 / :1180
            else {
                p_d = p_d * (r_new / r_old) + r_d;
                r_old = r_new;
            }
        }
        if (wr == 0 && lane < k) arow[lane] = a_d;
        const int r3 = (W == 1) ? cbase + CG_NCOUNTERS * __builtin_amdgcn_readfirstlane(pend) : s_claim[it & 1];
        dcur = dnxt; dnxt = dnn; pcur = pnxt;
        rix = rnxt; rnxt = rnn; rnn = r3;
        pend = issue_claim();
    }
}

}  // namespace cmfhip
