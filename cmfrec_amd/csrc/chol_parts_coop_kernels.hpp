// chol_parts_coop_kernels.hpp -- round 6: the rank-k update of the eight-block rows in double precision (k_t = 128 / 129: config 3) by
// two wavefronts per row that GATHER THE ROW'S ENTRIES ONCE BETWEEN THEM.
//
// chol_parts_producer_kernel (chol_wave_kernels.hpp) gives each of the two wavefronts 18 of the 36 tiles, and each wavefront loads every
// gathered row for itself -- 8 and 7 of its 8 column blocks -- straight into its MFMA operands: 1024 workgroups pulling 2 x 4 KB per
// step of four entries, 4.8 TB/s of requests in the item step.  What sharing the gather buys on config 3 (profiles/r06/r06_t_*, same
// box, alternating): 12.97 -> 12.80 ms -- little, and the same kernel with every gather confined to 1024 rows of the opposing matrix
// (-DCMF_COOP_IDXMASK=1023) runs no faster: the rank-k update is NOT bound by its gather, neither by its bandwidth nor by its latency.
// The counters say the matrix pipe is busy 0.68 of the launch with two wavefronts per SIMD (SQ_VALU_MFMA_BUSY_CYCLES against
// SQ_WAVE_CYCLES / 2); what keeps it from the 0.95 a bare loop of the same instruction reaches is inside the wavefront.
//
// Here the 128 threads of the workgroup load a step's four rows together -- thread t the doubles (t & 31) + 32 j, j < 4, of entry t >> 5
// (256 contiguous bytes per entry and load instruction, no alignment beyond the element's) -- G steps ahead in registers, write them
// to one of two LDS slots a step before their use, and both wavefronts read their operands from the slot (rows padded to 144 doubles: the two halves of a ds_read_b64 land on disjoint banks).  The entry's value,
// the opposing bias and the border column ride as one more load of the threads 0 / 1 / 2 of every entry's 32.  One s_barrier per step.
// Arithmetic, tile assignment and output are chol_part_rank_k's (the sums are taken in the same order: the partials are bit for bit the
// same), so the factorisation kernels consume them unchanged.
#pragma once
#include <hip/hip_runtime.h>
#include "chol_wave_kernels.hpp"

namespace cmfhip {

constexpr int PC_ROW = 144;                      // doubles per staged row: 128 + 16
constexpr int PC_SLOT = 4 * PC_ROW + 16;         // a step's four rows + their meta values (4 x {x, opposing bias, border, valid})

// one work item (row or slice): entries [st, st + nnz), partial written to pp.  Q: this wavefront's half of the tiles.
// EXPL: explicit-feedback weights (an entry's matrix weight is one): the operand is the gathered row itself, nothing is multiplied or
// selected per step -- the entries past the end of the item are zeroed where they are written to the slot (the item's last step
// only).  Vector instructions run on the datapath the double-precision matrix instructions use, so each one off the step is
// matrix-pipe time.  The sums are the general form's bit for bit (a product with one, a product with zero).
template <typename T, bool BORDER, int Q, int G, bool EXPL>
__device__ __forceinline__ void chol_part_rank_k_coop(const CholParams<T> &P, size_t st, int nnz, T *__restrict__ pp, int lane, int tid,
                                                      T *__restrict__ ring)
{
    using Mf = CholMfma<T>;
    using vec = typename Mf::vec;
    constexpr int NB = 8, NP = 2;
    constexpr int NR = NB / NP;
    constexpr int R0 = cq_row(NP, Q, 0);
    constexpr int NTP = cq_off(NP, Q, NR);
    constexpr int NT = NB * (NB + 1) / 2;
    vec acc[NTP];
#pragma unroll
    for (int i = 0; i < NTP; i++) acc[i] = vec{0, 0, 0, 0};
    const int lm = lane & 15, g = lane >> 4;
    const int bcol = P.kt - P.koff - 1;
    const bool impl_w = (P.mode == CHOL_IMPLICIT || P.mode == CHOL_COLLECTIVE_IMPLICIT);
    T rp[NR], gp[NR], gam = T(0), rbs = T(0);
#pragma unroll
    for (int i = 0; i < NR; i++) { rp[i] = T(0); gp[i] = T(0); }
    const int nsteps = (nnz + 3) >> 2;
    // ---- loader side: this thread's entry of a step, its four columns, its role among the entry's meta values ----
    const int le = tid >> 5, lc = tid & 31, role = tid & 31;       // columns lc + 32 j: 256 contiguous bytes per entry and load
    // (the memory counter retires loads in order: an index is loaded when its register set is filled for the step BEFORE -- G steps before
    //  the rows it addresses are issued -- so waiting for it drains nothing that should still be in flight)
    T sv[G][4], sx[G];
    int idxq[G];
    // (e >= 0 and nnz >= 1 wherever these run: one clamp; the row offset as one 32 x 32 -> 64-bit multiply-add -- indices and leading
    //  dimensions are below 2^31 -- instead of the 64 x 64-bit product's three quarter-rate instructions)
    const unsigned ldb32 = (unsigned)P.ldb;
    auto load_idx = [&](int s, int step) {
        const int e = 4 * step + le;
        idxq[s] = P.indices[st + (unsigned)min(e, nnz - 1)];
    };
    auto issue_rows = [&](int s, int step) {
        const int e = 4 * step + le;
        const size_t pos = st + (unsigned)min(e, nnz - 1);
#ifdef CMF_COOP_IDXMASK
        const int idxn = idxq[s] & CMF_COOP_IDXMASK;      // experiment: every gather inside a few rows (what the kernel does without memory latency)
#else
        const int idxn = idxq[s];
#endif
        const T *rowp = P.B + (unsigned long long)(unsigned)idxn * ldb32;
#pragma unroll
        for (int j = 0; j < 4; j++) sv[s][j] = rowp[lc + 32 * j];
        // one more value per thread, always from a valid address and untouched until it is written to the slot (a use here would wait
        // for the load just issued): 0 the entry's value, 1 the opposing bias, 2 the border column
        const T *xp = (role == 1 && P.bias_sub != nullptr) ? P.bias_sub + idxn : (BORDER && role == 2) ? rowp + bcol : P.values + pos;
        sx[s] = *xp;
    };
    // meta values of the slot: {x, opposing bias, border, 1 if the entry exists}
    auto write_slot = [&](int s, int slot, int step) {
        T *dst = ring + (size_t)slot * PC_SLOT;
#pragma unroll
        for (int j = 0; j < 4; j++) dst[le * PC_ROW + lc + 32 * j] = sv[s][j];
        T mv = sx[s];
        if (role == 1 && P.bias_sub == nullptr) mv = T(0);
        if constexpr (EXPL) {
            if (4 * step + 4 > nnz && 4 * step + le >= nnz) {          // (uniform first test: the last step of the item and beyond)
#pragma unroll
                for (int j = 0; j < 4; j++) dst[le * PC_ROW + lc + 32 * j] = T(0);
                mv = T(0);                                             // ... and its value, bias and border element
            }
            if (role < 3) dst[4 * PC_ROW + 4 * le + role] = mv;
        } else {
            if (role == 3) mv = (4 * step + le < nnz) ? T(1) : T(0);
            if (role < 4) dst[4 * PC_ROW + 4 * le + role] = mv;
        }
    };
    if (nsteps > 0) {
        static_for<0, G>([&](auto sc) { constexpr int s = decltype(sc)::value; load_idx(s, s); });
        static_for<0, G>([&](auto sc) { constexpr int s = decltype(sc)::value; issue_rows(s, s); load_idx(s, s + G); });
        write_slot(0, 0, 0);
        issue_rows(0, G);
        load_idx(0, 2 * G);
    }
    __syncthreads();
    const int niter = (nsteps + G - 1) / G;
    for (int it = 0; it < niter; it++) {
        static_for<0, G>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            const int i = it * G + s;                  // the step computed now, in slot i & 1
            // the next step's rows into the other slot, its registers refilled G steps ahead -- unconditionally (clamped addresses past the
            // end of the item): one path through the loads, so the compiler's memory-counter waits stay counted
            constexpr int sn = (s + 1) % G;
            write_slot(sn, (i + 1) & 1, i + 1);
            issue_rows(sn, i + 1 + G);
            load_idx(sn, i + 1 + 2 * G);
            if (i < nsteps) {
                const T *src = ring + (size_t)(i & 1) * PC_SLOT;
                const T *mrow = src + 4 * PC_ROW + 4 * g;
                const T x = mrow[0] - mrow[1];
                const T bv = BORDER ? mrow[2] : T(0);
                T ws = T(1), xw = x;
                if constexpr (!EXPL) {
                    const bool vld = mrow[3] != T(0);
                    ws = impl_w ? x : T(1);               // common.c:2091-2095, collective.c:2103-2108 vs common.c:1007-1012
                    xw = impl_w ? x + T(1) : x;           // common.c:2082-2085, collective.c:2097-2101 vs common.c:991-996
                    if (!vld) { ws = T(0); xw = T(0); }
                }
                const T *orow = src + g * PC_ROW + lm;
                T o[NB];
                static_for<R0, NB>([&](auto bc) {
                    constexpr int b = decltype(bc)::value;
                    o[b] = orow[16 * b];
                });
                static_for<0, NR>([&](auto ic) {
                    constexpr int ii = decltype(ic)::value;
                    constexpr int R = cq_row(NP, Q, ii), OFF = cq_off(NP, Q, ii);
                    const T a = EXPL ? o[R] : o[R] * ws;
                    static_for<R, NB>([&](auto bjc) {
                        constexpr int bj = decltype(bjc)::value;
                        acc[OFF + bj - R] = Mf::mma(a, o[bj], acc[OFF + bj - R]);
                    });
                    rp[ii] += xw * o[R];
                    if (BORDER) gp[ii] += (EXPL ? bv : ws * bv) * o[R];
                });
                if (BORDER && Q == 0) { gam += (EXPL ? bv : ws * bv) * bv; rbs += xw * bv; }
            }
            __syncthreads();
        });
    }
    static_for<0, NR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int R = cq_row(NP, Q, i), OFF = cq_off(NP, Q, i);
#pragma unroll
        for (int j = 0; j < NB - R; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) pp[(size_t)wtix(R, R + j, NB) * 256 + r * 64 + lane] = acc[OFF + j][r];
    });
    T *pv = pp + (size_t)NT * 256;
    auto over_groups = [&](T v) -> T { v = lanes::tswap16_add(v, v); return lanes::tswap32_add(v, v); };
    static_for<0, NR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int R = cq_row(NP, Q, i);
        const T v = over_groups(rp[i]);
        if (lane < 16) pv[16 * R + lane] = v;
        if (BORDER) {
            const T w = over_groups(gp[i]);
            if (lane < 16) pv[16 * NB + 16 * R + lane] = w;
        }
    });
    if (BORDER && Q == 0) {
        gam = over_groups(gam); rbs = over_groups(rbs);
        if (lane == 0) { pv[32 * NB] = gam; pv[32 * NB + 1] = rbs; }
    }
}

// workgroup = two wavefronts = one work item at a time; two wavefronts per SIMD (four workgroups per CU)
template <typename T, bool BORDER, int G, bool EXPL>
__global__ void __launch_bounds__(128, 2)
chol_parts_coop_kernel(const CholParams<T> P, const RowDesc *__restrict__ desc, const CholSlices<T> SL)
{
    constexpr size_t PART = chol_wave_part_elems(8);
    __shared__ int s_next;
    __shared__ __attribute__((aligned(16))) T ring[2 * PC_SLOT];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int rix = P.row_first + blockIdx.x;
    while (rix < P.nrows) {
        if (tid == 0) s_next = atomicAdd(P.counter, 1);
        int ritem, sfirst = 0, scount;
        if (rix < SL.n_slices) { ritem = SL.vrow[rix]; sfirst = SL.first[rix]; scount = SL.count[rix]; }
        else { ritem = SL.n_heavy + (rix - SL.n_slices); scount = -1; }
        const RowDesc d = desc[ritem];
        const int nnz_row = __builtin_amdgcn_readfirstlane(d.nnz);
        const size_t st_row = ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(d.st >> 32)) << 32) |
                              (unsigned)__builtin_amdgcn_readfirstlane((int)(d.st & 0xffffffffu));
        const int nnz = (scount >= 0) ? __builtin_amdgcn_readfirstlane(scount) : nnz_row;
        const size_t st = st_row + (size_t)__builtin_amdgcn_readfirstlane(sfirst);
        T *pp = SL.part + (size_t)(rix - SL.part_base) * PART;
        if (wave == 0) chol_part_rank_k_coop<T, BORDER, 0, G, EXPL>(P, st, nnz, pp, lane, tid, ring);
        else chol_part_rank_k_coop<T, BORDER, 1, G, EXPL>(P, st, nnz, pp, lane, tid, ring);
        __syncthreads();
        rix = P.row_first + (int)gridDim.x + s_next;
        __syncthreads();
    }
}

}  // namespace cmfhip
