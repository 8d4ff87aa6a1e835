// cg_mm_kernels.hpp -- the implicit model's short rows with the k x k Gramian product on the MATRIX PIPE.
//
// Reference: factors_implicit_cg, /root/reference/src/common.c:1914-1986 -- the symv of the first residual (:1932) and the
// symv of every CG step (:1958) -- called from the row loop of optimizeA_implicit (:3349-3368).
//
// The row kernels of cg_kernels.hpp compute `BtB v` row by row on the vector ALU from an LDS copy of the Gramian: 8 S FMAs,
// 4 S paired LDS reads and 16 cross-lane moves per lane and CG pass -- a quarter of the vector instructions of a row of at
// most 32 entries, and all of its LDS traffic.  A single row's `BtB v` is a matrix-VECTOR product and of no use to
// v_mfma_*_16x16x4; sixteen rows' are a matrix-MATRIX product  Y[k x 16] = BtB[k x k] P[k x 16].  So here a workgroup solves
// SIXTEEN rows in lock step -- sixteen wavefronts, one row each (17..32 entries), or eight wavefronts with two rows each
// (at most 16 entries) -- and every CG pass goes
//     every wavefront : its row's vector v (a for the first residual, p afterwards) -> LDS P[row][0..63]          | barrier A
//     every wavefront : replicated copy of v for the tile products straight from P (7 broadcast reads instead of 14 bpermutes)
//     every wavefront : its share (it, kq) of the product = (16-row block of the result, every MM_NQ-th 4-column step):
//                       chained v_mfma_*_16x16x4 with the Gramian block as A operand -- a few registers per lane, loaded once
//                       per launch, zero past k -- and P as B operand (lane l: P[row l & 15][4 kk + (l >> 4)], conflict-free
//                       with the row stride 66); the 16 x 16 partial result goes to LDS Y[kq][row][16 it + ..] in the
//                       accumulator layout.  Branch-free: every wavefront runs the same stream, the MFMAs are issued first and
//                       their results stored last, the tile products of the wavefront's own row (vector ALU) in between -- the
//                       matrix pipe works in the shadow of the vector ALU                                          | barrier B
//     every wavefront : y = sum over kq of Y[kq][row]; out = (tile part) -/+ y
// The rows of a batch take their data-dependent exits (1e-12 / 1e-8, common.c:1952, :1979) independently: a finished row keeps
// passing the barriers and its vector keeps riding through the MFMAs, its updates are masked; the batch ends when no row is
// active (a flag per row in LDS, read by everybody behind barrier A: uniform).
// Batches of sixteen consecutive positions of the processing order (rows sorted by length: equal work per wavefront) are
// claimed from the launch's padded counters like the rows of cg_rows_kernel, two batches ahead.
// Same arithmetic as the row kernels except for the order of the sums inside `BtB v` (the MFMA's fixed k order, then the two
// halves) and the point where that term joins the pass's total.
#pragma once
#include "cg_kernels.hpp"

namespace cmfhip {

template <typename T> struct MmOp;
template <> struct MmOp<double> {
    typedef double vec __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ vec mma(double a, double b, vec c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // accumulator register v of lane l: result row (l >> 4) + 4 v, column l & 15
    static __device__ __forceinline__ int drow(int lane, int v) { return (lane >> 4) + 4 * v; }
};
template <> struct MmOp<float> {
    typedef float vec __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ vec mma(float a, float b, vec c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    // accumulator register v of lane l: result row 4 (l >> 4) + v, column l & 15
    static __device__ __forceinline__ int drow(int lane, int v) { return 4 * (lane >> 4) + v; }
};

constexpr int MM_LDP = 66;        // row stride of P / Y in LDS: 16 rows x 2 columns of a half-wave land on 32 distinct 8-byte slots
constexpr int MM_ROWS = 16;       // rows per batch = the N dimension of the MFMA

constexpr int MM_ITILES = 4;      // 16-row blocks of the result (64 padded rows; the blocks past k multiply zeros)
template <int NWV> __host__ __device__ constexpr int mm_nq() { return NWV / MM_ITILES; }              // column-step classes
template <int S, int NWV> __host__ __device__ constexpr int mm_steps() { return (2 * S + mm_nq<NWV>() - 1) / mm_nq<NWV>(); }
template <typename T, int NWV> __host__ __device__ constexpr size_t mm_smem_bytes()
{
    return (size_t)(1 + mm_nq<NWV>()) * MM_ROWS * MM_LDP * sizeof(T) + (MM_ROWS + 4) * sizeof(int);
}

// Shared pieces of the kernels ----------------------------------------------------------------------------------------------
// the Gramian block of wavefront (it, kq) as A operands: step q covers the columns 4 (kq + NQ q) .. + 3
template <typename T, int QS, int NQ>
__device__ __forceinline__ void mm_load_gram(T (&ga)[QS], const T *__restrict__ BtB, int k, int it, int kq, int lane)
{
#pragma unroll
    for (int q = 0; q < QS; q++) {
        const int r = 16 * it + (lane & 15), c = 4 * (kq + NQ * q) + (lane >> 4);
        ga[q] = (r < k && c < k) ? BtB[(size_t)r * k + c] : T(0);
    }
}
// partial product of this wavefront's steps: Gram block x P (returned in the accumulator layout; mm_store puts it into Y)
template <typename T, int QS, int NQ>
__device__ __forceinline__ typename MmOp<T>::vec mm_issue(const T (&ga)[QS], const T *__restrict__ Pb, int kq, int lane)
{
    typedef MmOp<T> Op;
    T bop[QS];
#pragma unroll
    for (int q = 0; q < QS; q++) bop[q] = Pb[(lane & 15) * MM_LDP + 4 * (kq + NQ * q) + (lane >> 4)];
    typename Op::vec acc = {T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int q = 0; q < QS; q++) acc = Op::mma(ga[q], bop[q], acc);
    return acc;
}
template <typename T>
__device__ __forceinline__ void mm_store(const typename MmOp<T>::vec &acc, T *__restrict__ Yb, int it, int kq, int lane)
{
    typedef MmOp<T> Op;
    T *y = Yb + (size_t)kq * MM_ROWS * MM_LDP + (lane & 15) * MM_LDP + 16 * it;
#pragma unroll
    for (int v = 0; v < 4; v++) y[Op::drow(lane, v)] = acc[v];
}
// scheduling pipeline of a pass: QS x (1 MFMA, MM_VALU_PER_MFMA vector instructions), then the LDS writes
constexpr int MM_VALU_PER_MFMA = 24;
template <int QS>
__device__ __forceinline__ void mm_interleave()
{
#pragma unroll
    for (int q = 0; q < QS; q++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, MM_VALU_PER_MFMA, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x002, 200, 0);
    __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);
}
template <typename T, int NQ>
__device__ __forceinline__ T mm_result(const T *__restrict__ Yb, int row, int elem)
{
    T y = Yb[row * MM_LDP + elem];
#pragma unroll
    for (int q = 1; q < NQ; q++) y += Yb[(size_t)q * MM_ROWS * MM_LDP + row * MM_LDP + elem];
    return y;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Rows of 17 .. 32 entries (or the whole tiny bin): one row per wavefront, NW = 16 wavefronts per workgroup.
template <typename T, int S, int NW = MM_ROWS>
__global__ void __launch_bounds__(64 * NW, 4)
cg_rows_tiny_mm_kernel(const CgParams<T> P)
{
    constexpr int NQ = mm_nq<NW>(), QS = mm_steps<S, NW>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *Pb = reinterpret_cast<T *>(smem_raw);                 // [16][LDP]
    T *Yb = Pb + MM_ROWS * MM_LDP;                           // [NQ][16][LDP]
    int *flags = reinterpret_cast<int *>(Yb + NQ * MM_ROWS * MM_LDP);   // [16] row still active; [16..17] claimed batch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ll = lane & 7;
    const int k = P.k;
    const int it = wave % MM_ITILES, kq = wave / MM_ITILES;
    T ga[QS];
    mm_load_gram<T, QS, NQ>(ga, P.BtB, k, it, kq, lane);

    const int nbatches = (P.nrows + NW - 1) / NW;
    const int cslot = blockIdx.x % CG_NCOUNTERS;
    int *const my_counter = P.counter + cslot * CG_COUNTER_STRIDE;
    const int cbase = gridDim.x + cslot;
    auto issue_claim = [&]() -> int {
        int v = 0;
        if (tid == 0) v = atomicAdd(my_counter, 1);
        return v;
    };
    if (tid == 0) {
        flags[16] = cbase + CG_NCOUNTERS * atomicAdd(my_counter, 1);
        flags[17] = cbase + CG_NCOUNTERS * atomicAdd(my_counter, 1);
    }
    __syncthreads();
    int bcur = blockIdx.x, bnxt = flags[16], bnn = flags[17];
    __syncthreads();

    struct Pre { int idx; T x; T a; };
    auto load_desc = [&](int batch) -> RowDesc {
        RowDesc d; d.row = 0; d.nnz = 0; d.st = 0;
        const int pos = batch * NW + wave;
        if (batch < nbatches && pos < P.nrows) d = P.desc[pos];
        d.row = __builtin_amdgcn_readfirstlane(d.row);
        d.nnz = __builtin_amdgcn_readfirstlane(d.nnz);
        d.st = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(d.st >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)(d.st & 0xffffffffu));
        return d;
    };
    auto load_pre = [&](const RowDesc &d) -> Pre {
        Pre q; q.idx = 0; q.x = T(0); q.a = T(0);
        if ((lane >> 1) < d.nnz) {
            const size_t pos = d.st + (size_t)(lane >> 1);
            q.idx = P.indices[pos];
            q.x = P.values[pos];
        }
        if (d.nnz > 0 && lane < k) q.a = P.A[(size_t)d.row * P.lda + lane];
        return q;
    };

    RowDesc d0 = load_desc(bcur), d1 = load_desc(bnxt);
    Pre p0 = load_pre(d0);
    int pend = issue_claim();
    RegTile4<T, S> tile;
    while (bcur < nbatches) {
        load_tile4<T, S>(tile, P.B, P.ldb, k, p0.idx, d0.nnz, lane);
        const RowDesc d2 = load_desc(bnn);
        const Pre p1 = load_pre(d1);

        const int nnz = d0.nnz;
        const T lam = P.lam;
        const bool valid = (lane >> 1) < nnz;
        T a_d = p0.a;
        bool any_active = true;
        // one pass: out = sum_j w_j B_j  -/+  BtB v   (lane f <- element f)
        auto run_pass = [&](T vdist, auto mode_tag, bool active) -> T {
            constexpr int MODE = decltype(mode_tag)::value;
            Pb[wave * MM_LDP + lane] = vdist;
            if (MODE == 1 && lane == 0) flags[wave] = active ? 1 : 0;
            __syncthreads();                                                        // barrier A
            if (MODE == 1) {
                const int f = (lane < NW) ? flags[lane] : 0;
                any_active = __builtin_amdgcn_ballot_w64(f != 0) != 0ull;
                if (!any_active) return T(0);
            }
            T vrep[S];
#pragma unroll
            for (int s = 0; s < S; s++) vrep[s] = Pb[wave * MM_LDP + ll + 8 * s];
#ifdef CMF_MM_NO_MFMA
            typename MmOp<T>::vec yacc = {T(0), T(0), T(0), T(0)};     // timing experiment: lock step and LDS traffic without the MFMAs
#else
            const typename MmOp<T>::vec yacc = mm_issue<T, QS, NQ>(ga, Pb, kq, lane);
#endif
            PassAcc<T> acc;
            acc.zero();
            tile_pass4<T, S, true, MODE>(tile, vrep, p0.x, valid, acc, lane, T(1));
            T out[8];
            acc.close(out);
            T tot = treduce8_high<T>(out, lane);
            mm_store<T>(yacc, Yb, it, kq, lane);
            // program order of this block: one MFMA, a share of the pass's vector instructions, the next MFMA ... and the stores of
            // the result last.  (In-order issue: a wavefront waiting at a dependent MFMA issues nothing else, and after barrier A
            // all wavefronts of a SIMD would queue at the matrix pipe together.)
            mm_interleave<QS>();
            if (MODE == 0 && tid == 0) flags[16] = cbase + CG_NCOUNTERS * pend;    // the batch after bnn, for everybody
            __syncthreads();                                                        // barrier B
            const T y = mm_result<T, NQ>(Yb, wave, lane);
            return (MODE == 0) ? tot - y : tot + y;
        };
        // ---- residual (common.c:1932-1943) ----
        T r_d = run_pass(a_d, std::integral_constant<int, 0>{}, true);
        const int b3 = flags[16];
        r_d -= lam * a_d;
        if (lane >= k) r_d = T(0);
        T p_d = r_d;
        T r_old = wave_sum(r_d * r_d);
        bool done = (r_old <= (T)1e-12) || nnz <= 0;                                // common.c:1952
        for (int step = 0; step < P.max_cg_steps; step++) {
            T Ap_d = run_pass(p_d, std::integral_constant<int, 1>{}, !done);
            if (!any_active) break;
            Ap_d += lam * p_d;
            if (lane >= k) Ap_d = T(0);
            if (!done) {
                const T alpha = r_old / wave_sum(Ap_d * p_d);
                a_d += alpha * p_d;
                r_d -= alpha * Ap_d;
                const T r_new = wave_sum(r_d * r_d);
                if (r_new <= (T)1e-8) done = true;                                  // common.c:1979
                else {
                    p_d = p_d * (r_new / r_old) + r_d;
                    r_old = r_new;
                }
            }
        }
        if (nnz > 0 && lane < k) P.A[(size_t)d0.row * P.lda + lane] = a_d;
        d0 = d1; p0 = p1; d1 = d2;
        bcur = bnxt; bnxt = bnn; bnn = b3;
        pend = issue_claim();
    }
}

}  // namespace cmfhip
