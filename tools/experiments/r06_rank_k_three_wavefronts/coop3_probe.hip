#include "../../../cmfrec_amd/csrc/chol_parts_coop_kernels.hpp"
namespace cmfhip {
__host__ __device__ constexpr int c3_nr(int Q) { return Q == 2 ? 4 : 2; }
__host__ __device__ constexpr int c3_row(int Q, int i) { return Q == 0 ? (i == 0 ? 0 : 4) : Q == 1 ? (i == 0 ? 1 : 3) : (i == 0 ? 2 : i == 1 ? 5 : i == 2 ? 6 : 7); }
__host__ __device__ constexpr int c3_off(int Q, int i) { int o = 0; for (int j = 0; j < i; j++) o += 8 - c3_row(Q, j); return o; }

template <typename T, bool BORDER, int Q, int G>
__device__ __forceinline__ void rank_k_coop3(const CholParams<T> &P, size_t st, int nnz, T *__restrict__ pp, int lane, int tid, T *__restrict__ ring)
{
    using Mf = CholMfma<T>;
    using vec = typename Mf::vec;
    constexpr int NB = 8;
    constexpr int NR = c3_nr(Q);
    constexpr int R0 = c3_row(Q, 0);
    constexpr int NTP = c3_off(Q, NR);
    constexpr int NT = NB * (NB + 1) / 2;
    static_assert(NTP == 12, "twelve tiles per wavefront");
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
    // loader: 48 threads per entry, thread w of them the values w, w + 48, w + 96 of the entry's 128 columns + 4 meta values
    const int le = tid / 48, lw = tid - 48 * le;
    const int c2 = lw + 96;                         // < 128: a column; 128 .. 131: meta role c2 - 128; beyond: nothing
    const bool d2 = c2 < 128, m2 = c2 >= 128 && c2 < 132;
    const int role = c2 - 128;
    T sv[G][3];
    int iq[G];
    auto load_idx = [&](int s, int step) { iq[s] = P.indices[st + max(min(4 * step + le, nnz - 1), 0)]; };
    auto issue_rows = [&](int s, int step) {
        const size_t pos = st + max(min(4 * step + le, nnz - 1), 0);
        const T *rowp = P.B + (size_t)iq[s] * P.ldb;
        sv[s][0] = rowp[lw];
        sv[s][1] = rowp[lw + 48];
        const T *xp = d2 ? rowp + c2 : (role == 1 && P.bias_sub != nullptr) ? P.bias_sub + iq[s] : (BORDER && role == 2) ? rowp + bcol : P.values + pos;
        sv[s][2] = *xp;
    };
    auto write_slot = [&](int s, int slot, int step) {
        T *dst = ring + (size_t)slot * PC_SLOT;
        dst[le * PC_ROW + lw] = sv[s][0];
        dst[le * PC_ROW + lw + 48] = sv[s][1];
        T mv = sv[s][2];
        if (!d2) {
            if (role == 1 && P.bias_sub == nullptr) mv = T(0);
            if (role == 3) mv = (4 * step + le < nnz) ? T(1) : T(0);
        }
        if (d2) dst[le * PC_ROW + c2] = mv;
        else if (m2) dst[4 * PC_ROW + 4 * le + role] = mv;
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
            const int i = it * G + s;
            constexpr int sn = (s + 1) % G;
            write_slot(sn, (i + 1) & 1, i + 1);
            issue_rows(sn, i + 1 + G);
            load_idx(sn, i + 1 + 2 * G);
            if (i < nsteps) {
                const T *src = ring + (size_t)(i & 1) * PC_SLOT;
                const T *mrow = src + 4 * PC_ROW + 4 * g;
                const T x = mrow[0] - mrow[1];
                const T bv = BORDER ? mrow[2] : T(0);
                const bool vld = mrow[3] != T(0);
                T ws = impl_w ? x : T(1);
                T xw = impl_w ? x + T(1) : x;
                if (!vld) { ws = T(0); xw = T(0); }
                const T *orow = src + g * PC_ROW + lm;
                static_for<0, NR>([&](auto ic) {
                    constexpr int ii = decltype(ic)::value;
                    constexpr int R = c3_row(Q, ii), OFF = c3_off(Q, ii);
                    const T oR = orow[16 * R];
                    const T a = oR * ws;
                    static_for<R, NB>([&](auto bjc) { constexpr int bj = decltype(bjc)::value; acc[OFF + bj - R] = Mf::mma(a, bj == R ? oR : orow[16 * bj], acc[OFF + bj - R]); });
                    rp[ii] += xw * oR;
                    if (BORDER) gp[ii] += (ws * bv) * oR;
                });
                if (BORDER && Q == 0) { gam += (ws * bv) * bv; rbs += xw * bv; }
            }
            __syncthreads();
        });
    }
    static_for<0, NR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int R = c3_row(Q, i), OFF = c3_off(Q, i);
#pragma unroll
        for (int j = 0; j < NB - R; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) pp[(size_t)wtix(R, R + j, NB) * 256 + r * 64 + lane] = acc[OFF + j][r];
    });
    T *pv = pp + (size_t)NT * 256;
    auto over_groups = [&](T v) -> T { v = lanes::tswap16_add(v, v); return lanes::tswap32_add(v, v); };
    static_for<0, NR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int R = c3_row(Q, i);
        const T v = over_groups(rp[i]);
        if (lane < 16) pv[16 * R + lane] = v;
        if (BORDER) { const T w = over_groups(gp[i]); if (lane < 16) pv[16 * NB + 16 * R + lane] = w; }
    });
    if (BORDER && Q == 0) {
        gam = over_groups(gam); rbs = over_groups(rbs);
        if (lane == 0) { pv[32 * NB] = gam; pv[32 * NB + 1] = rbs; }
    }
}

template <typename T, bool BORDER, int G>
__global__ void __launch_bounds__(192, 3)
coop3_kernel(const CholParams<T> P, const RowDesc *__restrict__ desc, const CholSlices<T> SL)
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
        const size_t st_row = ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(d.st >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(d.st & 0xffffffffu));
        const int nnz = (scount >= 0) ? __builtin_amdgcn_readfirstlane(scount) : nnz_row;
        const size_t st = st_row + (size_t)__builtin_amdgcn_readfirstlane(sfirst);
        T *pp = SL.part + (size_t)(rix - SL.part_base) * PART;
        if (wave == 0) rank_k_coop3<T, BORDER, 0, G>(P, st, nnz, pp, lane, tid, ring);
        else if (wave == 1) rank_k_coop3<T, BORDER, 1, G>(P, st, nnz, pp, lane, tid, ring);
        else rank_k_coop3<T, BORDER, 2, G>(P, st, nnz, pp, lane, tid, ring);
        __syncthreads();
        rix = P.row_first + (int)gridDim.x + s_next;
        __syncthreads();
    }
}
template __global__ void coop3_kernel<double, true, 3>(const CholParams<double>, const RowDesc *, const CholSlices<double>);
template __global__ void coop3_kernel<double, true, 2>(const CholParams<double>, const RowDesc *, const CholSlices<double>);
template __global__ void coop3_kernel<double, false, 3>(const CholParams<double>, const RowDesc *, const CholSlices<double>);
}
