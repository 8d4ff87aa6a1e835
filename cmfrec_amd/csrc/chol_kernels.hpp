// chol_kernels.hpp -- closed-form (Cholesky) ALS row updates for gfx950.
//
// Device-side replacement of
//   explicit  : factors_closed_form, sparse branch   /root/reference/src/common.c:978-1013,1060-1070
//               (row loop common.c:3259-3299, lambda scaling :679-723)
//   implicit  : factors_implicit_chol                 /root/reference/src/common.c:2063-2126
//               (row loop common.c:3397-3417)
//   collective: collective_closed_form_block          /root/reference/src/collective.c:1534-1846
//               (row loop collective.c:5865-5965), sparse X + dense full U, add_X=true add_U=false
//
// One workgroup (256 threads) per row.  The k_t x k_t normal matrix lives in LDS for its whole
// life: initialised (zeros | BtB+lam*I | w*CtC in the upper-left block), accumulated from the
// gathered rows of the opposing factor matrix (staged through LDS in chunks, 4x4 register blocks
// per thread over the upper triangle), factorised in place (right-looking Cholesky) and used for
// the two triangular solves.  Only the upper triangle is referenced, as in the reference
// (tposv_ 'L' on the column-major view == upper of the row-major one).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace cmfhip {

enum CholMode { CHOL_EXPLICIT = 0, CHOL_IMPLICIT = 1, CHOL_COLLECTIVE = 2,
                CHOL_PREFILLED = 3 /* M = Minit[kt,kt] (diag included), rhs = the row itself, no gather:
                                      the multi-RHS posv of the C / D update, common.c:2872-2875 */ };

template <typename T>
struct CholParams {
    T *A; size_t lda;          // row r -> A + r*lda : the k_t unknowns (in/out)
    const T *B; size_t ldb;    // opposing factors, first used column; kb = kt - koff columns are used
    int kt;                    // unknowns per row
    int koff;                  // offset of the X-block inside the unknowns (k_user), 0 otherwise
    const size_t *indptr; const int *indices; const T *values;
    const T *bias_sub;         // x_j := x_j - bias_sub[idx_j], or null
    const int *order; int nrows;
    const T *Minit;            // implicit: BtB + lam*I [kt,kt];  collective: w*CtC [kc,kc];  explicit: null
    int kc;                    // collective: size of the side-info block (k_user + k), else 0
    int rows_with_u;           // collective: rows < rows_with_u carry side information
    int p_side;                // collective: number of side-info columns (scale_lam_sideinfo)
    T lam, lam_last;
    int scale_lam, scale_lam_sideinfo, scale_bias_const;
    int mode;
};

constexpr int CHOL_CHUNK = 16;     // gathered rows staged per round (4 MFMA k-steps)
__host__ __device__ inline int chol_ldm(int kt) { return kt | 1; }     // odd leading dimension
__host__ __device__ inline int chol_tiles(int kt) { return (kt + 15) / 16; }
// staged chunk: [CHUNK][lds] in the coordinates of the unknowns, lds == 16 (mod 32) so that the
// 4 rows x 16 columns an MFMA operand read touches land on distinct banks
__host__ __device__ inline int chol_lds_chunk(int kt) { int T = chol_tiles(kt); return 16 * T + ((T % 2 == 0) ? 16 : 0); }
__host__ __device__ inline size_t chol_lds_elems(int kt, int NTT)
{
    const int ldc = 16 * NTT + ((NTT % 2 == 0) ? 16 : 0);
    return (size_t)kt * chol_ldm(kt) + (size_t)CHOL_CHUNK * ldc + 2 * (size_t)kt + 2 * CHOL_CHUNK + 8;
}

template <typename T> struct CholMfma;
template <> struct CholMfma<double> {
    typedef double vec __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ vec mma(double a, double b, vec c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row_of(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct CholMfma<float> {
    typedef float vec __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ vec mma(float a, float b, vec c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row_of(int lane, int r) { return (lane >> 4) * 4 + r; }
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// tile t of the packed upper triangle of an NTT x NTT tile grid -> (bi, bj), bj >= bi
__host__ __device__ constexpr int tile_bi(int t, int NTT)
{
    int bi = 0, rem = t;
    while (rem >= NTT - bi) { rem -= NTT - bi; bi++; }
    return bi;
}
__host__ __device__ constexpr int tile_bj(int t, int NTT)
{
    int bi = 0, rem = t;
    while (rem >= NTT - bi) { rem -= NTT - bi; bi++; }
    return bi + rem;
}
// index of entry (a, b), b >= a, in a packed upper triangle of T x T register blocks
__host__ __device__ constexpr int tri_index(int a, int b, int T) { return a * T - a * (a - 1) / 2 + (b - a); }

// wave-uniform broadcast of one lane's value (v_readlane_b32, no LDS round trip)
__device__ __forceinline__ float bcast_lane(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ double bcast_lane(double v, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// TPW = upper 16x16 tiles of the normal matrix owned by one wave (tiles t = wave, wave+4, ...).
template <typename T, int TPW, int NTT>
__global__ void __launch_bounds__(256)
chol_rows_kernel(const CholParams<T> P)
{
    using Mf = CholMfma<T>;
    using vec = typename Mf::vec;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int kt = P.kt, koff = P.koff, kb = kt - koff;
    const int ldm = chol_ldm(kt);
    constexpr int ldc = 16 * NTT + ((NTT % 2 == 0) ? 16 : 0);   // == 16 (mod 32): conflict-free MFMA operand reads
    constexpr int NTALL = NTT * (NTT + 1) / 2;      // all upper tiles of the NTT x NTT grid (columns >= kt are zero padded)
    static_assert(TPW * 4 >= NTALL, "tiles per wave");
    T *M = reinterpret_cast<T *>(smem_raw);                // [kt][ldm]
    T *Bs = M + (size_t)kt * ldm;                          // [CHUNK][ldc], unknown coordinates, zero padded
    T *rhs = Bs + (size_t)CHOL_CHUNK * ldc;                // [kt]
    T *rdiag = rhs + kt;                                   // [kt] reciprocals of the Cholesky diagonal
    T *wsc = rdiag + kt;                                   // [CHUNK] rank-1 weights
    T *wrh = wsc + CHOL_CHUNK;                             // [CHUNK] rhs weights
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    // this wave's tiles (wave-uniform): t = wave, wave+4, ...
    int t_bi[TPW], t_bj[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; tt++) {
        const int t = wave + 4 * tt;
        t_bi[tt] = (t < NTALL) ? tile_bi(min(t, NTALL - 1), NTT) : -1;
        t_bj[tt] = (t < NTALL) ? tile_bj(min(t, NTALL - 1), NTT) : -1;
    }

    // staging map of this thread: element e = tid + 256u of the [CHUNK][ldc] chunk -> (row, column)
    constexpr int NPRE = NTT + 1;            // 16 * ldc / 256 staged elements per thread (ldc <= 16*NTT + 16)
    int st_r[NPRE], st_c[NPRE];
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        const int e = tid + 256 * u;
        st_r[u] = e / ldc;
        st_c[u] = e - st_r[u] * ldc;
        if (st_r[u] >= CHOL_CHUNK) { st_r[u] = CHOL_CHUNK; st_c[u] = 0; }    // out of the chunk
    }

    for (int rix = blockIdx.x; rix < P.nrows; rix += gridDim.x) {
        const int row = (P.order != nullptr) ? P.order[rix] : rix;
        const size_t st = (P.mode == CHOL_PREFILLED) ? 0 : P.indptr[row];
        const int nnz = (P.mode == CHOL_PREFILLED) ? 0 : (int)(P.indptr[row + 1] - st);
        T *arow = P.A + (size_t)row * P.lda;
        const bool has_u = (P.mode == CHOL_PREFILLED) || ((P.mode == CHOL_COLLECTIVE) && row < P.rows_with_u);
        if (P.mode == CHOL_COLLECTIVE && nnz == 0 && !has_u) {          // collective.c:1258-1268
            for (int e = tid; e < kt; e += 256) arow[e] = T(0);
            continue;
        }
        T lam = P.lam, lam_last = P.lam_last;
        if (P.mode == CHOL_EXPLICIT) {
            if (P.scale_lam) {                                           // common.c:679-723
                lam *= (T)nnz;
                if (!P.scale_bias_const) lam_last *= (T)nnz;
            }
        } else if (P.mode == CHOL_COLLECTIVE) {
            if (P.scale_lam || P.scale_lam_sideinfo) {                   // collective.c:1285-1355
                T mult = (nnz > 0) ? (T)nnz : T(1);
                if (P.scale_lam_sideinfo && has_u) mult += (T)P.p_side;
                lam *= mult;
                lam_last *= mult;
            }
        }
        __syncthreads();          // previous row's LDS readers are done
        // ---- initialise M (incl. the diagonal shift) and rhs ----
        for (int e = tid; e < kt * ldm; e += 256) {
            int i = e / ldm, j = e % ldm;
            T v = T(0);
            if (j < kt) {
                if (P.mode == CHOL_IMPLICIT || P.mode == CHOL_PREFILLED) v = P.Minit[(size_t)i * kt + j];
                else {
                    if (has_u && i < P.kc && j < P.kc) v = P.Minit[(size_t)i * P.kc + j];   // collective.c:1566-1571
                    if (i == j) v += (i == kt - 1) ? lam_last : lam;   // add_to_diag2: common.c:1060-1062, collective.c:1819
                }
            }
            M[e] = v;
        }
        for (int e = tid; e < kt; e += 256)
            rhs[e] = has_u ? arow[e] : T(0);   // w*U*C prefilled (collective.c:5768-5773)
        // ---- rank-k update on the matrix cores: M[koff:, koff:] += sum_j w_j B_j B_j^T ----
        vec acc[TPW];
#pragma unroll
        for (int tt = 0; tt < TPW; tt++) acc[tt] = vec{0, 0, 0, 0};
        // The gathered rows of chunk c+1 are fetched into registers while the MFMAs of chunk c run,
        // so the gather latency (index load -> row load) is off the critical path.
        const int nstage = min(NPRE, (CHOL_CHUNK * ldc + 255) / 256);
        T pre[NPRE];
        T pre_wsyr = T(0), pre_wrhs = T(0);
        auto fetch = [&](int c0) {
            const int nr = min(CHOL_CHUNK, nnz - c0);
            // all index loads first, then all row loads: two memory latencies per chunk instead of
            // one dependent (index -> row) pair after the other
            int idxs[NPRE];
#pragma unroll
            for (int u = 0; u < NPRE; u++) {
                const bool ok = u < nstage && st_r[u] < nr && st_c[u] >= koff && st_c[u] < kt;
                idxs[u] = ok ? P.indices[st + c0 + st_r[u]] : -1;
            }
#pragma unroll
            for (int u = 0; u < NPRE; u++)
                pre[u] = (idxs[u] >= 0) ? P.B[(size_t)idxs[u] * P.ldb + (st_c[u] - koff)] : T(0);
            pre_wsyr = T(0); pre_wrhs = T(0);
            if (tid < nr) {
                const int idx = P.indices[st + c0 + tid];
                T x = P.values[st + c0 + tid];
                if (P.bias_sub != nullptr) x -= P.bias_sub[idx];
                pre_wsyr = (P.mode == CHOL_IMPLICIT) ? x : T(1);           // common.c:2091-2095 vs :1007-1012
                pre_wrhs = (P.mode == CHOL_IMPLICIT) ? x + T(1) : x;       // common.c:2082-2085 vs :991-996
            }
        };
        if (nnz > 0) fetch(0);
        for (int c0 = 0; c0 < nnz; c0 += CHOL_CHUNK) {
            const int nr = min(CHOL_CHUNK, nnz - c0);
            __syncthreads();                                  // previous chunk consumed (and M/rhs init visible)
#pragma unroll
            for (int u = 0; u < NPRE; u++)
                if (u < nstage && st_r[u] < CHOL_CHUNK) Bs[tid + 256 * u] = pre[u];
            if (tid < CHOL_CHUNK) { wsc[tid] = pre_wsyr; wrh[tid] = pre_wrhs; }
            __syncthreads();
            if (c0 + CHOL_CHUNK < nnz) fetch(c0 + CHOL_CHUNK);          // next chunk: loads in flight during the MFMAs
            for (int e = koff + tid; e < kt; e += 256) {                   // rhs[e] += sum_r wrhs_r B_r[e]
                T s = rhs[e];
                for (int r = 0; r < nr; r++) s += wrh[r] * Bs[r * ldc + e];
                rhs[e] = s;
            }
#pragma unroll
            for (int q = 0; q < CHOL_CHUNK / 4; q++) {
                const int rr = 4 * q + (lane >> 4);
                const T w = wsc[rr];
                const T *brow = Bs + rr * ldc + (lane & 15);
                T opa[TPW], opb[TPW];                          // all operand reads first, then the MFMAs back to back
#pragma unroll
                for (int tt = 0; tt < TPW; tt++) {
                    opa[tt] = (t_bi[tt] >= 0) ? brow[16 * t_bi[tt]] : T(0);
                    opb[tt] = (t_bi[tt] >= 0) ? brow[16 * t_bj[tt]] : T(0);
                }
#pragma unroll
                for (int tt = 0; tt < TPW; tt++)
                    if (t_bi[tt] >= 0) acc[tt] = Mf::mma(w * opa[tt], opb[tt], acc[tt]);
            }
        }
        __syncthreads();
        // accumulators -> M (each upper-triangle entry has exactly one owner)
#pragma unroll
        for (int tt = 0; tt < TPW; tt++) {
            if (t_bi[tt] >= 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int gi = 16 * t_bi[tt] + Mf::row_of(lane, r), gj = 16 * t_bj[tt] + (lane & 15);
                    if (gi < kt && gj < kt && gj >= gi) M[(size_t)gi * ldm + gj] += acc[tt][r];
                }
            }
        }
        __syncthreads();
        // ---- Cholesky M = R^T R in registers.  Thread (ty, tx) of the 16 x 16 grid owns the entries
        //      (i = ty + 16a, j = tx + 16b), b >= a  (cyclic distribution: the work stays balanced as
        //      the trailing matrix shrinks).  Column c: the owners of row c publish their (unscaled) row
        //      and y_c through a double-buffered LDS line, one barrier, then every thread updates its
        //      own registers with  M_ij -= (M_ci / piv) M_cj ; the forward substitution of the right-hand
        //      side rides along as an extra column.  Rows are scaled by 1/sqrt(piv) at the end.
        {
            constexpr int NTRI = NTT * (NTT + 1) / 2;
            const int ty = tid >> 4, tx = tid & 15;
            T mreg[NTRI];
            T yv[NTT];
            static_for<0, NTT>([&](auto ac) {
                constexpr int a = decltype(ac)::value;
                const int i = ty + 16 * a;
                static_for<a, NTT>([&](auto bc) {
                    constexpr int b = decltype(bc)::value;
                    const int j = tx + 16 * b;
                    T v = (i == j) ? T(1) : T(0);                       // rows / columns >= kt: identity padding
                    if (i < kt && j < kt) v = (j >= i) ? M[(size_t)i * ldm + j] : T(0);
                    mreg[tri_index(a, b, NTT)] = v;
                });
                yv[a] = (i < kt) ? rhs[i] : T(0);
            });
            T *line = Bs;                                               // 2 x (16*NTT + 16) elements
            constexpr int LN = 16 * NTT + 16;
            __syncthreads();
            static_for<0, NTT>([&](auto a0c) {
                constexpr int A0 = decltype(a0c)::value;
                for (int cc = 0; cc < 16; cc++) {
                    const int c = 16 * A0 + cc;
                    if (c >= kt) break;
                    T *buf = line + (c & 1) * LN;
                    if (ty == cc) {
                        static_for<A0, NTT>([&](auto bc) {
                            constexpr int b = decltype(bc)::value;
                            buf[tx + 16 * b] = mreg[tri_index(A0, b, NTT)];
                        });
                        if (tx == 0) buf[16 * NTT] = yv[A0];
                    }
                    __syncthreads();
                    const T piv = buf[c];
                    const T inv = T(1) / piv;
                    const T ycv = buf[16 * NTT];
                    if (tid == 0) rdiag[c] = piv;
                    T colv[NTT];
                    static_for<A0, NTT>([&](auto bc) {
                        constexpr int b = decltype(bc)::value;
                        colv[b] = buf[tx + 16 * b];
                    });
                    static_for<A0, NTT>([&](auto ac) {
                        constexpr int a = decltype(ac)::value;
                        const int i = ty + 16 * a;
                        const T fi = (i > c && i < kt) ? buf[i] * inv : T(0);
                        static_for<a, NTT>([&](auto bc) {
                            constexpr int b = decltype(bc)::value;
                            mreg[tri_index(a, b, NTT)] -= fi * colv[b];
                        });
                        yv[a] -= fi * ycv;
                    });
                }
            });
            __syncthreads();                                            // all pivots are in rdiag[]
            static_for<0, NTT>([&](auto ac) {
                constexpr int a = decltype(ac)::value;
                const int i = ty + 16 * a;
                if (i < kt) {
                    const T isq = T(1) / sqrt(rdiag[i]);
                    static_for<a, NTT>([&](auto bc) {
                        constexpr int b = decltype(bc)::value;
                        const int j = tx + 16 * b;
                        if (j < kt && j >= i) M[(size_t)i * ldm + j] = mreg[tri_index(a, b, NTT)] * isq;
                    });
                    if (tx == 0) rhs[i] = yv[a] * isq;                  // y = R^-T rhs
                }
            });
            __syncthreads();
            for (int e = tid; e < kt; e += 256) rdiag[e] = T(1) / M[(size_t)e * ldm + e];
            __syncthreads();
        }
        // ---- backward substitution R x = y by one wavefront, vector in registers (lane l owns
        //      elements l, l+64, ...): no workgroup barriers on this serial chain
        if (wave == 0) {
            constexpr int NE = 5;                         // kt <= 320
            T x[NE];
#pragma unroll
            for (int e = 0; e < NE; e++) x[e] = (lane + 64 * e < kt) ? rhs[lane + 64 * e] : T(0);
            for (int c = kt - 1; c >= 0; c--) {           // R x = y
                T xc = T(0);
#pragma unroll
                for (int e = 0; e < NE; e++) if ((c >> 6) == e) xc = bcast_lane(x[e], c & 63);
                xc *= rdiag[c];
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    const int i = lane + 64 * e;
                    if (i == c) x[e] = xc;
                    else if (i < c) x[e] -= M[(size_t)i * ldm + c] * xc;
                }
            }
#pragma unroll
            for (int e = 0; e < NE; e++) if (lane + 64 * e < kt) arow[lane + 64 * e] = x[e];
        }
    }
}

}  // namespace cmfhip
