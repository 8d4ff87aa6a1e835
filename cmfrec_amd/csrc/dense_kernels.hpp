// dense_kernels.hpp -- small dense helpers of the ALS path for gfx950.
//
//   gram_*        : G = B[:, :k]^T B[:, :k]            (cblas_tsyrk call sites, reference
//                   src/common.c:2824, :3328; src/collective.c:6287-6296)
//   column ops    : strided column copy / fill on the [rows, k+1] "factor + bias column" layout
//                   (reference src/collective.c:8538-8543, :8723-8732, :8882-8884)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lanes.hpp"

namespace cmfhip {

// Second stage of the Gramian: partial[b][i * k + j] holds block b's sum for the entries of the UPPER triangle (i <= j; a block's
// k x k values are contiguous, so the first stage writes them in runs and this stage reads 128-byte runs).  A workgroup owns
// 16 consecutive entries; thread (ph, e) adds the blocks ph, ph + 16, ... of entry e in order, thread (0, e) the sixteen
// phase sums in order: the summation order depends only on (n, k), never on scheduling.  Both triangles of `out` are written.
constexpr int GRAM_RED_ENT = 16;
template <typename T>
__global__ void __launch_bounds__(256)
gram_reduce_kernel(const T *__restrict__ partial, int nblocks, int nent,
                   T *__restrict__ out, T scale, T add_diag, int k)
{
    __shared__ T ph_sum[16][GRAM_RED_ENT];
    const int e = threadIdx.x & (GRAM_RED_ENT - 1), ph = threadIdx.x / GRAM_RED_ENT;
    const int ent = blockIdx.x * GRAM_RED_ENT + e;
    const int i = ent / k, j = ent % k;
    const bool live = ent < nent && i <= j;
    T s = T(0);
    if (live) {
        const T *src = partial + ent;
        int b = ph;
        for (; b + 48 < nblocks; b += 64) {
            const T v0 = src[(size_t)b * nent], v1 = src[(size_t)(b + 16) * nent], v2 = src[(size_t)(b + 32) * nent],
                    v3 = src[(size_t)(b + 48) * nent];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; b < nblocks; b += 16) s += src[(size_t)b * nent];
    }
    ph_sum[ph][e] = s;
    __syncthreads();
    if (ph == 0 && live) {
        T tot = ph_sum[0][e];
#pragma unroll
        for (int q = 1; q < 16; q++) tot += ph_sum[q][e];
        tot *= scale;
        if (i == j) tot += add_diag;
        out[(size_t)i * k + j] = tot;
        if (i != j) out[(size_t)j * k + i] = tot;
    }
}

// MFMA Gramian (k <= 64): the dense B^T B precompute on the matrix cores.
//   fp64: v_mfma_f64_16x16x4_f64   A[i=l&15][kk=l>>4], B[kk=l>>4][j=l&15], D col=l&15, row=(l>>4)+4*reg
//   fp32: v_mfma_f32_16x16x4_f32   same A/B operand map,                   D col=l&15, row=(l>>4)*4+reg
// One MFMA step consumes 4 rows of B: every lane loads B[row0 + (l>>4)][16*cb + (l&15)] for the
// (up to 4) column blocks -- 4 rows x 128 B per block, coalesced -- and the 10 upper 16x16 tiles
// are updated with tile(bi,bj) += val[bi]^T val[bj].  A workgroup (4 waves) owns a slab of rows,
// its waves' accumulators are added in wave order through LDS and written as [block][entry]
// partials (upper triangle) for the deterministic second stage above.
template <typename T> struct MfmaAcc;
template <> struct MfmaAcc<double> {
    typedef double vec __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ vec mma(double a, double b, vec c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row_of(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct MfmaAcc<float> {
    typedef float vec __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ vec mma(float a, float b, vec c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row_of(int lane, int r) { return (lane >> 4) * 4 + r; }
};

// REM > 0 (double precision, 48 < k <= 52: k = 50 of the headline configuration): the last column block holds REM live columns,
// so its four tiles would run the matrix pipe at REM / 16 of its width -- 4 of the 10 MFMAs of a step, in the precision whose
// MFMA rate is the limit of this kernel.  Those REM columns go through the vector ALU instead: column 48 + q of the step's row
// is lane q of the lane's 16-lane row (one v_mov_b64_dpp row_newbcast), times the lane's own four values, per-lane partial sums
// over the rows the lane sees, added over the four row groups and the four waves in a fixed order at the end.  6 MFMAs + 5 REM
// vector instructions per step instead of 10 MFMAs (C2: 0.105 / 0.050 -> see profiles/r03 for the two matrices).
template <typename T, int REM = 0>
__global__ void __launch_bounds__(256)
gram_mfma_partial_kernel(const T *__restrict__ B, size_t ldb, int n, int k, int rows_per_block,
                         T *__restrict__ partial)
{
    using Acc = MfmaAcc<T>;
    using vec = typename Acc::vec;
    constexpr int NB = REM ? 3 : 4;                       // column blocks on the matrix pipe
    constexpr int NTL = NB * (NB + 1) / 2;                // their upper-triangle tiles
    constexpr int NX = REM ? REM : 1;
    __shared__ T red[4][NTL][4][64];                      // [wave][tile][reg][lane]
    __shared__ T redx[REM ? 4 : 1][NX][4][64];            // [wave][column 48 + q][column block][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(n, r0 + rows_per_block);
    const int kk = lane >> 4, cc = lane & 15;
    vec acc[NTL];
    T ex[NX][4];
#pragma unroll
    for (int t = 0; t < NTL; t++) acc[t] = vec{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < NX; q++)
#pragma unroll
        for (int cb = 0; cb < 4; cb++) ex[q][cb] = T(0);
    // four steps (16 rows of this wave) per trip: their 16 loads are in flight together (one step at a time exposed a full
    // memory latency per 10 MFMAs: 0.065 ms for the 359 k x 50 matrix of C2, twice what streaming it takes)
    constexpr int UNR = 4;
    for (int rb = r0 + 4 * wave; rb < r1; rb += 16 * UNR) {
        T val[UNR][4];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int row = rb + 16 * u + kk;
#pragma unroll
            for (int cb = 0; cb < 4; cb++) {
                const int col = 16 * cb + cc;
                val[u][cb] = (row < r1 && col < k) ? B[(size_t)row * ldb + col] : T(0);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            int t = 0;
#pragma unroll
            for (int bi = 0; bi < NB; bi++)
#pragma unroll
                for (int bj = bi; bj < NB; bj++) { acc[t] = Acc::mma(val[u][bi], val[u][bj], acc[t]); t++; }
            if constexpr (REM > 0) {
                static_for<0, REM>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    const T b = lanes::row_bcast16<q>(val[u][3]);
#pragma unroll
                    for (int cb = 0; cb < 4; cb++) ex[q][cb] += val[u][cb] * b;
                });
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NTL; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[wave][t][r][lane] = acc[t][r];
    if constexpr (REM > 0) {
#pragma unroll
        for (int q = 0; q < REM; q++)
#pragma unroll
            for (int cb = 0; cb < 4; cb++) redx[wave][q][cb][lane] = ex[q][cb];
    }
    __syncthreads();
    T *__restrict__ pblk = partial + (size_t)blockIdx.x * k * k;          // this block's k x k values (upper triangle written)
    // entry (i,j) with i in tile-row bi, j in tile-col bj: thread -> (tile, reg, lane)
    for (int e = threadIdx.x; e < NTL * 4 * 64; e += 256) {
        const int t = e / 256, r = (e / 64) % 4, l = e % 64;
        int bi = 0, rem = t;
        while (rem >= NB - bi) { rem -= NB - bi; bi++; }
        const int bj = bi + rem;
        const int i = 16 * bi + Acc::row_of(l, r), j = 16 * bj + (l & 15);
        if (i <= j && j < k) {                             // (a diagonal tile holds both triangles: the upper one is kept)
            T sum = red[0][t][r][l];
            sum += red[1][t][r][l]; sum += red[2][t][r][l]; sum += red[3][t][r][l];
            pblk[i * k + j] = sum;
        }
    }
    if constexpr (REM > 0) {
        // entry (48 + q, 16 cb + c): row groups in order inside a wave, waves in order
        for (int e = threadIdx.x; e < REM * 4 * 16; e += 256) {
            const int q = e / 64, cb = (e / 16) % 4, c = e % 16;
            const int i = 16 * NB + q, j = 16 * cb + c;
            if (j < k && (cb < NB || j >= i)) {                // entry (j, i) of the upper triangle, or (i, j) inside the last block
                T sum = T(0);
#pragma unroll
                for (int w = 0; w < 4; w++)
#pragma unroll
                    for (int g = 0; g < 4; g++) sum += redx[w][q][cb][16 * g + c];
                if (cb < NB) pblk[j * k + i] = sum;
                else pblk[i * k + j] = sum;
            }
        }
    }
}

// generic-k fallback for the Gramian (k > 64): one thread per entry, rows streamed from L2.
template <typename T>
__global__ void gram_naive_kernel(const T *__restrict__ B, size_t ldb, int n, int k,
                                  T *__restrict__ out, T scale, T add_diag)
{
    int ent = blockIdx.x * blockDim.x + threadIdx.x;
    if (ent >= k * k) return;
    int i = ent / k, j = ent % k;
    int lo = min(i, j), hi = max(i, j);
    T s = T(0);
    for (int r = 0; r < n; r++) s += B[(size_t)r * ldb + lo] * B[(size_t)r * ldb + hi];
    s *= scale;
    if (i == j) s += add_diag;
    out[ent] = s;
}

// out_part[b][c] = sum over the rows r of block b of (bias[r] + add) * M[r, c]  (rows in order, one thread per column):
// first stage of the right-hand-side constant of the missing-as-zero half-steps (collective.c:8573-8600, :8756-8787)
constexpr int COLSUM_ROWS = 256;
template <typename T>
__global__ void __launch_bounds__(256)
weighted_colsum_partial_kernel(const T *__restrict__ M, size_t ld, int rows, int cols, const T *__restrict__ bias, T add,
                               T *__restrict__ out_part)
{
    const int r0 = blockIdx.x * COLSUM_ROWS, r1 = min(rows, r0 + COLSUM_ROWS);
    for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        T acc = T(0);
        for (int r = r0; r < r1; r++) acc += ((bias != nullptr ? bias[r] : T(0)) + add) * M[(size_t)r * ld + c];
        out_part[(size_t)blockIdx.x * cols + c] = acc;
    }
}
// out[c] = scale * sum_b part[b][c], blocks in order
template <typename T>
__global__ void colsum_finish_kernel(const T *__restrict__ part, int nblocks, int cols, T scale, T *__restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    T acc = T(0);
    for (int b = 0; b < nblocks; b++) acc += part[(size_t)b * cols + c];
    out[c] = scale * acc;
}
// M[r, c] += v[c] for the first `cols` columns of every row
template <typename T>
__global__ void add_rowvec_kernel(T *__restrict__ M, size_t ld, size_t rows, int cols, const T *__restrict__ v)
{
    const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (e >= rows * (size_t)cols) return;
    const size_t r = e / cols; const int c = (int)(e % cols);
    M[r * ld + c] += v[c];
}

// M[r, c] = v[c] (or zero) for the first `cols` columns of every row
template <typename T>
__global__ void set_rowvec_kernel(T *__restrict__ M, size_t ld, size_t rows, int cols, const T *__restrict__ v)
{
    const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (e >= rows * (size_t)cols) return;
    const size_t r = e / cols; const int c = (int)(e % cols);
    M[r * ld + c] = (v != nullptr) ? v[c] : T(0);
}

// the refinement step of the shared-matrix solve (session.hip, launch_potrs_rows)
// out = the upper triangle of the row-major k x k factor R, zeros below (the factorisation leaves the lower triangle as it was)
template <typename T>
__global__ void upper_only_kernel(const T *__restrict__ R, int k, T *__restrict__ out)
{
    const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (e >= (size_t)k * k) return;
    const int r = (int)(e / k), c = (int)(e % k);
    out[e] = (c >= r) ? R[e] : T(0);
}
// res[r, c] = b[r, c] - res[r, c]
template <typename T>
__global__ void residual_rows_kernel(T *__restrict__ res, size_t ld_res, const T *__restrict__ b, size_t ld_b, size_t rows, int cols)
{
    const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (e >= rows * (size_t)cols) return;
    const size_t r = e / cols; const int c = (int)(e % cols);
    res[r * ld_res + c] = b[r * ld_b + c] - res[r * ld_res + c];
}
// x[r, c] += a[r, c]
template <typename T>
__global__ void add_rows_kernel(T *__restrict__ x, size_t ld_x, const T *__restrict__ a, size_t ld_a, size_t rows, int cols)
{
    const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (e >= rows * (size_t)cols) return;
    const size_t r = e / cols; const int c = (int)(e % cols);
    x[r * ld_x + c] += a[r * ld_a + c];
}

// dst[order[i], c] = src[order[i], c] for the first n positions of a processing order
template <typename T>
__global__ void copy_rows_by_order_kernel(const T *__restrict__ src, size_t ld_src, T *__restrict__ dst, size_t ld_dst, const int *__restrict__ order,
                                          int n, int cols)
{
    const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (e >= (size_t)n * cols) return;
    const int r = order[e / cols]; const int c = (int)(e % cols);
    dst[(size_t)r * ld_dst + c] = src[(size_t)r * ld_src + c];
}

// Missing-as-zero main matrix WITH observation weights (optimizeA Case 4, NA_as_zero && weight): per entry e of X (CSR or CSC
// order) the rank-1 weight of its correction and the bracket of its right-hand side,
//     g[e] = w[e] - 1,    xt[e] = w[e] x[e] - (w[e] - 1) (mean + bias[idx[e]])
// (factors_closed_form, /root/reference/src/common.c:866-885; factors_explicit_cg_NA_as_zero_weighted :1325-1338)
template <typename T>
__global__ void naz_entry_transform_kernel(const T *__restrict__ x, const T *__restrict__ w, const int *__restrict__ idx, size_t nnz,
                                           const T *__restrict__ bias, T mean, T *__restrict__ g, T *__restrict__ xt)
{
    const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const T we = w[e];
    const T cst = mean + ((bias != nullptr) ? bias[idx[e]] : T(0));
    g[e] = we - T(1);
    xt[e] = we * x[e] - (we - T(1)) * cst;
}
// the lambda multipliers of those rows under scale_lam: the sum of the row's weights (in double, entry order) plus the number of
// its absent entries (collective.c:7991-8022)
template <typename T>
__global__ void naz_wsum_kernel(const size_t *__restrict__ p, const T *__restrict__ w, int nrows, int others, T *__restrict__ wsum)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    double acc = 0;
    for (size_t e = p[r]; e < p[r + 1]; e++) acc += (double)w[e];
    wsum[r] = (T)acc + (T)(others - (int)(p[r + 1] - p[r]));
}
// CG on a matrix every row shares, one wavefront per row, for the rows of that model WITHOUT entries (they are solved when the
// bias / mean constant exists, common.c:3270-3271): (G + diag(lam_i .. lam_last_i)) a = cst from the row's current a --
// factors_explicit_cg_NA_as_zero_weighted with nnz = 0 (:1321-1324 symv, :1368-1371 constant, the steps :1388-1438).  k <= 64.
template <typename T>
__global__ void __launch_bounds__(256)
cg_shared_matrix_rows_kernel(T *__restrict__ A, size_t lda, const int *__restrict__ rows, int nrows, int k, const T *__restrict__ G,
                             const T *__restrict__ cst, T lam, T lam_last, const T *__restrict__ mult, int scale_bias_const, int max_cg_steps)
{
    const int lane = threadIdx.x & 63;
    const int pos = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pos >= nrows) return;
    const int row = rows[pos];
    T li = lam, ll = lam_last;
    if (mult != nullptr) { li *= mult[row]; if (!scale_bias_const) ll *= mult[row]; }
    const T dg = (lane == k - 1) ? ll : li;
    T *arow = A + (size_t)row * lda;
    T a = (lane < k) ? arow[lane] : T(0);
    auto mv = [&](T v) -> T {                        // (G v)[lane], G symmetric
        T acc = T(0);
        for (int j = 0; j < k; j++) acc += ((lane < k) ? G[(size_t)j * k + lane] : T(0)) * __shfl(v, j);
        return acc;
    };
    T r = -mv(a);
    if (cst != nullptr && lane < k) r += cst[lane];
    r -= dg * a;
    if (lane >= k) r = T(0);
    T p = r;
    T r_old = lanes::wave_sum(r * r);
    if (r_old > (T)1e-12) {
        for (int step = 0; step < max_cg_steps; step++) {
            T Ap = mv(p) + dg * p;
            if (lane >= k) Ap = T(0);
            const T alpha = r_old / lanes::wave_sum(p * Ap);
            a += alpha * p;
            r -= alpha * Ap;
            const T r_new = lanes::wave_sum(r * r);
            if (r_new <= (T)1e-8) break;
            p = p * (r_new / r_old) + r;
            r_old = r_new;
        }
    }
    if (lane < k) arow[lane] = a;
}

template <typename T>
__global__ void col_fill_kernel(T *__restrict__ M, size_t ld, int rows, int col, T value)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) M[(size_t)r * ld + col] = value;
}

template <typename T>
__global__ void col_extract_kernel(const T *__restrict__ M, size_t ld, int rows, int col, T *__restrict__ out)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) out[r] = M[(size_t)r * ld + col];
}

template <typename T>
__global__ void col_insert_kernel(T *__restrict__ M, size_t ld, int rows, int col, const T *__restrict__ in)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) M[(size_t)r * ld + col] = in[r];
}

// dst[r, 0:cols] = src[r, 0:cols] with different leading dimensions (copy_mat, helpers.c:1232)
template <typename T>
__global__ void copy_mat_kernel(const T *__restrict__ src, size_t lds, T *__restrict__ dst, size_t ldd,
                                size_t rows, int cols)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * (size_t)cols) return;
    size_t r = e / cols;
    int c = (int)(e % cols);
    dst[r * ldd + c] = src[r * lds + c];
}

template <typename T>
__global__ void fill_kernel(T *__restrict__ p, size_t n, T value)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) p[e] = value;
}

// ---- dense contractions of the side-information path on the matrix cores (round 3) ------------------------------------------
// C[M,N] (+)= alpha * op(A) * B, row-major, op(A) = A[M,K] or (TRANSA) A stored as [K,M]: the w U C / I D products (tall M,
// K = p, N = k_user + k), U^T A / A^T A (short M and N, K = the rows of the factor matrix: split over gridDim.z, partial
// products summed in split order by gemm_splitk_reduce_kernel -- bit-reproducible, no floating-point atomics).
// Workgroup: 128 x 128 output tile, four wavefronts in a 2 x 2 grid, each a 64 x 64 tile = 4 x 4 accumulator tiles of
// v_mfma_{f32,f64}_16x16x4; the K dimension in steps of BK = 16 staged through LDS (k-major, so that both MFMA operands are 16
// consecutive elements of 4 consecutive rows), the next step's global loads in flight during the MFMAs of the current one.
// 64 MFMAs per wave and step against 16 operand reads: the matrix pipe is the limit.
template <typename T> struct GemmMfma;
template <> struct GemmMfma<double> {
    typedef double vec __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ vec mma(double a, double b, vec c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row_of(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct GemmMfma<float> {
    typedef float vec __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ vec mma(float a, float b, vec c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row_of(int lane, int r) { return (lane >> 4) * 4 + r; }
};
constexpr int GEMM_BM = 128, GEMM_BN = 128;
constexpr int GEMM_BK = 32;                   // K-chunks of the split are multiples of this (the kernel's own step: 32 / 16)
template <typename T> struct GemmVec;
template <> struct GemmVec<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct GemmVec<double> { typedef double type __attribute__((ext_vector_type(2))); };

template <typename T, bool TRANSA>
__global__ void __launch_bounds__(256, 2)
gemm_mfma_kernel(int M, int N, int K, int kchunk, T alpha, const T *__restrict__ A, size_t lda, const T *__restrict__ B, size_t ldb,
                 T *__restrict__ C, size_t ldc, size_t split_stride)
{
    using Mf = GemmMfma<T>;
    using vec = typename Mf::vec;
    using lvec = typename GemmVec<T>::type;                    // 16 bytes of operand per load
    constexpr int BM = GEMM_BM, BN = GEMM_BN;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int BK = (sizeof(T) == 4) ? 32 : 16;             // 16 KB of each operand per step
    constexpr int NV = BM * BK / 256 / VEC;                    // 16-byte loads per thread, operand and step (4)
    // LDS images are k-major ([BK][row stride]): both MFMA operands are 16 consecutive elements of 4 consecutive k-rows.
    // Row stride 144 == 16 (mod 32): conflict-free operand reads and aligned vector stores where the global layout is k-major
    // too (B always, A when TRANSA).  A[M, K] arrives m-major: its 16-byte loads run along k and are stored as scalars down a
    // column -- stride 130, so that the 8 k-groups of a wave spread over the banks (2-way on stores and reads instead of 8-way).
    constexpr int LSA = TRANSA ? 144 : 130, LSB = 144;
    __shared__ __attribute__((aligned(16))) T As[BK * LSA];
    __shared__ __attribute__((aligned(16))) T Bs[BK * LSB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave >> 1, wx = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
    // whole tiles at 16-byte aligned addresses take the vector path; edges (and odd leading dimensions) the clamped scalar one
    const bool a_vec_ok = ((size_t)A % 16 == 0) && ((lda * sizeof(T)) % 16 == 0) && (m0 + BM <= M);
    const bool b_vec_ok = ((size_t)B % 16 == 0) && ((ldb * sizeof(T)) % 16 == 0) && (n0 + BN <= N);
    vec acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = vec{0, 0, 0, 0};
    lvec ra[NV], rb[NV];
    // thread -> (row, 16-byte group) of a tile whose rows are RL elements long
    auto load_step = [&](int k0) {
        const bool full_k = k0 + BK <= kend;
        if (a_vec_ok && full_k) {
            if (TRANSA) {                                      // A[k][m]: groups along m
                constexpr int GPR = BM / VEC;                  // groups per k-row
#pragma unroll
                for (int i = 0; i < NV; i++) {
                    const int kk = tid / GPR + (256 / GPR) * i, mq = tid % GPR;
                    ra[i] = *reinterpret_cast<const lvec *>(A + (size_t)(k0 + kk) * lda + m0 + mq * VEC);
                }
            } else {                                           // A[m][k]: groups along k
                constexpr int GPR = BK / VEC;
#pragma unroll
                for (int i = 0; i < NV; i++) {
                    const int mm = tid / GPR + (256 / GPR) * i, kq = tid % GPR;
                    ra[i] = *reinterpret_cast<const lvec *>(A + (size_t)(m0 + mm) * lda + k0 + kq * VEC);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV; i++)
#pragma unroll
                for (int j = 0; j < VEC; j++) {
                    int kk, mm;
                    if (TRANSA) { constexpr int GPR = BM / VEC; kk = tid / GPR + (256 / GPR) * i; mm = (tid % GPR) * VEC + j; }
                    else        { constexpr int GPR = BK / VEC; mm = tid / GPR + (256 / GPR) * i; kk = (tid % GPR) * VEC + j; }
                    const int gm = m0 + mm, gk = k0 + kk;
                    const size_t off = TRANSA ? (size_t)min(gk, kend - 1) * lda + (size_t)min(gm, M - 1)
                                              : (size_t)min(gm, M - 1) * lda + (size_t)min(gk, kend - 1);
                    const T v = A[off];
                    ra[i][j] = (gm < M && gk < kend) ? v : T(0);
                }
        }
        constexpr int GPRB = BN / VEC;
        if (b_vec_ok && full_k) {
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const int kk = tid / GPRB + (256 / GPRB) * i, nq = tid % GPRB;
                rb[i] = *reinterpret_cast<const lvec *>(B + (size_t)(k0 + kk) * ldb + n0 + nq * VEC);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV; i++)
#pragma unroll
                for (int j = 0; j < VEC; j++) {
                    const int kk = tid / GPRB + (256 / GPRB) * i, nn = (tid % GPRB) * VEC + j;
                    const int gk = k0 + kk, gn = n0 + nn;
                    const T v = B[(size_t)min(gk, kend - 1) * ldb + (size_t)min(gn, N - 1)];
                    rb[i][j] = (gk < kend && gn < N) ? v : T(0);
                }
        }
    };
    auto store_step = [&]() {
        if (TRANSA) {
            constexpr int GPR = BM / VEC;
#pragma unroll
            for (int i = 0; i < NV; i++)
                *reinterpret_cast<lvec *>(As + (tid / GPR + (256 / GPR) * i) * LSA + (tid % GPR) * VEC) = ra[i];
        } else {
            constexpr int GPR = BK / VEC;
#pragma unroll
            for (int i = 0; i < NV; i++)
#pragma unroll
                for (int j = 0; j < VEC; j++) As[((tid % GPR) * VEC + j) * LSA + tid / GPR + (256 / GPR) * i] = ra[i][j];
        }
        constexpr int GPRB = BN / VEC;
#pragma unroll
        for (int i = 0; i < NV; i++)
            *reinterpret_cast<lvec *>(Bs + (tid / GPRB + (256 / GPRB) * i) * LSB + (tid % GPRB) * VEC) = rb[i];
    };
    if (kbeg < kend) load_step(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();                                       // the previous step's operand reads are done
        store_step();
        __syncthreads();
        if (k0 + BK < kend) load_step(k0 + BK);                // lands behind the MFMAs below
        const T *ap = As + (lane >> 4) * LSA + 64 * wy + (lane & 15);
        const T *bp = Bs + (lane >> 4) * LSB + 64 * wx + (lane & 15);
#pragma unroll
        for (int q = 0; q < BK / 4; q++) {
            T av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; a++) av[a] = ap[(4 * q) * LSA + 16 * a];
#pragma unroll
            for (int b = 0; b < 4; b++) bv[b] = bp[(4 * q) * LSB + 16 * b];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = Mf::mma(av[a], bv[b], acc[a][b]);
        }
    }
    T *Cz = C + (size_t)blockIdx.z * split_stride;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int gm = m0 + 64 * wy + 16 * a + Mf::row_of(lane, r), gn = n0 + 64 * wx + 16 * b + (lane & 15);
                if (gm < M && gn < N) Cz[(size_t)gm * ldc + gn] = alpha * acc[a][b][r];
            }
}

// C[e] = sum_z partial[z][e], z ascending (a fixed order: the result does not depend on how the blocks were scheduled)
template <typename T>
__global__ void gemm_splitk_reduce_kernel(const T *__restrict__ partial, size_t split_stride, int nsplit, int M, int N, T *__restrict__ C, size_t ldc)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)M * N) return;
    T s = T(0);
    for (int z = 0; z < nsplit; z++) s += partial[(size_t)z * split_stride + e];
    C[(e / N) * ldc + (e % N)] = s;
}

// In-place Cholesky of one small matrix, M = R^T R, upper triangle of the row-major matrix (what LAPACK's
// potrf('L') leaves on the column-major view, src/collective.c:9241, :10107); the strict lower triangle is not
// touched.  One workgroup; the matrix stays in global memory (n <= a few hundred, called once per fit).
template <typename T>
__global__ void __launch_bounds__(256)
potrf_upper_kernel(T *__restrict__ M, int n)
{
    __shared__ T s_piv;
    const int tid = threadIdx.x;
    for (int c = 0; c < n; c++) {
        if (tid == 0) { s_piv = sqrt(M[(size_t)c * n + c]); M[(size_t)c * n + c] = s_piv; }
        __syncthreads();
        const T piv = s_piv;
        for (int j = c + 1 + tid; j < n; j += 256) M[(size_t)c * n + j] /= piv;
        __syncthreads();
        const int rem = n - c - 1;
        for (int e = tid; e < rem * rem; e += 256) {
            const int i = c + 1 + e / rem, j = c + 1 + e % rem;
            if (j >= i) M[(size_t)i * n + j] -= M[(size_t)c * n + i] * M[(size_t)c * n + j];
        }
        __syncthreads();
    }
}

// dst[off + i][off + j] (+)= scale * src[i][j] for an [ns, ns] block placed at (off, off) of an [nd, nd] matrix
template <typename T>
__global__ void add_block_kernel(const T *__restrict__ src, int ns, T scale, T *__restrict__ dst, int nd, int off)
{
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < ns * ns; e += gridDim.x * blockDim.x) {
        const int i = e / ns, j = e % ns;
        dst[(size_t)(off + i) * nd + (off + j)] += scale * src[e];
    }
}

// M[i][i] += value for i in [first, last)
template <typename T>
__global__ void add_diag_kernel(T *__restrict__ M, int n, int first, int last, T value)
{
    const int i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < last) M[(size_t)i * n + i] += value;
}

// out[kt,kt] := blockdiag(lam * I[ks], G[kk,kk])  (kt = ks + kk): the part of Be^T Be every row shares in the
// implicit model with side information, collective.c:6121-6135 (G = BtB + lam I already)
template <typename T>
__global__ void betbe_base_kernel(const T *__restrict__ G, int kk, int ks, T lam, T *__restrict__ out)
{
    const int kt = ks + kk;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < kt * kt; e += gridDim.x * blockDim.x) {
        const int i = e / kt, j = e % kt;
        T v = T(0);
        if (i >= ks && j >= ks) v = G[(size_t)(i - ks) * kk + (j - ks)];
        else if (i == j) v = lam;
        out[e] = v;
    }
}

// M[r, :] := keep[r, :] for the rows whose mask byte is zero (both [rows, ld])
template <typename T>
__global__ void restore_rows_kernel(T *__restrict__ M, const T *__restrict__ keep, size_t ld, size_t rows, const unsigned char *__restrict__ mask)
{
    const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (e >= rows * ld) return;
    if (!mask[e / ld]) M[e] = keep[e];
}

// M[r, :] := keep[r, :] for the rows whose mask byte is NOT `value`
template <typename T>
__global__ void keep_rows_unless_kernel(T *__restrict__ M, const T *__restrict__ keep, size_t ld, size_t rows, const unsigned char *__restrict__ mask,
                                        unsigned char value)
{
    const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (e >= rows * ld) return;
    if (mask[e / ld] != value) M[e] = keep[e];
}

// M[rows[e], 0 .. ncols) := 0
template <typename T>
__global__ void zero_rows_kernel(T *__restrict__ M, size_t ld, int ncols, const int *__restrict__ rows, int count)
{
    const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (e >= (size_t)count * ncols) return;
    M[(size_t)rows[e / ncols] * ld + (e % ncols)] = T(0);
}

// Linv [n, n] row-major := (R^T)^-1 for the row-major upper Cholesky factor R (M = R^T R): lower triangular, zeros above the
// diagonal.  One workgroup; thread j solves R x = e_j by back substitution (x = column j of R^-1 = row j of Linv): the
// elements of R it reads are the same for every thread (broadcast reads), its own x stays in its row of the output.
// LDS = true (launched with 2 n (n + 1) elements of dynamic LDS when that fits): R and the rows being built are staged in LDS,
// so that the dependent chain of a column is LDS latency instead of a global round trip per element (k = 50: 110 -> ~25 us).
template <typename T, bool LDS>
__global__ void __launch_bounds__(256) trtri_from_upper_kernel(const T *__restrict__ R, int n, T *__restrict__ Linv)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char trtri_smem[];
    const int ld = n + 1;
    T *Rs = reinterpret_cast<T *>(trtri_smem), *Xs = Rs + (size_t)n * ld;
    if (LDS) {
        for (int e = threadIdx.x; e < n * n; e += 256) Rs[(e / n) * ld + (e % n)] = R[e];
        __syncthreads();
    }
    for (int j = threadIdx.x; j < n; j += 256) {
        T *x = LDS ? Xs + (size_t)j * ld : Linv + (size_t)j * n;
        const T *Rm = LDS ? Rs : R;
        const int ldr = LDS ? ld : n;
        for (int i = j + 1; i < n; i++) x[i] = T(0);
        x[j] = T(1) / Rm[(size_t)j * ldr + j];
        for (int i = j - 1; i >= 0; i--) {
            T acc = T(0);
            for (int l = i + 1; l <= j; l++) acc += Rm[(size_t)i * ldr + l] * x[l];
            x[i] = -acc / Rm[(size_t)i * ldr + i];
        }
        if (LDS)
            for (int i = 0; i < n; i++) Linv[(size_t)j * n + i] = x[i];
    }
}

// out[kt, kt] = blockdiag(lam I [ks, ks], G [kb, kb]) + CtC [kc, kc] in the upper-left corner, kt = ks + kb: the matrix every row of
// a missing-as-zero half-step with dense side information shares (collective.c:5700-5716; G = B^T B with its own diagonal
// already added, CtC already times w)
template <typename T>
__global__ void naz_block_matrix_kernel(const T *__restrict__ G, int kb, const T *__restrict__ CtC, int kc, int ks, T lam, T *__restrict__ out)
{
    const int kt = ks + kb;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < kt * kt; e += gridDim.x * blockDim.x) {
        const int i = e / kt, j = e % kt;
        T v = T(0);
        if (i >= ks && j >= ks) v = G[(size_t)(i - ks) * kb + (j - ks)];
        else if (i == j) v = lam;
        if (i < kc && j < kc) v += CtC[(size_t)i * kc + j];
        out[e] = v;
    }
}

// out[kt, kt] = 0 except the [kb, kb] block at (ks, ks), which takes G  (sum_mat of BiTBi into the X block of the row's
// system, collective.c:1704-1707)
template <typename T>
__global__ void embed_block_kernel(const T *__restrict__ G, int kb, int ks, int kt, T *__restrict__ out)
{
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < kt * kt; e += gridDim.x * blockDim.x) {
        const int i = e / kt - ks, j = e % kt - ks;
        out[e] = (i >= 0 && j >= 0 && i < kb && j < kb) ? G[(size_t)i * kb + j] : T(0);
    }
}

// Right-hand sides of the shared-matrix update (optimizeA Case 3 on the binary indicator, tgemm_sp_dense with unit values,
// common.c:3145-3151): out[row, :] = sum over the row's entries j of F[idx_j, :].  Two stages so that a row of tens of
// thousands of entries is not one wavefront's chain and the sum order stays fixed: (1) one wavefront per segment of up to
// GSUM_SEG entries, lane <-> column, four entries in flight; (2) one wavefront per row adds its segments in order.
constexpr int GSUM_SEG = 256;
constexpr int GSUM_MAXC = 5;                 // columns per lane: k + k_main <= 320

template <typename T>
__global__ void __launch_bounds__(256)
gather_sum_segments_kernel(const size_t *__restrict__ indptr, const int *__restrict__ indices, const T *__restrict__ F, size_t ldf,
                           int kk, const int *__restrict__ seg_row, const int *__restrict__ seg_off, int nseg, T *__restrict__ partial)
{
    const int seg = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (seg >= nseg) return;
    const int row = seg_row[seg];
    const size_t st = indptr[row] + (size_t)seg_off[seg];
    const int len = (int)min((size_t)GSUM_SEG, indptr[row + 1] - st);
    T acc[GSUM_MAXC];
#pragma unroll
    for (int c = 0; c < GSUM_MAXC; c++) acc[c] = T(0);
    for (int j0 = 0; j0 < len; j0 += 4) {
        int idx[4];
#pragma unroll
        for (int u = 0; u < 4; u++) idx[u] = indices[st + min(j0 + u, len - 1)];
        T v[4][GSUM_MAXC];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int c = 0; c < GSUM_MAXC; c++) {
                const int f = lane + 64 * c;
                v[u][c] = (f < kk && j0 + u < len) ? F[(size_t)idx[u] * ldf + f] : T(0);
            }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int c = 0; c < GSUM_MAXC; c++) acc[c] += v[u][c];
    }
#pragma unroll
    for (int c = 0; c < GSUM_MAXC; c++) {
        const int f = lane + 64 * c;
        if (f < kk) partial[(size_t)seg * kk + f] = acc[c];
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
gather_sum_rows_kernel(const T *__restrict__ partial, const int *__restrict__ row_seg_first, int rows, int kk, T *__restrict__ out,
                       size_t ldo)
{
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int s0 = row_seg_first[row], s1 = row_seg_first[row + 1];
    for (int f = lane; f < kk; f += 64) {
        T acc = T(0);
        for (int sg = s0; sg < s1; sg++) acc += partial[(size_t)sg * kk + f];
        out[(size_t)row * ldo + f] = acc;
    }
}

// dst[r, off + f] += scale * src[r, f]  for f < kk: the gathered implicit-features term joins the prefilled right-hand sides
template <typename T>
__global__ void add_cols_scaled_kernel(T *__restrict__ dst, size_t ldd, int off, const T *__restrict__ src, int kk, T scale, size_t rows)
{
    const size_t total = rows * (size_t)kk;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / (size_t)kk; const int f = (int)(e % (size_t)kk);
        dst[r * ldd + off + f] += scale * src[e];
    }
}

// U[r, c] -= colmeans[c]  (preprocess_vec on the rows of new side information, collective.c:6337-6349)
template <typename T>
__global__ void sub_colmeans_kernel(T *__restrict__ U, size_t rows, int p, const T *__restrict__ colmeans)
{
    const size_t total = rows * (size_t)p;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
        U[e] -= colmeans[e % (size_t)p];
}

// rows of A that have side information but no entries in X take the side-information-only ("cold") solution:
// A[r, :kc] = cold[r, :kc], the remaining unknowns (k_main, bias) are zero  (collective.c:10683-10703, :3639-3660)
template <typename T>
__global__ void cold_select_kernel(T *__restrict__ A, size_t lda, int kt, int kc, const T *__restrict__ cold,
                                   const size_t *__restrict__ indptr, int rows_u)
{
    const int r = blockIdx.x;
    if (r >= rows_u || indptr[r + 1] != indptr[r]) return;
    for (int e = threadIdx.x; e < kt; e += blockDim.x) A[(size_t)r * lda + e] = (e < kc) ? cold[(size_t)r * kc + e] : T(0);
}

// Block systems on the tiled CG kernels (cg_kernels.hpp, GRAMX): the weighted Gramian  w C^T C (+) 0 + w_i Bi^T Bi (+) 0  as one
// k x k matrix, and the per-row constants  w (U C)_row (+) 0 + w_i gsum_row (+) 0.
template <typename T>
__global__ void block_gram_kernel(const T *__restrict__ CtC, int kc, T w_side, const T *__restrict__ BiTBi, int ki, T w_imp, int k,
                                  T *__restrict__ out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= k * k) return;
    const int i = e / k, j = e % k;
    T v = T(0);
    if (CtC != nullptr && i < kc && j < kc) v += w_side * CtC[(size_t)i * kc + j];
    if (BiTBi != nullptr && i < ki && j < ki) v += w_imp * BiTBi[(size_t)i * ki + j];
    out[e] = v;
}
template <typename T>
__global__ void block_rconst_kernel(const T *__restrict__ UC, int kc, T w_side, const T *__restrict__ gsum, int ki, T w_imp, int k, size_t rows,
                                    T *__restrict__ out)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * (size_t)k) return;
    const size_t r = e / k; const int f = (int)(e % k);
    T v = T(0);
    if (UC != nullptr && f < kc) v += w_side * UC[r * kc + f];
    if (gsum != nullptr && f < ki) v += w_imp * gsum[r * ki + f];
    out[e] = v;
}

}  // namespace cmfhip
