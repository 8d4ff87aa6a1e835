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

constexpr int CHOL_CHUNK = 16;     // gathered rows staged per round
__host__ __device__ inline int chol_ldm(int kt) { return kt | 1; }     // odd leading dimension
__host__ __device__ inline size_t chol_lds_elems(int kt)
{
    return (size_t)kt * chol_ldm(kt) + (size_t)CHOL_CHUNK * (kt + 1) + 2 * (size_t)kt + CHOL_CHUNK + 8;
}

template <typename T>
__global__ void __launch_bounds__(256)
chol_rows_kernel(const CholParams<T> P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int kt = P.kt, koff = P.koff, kb = kt - koff;
    const int ldm = chol_ldm(kt);
    T *M = reinterpret_cast<T *>(smem_raw);                // [kt][ldm]
    T *Bs = M + (size_t)kt * ldm;                          // [CHUNK][kb+1]
    T *rhs = Bs + (size_t)CHOL_CHUNK * (kt + 1);           // [kt]
    T *wsc = rhs + kt;                                     // [CHUNK] syr weights
    T *xsc = wsc + CHOL_CHUNK;                             // (unused tail / scratch)
    (void)xsc;
    const int tid = threadIdx.x;
    const int ldbs = kb + 1;
    // 4x4 blocks of the upper triangle of the X-block (size kb): block (bi,bj), bj >= bi
    const int nb = (kb + 3) / 4;
    const int nblocks_tri = nb * (nb + 1) / 2;

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
        // ---- initialise M and rhs ----
        for (int e = tid; e < kt * ldm; e += 256) {
            int i = e / ldm, j = e % ldm;
            T v = T(0);
            if (j < kt) {
                if (P.mode == CHOL_IMPLICIT || P.mode == CHOL_PREFILLED) v = P.Minit[(size_t)i * kt + j];
                else if (has_u && i < P.kc && j < P.kc) v = P.Minit[(size_t)i * P.kc + j];   // collective.c:1566-1571
            }
            M[e] = v;
        }
        for (int e = tid; e < kt; e += 256)
            rhs[e] = has_u ? arow[e] : T(0);   // w*U*C prefilled (collective.c:5768-5773)
        __syncthreads();
        // ---- accumulate the gathered rows ----
        for (int c0 = 0; c0 < nnz; c0 += CHOL_CHUNK) {
            const int nr = min(CHOL_CHUNK, nnz - c0);
            for (int e = tid; e < nr * kb; e += 256) {
                int r = e / kb, c = e % kb;
                int idx = P.indices[st + c0 + r];
                Bs[r * ldbs + c] = P.B[(size_t)idx * P.ldb + c];
            }
            if (tid < nr) {
                int idx = P.indices[st + c0 + tid];
                T x = P.values[st + c0 + tid];
                if (P.bias_sub != nullptr) x -= P.bias_sub[idx];
                // weight of the rank-1 update / of the rhs contribution
                T wsyr = (P.mode == CHOL_IMPLICIT) ? x : T(1);           // common.c:2091-2095 vs :1007-1012
                T wrhs = (P.mode == CHOL_IMPLICIT) ? x + T(1) : x;       // common.c:2082-2085 vs :991-996
                wsc[tid] = wsyr;
                Bs[tid * ldbs + kb] = wrhs;
            }
            __syncthreads();
            for (int e = tid; e < kb; e += 256) {                        // rhs[koff+e] += sum_r wrhs_r B_r[e]
                T s = rhs[koff + e];
                for (int r = 0; r < nr; r++) s += Bs[r * ldbs + kb] * Bs[r * ldbs + e];
                rhs[koff + e] = s;
            }
            for (int blk = tid; blk < nblocks_tri; blk += 256) {
                // unrank blk -> (bi, bj) with bj >= bi, rows of the block triangle in order
                int bi = 0, rem = blk;
                while (rem >= nb - bi) { rem -= nb - bi; bi++; }
                int bj = bi + rem;
                const int i0 = bi * 4, j0 = bj * 4;
                T acc[4][4];
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++) acc[a][b] = T(0);
                for (int r = 0; r < nr; r++) {
                    T w = wsc[r];
                    T vi[4], vj[4];
#pragma unroll
                    for (int a = 0; a < 4; a++) vi[a] = (i0 + a < kb) ? w * Bs[r * ldbs + i0 + a] : T(0);
#pragma unroll
                    for (int b = 0; b < 4; b++) vj[b] = (j0 + b < kb) ? Bs[r * ldbs + j0 + b] : T(0);
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b = 0; b < 4; b++) acc[a][b] += vi[a] * vj[b];
                }
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        int i = i0 + a, j = j0 + b;
                        if (i < kb && j < kb && j >= i) M[(size_t)(koff + i) * ldm + koff + j] += acc[a][b];
                    }
            }
            __syncthreads();
        }
        // ---- + diag (add_to_diag / add_to_diag2: common.c:1060-1062, collective.c:1819) ----
        if (P.mode == CHOL_EXPLICIT || P.mode == CHOL_COLLECTIVE)
            for (int e = tid; e < kt; e += 256) M[(size_t)e * ldm + e] += (e == kt - 1) ? lam_last : lam;
        __syncthreads();
        // ---- in-place Cholesky of the upper triangle: M = R^T R ----
        for (int c = 0; c < kt; c++) {
            T d = sqrt(M[(size_t)c * ldm + c]);
            __syncthreads();
            for (int j = c + tid; j < kt; j += 256) M[(size_t)c * ldm + j] = (j == c) ? d : M[(size_t)c * ldm + j] / d;
            __syncthreads();
            const int rem = kt - c - 1;
            // trailing update of rows c+1.., entries j >= i
            for (int e = tid; e < rem * rem; e += 256) {
                int i = c + 1 + e / rem, j = c + 1 + e % rem;
                if (j >= i) M[(size_t)i * ldm + j] -= M[(size_t)c * ldm + i] * M[(size_t)c * ldm + j];
            }
            __syncthreads();
        }
        // ---- R^T y = rhs ----
        for (int c = 0; c < kt; c++) {
            if (tid == 0) rhs[c] = rhs[c] / M[(size_t)c * ldm + c];
            __syncthreads();
            T yc = rhs[c];
            for (int j = c + 1 + tid; j < kt; j += 256) rhs[j] -= M[(size_t)c * ldm + j] * yc;
            __syncthreads();
        }
        // ---- R x = y ----
        for (int c = kt - 1; c >= 0; c--) {
            if (tid == 0) rhs[c] = rhs[c] / M[(size_t)c * ldm + c];
            __syncthreads();
            T xc = rhs[c];
            for (int i = tid; i < c; i += 256) rhs[i] -= M[(size_t)i * ldm + c] * xc;
            __syncthreads();
        }
        for (int e = tid; e < kt; e += 256) arow[e] = rhs[e];
    }
}

}  // namespace cmfhip
