/*
 * cmfrec_hip.h -- C ABI of the MI355X-native ALS factor-update path behind cmfrec's interface.
 *
 * One shared object per precision, like the reference (setup.py:389-412 builds wrapper_double /
 * wrapper_float from the same sources):
 *     libcmfrec_hip_double.so : real_t = double        libcmfrec_hip_float.so : real_t = float
 * (compile the including translation unit with -DCMFREC_HIP_FLOAT for the float library).
 * int_t is the 32-bit `int` of the reference's default build (src/cmfrec.h:300-305); CSR/CSC
 * offsets are size_t (src/cmfrec.h:991).  Dense matrices are row-major; missing optional pointers
 * are NULL and optional sizes 0 (include/cmfrec.h.in:238-241).
 *
 * Three levels, from the outside in:
 *   1. the two drop-in fit entry points, byte-for-byte the reference's positional signatures;
 *   2. the operator boundary: one factor update with host buffers (what the reference's internal
 *      optimizeA* functions compute), used by the parity tests;
 *   3. a device-resident session (factors + CSR/CSC stay in HBM across half-iterations), which is
 *      what level 1 drives and what bench.py times.
 *
 * Return codes (include/cmfrec.h.in:210-213): 0 ok, 1 out of memory (host or device), 2 invalid or
 * unsupported input, 3 interrupted.  Additionally 4 = HIP runtime failure (no usable gfx950
 * device, kernel launch error); there is NO CPU fallback.
 */
#ifndef CMFREC_HIP_H
#define CMFREC_HIP_H
#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>

#ifdef CMFREC_HIP_FLOAT
typedef float real_t;
#else
typedef double real_t;
#endif
typedef int int_t;

#ifdef __cplusplus
extern "C" {
#endif

/* ============================ level 1: drop-in fit entry points ============================ */

/* Replaces fit_collective_implicit_als, /root/reference/src/cmfrec.h:1893-1921 (body
 * src/collective.c:9375-10207; documented in include/cmfrec.h.in:927-959).
 * Supported: CG / PCG / Cholesky, optionally with DENSE side information U[m_u, p] / II[n_i, q] without NaN
 * (m_u, n_i may exceed m, n: A, B then have max(m, m_u) / max(n, n_i) rows, src/collective.c:9437-9440) (k_user, k_item, k_main, w_main, w_user, w_item as the reference): Cholesky
 * (optimizeA_collective_implicit, src/collective.c:5971-6244) or block CG / PCG
 * (collective_block_cg_implicit, :2905-3303).  No adjust_weight.
 * or SPARSE side information as COO triplets (U_row / U_col / U_sp / nnz_U and the I_* twins; missing = absent, rows
 * within X's; collective.c:1849-2131 / :2905-3303 with u_vec_sp); nonneg / nonneg_C / nonneg_D with max_cd_steps
 * (solve_nonneg, common.c:2131-2179; k_t <= 140 / 199), l1_lam (solve_elasticnet, :2228-2294) and the per-matrix
 * lam_unique / l1_lam_unique (entries 2..5: A, B, C, D; :9793-9809, :9855-10016); dense U / II with NaN (= missing) under the
 * plain Cholesky solver; precompute_for_predictions.  Anything else returns 2. */
int_t fit_collective_implicit_als(
    real_t *A, real_t *B,
    real_t *C, real_t *D,
    bool reset_values, int_t seed,
    real_t *U_colmeans, real_t *I_colmeans,
    int_t m, int_t n, int_t k,
    int_t ixA[], int_t ixB[], real_t *X, size_t nnz,
    real_t lam, real_t *lam_unique,
    real_t l1_lam, real_t *l1_lam_unique,
    real_t *U, int_t m_u, int_t p,
    real_t *II, int_t n_i, int_t q,
    int_t U_row[], int_t U_col[], real_t *U_sp, size_t nnz_U,
    int_t I_row[], int_t I_col[], real_t *I_sp, size_t nnz_I,
    bool NA_as_zero_U, bool NA_as_zero_I,
    int_t k_main, int_t k_user, int_t k_item,
    real_t w_main, real_t w_user, real_t w_item,
    real_t *w_main_multiplier,
    real_t alpha, bool adjust_weight, bool apply_log_transf,
    int_t niter, int nthreads,
    bool verbose, bool handle_interrupt,
    bool use_cg, int_t max_cg_steps, bool precondition_cg, bool finalize_chol,
    bool nonneg, int_t max_cd_steps, bool nonneg_C, bool nonneg_D,
    bool precompute_for_predictions,
    real_t *precomputedBtB,
    real_t *precomputedBeTBe,
    real_t *precomputedBeTBeChol,
    real_t *precomputedCtUbias);

/* Replaces fit_collective_explicit_als, /root/reference/src/cmfrec.h:1851-1892 (body
 * src/collective.c:7263-9370; include/cmfrec.h.in:885-926).
 * Supported: sparse X with missing-as-NA, biases/centering/scale_lam,
 * CG / PCG or Cholesky, optional DENSE side information U[m_u, p] / II[n_i, q] without NaN (m_u, n_i may exceed
 * m, n: rows known from side information only are fitted to it alone and get a zero bias, :4967-5101, :8296)
 * (Cholesky: collective_closed_form_block, src/collective.c:1223-1847; CG: collective_block_cg, :2134-2903),
 * k_main/k_user/k_item, w_user/w_item; also SPARSE side information as COO triplets (missing = absent, rows within
 * X's; collective_closed_form_block / collective_block_cg with u_vec_sp, :1636-1653, :1719-1731, :2609-2621); nonneg /
 * nonneg_C / nonneg_D with max_cd_steps (solve_nonneg, common.c:2131-2179), l1_lam (solve_elasticnet) and the per-matrix
 * lam_unique / l1_lam_unique (user bias, item bias, A, B, C, D; :8178-8219, :8367-8423, :8649-8654, :8820-8825, :9066-9238);
 * add_implicit_features with Ai, Bi, w_implicit (:8448-8534, :1704-1771, :2301-2304, :2624-2643, :2862-2868; Cholesky or
 * CG / PCG, dense or no side information inside X, not with nonneg / L1 / precompute_for_predictions); dense U / II with
 * NaN (= missing) under the Cholesky solver with unscaled lambda; scale_bias_const (scaling_biasA / scaling_biasB are
 * outputs); observation weights (weight, one per entry of X; src/common.c:1098-1291, :679-723); NA_as_zero_X on sparse X for
 * the model without side information (optimizeA Case 3, src/common.c:3118-3205; src/collective.c:8573-8600); dense X
 * (Xfull [m, n], NaN = missing, weight [m, n]) for the model without side information, with the reference's choice of solver
 * per half-step (optimizeA Cases 1-2, src/common.c:2787-3116).  Not built: NA_as_zero_U / NA_as_zero_I.  Anything else
 * returns 2. */
int_t fit_collective_explicit_als(
    real_t *biasA, real_t *biasB,
    real_t *A, real_t *B,
    real_t *C, real_t *D,
    real_t *Ai, real_t *Bi,
    bool add_implicit_features,
    bool reset_values, int_t seed,
    real_t *glob_mean,
    real_t *U_colmeans, real_t *I_colmeans,
    int_t m, int_t n, int_t k,
    int_t ixA[], int_t ixB[], real_t *X, size_t nnz,
    real_t *Xfull,
    real_t *weight,
    bool user_bias, bool item_bias, bool center,
    real_t lam, real_t *lam_unique,
    real_t l1_lam, real_t *l1_lam_unique,
    bool scale_lam, bool scale_lam_sideinfo, bool scale_bias_const,
    real_t *scaling_biasA, real_t *scaling_biasB,
    real_t *U, int_t m_u, int_t p,
    real_t *II, int_t n_i, int_t q,
    int_t U_row[], int_t U_col[], real_t *U_sp, size_t nnz_U,
    int_t I_row[], int_t I_col[], real_t *I_sp, size_t nnz_I,
    bool NA_as_zero_X, bool NA_as_zero_U, bool NA_as_zero_I,
    int_t k_main, int_t k_user, int_t k_item,
    real_t w_main, real_t w_user, real_t w_item, real_t w_implicit,
    int_t niter, int nthreads,
    bool verbose, bool handle_interrupt,
    bool use_cg, int_t max_cg_steps, bool precondition_cg, bool finalize_chol,
    bool nonneg, int_t max_cd_steps, bool nonneg_C, bool nonneg_D,
    bool precompute_for_predictions,
    bool include_all_X,
    real_t *B_plus_bias,
    real_t *precomputedBtB,
    real_t *precomputedTransBtBinvBt,
    real_t *precomputedBtXbias,
    real_t *precomputedBeTBeChol,
    real_t *precomputedBiTBi,
    real_t *precomputedTransCtCinvCt,
    real_t *precomputedCtCw,
    real_t *precomputedCtUbias);

/* ============================ level 2: operator boundary (host buffers) ==================== */

/* One iALS half-step; replaces optimizeA_implicit, src/cmfrec.h:1014-1027 / src/common.c:3305-3421.
 * A[m,lda] in/out, B[n,ldb]; the k solved columns start at A / B.  BtB_out (k*k) optional. */
int cmfrec_hip_optimizeA_implicit(
    real_t *A, size_t lda, const real_t *B, size_t ldb,
    int_t m, int_t n, int_t k,
    const size_t Xcsr_p[], const int_t Xcsr_i[], const real_t *Xcsr,
    real_t lam,
    bool use_cg, bool precondition_cg, int_t max_cg_steps,
    real_t *BtB_out);

/* One explicit-feedback half-step on sparse X (missing = NA, unweighted); replaces optimizeA
 * Case 4, src/cmfrec.h:986-1008 / src/common.c:3209-3302.  bias_sub (length n, optional) is
 * subtracted from the stored values on the fly, replacing the host sweeps collective.c:8566-8570 /
 * :8750-8754. */
int cmfrec_hip_optimizeA_explicit(
    real_t *A, size_t lda, const real_t *B, size_t ldb,
    int_t m, int_t n, int_t k,
    const size_t Xcsr_p[], const int_t Xcsr_i[], const real_t *Xcsr,
    const real_t *bias_sub,
    real_t lam, real_t lam_last, bool scale_lam, bool scale_bias_const,
    bool use_cg, bool precondition_cg, int_t max_cg_steps);

/* The same with observation weights (optimizeA Case 4 with weight != NULL, src/common.c:3268-3299 -> the weighted branches
 * of factors_closed_form :985-1012 and factors_explicit_cg / _pcg :1126-1135, :1162-1171, :1222-1254): weight[nnz] in the
 * entry order of Xcsr; wsum[m] (optional) = the lambda multipliers of scale_lam the driver passes (wsumA,
 * src/collective.c:7978-8008), NULL = every row's own sum of weights.  weight == NULL: cmfrec_hip_optimizeA_explicit. */
int cmfrec_hip_optimizeA_explicit_weighted(
    real_t *A, size_t lda, const real_t *B, size_t ldb,
    int_t m, int_t n, int_t k,
    const size_t Xcsr_p[], const int_t Xcsr_i[], const real_t *Xcsr, const real_t *weight, const real_t *wsum,
    const real_t *bias_sub,
    real_t lam, real_t lam_last, bool scale_lam, bool scale_bias_const,
    bool use_cg, bool precondition_cg, int_t max_cg_steps);

/* Dense full side-information update (the C / D step); replaces optimizeA Case 1,
 * src/common.c:2793-2991.  do_B: Xfull is [n, ldX] and used transposed (common.c:2852-2855). */
int cmfrec_hip_optimizeA_dense_full(
    real_t *A, size_t lda, const real_t *B, size_t ldb,
    int_t m, int_t n, int_t k,
    const real_t *Xfull, size_t ldX, bool do_B,
    real_t lam, real_t lam_last, bool scale_lam);

/* Collective (block-system) half-step with dense full U, Cholesky; replaces optimizeA_collective
 * general branch, src/cmfrec.h:1646-1683 / src/collective.c:5566-5968 ->
 * collective_closed_form_block (:1223-1847). */
int cmfrec_hip_optimizeA_collective(
    real_t *A, size_t lda, const real_t *B, size_t ldb, const real_t *C,
    int_t m, int_t m_u, int_t n, int_t p,
    int_t k, int_t k_main, int_t k_user, int_t k_item,
    const size_t Xcsr_p[], const int_t Xcsr_i[], const real_t *Xcsr,
    const real_t *bias_sub,
    const real_t *U,
    real_t lam, real_t w_user, real_t lam_last,
    bool scale_lam, bool scale_lam_sideinfo);

/* Factors of rows that were not part of the fit, one batched pass (level 2 under the two drop-in entry points below):
 * the closed-form row update of the fit with B (and C) fixed -- collective_factors_warm / _cold and their implicit
 * twins, src/collective.c:3309-4087.  X: COO (ixA, ixB, X, nnz) or CSR (Xcsr_*, m_x+1 offsets), values already
 * shifted by the global mean / scaled by alpha; biasB is subtracted inside the gather.  U raw, U_colmeans subtracted on
 * the device.  A is [max(m_x, m_u), k_user+k+k_main], biasA (optional) one value per row.  lam_x: what the implicit
 * model puts on the diagonal of the X block (see factors_collective_implicit_multiple below); BtB_pre (implicit,
 * lam included) and TransCtCinvCt_pre ([p, k_user+k]) replace the matrices otherwise rebuilt from B and C. */
int cmfrec_hip_factors_multiple(
    real_t *A, real_t *biasA, int_t m_x, int_t m_u, int_t p, const real_t *U, const real_t *U_colmeans,
    const int_t ixA[], const int_t ixB[], const real_t *X, size_t nnz,
    const size_t Xcsr_p[], const int_t Xcsr_i[], const real_t *Xcsr,
    const real_t *B, int_t n, const real_t *C, const real_t *biasB,
    int_t k, int_t k_user, int_t k_item, int_t k_main,
    real_t lam, real_t lam_bias, real_t lam_x, real_t w_user,
    bool implicit, bool scale_lam, bool scale_lam_sideinfo, bool scale_bias_const,
    const real_t *BtB_pre, const real_t *TransCtCinvCt_pre,
    /* sparse side information instead of U (NULL): COO triplets or CSR over the m_u rows, missing = absent */
    const int_t U_row[], const int_t U_col[], const real_t *U_sp, size_t nnz_U,
    const size_t U_csr_p[], const int_t U_csr_i[], const real_t *U_csr,
    bool nonneg /* solve_nonneg instead of the Cholesky solves, 10 (k_user+k+k_main[+1]) sweeps at most */);
int cmfrec_hip_factors_multiple_l1(
    real_t *A, real_t *biasA, int_t m_x, int_t m_u, int_t p, const real_t *U, const real_t *U_colmeans,
    const int_t ixA[], const int_t ixB[], const real_t *X, size_t nnz,
    const size_t Xcsr_p[], const int_t Xcsr_i[], const real_t *Xcsr,
    const real_t *B, int_t n, const real_t *C, const real_t *biasB,
    int_t k, int_t k_user, int_t k_item, int_t k_main,
    real_t lam, real_t lam_bias, real_t lam_x, real_t w_user,
    bool implicit, bool scale_lam, bool scale_lam_sideinfo, bool scale_bias_const,
    const real_t *BtB_pre, const real_t *TransCtCinvCt_pre,
    /* sparse side information instead of U (NULL): COO triplets or CSR over the m_u rows, missing = absent */
    const int_t U_row[], const int_t U_col[], const real_t *U_sp, size_t nnz_U,
    const size_t U_csr_p[], const int_t U_csr_i[], const real_t *U_csr,
    bool nonneg /* solve_nonneg instead of the Cholesky solves, 10 (k_user+k+k_main[+1]) sweeps at most */,
    /* L1 penalty of the row systems / of the bias unknown, after the w_main rescaling (solve_elasticnet,
     * /root/reference/src/common.c:2228-2294; collective.c:3571-3931); TransCtCinvCt_pre must be NULL */
    real_t l1_lam, real_t l1_lam_bias);

/* Replace factors_collective_explicit_multiple / factors_collective_implicit_multiple,
 * /root/reference/src/cmfrec.h:2004-2047 and :2048-2071 (bodies src/collective.c:10865-11174, :11176-11340): same
 * positional parameters, same return codes (0 ok, 1 out of memory, 2 invalid / unsupported).  Supported: sparse X
 * (COO or CSR), dense U without NaN (rows beyond m get the side-information-only solution, rows beyond m_u the plain
 * one), user bias, lam_unique, scale_lam / scale_lam_sideinfo / scale_bias_const, w_main / w_user, alpha and
 * apply_log_transf; sparse side information (COO or CSR; not together with scale_lam_sideinfo in the explicit
 * version).  NA_as_zero, L1, weights, dense X, binary side information and implicit features return 2; nonneg is
 * supported (solve_nonneg on every row system).  Of the precomputed matrices only BtB (implicit) and TransCtCinvCt (explicit) are read -- the
 * ones that change the result; the others are rebuilt on the device from B and C.
 * Two details of the reference that are kept: (1) without a precomputed BtB the implicit version puts the lam of the
 * call, *not* lam / w_main, on the diagonal of the X block (collective.c:11270-11280) while the k_user block gets
 * lam / w_main; (2) the side-information-only solution of the explicit version under scale_lam_sideinfo scales lam by
 * p on all but the last of the k_user+k unknowns (:3397-3411). */
int_t factors_collective_explicit_multiple(
    real_t *A, real_t *biasA, int_t m,
    real_t *U, int_t m_u, int_t p,
    bool NA_as_zero_U, bool NA_as_zero_X,
    bool nonneg,
    int_t U_row[], int_t U_col[], real_t *U_sp, size_t nnz_U,
    size_t U_csr_p[], int_t U_csr_i[], real_t *U_csr,
    real_t *Ub, int_t m_ubin, int_t pbin,
    real_t *C, real_t *Cb,
    real_t glob_mean, real_t *biasB,
    real_t *U_colmeans,
    real_t *X, int_t ixA[], int_t ixB[], size_t nnz,
    size_t *Xcsr_p, int_t *Xcsr_i, real_t *Xcsr,
    real_t *Xfull, int_t n,
    real_t *weight,
    real_t *B,
    real_t *Bi, bool add_implicit_features,
    int_t k, int_t k_user, int_t k_item, int_t k_main,
    real_t lam, real_t *lam_unique,
    real_t l1_lam, real_t *l1_lam_unique,
    bool scale_lam, bool scale_lam_sideinfo,
    bool scale_bias_const, real_t scaling_biasA,
    real_t w_main, real_t w_user, real_t w_implicit,
    int_t n_max, bool include_all_X,
    real_t *BtB, real_t *TransBtBinvBt, real_t *BtXbias, real_t *BeTBeChol, real_t *BiTBi,
    real_t *TransCtCinvCt, real_t *CtCw, real_t *CtUbias, real_t *B_plus_bias,
    int nthreads);

int_t factors_collective_implicit_multiple(
    real_t *A, int_t m,
    real_t *U, int_t m_u, int_t p,
    bool NA_as_zero_U,
    bool nonneg,
    int_t U_row[], int_t U_col[], real_t *U_sp, size_t nnz_U,
    size_t U_csr_p[], int_t U_csr_i[], real_t *U_csr,
    real_t *X, int_t ixA[], int_t ixB[], size_t nnz,
    size_t *Xcsr_p, int_t *Xcsr_i, real_t *Xcsr,
    real_t *B, int_t n,
    real_t *C,
    real_t *U_colmeans,
    int_t k, int_t k_user, int_t k_item, int_t k_main,
    real_t lam, real_t l1_lam, real_t alpha, real_t w_main, real_t w_user,
    real_t w_main_multiplier,
    bool apply_log_transf,
    real_t *BeTBe, real_t *BtB, real_t *BeTBeChol, real_t *CtUbias,
    int nthreads);

/* The same half-step with SPARSE side information: U as CSR over its m_u rows (missing attributes are simply absent,
 * !NA_as_zero_U).  Explicit (implicit = false: collective_closed_form_block, src/collective.c:1223-1847) or implicit
 * model (collective_closed_form_block_implicit, :1849-2131), Cholesky.  A is overwritten ([m, lda], first
 * k_user+k+k_main columns). */
int cmfrec_hip_optimizeA_collective_sparse(
    real_t *A, size_t lda, const real_t *B, size_t ldb, const real_t *C,
    int_t m, int_t m_u, int_t n, int_t p,
    int_t k, int_t k_main, int_t k_user, int_t k_item,
    const size_t Xcsr_p[], const int_t Xcsr_i[], const real_t *Xcsr,
    const real_t *bias_sub,
    const size_t Ucsr_p[], const int_t Ucsr_i[], const real_t *Ucsr,
    real_t lam, real_t w_user, real_t lam_last,
    bool scale_lam, bool scale_lam_sideinfo, bool implicit);

/* ============================ level 3: device-resident session ============================= */

typedef struct cmfrec_hip_session cmfrec_hip_session;

typedef struct cmfrec_hip_model {
    int32_t implicit;                 /* 1: iALS (optimizeA_implicit), 0: explicit (optimizeA / _collective) */
    int32_t m, n, k;                  /* global sizes, shared factors */
    int32_t k_main, k_user, k_item;
    int32_t user_bias, item_bias;     /* explicit only: bias rides as an extra column (collective.c:7651-7663) */
    int32_t scale_lam, scale_lam_sideinfo;
    int32_t use_cg, precondition_cg, max_cg_steps;
    int32_t p, q, m_u, n_i;           /* dense side info widths / rows (0 = none) */
    real_t lam, w_user, w_item;
    /* row-block shard owned by this process (single GPU: [0,m) and [0,n)) */
    int32_t row_begin, row_end, col_begin, col_end;
    /* m, n are the rows of A and B.  Side information may describe more users / items than X has rows / columns
     * (m_u > m_x: the reference's m_max = max(m, m_u), src/collective.c:7332-7335): then m_x / n_x give the shape
     * of X; 0 = same as m / n.  Rows beyond m_x are solved from their side information alone. */
    int32_t m_x, n_x;
} cmfrec_hip_model;

/* device < 0: use the current HIP device. Returns NULL on failure (see cmfrec_hip_last_error). */
cmfrec_hip_session *cmfrec_hip_session_create(const cmfrec_hip_model *model, int device);
void cmfrec_hip_session_destroy(cmfrec_hip_session *s);
const char *cmfrec_hip_last_error(void);
/* return code (1 out of memory, 2 invalid input, 4 HIP failure) that goes with the last NULL from cmfrec_hip_session_create */
int cmfrec_hip_last_error_code(void);

/* CSR of the local user rows [row_begin,row_end) (indptr rebased to 0, column ids global) and CSC
 * of the local item columns [col_begin,col_end) (row ids global).  Host buffers; copied to HBM. */
int cmfrec_hip_session_set_X(cmfrec_hip_session *s,
                             const size_t *csr_p, const int_t *csr_i, const real_t *csr_v,
                             const size_t *csc_p, const int_t *csc_i, const real_t *csc_v);
/* Same from the COO triplet (host buffers): CSR and CSC are built in HBM by a stable sort, i.e. with the
 * entry order of coo_to_csr_and_csc (src/helpers.c:1375-1491); values become (x - subtract) * alpha on the
 * way (centring src/common.c:3603-3613; confidence src/collective.c:9606-9611).  Only for sessions that own
 * all rows and columns. */
/* Multi-GPU overlap (optional, after set_X with the same CSR): cut the local rows of A into `nparts` contiguous parts
 * with their own processing orders.  update('A') then finishes part after part and records one event per part;
 * stream_wait_part makes another stream (the one the all-gather of that part is issued on) wait for it, so the
 * collective of a finished part runs beside the kernels of the following ones.  Sessions without user side
 * information only.  nparts <= 1 removes the split. */
int cmfrec_hip_session_set_A_parts(cmfrec_hip_session *s, const size_t *csr_p, const int_t *csr_i,
                                   const real_t *csr_v, int nparts);
int cmfrec_hip_session_nparts(cmfrec_hip_session *s);
int cmfrec_hip_session_part_range(cmfrec_hip_session *s, int part, int *begin, int *end);   /* local row offsets */
int cmfrec_hip_session_stream_wait_part(cmfrec_hip_session *s, int part, void *stream);

int cmfrec_hip_session_set_X_coo(cmfrec_hip_session *s, const int_t *row, const int_t *col, const real_t *val,
                                 size_t nnz, real_t subtract, real_t alpha);
/* The same with observation weights (explicit model): weight[nnz], one per entry, travels through the same stable sort as
 * the values (weightR / weightC of coo_to_csr_and_csc, src/helpers.c:1375-1491).  Every row solver then weights the entry's
 * rank-1 term and right-hand side (src/common.c:985-1012, :1126-1135, :1162-1171, :1246-1254), scale_lam multiplies lambda by
 * the row's sum of weights (:679-723; src/collective.c:7978-8008) and init_biases takes weighted means (src/common.c:4180-4205,
 * :4672-4692, :4826-4847).  weight == NULL: as cmfrec_hip_session_set_X_coo.  Returns 2 for the implicit model.
 * cmfrec_hip_session_set_X_weighted: the same from host CSR + CSC with their weights in the same entry order. */
int cmfrec_hip_session_set_X_coo_weighted(cmfrec_hip_session *s, const int_t *row, const int_t *col, const real_t *val,
                                          const real_t *weight, size_t nnz, real_t subtract, real_t alpha);
int cmfrec_hip_session_set_X_weighted(cmfrec_hip_session *s,
                                      const size_t *csr_p, const int_t *csr_i, const real_t *csr_v, const real_t *csr_w,
                                      const size_t *csc_p, const int_t *csc_i, const real_t *csc_v, const real_t *csc_w);
/* One shard from a COO triplet already resident in HBM (device pointers): which = 'r': CSR of the local rows, d_key = row -
   row_begin, d_other = global column; 'c': CSC of the local columns, d_key = column - col_begin, d_other = global row.
   Multi-GPU set-up path (the triplets come out of an all-to-all between the ranks); no reference counterpart: the
   reference converts on the host (src/helpers.c:1375-1491). */
int cmfrec_hip_session_set_X_coo_device(cmfrec_hip_session *s, int which, const int_t *d_key, const int_t *d_other,
                                        const real_t *d_val, size_t nnz, real_t subtract, real_t alpha);
/* NA_as_zero for the main matrix (explicit model, sparse X whose absent entries are zeros; plain model on one device): from
 * here on every update('A' / 'B') is optimizeA Case 3 (src/common.c:3118-3205) -- one shared matrix opp^T opp + lambda (x rows
 * under scale_lam), right-hand sides sum_j x_j opp_j over the row's entries plus the constant -sum over ALL opposing rows of
 * (their bias + glob_mean [when center]) x row (src/collective.c:8573-8600, :8756-8787), one factorisation and a triangular
 * solve for every row, with or without entries.  A side with dense complete side information on exactly the rows of X:
 * optimizeA_collective's factorised block matrix (src/collective.c:5607-5617, :5700-5716), closed form only.  X must have been
 * set uncentred. */
int cmfrec_hip_session_set_NA_as_zero_X(cmfrec_hip_session *s, int on, int center, real_t glob_mean);
/* Rows of A (which = 'A') or B ('B') that every update of that matrix leaves at ZERO, bias included, instead of solving them:
 * with NA_as_zero_U / NA_as_zero_I the reference does not solve a row that has neither an entry of X nor an entry of the side
 * information (collective_closed_form_block, src/collective.c:1262-1271; _implicit: :1876-1884), while the zero-filled dense
 * matrix the fit runs on (fit.hip, ZeroFilledSide) would.  count = 0 clears the list. */
int cmfrec_hip_session_set_zero_rows(cmfrec_hip_session *s, int which, const int_t *rows, int count);
/* Rows of A ('A') or B ('B') that take the CLOSED FORM in every update of that matrix, also when the update runs CG for the
 * others: a dense X whose half-step (optimizeA Case 2, src/common.c:2992-3116) has rows missing fewer than 2 k entries (closed form
 * from the precomputed B^T B, factors_closed_form :662, :759-790) next to rows missing more (the solver asked for).  mask: one byte
 * per row of the matrix, non-zero = closed form; NULL clears it.  A CG update then solves every row by CG, keeps a copy, solves
 * every row in closed form and puts the copy back for the rows whose byte is zero.
 * 'C' / 'D' (round 5): the attributes of DENSE side information with NaN, which the session holds as the sparse matrix of its
 * present, centred values (cmfrec_hip_session_set_sideinfo_sparse).  The reference's dense C / D update (optimizeA Cases 1-2 on
 * the transposed matrix, src/common.c:2793-3116) treats an attribute by the number of values it misses; one byte per attribute:
 * 0 = the solver asked for, 1 = closed form (fewer than 2 (k_side + k) missing values, :759-790; the complete attributes of a
 * matrix with at least 75 % complete ones), 2 = CG from zero with k_side + k steps (the other attributes of such a matrix,
 * :2953-2985).  fit_collective_*_als sets them (fit.hip, DenseNanSide::rules). */
int cmfrec_hip_session_set_closed_form_rows(cmfrec_hip_session *s, int which, const unsigned char *mask);
/* The lambda multipliers of the rows of A ('A') / B ('B') under scale_lam, instead of the rows' own sums of weights: a dense X
 * under scale_lam, where a row that misses fewer than 2 k entries keeps the n lam of a complete row (its matrix is the precomputed
 * B^T B + n lam I minus the missing rows, src/common.c:759-790, :3031-3032) while the others take lam times their present entries.
 * The session must hold X with observation weights (unit weights for this use); call after the bias start values.  mult: one
 * value per row of the matrix.
 * 'C' / 'D' (round 5): the multipliers of the attributes of dense side information with NaN under scale_lam (the same rule:
 * the rows of U x lam for an attribute that misses fewer than 2 (k_side + k) values, its present values otherwise); the
 * session puts unit weights on its attribute-major copy of the sparse side information. */
int cmfrec_hip_session_set_lambda_multipliers(cmfrec_hip_session *s, int which, const real_t *mult);
/* Bias start values of the explicit model from the resident X, as initialize_biases_twosided /
 * _onesided (src/common.c:4410-4909, 4265-4289; call sites src/collective.c:8166-8220): written to the
 * session's bias vectors and the bias columns of A / B.  Call after set_X* and set_factors. */
int cmfrec_hip_session_init_biases(cmfrec_hip_session *s, real_t lam_user, real_t lam_item);
/* Copies the resident CSR (which = 'r') or CSC ('c') back: indptr[rows+1], indices[nnz], values[nnz],
 * order[rows] = rows in processing order.  NULL = skip. */
int cmfrec_hip_session_get_X(cmfrec_hip_session *s, int which, size_t *indptr, int_t *indices, real_t *values,
                             int_t *order);
/* Full factor matrices A[m, k_user+k+k_main], B[n, k_item+k+k_main] (+ biasA[m], biasB[n] when the
 * model has them; C[p,k_user+k], D[q,k_item+k]); host buffers.  NULL = leave unchanged. */
int cmfrec_hip_session_set_factors(cmfrec_hip_session *s, const real_t *A, const real_t *B,
                                   const real_t *biasA, const real_t *biasB,
                                   const real_t *C, const real_t *D);
int cmfrec_hip_session_get_factors(cmfrec_hip_session *s, real_t *A, real_t *B,
                                   real_t *biasA, real_t *biasB, real_t *C, real_t *D);
/* Dense side information, already centred by column (host does column means like
 * common.c:4911-4997): U rows [0,m_u), II rows [0,n_i). */
int cmfrec_hip_session_set_sideinfo(cmfrec_hip_session *s, const real_t *U, const real_t *II);
/* Non-negativity constraints: nonneg for A and B, nonneg_C / nonneg_D for the side-information factors.  The closed-form
 * row systems are then solved by the reference's cyclic coordinate descent (solve_nonneg, src/common.c:2131-2179,
 * at most max_cd_steps sweeps, 0 = until converged) and the CG is switched off for that matrix (common.c:725, :2781).
 * The k_t x k_t system lives in LDS: k_t <= 140 (double) / 199 (single). */
/* Row-block shards of the explicit / collective model (one session per GPU): side information restricted to
   the rows of the block -- U_local = rows [row_begin, min(row_end, m_u)) of U, I_local = rows [col_begin, min(col_end, n_i))
   of I (already centred) -- and the C / D update (optimizeA Case 1, src/common.c:2793-2991) cut in two:
   _partial leaves [F_loc^T F_loc | U_loc^T F_loc] in a device buffer (cmfrec_hip_session_device_ptr(s, 'P', &elems, NULL)),
   the caller all-reduces it over the ranks, _finish adds lambda and solves -- the same small system on every rank. */
int cmfrec_hip_session_set_sideinfo_local(cmfrec_hip_session *s, const real_t *U_local, const real_t *I_local);
int cmfrec_hip_session_sideinfo_partial(cmfrec_hip_session *s, int which /* 'C' | 'D' */);
int cmfrec_hip_session_sideinfo_finish(cmfrec_hip_session *s, int which /* 'C' | 'D' */);
int cmfrec_hip_session_set_nonneg(cmfrec_hip_session *s, int nonneg, int nonneg_C, int nonneg_D, int max_cd_steps);
/* Implicit features of the explicit model (add_implicit_features of fit_collective_explicit_als,
 * /root/reference/src/collective.c:7269, :8448-8534): Ai [m, k+k_main] and Bi [n, k+k_main] factorise the binary
 * "was observed" pattern of X with weight w_implicit (already divided by w_main).  After this call the session's
 * updates 'b' (Bi from A) and 'a' (Ai from B) exist, iterate() runs them between D and B like the reference, and the
 * A / B updates carry the extra term (collective.c:1704-1707, :1757-1771).  Ai / Bi may be NULL (start values are never
 * read by the closed-form updates).  Ai / Bi are always closed-form; A / B run Cholesky or the block CG like the rest of
 * the model.  Whole-matrix sessions, dense or no side information. */
int cmfrec_hip_session_set_implicit_features(cmfrec_hip_session *s, real_t w_implicit, const real_t *Ai, const real_t *Bi);
int cmfrec_hip_session_get_implicit_features(cmfrec_hip_session *s, real_t *Ai, real_t *Bi);

/* scale_bias_const of the explicit model (/root/reference/src/collective.c:7555, :8110-8160): rows solved without side
 * information keep the bias' lambda as given (lam_unique[0] / [1], which the caller has multiplied by scaling_biasA /
 * scaling_biasB) instead of scaling it by the row's count (common.c:679-723). */
int cmfrec_hip_session_set_scale_bias_const(cmfrec_hip_session *s, int on);

/* Per-matrix penalties (lam_unique / l1_lam_unique of the fit entry points, /root/reference/src/collective.c:430):
 * six values each in the order user bias, item bias, A, B, C, D, already divided by w_main.  Either pointer may be NULL
 * (that family keeps the scalar of the model / of set_l1).  The A / B updates use [2] / [3] and, for a fitted bias, [0] /
 * [1] on the last unknown (:8649-8654, :8820-8825); C / D use [4] / w_user, [5] / w_item (:8367, :8418); the implicit
 * model ignores [0], [1]. */
int cmfrec_hip_session_set_lam_unique(cmfrec_hip_session *s, const real_t *lam_unique, const real_t *l1_lam_unique,
                                      int max_cd_steps);

/* L1 penalty (one value for every matrix: A, B and the bias columns take l1_lam, C / D l1_lam / w_user, / w_item;
 * scaled per row like lambda).  Systems are then solved by solve_elasticnet (src/common.c:2228-2294), or by
 * solve_nonneg with the penalty on the right-hand side where a non-negativity constraint applies; no CG. */
int cmfrec_hip_session_set_l1(cmfrec_hip_session *s, real_t l1_lam, int max_cd_steps);
/* SPARSE side information as COO triplets (which = 'U': [m_u, p], 'I': [n_i, q]; missing = absent, no centring):
 * CSR by row and CSC by attribute are built on the device.  Cholesky updates: the row's attributes are a second gather
 * source of the row kernel; CG / PCG: a second gathered term of the block CG (generic kernel).  m_u <= rows of X. */
int cmfrec_hip_session_set_sideinfo_sparse(cmfrec_hip_session *s, int which, const int_t *row, const int_t *col,
                                           const real_t *val, size_t nnz);

/* One update of the local block, asynchronous on the session stream. which: 'A','B','C','D'.
 * use_cholesky != 0 forces the Cholesky solver for this call (finalize_chol). */
int cmfrec_hip_session_update(cmfrec_hip_session *s, int which, int use_cholesky);
/* niter full ALS iterations in the reference's order C, D, B, A (collective.c:8334-8898 /
 * :9827-10045) on a single-GPU session. */
int cmfrec_hip_session_iterate(cmfrec_hip_session *s, int niter, int finalize_chol);
int cmfrec_hip_session_sync(cmfrec_hip_session *s);

/* Device pointers for torch.distributed / RCCL plumbing (factor all-gather between half-steps).
 * which 'A'/'B': the [rows, ld] matrix incl. the bias column; returns NULL if unknown. */
void *cmfrec_hip_session_device_ptr(cmfrec_hip_session *s, int which, size_t *rows, size_t *ld);
void *cmfrec_hip_session_stream(cmfrec_hip_session *s);
/* To be called after the local block of `which` was refreshed on all ranks (after the all-gather):
 * refreshes what is derived from the full matrix (bias vectors). */
int cmfrec_hip_session_after_gather(cmfrec_hip_session *s, int which);
/* The "precompute_for_predictions" epilogue of the fit drivers (src/collective.c:8936-9249, 10056-10115) from
 * the resident factors.  Host buffers, NULL = skip.  With kp = k + k_main (+1 with a user bias, explicit),
 * kc = k_user + k, kq = k_user + kp:  BtB[kp, kp] (implicit: + lam I);  explicit only: TransBtBinvBt[n, kp],
 * TransCtCinvCt[p, kc];  with user side information: CtCw[kc, kc] (explicit), BeTBe[kq, kq] (implicit),
 * BeTBeChol[kq, kq] (upper triangle = the factor R, M = R^T R).
 * include_all_X (explicit model): use all max(n, n_i) rows of B, else the n columns X has (collective.c:9013).
 * last_step_cholesky: whether the last iteration of the fit used the Cholesky solver (use_cg=false or
 * finalize_chol).  It only matters for quirk Q9 of the reference, which this library reproduces: in the implicit
 * model with user side information, after a CG last step, BeTBe / BeTBeChol carry the UNWEIGHTED C^T C when
 * w_user != 1 (src/collective.c:10077 tests `w_user == 1.` where `!=` was meant); see DESIGN.md, Parity. */
int cmfrec_hip_session_precompute(cmfrec_hip_session *s, int last_step_cholesky, int include_all_X, real_t *BtB,
                                  real_t *TransBtBinvBt, real_t *BeTBe, real_t *BeTBeChol, real_t *CtCw,
                                  real_t *TransCtCinvCt);

/* HIP-event time (ms) and launch count of the row-update kernels of `which` ('A' or 'B') since
 * the last reset; synchronises the stream. */
int cmfrec_hip_session_kernel_time(cmfrec_hip_session *s, int which, double *ms, long *launches);
/* Per-kernel figures of the CG row-update launches of `which`.  Rows are scheduled in six nnz bins,
 * one persistent launch each: bin 1: 257..2048 nnz (8 waves/row), 2: 129..256 (4 waves/row),
 * 3: 65..128 (2 waves/row), 4: 33..64 (1 wave/row), 5: 1..32 (1 wave/row, half tiles, double-buffered
 * gather); bin 0: rows > 2048 nnz, whose CG passes are
 * split over many workgroups (one launch pair per pass; the figure covers the whole sequence).
 * ms = summed HIP-event time of that bin's launches since the last reset. */
int cmfrec_hip_session_bin_stats(cmfrec_hip_session *s, int which, int bin, double *ms, long *launches,
                                 long *rows, unsigned long long *nnz);
/* 1 if the launches of that bin run beside other kernels (few split rows on the second stream; every bin of a block
 * that is updated in parts): their event timings then include the neighbours' work and are not a kernel duration. */
int cmfrec_hip_session_bin_overlaps(cmfrec_hip_session *s, int which, int bin);
/* How the split rows (> 1024 entries) of that side are solved: 0 none, 1 streamed once per CG pass, 2 read once, CG on the
 * row's own Gramian (chosen when few split rows share an opposing row, e.g. a rank's item block of a multi-GPU run). */
int cmfrec_hip_session_vh_mode(cmfrec_hip_session *s, int which);
/* rows of X's CSR ('A') / CSC ('B') shard with at least this many entries are "split rows": their entries are stored sorted
   by the opposing index (a permutation of the row's sums) and take the split-row kernels (1025; double precision: 513 where
   the split rows go through their own Gramian) */
int cmfrec_hip_session_vh_min(cmfrec_hip_session *s, int which);
/* the most recent collective Cholesky half-step: rows solved by the low-rank kernels (0: path not taken) and the
   eigen-decomposition behind them (3 tridiagonalisation + implicit QL, 2 the one-workgroup Jacobi kernel; CMFREC_HIP_EIG=jacobi forces 2) */
int cmfrec_hip_session_lowrank_info(cmfrec_hip_session *s, int *rows, int *eig);
void cmfrec_hip_session_reset_timers(cmfrec_hip_session *s);

/* Batched top-N (the step after the path; the reference ranks one user per call: topN, src/common.c:5127-5380).
 * score(u, i) = A_u . B_i (+ biasB[i]) over the k columns starting at A / B (skip k_user / k_item by offsetting the
 * pointers), items in the user's exclusion list (CSR over the nu users, each list sorted ascending; NULL = none)
 * are skipped, the n_top best item ids per user are returned in descending score (ties: lower id first), -1 where
 * fewer than n_top items remain.  out_scores (optional) holds the scores as defined above; the reference adds
 * glob_mean + biasA[u] afterwards, which does not change the order.  k <= 64, n_top <= 128. */
int cmfrec_hip_topN_batch(const real_t *A, size_t lda, int_t nu, const real_t *B, size_t ldb, int_t n, int_t k,
                          const real_t *biasB, const size_t excl_p[], const int_t excl_i[], int_t n_top,
                          int_t *out_ids, real_t *out_scores);

/* Replace precompute_collective_explicit / precompute_collective_implicit, /root/reference/src/cmfrec.h:1922-1960 (bodies
 * src/collective.c:10209-10485, :10487-10566): the matrices the prediction functions reuse, from factors the caller holds -- the
 * same positional parameters, the same return codes.  Host buffers in and out; the Gramians run on the library's MFMA kernels,
 * the solves with many right-hand sides (TransBtBinvBt, TransCtCinvCt) on the row Cholesky kernel, BeTBeChol on the small
 * factorisation kernel.  As in the reference the symmetric outputs carry their upper triangle (row-major), the strictly lower
 * part is zero.  extra_precision (the reference's second summation order) is accepted and ignored. */
int_t precompute_collective_explicit(
    real_t *B, int_t n, int_t n_max, bool include_all_X,
    real_t *C, int_t p,
    real_t *Bi, bool add_implicit_features,
    real_t *biasB, real_t glob_mean, bool NA_as_zero_X,
    real_t *U_colmeans, bool NA_as_zero_U,
    int_t k, int_t k_user, int_t k_item, int_t k_main,
    bool user_bias,
    bool nonneg,
    real_t lam, real_t *lam_unique,
    bool scale_lam, bool scale_lam_sideinfo,
    bool scale_bias_const, real_t scaling_biasA,
    real_t w_main, real_t w_user, real_t w_implicit,
    real_t *B_plus_bias,
    real_t *BtB,
    real_t *TransBtBinvBt,
    real_t *BtXbias,
    real_t *BeTBeChol,
    real_t *BiTBi,
    real_t *TransCtCinvCt,
    real_t *CtCw,
    real_t *CtUbias);
int_t precompute_collective_implicit(
    real_t *B, int_t n,
    real_t *C, int_t p,
    real_t *U_colmeans, bool NA_as_zero_U,
    int_t k, int_t k_user, int_t k_item, int_t k_main,
    real_t lam, real_t w_main, real_t w_user, real_t w_main_multiplier,
    bool nonneg,
    bool extra_precision,
    real_t *BtB,
    real_t *BeTBe,
    real_t *BeTBeChol,
    real_t *CtUbias);

/* Replace topN_old_collective_explicit / topN_old_collective_implicit, /root/reference/src/cmfrec.h:2104-2127 (bodies
 * src/collective.c:11546-11613 over topN, src/common.c:5127-5380): the n_top best items of ONE user whose factors exist, with an
 * include list or an exclude list, scores (+ glob_mean + the user's bias) on request.  Same argument checks and return codes.
 * Scores of the candidates on the device, a stable descending radix sort; ties go to the earlier candidate (the lower item id
 * without an include list) where the reference's qsort leaves their order open.  Many users at once: cmfrec_hip_topN_batch. */
int_t topN_old_collective_explicit(
    real_t *a_vec, real_t a_bias,
    real_t *A, real_t *biasA, int_t row_index,
    real_t *B,
    real_t *biasB,
    real_t glob_mean,
    int_t k, int_t k_user, int_t k_item, int_t k_main,
    int_t *include_ix, int_t n_include,
    int_t *exclude_ix, int_t n_exclude,
    int_t *outp_ix, real_t *outp_score,
    int_t n_top, int_t n, int_t n_max, bool include_all_X, int nthreads);
int_t topN_old_collective_implicit(
    real_t *a_vec,
    real_t *A, int_t row_index,
    real_t *B,
    int_t k, int_t k_user, int_t k_item, int_t k_main,
    int_t *include_ix, int_t n_include,
    int_t *exclude_ix, int_t n_exclude,
    int_t *outp_ix, real_t *outp_score,
    int_t n_top, int_t n, int nthreads);

/* Runs the on-device self-test of the cross-lane primitives (DPP / permlane swaps) the row kernels
 * are built on; returns the number of mismatching lanes (0 = ok), negative = HIP failure. */
int cmfrec_hip_selftest_lanes(void);
/* timing / agreement probe of the dense contraction C[M, N] = op(A) B (transa: A stored [K, M]) on random operands: ms per call of
   the library's own MFMA kernel and of rocBLAS (loaded at run time if it is there; -1 and a plain reference kernel otherwise),
   largest difference of the two results relative to the largest entry */
int cmfrec_hip_gemm_probe(int M, int N, int K, int transa, int reps, double *ms_own, double *ms_rocblas, double *max_rel_diff);
/* The symmetric eigen-decomposition the low-rank row path runs once per half-step on w C^T C (replaces what the reference gets
 * from a k_t x k_t dposv per row, src/collective.c:1823, for rows of few entries): A [n, n] symmetric (2 <= n <= 320) ->
 * Q [n, n] with Q[i][c] = component i of eigenvector c, lam [n] clamped at zero.  method 0: Householder tridiagonalisation +
 * implicit QL (the default of the path), 1: one-workgroup Jacobi (cross-check).  ms: device milliseconds per run over `reps`. */
int cmfrec_hip_sym_eig(int n, const real_t *A, real_t *Q, real_t *lam, int method, int reps, double *ms);

/* Start values as the reference's random_parallel draws them (src/helpers.c:927-1043; xoshiro256++
 * seeded by splitmix64, truncated ziggurat normals or uniforms, scaled 2^-7): A <- stream(seed),
 * B <- the stream jumped once.  Host-only.  Returns 1 if the ziggurat tables are NumPy's exact
 * constants, 0 if they were recomputed (<= 1 ulp apart). */
int cmfrec_hip_random_parallel(real_t *A, size_t sizeA, real_t *B, size_t sizeB, int_t seed, bool normal);

/* The point-to-point schedule MultiDev::exchange issues for one half-step (direct placement): the operations of all
 * D devices for the block boundaries bb[0 .. D], five ints each -- {device, peer, send (1) / receive (0), first row, rows} -- in
 * issue order, at most `cap` of them written to `out`.  Host-only, no device needed: lets a test check that every send has its
 * matching receive with the same count and that the received blocks tile the replica.  Returns the number of operations. */
int cmfrec_hip_exchange_plan(int D, const int *bb, int *out, int cap);

/* Build info: sizeof(real_t), and the gfx target the kernels were compiled for. */
int cmfrec_hip_sizeof_real(void);
/* sizeof(cmfrec_hip_model) as compiled: lets a binding verify its mirror of the struct before the first call. */
int cmfrec_hip_sizeof_model(void);
const char *cmfrec_hip_build_info(void);
/* The CMFREC_HIP_* environment switches are read when a session is created; this reads them again for the
 * sessions that already exist (bench.py's pass with the nnz bins in line). */
void cmfrec_hip_reload_switches(void);

#ifdef __cplusplus
}
#endif
#endif
