/*
 * cmf_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99 + OpenMP, no BLAS) of the ALS factor-update hot path of
 * david-cortes/cmfrec, written from the reading of the reference sources.  Every function cites
 * the reference file:line it follows.  It is the parity checker for the HIP path and the "port"
 * CPU baseline of bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it; the product (cmfrec_amd/) never does.
 *
 * Pinned against the real reference (oracle/_ref, built from /root/reference by oracle/Makefile)
 * by tests/test_oracle_vs_ref.py and against the committed fixtures in tests/golden/ (generated
 * from the real reference by tests/golden/make_golden.py).
 *
 * Precision is fixed per shared object like the reference (src/cmfrec.h:232-294):
 *   libcmf_oracle_double.so : real_t = double      libcmf_oracle_float.so : real_t = float
 * int_t = int (32 bit), CSR/CSC offsets are size_t (src/cmfrec.h:300-305, :991).
 */
#ifndef CMF_ORACLE_H
#define CMF_ORACLE_H
#include <stddef.h>
#include <stdbool.h>
#include <stdint.h>

#ifdef ORACLE_FLOAT
typedef float real_t;
#else
typedef double real_t;
#endif
typedef int int_t;

#ifdef __cplusplus
extern "C" {
#endif

int oracle_sizeof_real(void);

/* helpers.c:1375-1491 -- stable counting sort COO -> CSR and CSC (entries keep COO order). */
void oracle_coo_to_csr_and_csc(const int_t *Xrow, const int_t *Xcol, const real_t *Xval,
                               int_t m, int_t n, size_t nnz,
                               size_t *csr_p, int_t *csr_i, real_t *csr_v,
                               size_t *csc_p, int_t *csc_i, real_t *csc_v);

/* cblas_tsyrk(RowMajor, Upper, Trans) call sites common.c:2824,3328: out[k*k] = B[:, :k]^T B[:, :k].
 * Both triangles are filled. */
void oracle_gram(const real_t *B, size_t ldb, int_t n, int_t k, real_t *out, int nthreads);

/* common.c:3305-3421 (optimizeA_implicit).  use_cg: factors_implicit_cg (:1914-1986) or, with
 * precondition_cg, factors_implicit_pcg (:1988-2061); else factors_implicit_chol (:2063-2126).
 * BtB_out (k*k, may be NULL) receives BtB exactly as the reference leaves precomputedBtB
 * (with +lam on the diagonal in Cholesky mode). */
void oracle_optimizeA_implicit(real_t *A, size_t lda, const real_t *B, size_t ldb,
                               int_t m, int_t n, int_t k,
                               const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                               real_t lam, int nthreads,
                               bool use_cg, bool precondition_cg, int_t max_cg_steps,
                               real_t *BtB_out);

/* common.c:3209-3302 (optimizeA, Case 4: sparse X, missing-as-NA, no weights) ->
 * factors_closed_form sparse branches (:631-1095): CG (:1098-1188), PCG (:1190-1291) or
 * Cholesky (:978-1013,1060-1070). */
void oracle_optimizeA_explicit(real_t *A, size_t lda, const real_t *B, size_t ldb,
                               int_t m, int_t n, int_t k,
                               const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                               real_t lam, real_t lam_last,
                               bool scale_lam, bool scale_bias_const,
                               int nthreads,
                               bool use_cg, bool precondition_cg, int_t max_cg_steps);

/* Observation weights of the oracle_optimizeA_explicit call that follows (optimizeA Case 4 with weight != NULL: the weighted
 * branches of common.c:985-1012, :1126-1135, :1162-1171, :1222-1254): weights_csr_order[nnz] in the entry order of Xcsr and
 * wsum[m] = the driver's lambda multipliers under scale_lam (wsumA, collective.c:7978-8008), or NULL: the row's own sum
 * (common.c:696-707).  Cleared by the call. */
void oracle_set_row_weights(const real_t *weights_csr_order, const real_t *wsum);
/* NA_as_zero_X of the oracle_fit_explicit_als call that follows (sparse X, absent = zero; model without side information and
 * without weights, returns 2 otherwise): mean over all m x n cells (common.c:3517-3523), missing-as-zero bias start values
 * (common.c:4207-4237, :4453-4476, :4693-4710, :4849-4868), half-steps through optimizeA Case 3 (common.c:3118-3205) with the
 * right-hand-side constant of the driver (collective.c:8573-8600, :8756-8787).  Cleared by the call. */
void oracle_set_fit_NA_as_zero_X(bool on);
void oracle_set_closed_form_rows(const unsigned char *maskA, const unsigned char *maskB);   /* dense X, Case 2: rows solved in closed form inside a CG half-step */
void oracle_set_lambda_multipliers(const real_t *multA, const real_t *multB);   /* dense X under scale_lam: per-row lambda multipliers of the next (unit-weighted) fit */
void oracle_set_zero_rows(const int_t *rowsA, int_t nA, const int_t *rowsB, int_t nB);   /* NA_as_zero_U / _I: rows the next fit zeroes after their update */
/* bias_BtX[k] of the oracle_optimizeA_naz call that follows (common.c:3152-3157).  Cleared by the call. */
void oracle_set_naz_bias_BtX(const real_t *bias_BtX);

/* Observation weights (COO order) of the oracle_fit_explicit_als call that follows: weighted mean (common.c:3574-3584),
 * weightR / weightC, wsumA / wsumB, weighted bias start values (common.c:4672-4692, :4826-4847), weighted row solvers.  Model
 * without side information only (returns 2 otherwise).  Cleared by the call. */
void oracle_set_fit_weights(const real_t *weight);
void oracle_optimizeA_naz_weighted(real_t *A, size_t lda, const real_t *B, size_t ldb, int_t m, int_t n, int_t k,
                                   const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr, const real_t *weight,
                                   const real_t *wsum, const real_t *bias_BtX, const real_t *bias_X, real_t bias_X_glob,
                                   real_t lam, real_t lam_last, bool scale_lam, bool scale_bias_const,
                                   bool use_cg, bool precondition_cg, int_t max_cg_steps, int nthreads);
real_t oracle_calc_mean_and_center_weighted(real_t *X, const real_t *weight, size_t nnz, int nthreads);
void oracle_initialize_biases_twosided_weighted(int_t m, int_t n,
                                                const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr, const real_t *weightR,
                                                const size_t *Xcsc_p, const int_t *Xcsc_i, const real_t *Xcsc, const real_t *weightC,
                                                real_t lam_user, real_t lam_item, const real_t *wsumA, const real_t *wsumB,
                                                real_t *biasA, real_t *biasB);

/* common.c:2793-2991 (optimizeA, Case 1: dense full X, no weights) -- the C / D update.
 * do_B=false: A[m,k] = Xfull[m,n] B[n,k] (BtB+diag)^-1 ; do_B=true: Xfull is [n, ldX>=m] and is
 * used transposed (:2852-2855). */
void oracle_optimizeA_dense_full(real_t *A, size_t lda, const real_t *B, size_t ldb,
                                 int_t m, int_t n, int_t k,
                                 const real_t *Xfull, size_t ldX, bool do_B,
                                 real_t lam, real_t lam_last, bool scale_lam, int nthreads);

/* collective.c:4720-5969 general branch (:5566-5968) with sparse X (missing-as-NA), dense full
 * U (no NaN), Cholesky -> collective_closed_form_block (:1223-1847).  m_u may be < m (rows
 * >= m_u have no side info: u_vec == NULL, collective.c:5935 guarded by the m>m_u split :4832). */
void oracle_optimizeA_collective_chol(real_t *A, size_t lda, const real_t *B, size_t ldb,
                                      const real_t *C,
                                      int_t m, int_t m_u, int_t n, int_t p,
                                      int_t k, int_t k_main, int_t k_user, int_t k_item,
                                      const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                      const real_t *U,
                                      real_t lam, real_t w_user, real_t lam_last,
                                      bool scale_lam, bool scale_lam_sideinfo,
                                      int nthreads);

/* common.c:3423-3648 (sparse, unweighted branch :3494-3524, :3599-3604). Returns the mean and
 * subtracts it from X in place. nthreads>=8 switches from the running mean to sum/cnt (:3497). */
real_t oracle_calc_mean_and_center(real_t *X, size_t nnz, int nthreads);

/* common.c:4410-4909 (sparse, unweighted, !NA_as_zero branches :4643-4669, :4799-4825). */
void oracle_initialize_biases_twosided(int_t m, int_t n,
                                       const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                       const size_t *Xcsc_p, const int_t *Xcsc_i, const real_t *Xcsc,
                                       real_t lam_user, real_t lam_item, bool scale_lam,
                                       real_t *biasA, real_t *biasB);

/* collective.c:9375-10207 restricted to: no side info, k_main=k_user=k_item=0 allowed only as 0,
 * w_main=1, no L1/nonneg.  reset_values must be false (caller injects A, B). */
/* collective.c:5971-6244 + :1849-2131: implicit X + dense full U (no NaN), Cholesky, m_u <= m */
void oracle_optimizeA_collective_implicit_chol(real_t *A, size_t lda, const real_t *B, size_t ldb,
                                               const real_t *C,
                                               int_t m, int_t m_u, int_t n, int_t p,
                                               int_t k, int_t k_main, int_t k_user, int_t k_item,
                                               const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                               const real_t *U, real_t lam, real_t w_user, int nthreads);

int oracle_fit_implicit_als_sideinfo(real_t *A, real_t *B, real_t *C, real_t *D,
                                     real_t *U_colmeans, real_t *I_colmeans,
                                     int_t m, int_t n, int_t k,
                                     const int_t *ixA, const int_t *ixB, const real_t *X, size_t nnz,
                                     const real_t *U, int_t m_u, int_t p, const real_t *II, int_t n_i, int_t q,
                                     int_t k_main, int_t k_user, int_t k_item,
                                     real_t w_main, real_t w_user, real_t w_item,
                                     real_t lam, real_t alpha, bool apply_log_transf,
                                     int_t niter, int nthreads,
                                     bool use_cg, int_t max_cg_steps, bool precondition_cg, bool finalize_chol);

int oracle_fit_implicit_als(real_t *A, real_t *B, int_t m, int_t n, int_t k,
                            const int_t *ixA, const int_t *ixB, const real_t *X, size_t nnz,
                            real_t lam, real_t alpha, bool apply_log_transf,
                            int_t niter, int nthreads,
                            bool use_cg, int_t max_cg_steps, bool precondition_cg, bool finalize_chol);

/* collective.c:7263-9370 restricted to: sparse X missing-as-NA, no weights, optional dense full
 * U[m_u,p] / II[n_i,q] (m_u<=m, n_i<=n), w_main=1, no L1/nonneg/implicit features.
 * reset_values=false semantics: caller injects A, B (and biasA/biasB start values, C, D).
 * With side info only Cholesky is restated (block-CG is SURVEY 8f). Returns 0 ok, 2 unsupported. */
void oracle_optimizeA_naz(real_t *A, size_t lda, const real_t *B, size_t ldb, int_t m, int_t n, int_t k,
                          const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                          real_t lam, real_t lam_last, bool scale_lam, int nthreads);
int oracle_fit_explicit_als_implicit_features(real_t *biasA, real_t *biasB, real_t *A, real_t *B, real_t *C, real_t *D,
                            real_t *Ai, real_t *Bi, real_t w_implicit,
                            real_t *glob_mean, real_t *U_colmeans, real_t *I_colmeans,
                            int_t m, int_t n, int_t k,
                            const int_t *ixA, const int_t *ixB, const real_t *X, size_t nnz,
                            bool user_bias, bool item_bias, bool center,
                            real_t lam, bool scale_lam, bool scale_lam_sideinfo,
                            const real_t *U, int_t m_u, int_t p,
                            const real_t *II, int_t n_i, int_t q,
                            int_t k_main, int_t k_user, int_t k_item,
                            real_t w_user, real_t w_item,
                            int_t niter, int nthreads,
                            bool use_cg, int_t max_cg_steps, bool precondition_cg, bool finalize_chol,
                            bool init_biases);
void oracle_set_scale_bias_const(bool on);
void oracle_set_lam_unique(const real_t *lam6, const real_t *l16);
int oracle_fit_explicit_als(real_t *biasA, real_t *biasB, real_t *A, real_t *B, real_t *C, real_t *D,
                            real_t *glob_mean, real_t *U_colmeans, real_t *I_colmeans,
                            int_t m, int_t n, int_t k,
                            const int_t *ixA, const int_t *ixB, const real_t *X, size_t nnz,
                            bool user_bias, bool item_bias, bool center,
                            real_t lam, bool scale_lam, bool scale_lam_sideinfo,
                            const real_t *U, int_t m_u, int_t p,
                            const real_t *II, int_t n_i, int_t q,
                            int_t k_main, int_t k_user, int_t k_item,
                            real_t w_user, real_t w_item,
                            int_t niter, int nthreads,
                            bool use_cg, int_t max_cg_steps, bool precondition_cg, bool finalize_chol,
                            bool init_biases);

/* Factors of new rows: factors_collective_explicit_multiple (collective.c:10865-11174) -> per row
 * factors_collective_explicit_single (:10575-10739) -> collective_factors_warm (:3555-3964) / collective_factors_cold
 * (:3309-3440), restricted to sparse X (CSR of the new rows, m rows) and dense U[m_u, p] without NaN.
 * A[max(m, m_u), k_user+k+k_main]; biasA (nullable) switches the user bias on ([B | 1] layout, lam_bias on the last
 * unknown).  TransCtCinvCt (nullable, [p, k_user+k]) is what the cold rows use when given (:3380-3386). */
void oracle_factors_explicit_multiple(real_t *A, real_t *biasA, int_t m,
                                      const real_t *U, int_t m_u, int_t p, const real_t *C,
                                      real_t glob_mean, const real_t *biasB, const real_t *U_colmeans,
                                      const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                      int_t n, const real_t *B,
                                      int_t k, int_t k_user, int_t k_item, int_t k_main,
                                      real_t lam, real_t lam_bias,
                                      bool scale_lam, bool scale_lam_sideinfo, bool scale_bias_const, real_t scaling_biasA,
                                      real_t w_main, real_t w_user, const real_t *TransCtCinvCt, int nthreads);

/* factors_collective_implicit_multiple (collective.c:11176-11340) -> factors_collective_implicit_single (:10741-10863)
 * -> collective_factors_warm_implicit (:3966-4087) / collective_factors_cold_implicit (:3442-3553); same restrictions.
 * BtB (nullable, [k+k_main]^2, lam included) as the caller precomputed it; when NULL it is built with the lam of the
 * call *before* the w_main rescaling (:11270-11280). */
void oracle_factors_implicit_multiple(real_t *A, int_t m,
                                      const real_t *U, int_t m_u, int_t p, const real_t *C, const real_t *U_colmeans,
                                      const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                      int_t n, const real_t *B,
                                      int_t k, int_t k_user, int_t k_item, int_t k_main,
                                      real_t lam, real_t alpha, real_t w_main, real_t w_user, real_t w_main_multiplier,
                                      bool apply_log_transf, const real_t *BtB, int nthreads);

/* Collective half-step with SPARSE side information (U as CSR over m_u rows, missing = absent), Cholesky:
 * collective_closed_form_block (collective.c:1223-1847) / collective_closed_form_block_implicit (:1849-2131),
 * branches u_vec == NULL && u_vec_sp != NULL && !NA_as_zero_U. */
void oracle_optimizeA_collective_sparse_chol(real_t *A, size_t lda, const real_t *B, size_t ldb, const real_t *C,
                                             int_t m, int_t m_u, int_t n, int_t p,
                                             int_t k, int_t k_main, int_t k_user, int_t k_item,
                                             const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                             const size_t *Ucsr_p, const int_t *Ucsr_i, const real_t *Ucsr,
                                             real_t lam, real_t w_user, real_t lam_last,
                                             bool scale_lam, bool scale_lam_sideinfo, bool implicit, int nthreads);

/* Block CG / PCG on the collective system with SPARSE side information (collective_block_cg :2134-2903 and
 * collective_block_cg_implicit :2905-3303, u_vec_sp branches). */
void oracle_optimizeA_collective_sparse_cg(real_t *A, size_t lda, const real_t *B, size_t ldb, const real_t *C,
                                           int_t m, int_t m_u, int_t n, int_t p,
                                           int_t k, int_t k_main, int_t k_user, int_t k_item,
                                           const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                           const size_t *Ucsr_p, const int_t *Ucsr_i, const real_t *Ucsr,
                                           real_t lam, real_t w_user, real_t lam_last,
                                           bool scale_lam, bool scale_lam_sideinfo, bool implicit,
                                           int_t max_cg_steps, bool precondition_cg, int nthreads);

/* Whole fits with SPARSE side information (COO, missing = absent), Cholesky or CG / PCG updates, m_u <= m, n_i <= n,
 * injected start values: fit_collective_explicit_als (collective.c:7263-9370) / fit_collective_implicit_als (:9375-10207). */
/* NA_as_zero_X of the next oracle_fit_als_sparse_sideinfo / oracle_optimizeA_collective_sparse_chol call (explicit model, closed form) */
void oracle_set_sparse_fit_NA_as_zero_X(bool on);
/* dense side information with NaN run as the sparse fit: per-attribute rules of the dense C / D update (see cmf_oracle.c) */
void oracle_set_sideinfo_dense_rules(const unsigned char *cfC, const real_t *multC, const unsigned char *cfD, const real_t *multD);
void oracle_set_collective_sparse_naz(bool on, const real_t *bias_BtX);
int oracle_fit_als_sparse_sideinfo(bool implicit, real_t *biasA, real_t *biasB, real_t *A, real_t *B, real_t *C, real_t *D,
                                   real_t *glob_mean, int_t m, int_t n, int_t k,
                                   const int_t *ixA, const int_t *ixB, const real_t *X, size_t nnz,
                                   bool user_bias, bool item_bias, bool center,
                                   real_t lam, real_t alpha, bool scale_lam, bool scale_lam_sideinfo,
                                   const int_t *U_row, const int_t *U_col, const real_t *U_sp, size_t nnz_U, int_t m_u, int_t p,
                                   const int_t *I_row, const int_t *I_col, const real_t *I_sp, size_t nnz_I, int_t n_i, int_t q,
                                   int_t k_main, int_t k_user, int_t k_item,
                                   real_t w_main, real_t w_user, real_t w_item, int_t niter, int nthreads,
                                   bool use_cg, int_t max_cg_steps, bool precondition_cg, bool finalize_chol);

/* Non-negativity option for everything above (process-wide, test infrastructure): the fits use solve_nonneg
 * (common.c:2131-2179) for A / B (nonneg), C (nonneg_C), D (nonneg_D) and switch the CG off like the reference
 * (collective.c:7474-7479); oracle_set_nonneg_now turns it on for operator-level calls outside a fit. */
void oracle_set_nonneg(bool nonneg, bool nonneg_C, bool nonneg_D, int_t max_cd_steps);
void oracle_set_nonneg_now(bool on, int_t max_cd_steps);
/* L1 penalty (scalar l1_lam) of the following fits / of operator-level calls: solve_elasticnet (common.c:2228-2294),
 * or the shifted right-hand side of solve_nonneg; scaled per row like lambda; no CG. */
void oracle_set_l1(real_t l1_lam, int_t max_cd_steps);
void oracle_set_l1_now(real_t l1, int_t max_cd_steps);

#ifdef __cplusplus
}
#endif
#endif
