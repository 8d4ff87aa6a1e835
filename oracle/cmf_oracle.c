/*
 * cmf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See cmf_oracle.h.
 *
 * Plain C99 + OpenMP restatement of cmfrec's ALS factor-update path (reference paths are
 * relative to /root/reference).  No BLAS: the BLAS-1/2 calls of the reference are written out as
 * sequential loops in the same order over the non-zeros, so that the only differences against the
 * real reference are the intra-vector summation order of OpenBLAS' SIMD kernels and FMA
 * contraction (tolerance-level, see tests/test_oracle_vs_ref.py).
 */
#include "cmf_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#else
static int omp_get_thread_num(void) { return 0; }
#endif

#ifdef ORACLE_FLOAT
#define sqrt_t sqrtf
#define fabs_t fabsf
#define log_t logf
#define EPSILON_T FLT_EPSILON
#else
#define sqrt_t sqrt
#define fabs_t fabs
#define log_t log
#define EPSILON_T DBL_EPSILON
#endif

int oracle_sizeof_real(void) { return (int)sizeof(real_t); }

/* ------------------------------------------------------------------------------------------ */
/* small dense helpers                                                                          */
/* ------------------------------------------------------------------------------------------ */
static real_t dot_(int_t k, const real_t *x, const real_t *y)
{
    real_t s = 0;
    for (int_t i = 0; i < k; i++) s += x[i] * y[i];
    return s;
}
static void axpy_(int_t k, real_t a, const real_t *x, real_t *y)
{
    for (int_t i = 0; i < k; i++) y[i] += a * x[i];
}
/* y = alpha * M x, M symmetric k x k with leading dimension ld (full storage) */
static void symv_(int_t k, real_t alpha, const real_t *M, int_t ld, const real_t *x, real_t *y)
{
    for (int_t i = 0; i < k; i++) {
        real_t s = 0;
        for (int_t j = 0; j < k; j++) s += M[(size_t)i * ld + j] * x[j];
        y[i] = alpha * s;
    }
}
/* upper-triangular rank-1 update, helpers.c:1729-1739 (custom_syr, explicit fma) */
static void syr_upper_(int_t k, real_t alpha, const real_t *x, real_t *M, int_t ld)
{
    for (int_t i = 0; i < k; i++) {
        real_t t = alpha * x[i];
        for (int_t j = i; j < k; j++)
#ifdef ORACLE_FLOAT
            M[(size_t)i * ld + j] = fmaf(t, x[j], M[(size_t)i * ld + j]);
#else
            M[(size_t)i * ld + j] = fma(t, x[j], M[(size_t)i * ld + j]);
#endif
    }
}
/* In-place Cholesky of the UPPER triangle of row-major M (== LAPACK 'L' on the column-major
 * view, which is how the reference calls tposv_/tpotrf_, e.g. common.c:1066-1070). M = R^T R. */
static int chol_upper_(int_t k, real_t *M, int_t ld)
{
    for (int_t i = 0; i < k; i++) {
        for (int_t j = i; j < k; j++) {
            real_t s = M[(size_t)i * ld + j];
            for (int_t t = 0; t < i; t++) s -= M[(size_t)t * ld + i] * M[(size_t)t * ld + j];
            if (i == j) {
                if (!(s > 0)) return 1;
                M[(size_t)i * ld + i] = sqrt_t(s);
            } else {
                M[(size_t)i * ld + j] = s / M[(size_t)i * ld + i];
            }
        }
    }
    return 0;
}
static void chol_solve_upper_(int_t k, const real_t *R, int_t ld, real_t *b)
{
    for (int_t i = 0; i < k; i++) {            /* R^T y = b */
        real_t s = b[i];
        for (int_t t = 0; t < i; t++) s -= R[(size_t)t * ld + i] * b[t];
        b[i] = s / R[(size_t)i * ld + i];
    }
    for (int_t i = k - 1; i >= 0; i--) {       /* R x = y */
        real_t s = b[i];
        for (int_t t = i + 1; t < k; t++) s -= R[(size_t)i * ld + t] * b[t];
        b[i] = s / R[(size_t)i * ld + i];
    }
}

/* solve_nonneg, common.c:2131-2179 (no L1): cyclic coordinate descent on the normal equations, a >= 0.
 * M: k x k, upper triangle valid (the lower one is filled here like fill_lower_triangle, helpers.c:1624). */
static void solve_nonneg_(int_t k, real_t *M, int_t ld, real_t *b, long max_cd_steps)
{
    for (int_t r = 1; r < k; r++)
        for (int_t c = 0; c < r; c++) M[(size_t)r * ld + c] = M[(size_t)c * ld + r];
    real_t a_prev[1024];
    for (int_t i = 0; i < k; i++) a_prev[i] = 0;
    if (max_cd_steps <= 0) max_cd_steps = 0x7fffffff;
    for (long iter = 0; iter < max_cd_steps; iter++) {
        real_t diff_iter = 0;
        for (int_t ix = 0; ix < k; ix++) {
            real_t newval = a_prev[ix] + b[ix] / M[(size_t)ix * ld + ix];
            newval = (newval >= 0) ? newval : 0;
            const real_t diff_val = newval - a_prev[ix];
            if (fabs((double)diff_val) > 1e-8) {
                diff_iter += (real_t)fabs((double)diff_val);
                for (int_t f = 0; f < k; f++) b[f] -= diff_val * M[(size_t)ix * ld + f];
                a_prev[ix] = newval;
            }
        }
        if (isnan(diff_iter) || !isfinite(diff_iter) || diff_iter < 1e-8) break;
    }
    for (int_t i = 0; i < k; i++) b[i] = a_prev[i];
}

/* solve_elasticnet, common.c:2228-2294: the same descent on a positive and a negative part, a = a+ - a-. */
static void solve_elasticnet_(int_t k, real_t *M, int_t ld, real_t *b, real_t l1, real_t l1_last, long max_cd_steps)
{
    for (int_t r = 1; r < k; r++)
        for (int_t c = 0; c < r; c++) M[(size_t)r * ld + c] = M[(size_t)c * ld + r];
    real_t bneg[1024], ap[1024], an[1024];
    for (int_t i = 0; i < k; i++) { ap[i] = 0; an[i] = 0; bneg[i] = -b[i] - l1; }
    for (int_t i = 0; i < k; i++) b[i] -= l1;
    if (l1 != l1_last) { b[k - 1] -= (l1_last - l1); bneg[k - 1] -= (l1_last - l1); }
    if (max_cd_steps <= 0) max_cd_steps = 0x7fffffff;
    for (long iter = 0; iter < max_cd_steps; iter++) {
        real_t diff_iter = 0;
        for (int_t ix = 0; ix < k; ix++) {
            real_t newval = ap[ix] + b[ix] / M[(size_t)ix * ld + ix];
            newval = (newval >= 0) ? newval : 0;
            const real_t d = newval - ap[ix];
            if (fabs((double)d) > 1e-8) {
                diff_iter += (real_t)fabs((double)d);
                for (int_t f = 0; f < k; f++) bneg[f] += d * M[(size_t)ix * ld + f];
                for (int_t f = 0; f < k; f++) b[f] -= d * M[(size_t)ix * ld + f];
                ap[ix] = newval;
            }
        }
        for (int_t ix = 0; ix < k; ix++) {
            real_t newval = an[ix] + bneg[ix] / M[(size_t)ix * ld + ix];
            newval = (newval >= 0) ? newval : 0;
            const real_t d = newval - an[ix];
            if (fabs((double)d) > 1e-8) {
                diff_iter += (real_t)fabs((double)d);
                for (int_t f = 0; f < k; f++) b[f] += d * M[(size_t)ix * ld + f];
                for (int_t f = 0; f < k; f++) bneg[f] -= d * M[(size_t)ix * ld + f];
                an[ix] = newval;
            }
        }
        if (isnan(diff_iter) || !isfinite(diff_iter) || diff_iter < 1e-8) break;
    }
    for (int_t i = 0; i < k; i++) b[i] = ap[i] - an[i];
}

/* L1 penalty of the step in progress (g_l1: base value of the matrix being updated) times the row's lambda multiplier
 * (t_l1_mult, set by the row loops where they scale lam): the reference scales both alike (common.c:716-722,
 * collective.c:1349-1354) */
static real_t g_l1 = 0, g_l1_base = 0;
static __thread real_t t_l1_mult = 1;
void oracle_set_l1(real_t l1_lam, int_t max_cd_steps);

/* What the closed-form row functions end with: posv, or solve_nonneg when the option is on (common.c:1066-1090,
 * :2101-2126, collective.c:1822-1846, :2107-2131).  The option is process-wide test-infrastructure state. */
static bool g_nonneg = false, g_nn_AB = false, g_nn_C = false, g_nn_D = false;
static long g_max_cd = 100;
void oracle_set_nonneg(bool nonneg, bool nonneg_C, bool nonneg_D, int_t max_cd_steps)
{
    g_nn_AB = nonneg; g_nn_C = nonneg_C; g_nn_D = nonneg_D; g_max_cd = max_cd_steps;
    g_nonneg = false; g_l1 = 0; t_l1_mult = 1;
}
/* for operator-level calls outside a fit */
void oracle_set_nonneg_now(bool on, int_t max_cd_steps) { g_nonneg = on; g_max_cd = max_cd_steps; }
void oracle_set_l1(real_t l1_lam, int_t max_cd_steps) { g_l1_base = l1_lam; g_l1 = 0; g_max_cd = max_cd_steps; }
void oracle_set_l1_now(real_t l1, int_t max_cd_steps) { g_l1 = l1; g_max_cd = max_cd_steps; }
/* per-matrix penalties (lam_unique / l1_lam_unique, collective.c:430): user bias, item bias, A, B, C, D -- as the caller
 * passes them (the fits divide by w_main like the reference, :7503-7520, :9793-9809) */
static bool g_has_lam6 = false, g_has_l16 = false;
static real_t g_lam6[6], g_l16[6];
void oracle_set_lam_unique(const real_t *lam6, const real_t *l16)
{
    g_has_lam6 = lam6 != NULL; g_has_l16 = l16 != NULL;
    for (int i = 0; i < 6; i++) { g_lam6[i] = lam6 ? lam6[i] : 0; g_l16[i] = l16 ? l16[i] : 0; }
}
/* scale_bias_const of the explicit fit (collective.c:7555, :8110-8160); rows solved without side information then keep the
 * bias' lambda unscaled (common.c:679-723), rows of the block system scale it anyway (collective.c:1347-1348) */
static bool g_scale_bias_const = false;
void oracle_set_scale_bias_const(bool on) { g_scale_bias_const = on; }
static bool g_l1_last_set = false;       /* the last unknown (a fitted bias) carries its own L1 penalty */
static real_t g_l1_last = 0;
static void solve_sym_(int_t k, real_t *M, int_t ld, real_t *b)
{
    const real_t l1 = g_l1 * t_l1_mult;
    const real_t l1_last = g_l1_last_set ? g_l1_last * t_l1_mult : l1;
    if (g_nonneg) {
        if (l1 != 0 || l1_last != 0) {                                          /* common.c:2148-2154 */
            for (int_t i = 0; i < k - 1; i++) b[i] -= l1;
            b[k - 1] -= l1_last;
        }
        solve_nonneg_(k, M, ld, b, g_max_cd);
        return;
    }
    if (l1 != 0 || l1_last != 0) { solve_elasticnet_(k, M, ld, b, l1, l1_last, g_max_cd); return; }
    if (chol_upper_(k, M, ld) == 0) { chol_solve_upper_(k, M, ld, b); return; }
    for (int_t i = 0; i < k; i++) b[i] = NAN;
}

/* ------------------------------------------------------------------------------------------ */
void oracle_coo_to_csr_and_csc(const int_t *Xrow, const int_t *Xcol, const real_t *Xval,
                               int_t m, int_t n, size_t nnz,
                               size_t *csr_p, int_t *csr_i, real_t *csr_v,
                               size_t *csc_p, int_t *csc_i, real_t *csc_v)
{
    /* helpers.c:1392-1402 */
    memset(csr_p, 0, ((size_t)m + 1) * sizeof(size_t));
    memset(csc_p, 0, ((size_t)n + 1) * sizeof(size_t));
    for (size_t ix = 0; ix < nnz; ix++) {
        csr_p[(size_t)Xrow[ix] + 1]++;
        csc_p[(size_t)Xcol[ix] + 1]++;
    }
    for (int_t r = 0; r < m; r++) csr_p[(size_t)r + 1] += csr_p[r];
    for (int_t c = 0; c < n; c++) csc_p[(size_t)c + 1] += csc_p[c];
    /* helpers.c:1419-1446: stable placement in COO order */
    size_t *cr = (size_t *)calloc((size_t)m + 1, sizeof(size_t));
    size_t *cc = (size_t *)calloc((size_t)n + 1, sizeof(size_t));
    for (size_t ix = 0; ix < nnz; ix++) {
        size_t r = (size_t)Xrow[ix], c = (size_t)Xcol[ix];
        size_t pr = csr_p[r] + cr[r]++;
        csr_v[pr] = Xval[ix];
        csr_i[pr] = Xcol[ix];
        size_t pc = csc_p[c] + cc[c]++;
        csc_v[pc] = Xval[ix];
        csc_i[pc] = Xrow[ix];
    }
    free(cr);
    free(cc);
}

void oracle_gram(const real_t *B, size_t ldb, int_t n, int_t k, real_t *out, int nthreads)
{
    (void)nthreads;
    /* accumulate in double then round: the reference's syrk is a blocked BLAS-3 kernel whose
       summation order is unspecified; tolerance-level agreement only */
    double *acc = (double *)calloc((size_t)k * k, sizeof(double));
    for (int_t r = 0; r < n; r++) {
        const real_t *b = B + (size_t)r * ldb;
        for (int_t i = 0; i < k; i++) {
            double bi = b[i];
            for (int_t j = i; j < k; j++) acc[(size_t)i * k + j] += bi * (double)b[j];
        }
    }
    for (int_t i = 0; i < k; i++)
        for (int_t j = i; j < k; j++) {
            out[(size_t)i * k + j] = (real_t)acc[(size_t)i * k + j];
            out[(size_t)j * k + i] = (real_t)acc[(size_t)i * k + j];
        }
    free(acc);
}

/* ------------------------------------------------------------------------------------------ */
/* implicit-feedback per-row solvers                                                            */
/* ------------------------------------------------------------------------------------------ */
/* common.c:1914-1986 */
static void implicit_cg_row(real_t *a, int_t k, const real_t *B, size_t ldb,
                            const real_t *Xa, const int_t *ixB, size_t nnz, real_t lam,
                            const real_t *BtB, int_t max_cg_steps, real_t *buf)
{
    real_t *Ap = buf, *r = Ap + k, *p = r + k;
    symv_(k, (real_t)-1, BtB, k, a, r);                                       /* :1932 */
    for (size_t ix = 0; ix < nnz; ix++) {                                     /* :1936-1942 */
        const real_t *b = B + (size_t)ixB[ix] * ldb;
        real_t coef = dot_(k, b, a);
        axpy_(k, -(coef - (real_t)1.) * Xa[ix] - coef, b, r);
    }
    axpy_(k, -lam, a, r);                                                     /* :1943 */
    memcpy(p, r, (size_t)k * sizeof(real_t));
    real_t r_old = dot_(k, r, r);
    if (r_old <= (real_t)1e-12) return;                                       /* :1952 */
    for (int_t step = 0; step < max_cg_steps; step++) {
        symv_(k, (real_t)1, BtB, k, p, Ap);                                   /* :1958 */
        for (size_t ix = 0; ix < nnz; ix++) {                                 /* :1962-1968 */
            const real_t *b = B + (size_t)ixB[ix] * ldb;
            real_t coef = dot_(k, b, p);
            axpy_(k, coef * (Xa[ix] - (real_t)1.) + coef, b, Ap);
        }
        axpy_(k, lam, p, Ap);
        real_t alpha = r_old / dot_(k, Ap, p);
        axpy_(k, alpha, p, a);
        axpy_(k, -alpha, Ap, r);
        real_t r_new = dot_(k, r, r);
        if (r_new <= (real_t)1e-8) break;                                     /* :1979 */
        real_t ratio = r_new / r_old;
        for (int_t i = 0; i < k; i++) p[i] = p[i] * ratio + r[i];             /* :1982-1983 */
        r_old = r_new;
    }
}

/* common.c:1988-2061 (Jacobi-preconditioned, fixed step count) */
static void implicit_pcg_row(real_t *a, int_t k, const real_t *B, size_t ldb,
                             const real_t *Xa, const int_t *ixB, size_t nnz, real_t lam,
                             const real_t *BtB, int_t max_cg_steps, real_t *buf)
{
    real_t *Ap = buf, *r = Ap + k, *p = r + k, *z = p + k, *PC = z + k;
    memset(PC, 0, (size_t)k * sizeof(real_t));
    for (size_t ix = 0; ix < nnz; ix++) {                                     /* :2009-2014 */
        const real_t *b = B + (size_t)ixB[ix] * ldb;
        for (int_t i = 0; i < k; i++) PC[i] += Xa[ix] * (b[i] * b[i]);
    }
    for (int_t i = 0; i < k; i++) PC[i] += BtB[(size_t)i * k + i];
    for (int_t i = 0; i < k; i++) PC[i] = (real_t)1 / PC[i];
    symv_(k, (real_t)-1, BtB, k, a, r);
    for (size_t ix = 0; ix < nnz; ix++) {
        const real_t *b = B + (size_t)ixB[ix] * ldb;
        real_t coef = dot_(k, b, a);
        axpy_(k, -(coef - (real_t)1.) * Xa[ix] - coef, b, r);
    }
    axpy_(k, -lam, a, r);
    for (int_t i = 0; i < k; i++) z[i] = r[i] * PC[i];
    real_t r_old = dot_(k, z, r);
    memcpy(p, z, (size_t)k * sizeof(real_t));
    for (int_t step = 0; step < max_cg_steps; step++) {
        symv_(k, (real_t)1, BtB, k, p, Ap);
        for (size_t ix = 0; ix < nnz; ix++) {
            const real_t *b = B + (size_t)ixB[ix] * ldb;
            real_t coef = dot_(k, b, p);
            axpy_(k, coef * (Xa[ix] - (real_t)1.) + coef, b, Ap);
        }
        axpy_(k, lam, p, Ap);
        real_t alpha = r_old / dot_(k, Ap, p);
        axpy_(k, alpha, p, a);
        axpy_(k, -alpha, Ap, r);
        for (int_t i = 0; i < k; i++) z[i] = r[i] * PC[i];
        real_t r_new = dot_(k, z, r);
        real_t ratio = r_new / r_old;
        for (int_t i = 0; i < k; i++) p[i] = p[i] * ratio + z[i];
        r_old = r_new;
    }
}

/* common.c:2063-2126 (BtB already holds +lam on its diagonal, :3333) */
static void implicit_chol_row(real_t *a, int_t k, const real_t *B, size_t ldb,
                              const real_t *Xa, const int_t *ixB, size_t nnz,
                              const real_t *BtB, real_t *buf)
{
    if (nnz == 0) { memset(a, 0, (size_t)k * sizeof(real_t)); return; }
    for (size_t ix = 0; ix < nnz; ix++)                                        /* :2082-2085 */
        axpy_(k, Xa[ix] + (real_t)1., B + (size_t)ixB[ix] * ldb, a);
    real_t *M = buf;
    memset(M, 0, (size_t)k * k * sizeof(real_t));
    for (size_t ix = 0; ix < nnz; ix++)                                        /* :2091-2095 */
        syr_upper_(k, Xa[ix], B + (size_t)ixB[ix] * ldb, M, k);
    for (int_t i = 0; i < k; i++)                                              /* :2097 sum_mat */
        for (int_t j = i; j < k; j++) M[(size_t)i * k + j] += BtB[(size_t)i * k + j];
    solve_sym_(k, M, k, a);
}

void oracle_optimizeA_implicit(real_t *A, size_t lda, const real_t *B, size_t ldb,
                               int_t m, int_t n, int_t k,
                               const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                               real_t lam, int nthreads,
                               bool use_cg, bool precondition_cg, int_t max_cg_steps,
                               real_t *BtB_out)
{
    if (nthreads < 1) nthreads = 1;
    real_t *BtB = (real_t *)malloc((size_t)k * k * sizeof(real_t));
    oracle_gram(B, ldb, n, k, BtB, nthreads);                                  /* :3328 */
    if (!use_cg) {                                                             /* :3332-3335 */
        for (int_t i = 0; i < k; i++) BtB[(size_t)i * k + i] += lam;
        for (size_t ix = 0; ix < (size_t)m * lda - (lda - (size_t)k); ix++) A[ix] = 0;
    }
    size_t szbuf = use_cg ? (size_t)(precondition_cg ? 5 : 3) * k : (size_t)k * k;
    real_t *bufs = (real_t *)malloc(szbuf * (size_t)nthreads * sizeof(real_t));
    #pragma omp parallel for schedule(dynamic) num_threads(nthreads)
    for (int_t ix = 0; ix < m; ix++) {
        t_l1_mult = 1;                                                         /* no row scaling of lambda / l1 in the implicit model */                                         /* :3349-3417 */
        size_t st = Xcsr_p[ix], en = Xcsr_p[(size_t)ix + 1];
        if (en <= st) continue;
        real_t *buf = bufs + szbuf * (size_t)omp_get_thread_num();
        real_t *a = A + (size_t)ix * lda;
        if (use_cg && !precondition_cg)
            implicit_cg_row(a, k, B, ldb, Xcsr + st, Xcsr_i + st, en - st, lam, BtB, max_cg_steps, buf);
        else if (use_cg)
            implicit_pcg_row(a, k, B, ldb, Xcsr + st, Xcsr_i + st, en - st, lam, BtB, max_cg_steps, buf);
        else
            implicit_chol_row(a, k, B, ldb, Xcsr + st, Xcsr_i + st, en - st, BtB, buf);
    }
    if (BtB_out != NULL) memcpy(BtB_out, BtB, (size_t)k * k * sizeof(real_t));
    free(bufs);
    free(BtB);
}

/* ------------------------------------------------------------------------------------------ */
/* explicit-feedback per-row solvers                                                            */
/* ------------------------------------------------------------------------------------------ */
/* common.c:1098-1188; wt: the row's observation weights (weighted branches :1126-1135, :1162-1171) or NULL */
static void explicit_cg_row(real_t *a, int_t k, const real_t *B, size_t ldb,
                            const real_t *Xa, const int_t *ixB, size_t nnz, const real_t *wt,
                            real_t lam, real_t lam_last, int_t max_cg_steps, real_t *buf)
{
    real_t *Ap = buf, *p = Ap + k, *r = p + k;
    memset(r, 0, (size_t)k * sizeof(real_t));
    for (size_t ix = 0; ix < nnz; ix++) {                                      /* :1119-1124 / :1128-1134 */
        const real_t *b = B + (size_t)ixB[ix] * ldb;
        real_t coef = dot_(k, b, a);
        coef -= Xa[ix];
        if (wt != NULL) coef *= wt[ix];
        axpy_(k, -coef, b, r);
    }
    axpy_(k, -lam, a, r);                                                      /* :1138 */
    if (lam != lam_last) r[k - 1] -= (lam_last - lam) * a[k - 1];
    real_t r_old = dot_(k, r, r);
    if (r_old <= (real_t)1e-12) return;                                        /* :1147 */
    memcpy(p, r, (size_t)k * sizeof(real_t));
    for (int_t step = 0; step < max_cg_steps; step++) {
        memset(Ap, 0, (size_t)k * sizeof(real_t));
        for (size_t ix = 0; ix < nnz; ix++) {                                  /* :1157-1160 / :1164-1170 */
            const real_t *b = B + (size_t)ixB[ix] * ldb;
            real_t coef = dot_(k, b, p);
            if (wt != NULL) coef *= wt[ix];
            axpy_(k, coef, b, Ap);
        }
        axpy_(k, lam, p, Ap);
        if (lam != lam_last) Ap[k - 1] += (lam_last - lam) * p[k - 1];
        real_t alpha = r_old / dot_(k, p, Ap);
        axpy_(k, alpha, p, a);
        axpy_(k, -alpha, Ap, r);
        real_t r_new = dot_(k, r, r);
        if (r_new <= (real_t)1e-8) break;                                      /* :1180 */
        real_t ratio = r_new / r_old;
        for (int_t i = 0; i < k; i++) p[i] = p[i] * ratio + r[i];
        r_old = r_new;
    }
}

/* common.c:1190-1291; wt: the row's observation weights or NULL */
static void explicit_pcg_row(real_t *a, int_t k, const real_t *B, size_t ldb,
                             const real_t *Xa, const int_t *ixB, size_t nnz, const real_t *wt,
                             real_t lam, real_t lam_last, int_t max_cg_steps, real_t *buf)
{
    real_t *Ap = buf, *p = Ap + k, *r = p + k, *z = r + k, *PC = z + k;
    memset(r, 0, (size_t)k * sizeof(real_t));
    for (size_t ix = 0; ix < nnz; ix++) {
        const real_t *b = B + (size_t)ixB[ix] * ldb;
        real_t coef = dot_(k, b, a);
        coef -= Xa[ix];
        if (wt != NULL) coef *= wt[ix];                                        /* :1222-1229 */
        axpy_(k, -coef, b, r);
    }
    axpy_(k, -lam, a, r);
    if (lam != lam_last) r[k - 1] -= (lam_last - lam) * a[k - 1];
    memset(PC, 0, (size_t)k * sizeof(real_t));
    for (size_t ix = 0; ix < nnz; ix++) {                                      /* :1238-1243 / :1246-1254 */
        const real_t *b = B + (size_t)ixB[ix] * ldb;
        const real_t w_this = (wt != NULL) ? wt[ix] : (real_t)1;
        for (int_t i = 0; i < k; i++) PC[i] += w_this * (b[i] * b[i]);
    }
    for (int_t i = 0; i < k; i++) PC[i] += lam;
    if (lam != lam_last) PC[k - 1] += (lam_last - lam);
    for (int_t i = 0; i < k; i++) PC[i] = (real_t)1 / PC[i];
    for (int_t i = 0; i < k; i++) z[i] = r[i] * PC[i];
    real_t r_old = dot_(k, z, r);
    memcpy(p, z, (size_t)k * sizeof(real_t));
    for (int_t step = 0; step < max_cg_steps; step++) {
        memset(Ap, 0, (size_t)k * sizeof(real_t));
        for (size_t ix = 0; ix < nnz; ix++) {
            const real_t *b = B + (size_t)ixB[ix] * ldb;
            real_t coef = dot_(k, b, p);
            if (wt != NULL) coef *= wt[ix];
            axpy_(k, coef, b, Ap);
        }
        axpy_(k, lam, p, Ap);
        if (lam != lam_last) Ap[k - 1] += (lam_last - lam) * p[k - 1];
        real_t alpha = r_old / dot_(k, p, Ap);
        axpy_(k, alpha, p, a);
        axpy_(k, -alpha, Ap, r);
        for (int_t i = 0; i < k; i++) z[i] = r[i] * PC[i];
        real_t r_new = dot_(k, z, r);
        real_t ratio = r_new / r_old;
        for (int_t i = 0; i < k; i++) p[i] = p[i] * ratio + z[i];
        r_old = r_new;
    }
}

/* common.c:978-1013 + :1060-1070; wt: the row's observation weights or NULL */
static void explicit_chol_row(real_t *a, int_t k, const real_t *B, size_t ldb,
                              const real_t *Xa, const int_t *ixB, size_t nnz, const real_t *wt,
                              real_t lam, real_t lam_last, real_t *buf)
{
    memset(a, 0, (size_t)k * sizeof(real_t));
    for (size_t ix = 0; ix < nnz; ix++)                      /* tgemv_dense_sp / _weighted, helpers.c:1175-1203 */
        axpy_(k, (wt != NULL) ? wt[ix] * Xa[ix] : Xa[ix], B + (size_t)ixB[ix] * ldb, a);
    real_t *M = buf;
    memset(M, 0, (size_t)k * k * sizeof(real_t));
    for (size_t ix = 0; ix < nnz; ix++)                      /* common.c:1007-1012 */
        syr_upper_(k, (wt != NULL) ? wt[ix] : (real_t)1, B + (size_t)ixB[ix] * ldb, M, k);
    for (int_t i = 0; i < k - 1; i++) M[(size_t)i * k + i] += lam;            /* add_to_diag2 */
    M[(size_t)(k - 1) * k + (k - 1)] += lam_last;
    solve_sym_(k, M, k, a);
}

/* observation weights of the call that follows (optimizeA Case 4 with weight != NULL, common.c:3268-3299): one per entry in
 * the order of Xcsr, and the per-row lambda multipliers of scale_lam (wsumA / wsumB of the driver, or NULL: the row's sum in
 * real_t, common.c:696-712).  Cleared by the call. */
static const real_t *g_row_weights = NULL, *g_row_wsum = NULL;
void oracle_set_row_weights(const real_t *weights_csr_order, const real_t *wsum)
{
    g_row_weights = weights_csr_order; g_row_wsum = wsum;
}

/* Dense X, optimizeA Case 2 (common.c:2992-3116): a row that misses fewer than 2 k entries is solved in closed form from the
 * precomputed B^T B whatever use_cg says (factors_closed_form, :662, :759-790), the others by the solver asked for.  The masks
 * (one byte per row / column, non-zero = closed form) of the next oracle_fit_explicit_als call; g_cf_now: the one of the
 * half-step about to run (consumed by oracle_optimizeA_explicit). */
static const unsigned char *g_cf_rows_A = NULL, *g_cf_rows_B = NULL, *g_cf_now = NULL;
void oracle_set_closed_form_rows(const unsigned char *maskA, const unsigned char *maskB) { g_cf_rows_A = maskA; g_cf_rows_B = maskB; }
/* ... and under scale_lam such a row keeps the n lam of a complete row (its matrix is the precomputed B^T B + n lam I minus the
 * missing rows: factors_closed_form :759-790 with BtB_has_diag, the diagonal added at :3031-3032 / :2832), the others lam times
 * their present entries: the per-row multipliers of the next oracle_fit_explicit_als call, which must carry (unit) weights --
 * they replace wsumA / wsumB after the bias start values. */
static const real_t *g_lam_mult_A = NULL, *g_lam_mult_B = NULL;
void oracle_set_lambda_multipliers(const real_t *multA, const real_t *multB) { g_lam_mult_A = multA; g_lam_mult_B = multB; }

void oracle_optimizeA_explicit(real_t *A, size_t lda, const real_t *B, size_t ldb,
                               int_t m, int_t n, int_t k,
                               const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                               real_t lam, real_t lam_last,
                               bool scale_lam, bool scale_bias_const,
                               int nthreads,
                               bool use_cg, bool precondition_cg, int_t max_cg_steps)
{
    (void)n;
    const real_t *wts = g_row_weights, *wsums = g_row_wsum;
    g_row_weights = NULL; g_row_wsum = NULL;
    const unsigned char *cf = g_cf_now;
    g_cf_now = NULL;
    if (nthreads < 1) nthreads = 1;
    size_t szbuf = (size_t)k * k;                                              /* :3244-3249 */
    if (use_cg && cf == NULL) szbuf = (size_t)(precondition_cg ? 5 : 3) * k;
    real_t *bufs = (real_t *)malloc(szbuf * (size_t)nthreads * sizeof(real_t));
    #pragma omp parallel for schedule(dynamic) num_threads(nthreads)
    for (int_t ix = 0; ix < m; ix++) {                                         /* :3268-3299 */
        size_t st = Xcsr_p[ix], en = Xcsr_p[(size_t)ix + 1];
        if (en <= st) continue;          /* empty rows are left untouched (:3270) */
        size_t nnz = en - st;
        real_t lam_i = lam, lam_last_i = lam_last;
        t_l1_mult = 1;
        const real_t *wt = (wts != NULL) ? wts + st : NULL;
        if (scale_lam) {                                                       /* :679-723 */
            real_t mult = (real_t)nnz;
            if (wt != NULL) {
                real_t wsum = (wsums != NULL) ? wsums[ix] : (real_t)0;
                if (wsum <= 0) { wsum = 0; for (size_t e = 0; e < nnz; e++) wsum += wt[e]; }   /* :696-707 */
                mult = wsum;
            }
            lam_i *= mult;
            if (!scale_bias_const) lam_last_i *= mult;
            t_l1_mult = mult;
        }
        real_t *buf = bufs + szbuf * (size_t)omp_get_thread_num();
        real_t *a = A + (size_t)ix * lda;
        const bool row_cg = use_cg && !(cf != NULL && cf[ix] == 1);
        int_t steps_i = max_cg_steps;
        if (row_cg && cf != NULL && cf[ix] == 2) {                             /* Case 1's fix-up loop: from zero, k steps (:2953-2985) */
            memset(a, 0, (size_t)k * sizeof(real_t));
            steps_i = k;
        }
        if (row_cg && !precondition_cg)
            explicit_cg_row(a, k, B, ldb, Xcsr + st, Xcsr_i + st, nnz, wt, lam_i, lam_last_i, steps_i, buf);
        else if (row_cg)
            explicit_pcg_row(a, k, B, ldb, Xcsr + st, Xcsr_i + st, nnz, wt, lam_i, lam_last_i, steps_i, buf);
        else
            explicit_chol_row(a, k, B, ldb, Xcsr + st, Xcsr_i + st, nnz, wt, lam_i, lam_last_i, buf);
    }
    free(bufs);
}

/* ------------------------------------------------------------------------------------------ */
void oracle_optimizeA_dense_full(real_t *A, size_t lda, const real_t *B, size_t ldb,
                                 int_t m, int_t n, int_t k,
                                 const real_t *Xfull, size_t ldX, bool do_B,
                                 real_t lam, real_t lam_last, bool scale_lam, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    real_t *BtB = (real_t *)malloc((size_t)k * k * sizeof(real_t));
    oracle_gram(B, ldb, n, k, BtB, nthreads);                                  /* :2824 */
    real_t dl = scale_lam ? lam * (real_t)n : lam;                             /* :2832-2833 */
    real_t dll = scale_lam ? lam_last * (real_t)n : lam_last;
    for (int_t i = 0; i < k - 1; i++) BtB[(size_t)i * k + i] += dl;
    BtB[(size_t)(k - 1) * k + (k - 1)] += dll;
    const real_t l1_rows = g_l1 * (scale_lam ? (real_t)n : (real_t)1);          /* :2882-2883, :2896-2897 */
    const bool nonneg = g_nonneg || l1_rows != 0;                               /* solve_*_batch, :2876-2902: the matrix is shared, not factored */
    int bad = nonneg ? 0 : chol_upper_(k, BtB, k);
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int_t i = 0; i < m; i++) {
        real_t *a = A + (size_t)i * lda;
        /* gemm, :2847-2855 -- accumulate in double (blocked BLAS-3 order is unspecified) */
        double acc[512];
        double *accp = (k <= 512) ? acc : (double *)calloc((size_t)k, sizeof(double));
        for (int_t c = 0; c < k; c++) accp[c] = 0;
        for (int_t j = 0; j < n; j++) {
            double x = do_B ? Xfull[(size_t)j * ldX + i] : Xfull[(size_t)i * ldX + j];
            const real_t *b = B + (size_t)j * ldb;
            for (int_t c = 0; c < k; c++) accp[c] += x * (double)b[c];
        }
        for (int_t c = 0; c < k; c++) a[c] = (real_t)accp[c];
        if (accp != acc) free(accp);
        if (nonneg) {
            real_t *Mc = (real_t *)malloc((size_t)k * k * sizeof(real_t));
            memcpy(Mc, BtB, (size_t)k * k * sizeof(real_t));
            if (g_nonneg) {
                if (l1_rows != 0) for (int_t c = 0; c < k; c++) a[c] -= l1_rows;
                solve_nonneg_(k, Mc, k, a, g_max_cd);
            } else solve_elasticnet_(k, Mc, k, a, l1_rows, l1_rows, g_max_cd);
            free(Mc);
        }
        else if (!bad) chol_solve_upper_(k, BtB, k, a);                        /* :2872 posv */
        else for (int_t c = 0; c < k; c++) a[c] = NAN;
    }
    free(BtB);
}

/* ------------------------------------------------------------------------------------------ */
/* Missing-as-zero rows on an unweighted sparse matrix (optimizeA Case 3, common.c:3116-3205): one matrix
 * B^T B + lam (x n under scale_lam) for every row, right-hand sides X B (tgemm_sp_dense).  Xcsr == NULL: unit values
 * (the binary indicator the Ai / Bi updates of add_implicit_features run on, collective.c:8448-8534). */
/* bias_BtX [k] (or NULL): the constant every row's right-hand side receives when the main matrix itself is missing-as-zero --
 * minus the sum over ALL rows of B of (their bias + the global mean) x the row (common.c:3152-3157; built by the driver,
 * collective.c:8573-8600, :8756-8787).  Cleared by the call. */
static const real_t *g_naz_bias_BtX = NULL;
void oracle_set_naz_bias_BtX(const real_t *bias_BtX) { g_naz_bias_BtX = bias_BtX; }

void oracle_optimizeA_naz(real_t *A, size_t lda, const real_t *B, size_t ldb, int_t m, int_t n, int_t k,
                          const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                          real_t lam, real_t lam_last, bool scale_lam, int nthreads)
{
    const real_t *bias_BtX = g_naz_bias_BtX;
    g_naz_bias_BtX = NULL;
    if (nthreads < 1) nthreads = 1;
    real_t *BtB = (real_t *)malloc((size_t)k * k * sizeof(real_t));
    oracle_gram(B, ldb, n, k, BtB, nthreads);                                  /* :3128-3131 */
    for (int_t i = 0; i < k - 1; i++) BtB[(size_t)i * k + i] += scale_lam ? lam * (real_t)n : lam;   /* :3137-3138 */
    BtB[(size_t)(k - 1) * k + (k - 1)] += scale_lam ? lam_last * (real_t)n : lam_last;
    const real_t l1_rows = g_l1 * (scale_lam ? (real_t)n : (real_t)1);          /* :3182-3183, :3195-3196 */
    const bool cd = g_nonneg || l1_rows != 0;
    int bad = cd ? 0 : chol_upper_(k, BtB, k);                                 /* :3171-3175 posv */
    #pragma omp parallel for schedule(dynamic) num_threads(nthreads)
    for (int_t ix = 0; ix < m; ix++) {
        real_t *a = A + (size_t)ix * lda;
        memset(a, 0, (size_t)k * sizeof(real_t));                              /* :3139-3144 */
        for (size_t jx = Xcsr_p[ix]; jx < Xcsr_p[(size_t)ix + 1]; jx++)       /* :3145-3151 */
            axpy_(k, Xcsr ? Xcsr[jx] : (real_t)1, B + (size_t)Xcsr_i[jx] * ldb, a);
        if (bias_BtX != NULL) axpy_(k, (real_t)1, bias_BtX, a);               /* :3152-3157 (multiplier_bias_BtX = 1) */
        if (cd) {
            real_t *Mc = (real_t *)malloc((size_t)k * k * sizeof(real_t));
            memcpy(Mc, BtB, (size_t)k * k * sizeof(real_t));
            if (g_nonneg) {
                if (l1_rows != 0) for (int_t c = 0; c < k; c++) a[c] -= l1_rows;
                solve_nonneg_(k, Mc, k, a, g_max_cd);
            } else solve_elasticnet_(k, Mc, k, a, l1_rows, l1_rows, g_max_cd);
            free(Mc);
        }
        else if (!bad) chol_solve_upper_(k, BtB, k, a);
        else for (int_t c = 0; c < k; c++) a[c] = NAN;
    }
    free(BtB);
}

/* ------------------------------------------------------------------------------------------ */
/* Missing-as-zero rows WITH observation weights (optimizeA Case 4 with NA_as_zero && weight, common.c:3209-3302; per row
 * factors_closed_form :631-1095 with its NA_as_zero + weight branches): absent entries count as zeros of weight one, so every
 * row's system is the shared  B^T B  plus the correction of its present entries,
 *     M_i = B^T B + sum_j (w_j - 1) B_j B_j^T + diag(lam_i .. lam_i, lam_last_i)
 *     rhs_i = sum_j [ w_j x_j - (w_j - 1) (bias_X_glob + bias_X[j]) ] B_j + bias_BtX
 * (closed form :846-907; CG factors_explicit_cg_NA_as_zero_weighted :1293-1441, its k < n branch; PCG :1443-1613), with
 * lam_i = lam x (sum of the row's weights + n - nnz_i) under scale_lam (the driver's wsumA / wsumB, collective.c:7991-8022, or
 * the row's own sum, common.c:696-712).  Rows are solved when they have entries OR when bias_BtX is given (:3270-3271).
 * bias_BtX [k]: minus the sum over all rows of B of (their bias + the mean) x the row; bias_X [n]: the opposing biases;
 * bias_X_glob: the mean (driver: collective.c:8573-8600, :8756-8787, :8701-8706, :8872-8877). */
static void naz_weighted_cg_row(real_t *a, int_t k, const real_t *B, size_t ldb, const real_t *Xa, const int_t *ixB, size_t nnz,
                                const real_t *wt, const real_t *BtB, const real_t *bias_BtX, const real_t *bias_X, real_t bias_X_glob,
                                real_t lam, real_t lam_last, int_t max_cg_steps, bool precond, real_t *buf)
{
    real_t *Ap = buf, *p = Ap + k, *r = p + k, *z = r + k, *PC = z + k;
    if (precond) {                                                             /* :1486-1497 */
        memset(PC, 0, (size_t)k * sizeof(real_t));
        for (size_t ix = 0; ix < nnz; ix++) {
            const real_t *b = B + (size_t)ixB[ix] * ldb;
            const real_t w_this = wt[ix] - (real_t)1;
            for (int_t i = 0; i < k; i++) PC[i] += w_this * b[i] * b[i];
        }
        for (int_t i = 0; i < k; i++) PC[i] += BtB[(size_t)i * k + i];
        for (int_t i = 0; i < k; i++) PC[i] += lam;
        if (lam != lam_last) PC[k - 1] += (lam_last - lam);
        for (int_t i = 0; i < k; i++) PC[i] = (real_t)1 / PC[i];
    }
    symv_(k, (real_t)-1, BtB, k, a, r);                                        /* :1321-1324 */
    for (size_t ix = 0; ix < nnz; ix++) {                                      /* :1325-1338 */
        const real_t *b = B + (size_t)ixB[ix] * ldb;
        const real_t coef = dot_(k, b, a);
        axpy_(k, -(wt[ix] - (real_t)1.) * (coef + bias_X_glob + ((bias_X == NULL) ? (real_t)0 : bias_X[ixB[ix]])) + (wt[ix] * Xa[ix]), b, r);
    }
    if (bias_BtX != NULL) axpy_(k, (real_t)1, bias_BtX, r);                    /* :1368-1371 (multiplier_bias_BtX = 1) */
    axpy_(k, -lam, a, r);
    if (lam != lam_last) r[k - 1] -= (lam_last - lam) * a[k - 1];
    real_t r_old;
    if (precond) {
        for (int_t i = 0; i < k; i++) z[i] = r[i] * PC[i];
        r_old = dot_(k, z, r);
        memcpy(p, z, (size_t)k * sizeof(real_t));
    } else {
        memcpy(p, r, (size_t)k * sizeof(real_t));
        r_old = dot_(k, r, r);
        if (r_old <= (real_t)1e-12) return;                                    /* :1384 */
    }
    for (int_t step = 0; step < max_cg_steps; step++) {
        symv_(k, (real_t)1, BtB, k, p, Ap);                                    /* :1391-1394 */
        for (size_t ix = 0; ix < nnz; ix++) {                                  /* :1395-1403 */
            const real_t *b = B + (size_t)ixB[ix] * ldb;
            const real_t coef = dot_(k, b, p);
            axpy_(k, (wt[ix] - (real_t)1.) * coef, b, Ap);
        }
        axpy_(k, lam, p, Ap);
        if (lam != lam_last) Ap[k - 1] += (lam_last - lam) * p[k - 1];
        const real_t alpha = r_old / dot_(k, p, Ap);
        axpy_(k, alpha, p, a);
        axpy_(k, -alpha, Ap, r);
        real_t r_new;
        if (precond) {
            for (int_t i = 0; i < k; i++) z[i] = r[i] * PC[i];
            r_new = dot_(k, z, r);
            const real_t ratio = r_new / r_old;
            for (int_t i = 0; i < k; i++) p[i] = p[i] * ratio + z[i];
        } else {
            r_new = dot_(k, r, r);
            if (r_new <= (real_t)1e-8) break;                                  /* :1432 */
            const real_t ratio = r_new / r_old;
            for (int_t i = 0; i < k; i++) p[i] = p[i] * ratio + r[i];
        }
        r_old = r_new;
    }
}

void oracle_optimizeA_naz_weighted(real_t *A, size_t lda, const real_t *B, size_t ldb, int_t m, int_t n, int_t k,
                                   const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr, const real_t *weight,
                                   const real_t *wsum, const real_t *bias_BtX, const real_t *bias_X, real_t bias_X_glob,
                                   real_t lam, real_t lam_last, bool scale_lam, bool scale_bias_const,
                                   bool use_cg, bool precondition_cg, int_t max_cg_steps, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (g_nonneg) use_cg = false;                                              /* common.c:725 */
    real_t *BtB = (real_t *)malloc((size_t)k * k * sizeof(real_t));
    oracle_gram(B, ldb, n, k, BtB, nthreads);                                  /* :3233-3236, no diagonal: it is added per row below */
    const real_t l1_base = g_l1;
    #pragma omp parallel for schedule(dynamic) num_threads(nthreads)
    for (int_t ix = 0; ix < m; ix++) {
        const size_t st = Xcsr_p[ix], nnz = Xcsr_p[(size_t)ix + 1] - st;
        if (!(nnz > 0 || bias_BtX != NULL)) continue;                          /* :3270-3271: left as it is */
        real_t *a = A + (size_t)ix * lda;
        real_t lam_i = lam, lam_last_i = lam_last, l1_i = l1_base;
        if (scale_lam) {                                                       /* :679-723 */
            real_t ws = (wsum != NULL) ? wsum[ix] : (real_t)0;
            if (ws <= 0) {
                ws = 0;
                for (size_t jx = 0; jx < nnz; jx++) ws += weight[st + jx];
                ws += (real_t)(n - (int_t)nnz);
            }
            if (fabs_t(ws) < EPSILON_T && bias_BtX == NULL) { memset(a, 0, (size_t)k * sizeof(real_t)); continue; }
            lam_i *= ws; l1_i *= ws;
            if (!scale_bias_const) lam_last_i *= ws;
        }
        real_t *buf = (real_t *)malloc(((size_t)k * k + 6 * (size_t)k) * sizeof(real_t));
        if (use_cg) {
            naz_weighted_cg_row(a, k, B, ldb, Xcsr + st, Xcsr_i + st, nnz, weight + st, BtB, bias_BtX, bias_X, bias_X_glob,
                                lam_i, lam_last_i, max_cg_steps, precondition_cg, buf);
        } else {                                                               /* :846-907, then :1060-1093 */
            real_t *M = buf;
            memset(a, 0, (size_t)k * sizeof(real_t));
            memset(M, 0, (size_t)k * k * sizeof(real_t));
            for (size_t jx = 0; jx < nnz; jx++) {
                const real_t *b = B + (size_t)Xcsr_i[st + jx] * ldb;
                const real_t w = weight[st + jx];
                syr_upper_(k, w - (real_t)1., b, M, k);
                axpy_(k, (w * Xcsr[st + jx]) - (w - (real_t)1.) * (bias_X_glob + ((bias_X == NULL) ? (real_t)0 : bias_X[Xcsr_i[st + jx]])), b, a);
            }
            for (int_t i = 0; i < k; i++)
                for (int_t j = i; j < k; j++) M[(size_t)i * k + j] += BtB[(size_t)i * k + j];
            for (int_t i = 0; i < k - 1; i++) M[(size_t)i * k + i] += lam_i;
            M[(size_t)(k - 1) * k + (k - 1)] += lam_last_i;
            if (bias_BtX != NULL) axpy_(k, (real_t)1, bias_BtX, a);
            if (g_nonneg) {
                for (int_t i = 0; i < k; i++) for (int_t j = 0; j < i; j++) M[(size_t)i * k + j] = M[(size_t)j * k + i];
                if (l1_i != 0) for (int_t c = 0; c < k; c++) a[c] -= l1_i;
                solve_nonneg_(k, M, k, a, g_max_cd);
            } else if (l1_i != 0) {
                for (int_t i = 0; i < k; i++) for (int_t j = 0; j < i; j++) M[(size_t)i * k + j] = M[(size_t)j * k + i];
                solve_elasticnet_(k, M, k, a, l1_i, l1_i, g_max_cd);
            } else {
                const int bad = chol_upper_(k, M, k);
                if (!bad) chol_solve_upper_(k, M, k, a);
                else for (int_t c = 0; c < k; c++) a[c] = NAN;
            }
        }
        free(buf);
    }
    free(BtB);
}

/* NA_as_zero_X of the collective_chol_impl call that follows (cleared by the call; used for the model with implicit features and no
 * side information, which the reference runs through optimizeA_collective's general branch, collective.c:8612 / :8783 ->
 * collective_closed_form_block :1534-1846 with prefer_BtB): the lower-right block is the whole B^T B (:1631-1640), the right-hand
 * side gains bias_BtX (:1774-1775), the lambda multiplier is n (:1300-1301), rows without entries are solved when bias_BtX exists
 * (:1258-1268). */
static bool g_cc_naz = false;
static const real_t *g_cc_btx = NULL;
static void collective_chol_impl(real_t *A, size_t lda, const real_t *B, size_t ldb,
                                      const real_t *C,
                                      int_t m, int_t m_u, int_t n, int_t p,
                                      int_t k, int_t k_main, int_t k_user, int_t k_item,
                                      const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                      const real_t *U,
                                      real_t lam, real_t w_user, real_t lam_last,
                                      bool scale_lam, bool scale_lam_sideinfo,
                                      int nthreads,
                                      const real_t *Bi, int_t k_main_i, real_t w_implicit);
void oracle_optimizeA_collective_chol(real_t *A, size_t lda, const real_t *B, size_t ldb,
                                      const real_t *C,
                                      int_t m, int_t m_u, int_t n, int_t p,
                                      int_t k, int_t k_main, int_t k_user, int_t k_item,
                                      const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                      const real_t *U,
                                      real_t lam, real_t w_user, real_t lam_last,
                                      bool scale_lam, bool scale_lam_sideinfo,
                                      int nthreads)
{
    collective_chol_impl(A, lda, B, ldb, C, m, m_u, n, p, k, k_main, k_user, k_item, Xcsr_p, Xcsr_i, Xcsr, U,
                         lam, w_user, lam_last, scale_lam, scale_lam_sideinfo, nthreads, NULL, 0, (real_t)1);
}

/* Bi != NULL: the implicit-features term -- w_i Bi^T Bi on the X block of every solved row (collective.c:5689-5695,
 * :1704-1707) and w_i sum_{j observed} Bi_j on its right-hand side (:1757-1771); Bi is [n, k + k_main_i]. */
static void collective_chol_impl(real_t *A, size_t lda, const real_t *B, size_t ldb,
                                      const real_t *C,
                                      int_t m, int_t m_u, int_t n, int_t p,
                                      int_t k, int_t k_main, int_t k_user, int_t k_item,
                                      const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                      const real_t *U,
                                      real_t lam, real_t w_user, real_t lam_last,
                                      bool scale_lam, bool scale_lam_sideinfo,
                                      int nthreads,
                                      const real_t *Bi, int_t k_main_i, real_t w_implicit)
{
    if (nthreads < 1) nthreads = 1;
    const bool naz = g_cc_naz;
    const real_t *btx = naz ? g_cc_btx : NULL;
    g_cc_naz = false; g_cc_btx = NULL;
    real_t *BtBn = NULL;
    if (naz) {
        BtBn = (real_t *)malloc((size_t)(k + k_main) * (k + k_main) * sizeof(real_t));
        oracle_gram(B + k_item, ldb, n, k + k_main, BtBn, nthreads);
    }
    const int_t kbi = k + k_main_i;
    real_t *BiTBi = NULL;
    if (Bi != NULL) {
        BiTBi = (real_t *)malloc((size_t)kbi * kbi * sizeof(real_t));
        oracle_gram(Bi, (size_t)kbi, n, kbi, BiTBi, nthreads);
        for (size_t i = 0; i < (size_t)kbi * kbi; i++) BiTBi[i] *= w_implicit;
    }
    int_t k_totA = k_user + k + k_main, k_totC = k_user + k, kb = k + k_main;
    /* collective.c:4817-4822: A := 0 when not CG */
    for (size_t ix = 0; ix < (size_t)m * lda - (lda - (size_t)k_totA); ix++) A[ix] = 0;
    /* collective.c:5658-5668 + :6292-6296 : CtCw = w_user * C^T C */
    real_t *CtCw = (real_t *)calloc((size_t)k_totC * k_totC + 1, sizeof(real_t));
    if (p > 0 && U != NULL) {
        oracle_gram(C, (size_t)k_totC, p, k_totC, CtCw, nthreads);
        for (size_t i = 0; i < (size_t)k_totC * k_totC; i++) CtCw[i] *= w_user;
    }
    /* collective.c:5768-5773 : A[:m_u, :k_totC] = w_user * U C  (add_U = false) */
    if (p > 0 && U != NULL) {
        #pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int_t i = 0; i < m_u; i++) {
            real_t *a = A + (size_t)i * lda;
            for (int_t c = 0; c < k_totC; c++) {
                double s = 0;
                for (int_t j = 0; j < p; j++) s += (double)U[(size_t)i * p + j] * (double)C[(size_t)j * k_totC + c];
                a[c] = (real_t)((double)w_user * s);
            }
        }
    }
    size_t szbuf = (size_t)k_totA * k_totA;
    real_t *bufs = (real_t *)malloc(szbuf * (size_t)nthreads * sizeof(real_t));
    #pragma omp parallel for schedule(dynamic) num_threads(nthreads)
    for (int_t ix = 0; ix < m; ix++) {                                         /* :5865-5965 */
        size_t st = Xcsr_p[ix], en = Xcsr_p[(size_t)ix + 1];
        size_t nnz = en - st;
        bool has_u = (U != NULL && p > 0 && ix < m_u);
        real_t *a = A + (size_t)ix * lda;
        if (nnz == 0 && !has_u && !(naz && btx != NULL)) { memset(a, 0, (size_t)k_totA * sizeof(real_t)); continue; } /* :1258-1268 */
        real_t lam_i = lam, lam_last_i = lam_last;
        if (scale_lam || scale_lam_sideinfo) {                                 /* :1285-1355 */
            real_t mult = (real_t)nnz;
            if (nnz == 0) mult = 1;                                            /* :1332-1336 */
            if (naz) mult = (real_t)n;                                         /* :1300-1301 */
            if (scale_lam_sideinfo && has_u) mult += (real_t)p;                /* :1338-1346 */
            lam_i *= mult;
            if (has_u || !g_scale_bias_const) lam_last_i *= mult;             /* rows >= m_u: plain optimizeA (:4832-4965) */
            t_l1_mult = mult;
        } else t_l1_mult = 1;
        real_t *M = bufs + szbuf * (size_t)omp_get_thread_num();
        memset(M, 0, szbuf * sizeof(real_t));                                  /* :1536 */
        if (has_u)                                                             /* :1566-1571 */
            for (int_t i = 0; i < k_totC; i++)
                for (int_t j = 0; j < k_totC; j++)
                    M[(size_t)i * k_totA + j] = CtCw[(size_t)i * k_totC + j];
        real_t *Mlr = M + (size_t)k_user + (size_t)k_user * k_totA;            /* :1360 */
        if (naz) {                                                             /* :1631-1640 */
            for (int_t i = 0; i < kb; i++)
                for (int_t j = i; j < kb; j++) Mlr[(size_t)i * k_totA + j] += BtBn[(size_t)i * kb + j];
        } else
        for (size_t jx = st; jx < en; jx++)                                    /* :1694-1699 */
            syr_upper_(kb, (real_t)1, B + (size_t)k_item + (size_t)Xcsr_i[jx] * ldb, Mlr, k_totA);
        /* :1542-1543 tail already zero; :1738-1742 rhs += B^T x */
        for (size_t jx = st; jx < en; jx++)
            axpy_(kb, Xcsr[jx], B + (size_t)k_item + (size_t)Xcsr_i[jx] * ldb, a + k_user);
        if (btx != NULL) axpy_(kb, (real_t)1, btx, a + k_user);               /* :1774-1775 */
        if (Bi != NULL) {
            for (int_t i = 0; i < kbi; i++)                                    /* :1704-1707 */
                for (int_t j = i; j < kbi; j++) Mlr[(size_t)i * k_totA + j] += BiTBi[(size_t)i * kbi + j];
            for (size_t jx = st; jx < en; jx++)                                /* :1764-1770 */
                axpy_(kbi, w_implicit, Bi + (size_t)Xcsr_i[jx] * kbi, a + k_user);
        }
        for (int_t i = 0; i < k_totA - 1; i++) M[(size_t)i * k_totA + i] += lam_i; /* :1819 */
        M[(size_t)(k_totA - 1) * k_totA + (k_totA - 1)] += lam_last_i;
        solve_sym_(k_totA, M, k_totA, a);
    }
    free(bufs);
    free(CtCw);
    free(BiTBi);
    free(BtBn);
}

/* Block CG on the collective system, dense full U without NaN (prefer_CtC branches):
 * collective_block_cg (collective.c:2134-2903; called from collective_closed_form_block :1223-1533 with the
 * row's lambda scaling :1285-1355) and collective_block_cg_implicit (:2905-3303).  Unknowns [k_user, k_totA)
 * couple to X, unknowns [0, k_user+k) to U.  Rows >= m_u are plain rows (optimizeA / optimizeA_implicit on the
 * X block, collective.c:4832-5101, :6037-6054): their first k_user unknowns are not touched. */
/* U dense [m_u, p] (prefer_CtC branches), or U == NULL and the row's attributes as CSR (u_vec_sp branches:
 * residual :2609-2621, products :2847-2860, preconditioner :2292-2298 -- each present attribute j contributes
 * w (u_j - C_j.a) C_j,  w (C_j.p) C_j  and  C_j^2 (unweighted); lambda counts them under scale_lam_sideinfo). */
/* implicit-features term of the block CG (collective.c:2301-2304, :2624-2643, :2862-2868), set by the explicit fit around
 * its A / B updates: Bi [n, ki] gathered at the observed positions (unit values), Bi^T Bi (unweighted) on the X block */
static const real_t *g_cg_Bi = NULL;
static int_t g_cg_ki = 0;
static real_t g_cg_wimp = 0;
static void collective_cg_impl(real_t *A, size_t lda, const real_t *B, size_t ldb, const real_t *C,
                               int_t m, int_t m_u, int_t n, int_t p,
                               int_t k, int_t k_main, int_t k_user, int_t k_item,
                               const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                               const real_t *U, const size_t *Ucsr_p, const int_t *Ucsr_i, const real_t *Ucsr,
                               real_t lam, real_t w_user, real_t lam_last,
                               bool scale_lam, bool scale_lam_sideinfo, bool implicit,
                               int_t max_cg_steps, bool precondition_cg, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    const bool sparse_u = (U == NULL && Ucsr_p != NULL);
    const int_t kt = k_user + k + k_main, kc = k_user + k, kb = k + k_main;
    real_t *CtC = (real_t *)calloc((size_t)kc * kc + 1, sizeof(real_t));
    if (!sparse_u) oracle_gram(C, (size_t)kc, p, kc, CtC, nthreads);             /* unweighted, collective.c:5855-5860 */
    real_t *BtB = NULL;
    if (implicit) {                                                              /* :6056-6061, no lambda with CG */
        BtB = (real_t *)calloc((size_t)kb * kb + 1, sizeof(real_t));
        oracle_gram(B + k_item, ldb, n, kb, BtB, nthreads);
    }
    const real_t *Bi = implicit ? NULL : g_cg_Bi;
    const int_t ki = g_cg_ki;
    const real_t wimp = g_cg_wimp;
    real_t *BiTBi = NULL;
    if (Bi != NULL) {
        BiTBi = (real_t *)calloc((size_t)ki * ki + 1, sizeof(real_t));
        oracle_gram(Bi, (size_t)ki, n, ki, BiTBi, nthreads);
    }
    #pragma omp parallel for schedule(dynamic) num_threads(nthreads)
    for (int_t ix = 0; ix < m; ix++) {
        const size_t st = Xcsr_p[ix], en = Xcsr_p[(size_t)ix + 1];
        const size_t nnz = en - st;
        const bool has_u = ix < m_u;
        const size_t us = (sparse_u && has_u) ? Ucsr_p[ix] : 0, ue = (sparse_u && has_u) ? Ucsr_p[(size_t)ix + 1] : 0;
        real_t *a = A + (size_t)ix * lda;
        if (nnz == 0 && !has_u) {                                                /* plain rows without entries stay untouched */
            if (Bi != NULL) memset(a, 0, (size_t)kt * sizeof(real_t));           /* ... unless optimizeA_collective ran them (:1258-1268) */
            continue;
        }
        if (nnz == 0 && sparse_u && ue == us) {                                  /* :1258-1268: neither observations nor attributes */
            memset(a, 0, (size_t)kt * sizeof(real_t));
            continue;
        }
        const int_t lo = has_u ? 0 : k_user;
        real_t lam_i = lam, lam_last_i = lam_last;
        if (!implicit) {
            if (has_u) {
                if (scale_lam || scale_lam_sideinfo) {                           /* :1285-1355 */
                    real_t mult = (nnz > 0) ? (real_t)nnz : (real_t)1;
                    if (scale_lam_sideinfo) mult += sparse_u ? (real_t)(ue - us) : (real_t)p;
                    lam_i *= mult; lam_last_i *= mult;
                }
            } else if (scale_lam) { lam_i *= (real_t)nnz; if (!g_scale_bias_const) lam_last_i *= (real_t)nnz; }   /* common.c:679-723 */
        }
        real_t r[512], pp[512], Ap[512], z[512], PC[512], ctu[512];
        /* Ap-type product: out = [BtB v_x] + sum_j w_j(B_j.v_x) B_j + w CtC v_u  (no lambda) */
        #define BLOCK_MATVEC(v, out, resid)                                                              \
            do {                                                                                          \
                for (int_t f = 0; f < kt; f++) out[f] = 0;                                                \
                if (implicit)                                                                             \
                    for (int_t i = 0; i < kb; i++) {                                                      \
                        double sacc = 0;                                                                  \
                        for (int_t j = 0; j < kb; j++) sacc += (double)BtB[(size_t)i * kb + j] * (double)v[k_user + j]; \
                        out[k_user + i] = (real_t)((resid) ? -sacc : sacc);                               \
                    }                                                                                     \
                for (size_t jx = st; jx < en; jx++) {                                                     \
                    const real_t *b = B + (size_t)k_item + (size_t)Xcsr_i[jx] * ldb;                      \
                    real_t coef = 0;                                                                      \
                    for (int_t f = 0; f < kb; f++) coef += b[f] * v[k_user + f];                          \
                    real_t wgt;                                                                           \
                    if (implicit) wgt = (resid) ? (-(coef - (real_t)1) * Xcsr[jx] - coef) : (coef * (Xcsr[jx] - (real_t)1) + coef); \
                    else          wgt = (resid) ? (-coef + Xcsr[jx]) : coef;                              \
                    for (int_t f = 0; f < kb; f++) out[k_user + f] += wgt * b[f];                         \
                    if (Bi != NULL && (resid))                                                            \
                        for (int_t f = 0; f < ki; f++) out[k_user + f] += wimp * Bi[(size_t)Xcsr_i[jx] * ki + f]; \
                }                                                                                         \
                if (Bi != NULL)                                                                           \
                    for (int_t i = 0; i < ki; i++) {                                                      \
                        double sacc = 0;                                                                  \
                        for (int_t j = 0; j < ki; j++) sacc += (double)BiTBi[(size_t)i * ki + j] * (double)v[k_user + j]; \
                        out[k_user + i] += (resid) ? -wimp * (real_t)sacc : wimp * (real_t)sacc;          \
                    }                                                                                     \
                if (has_u && sparse_u)                                                                    \
                    for (size_t jx = us; jx < ue; jx++) {                                                 \
                        const real_t *cj = C + (size_t)Ucsr_i[jx] * kc;                                   \
                        real_t coef = 0;                                                                  \
                        for (int_t f = 0; f < kc; f++) coef += cj[f] * v[f];                              \
                        const real_t wgt = (resid) ? w_user * (-coef + Ucsr[jx]) : w_user * coef;        \
                        for (int_t f = 0; f < kc; f++) out[f] += wgt * cj[f];                             \
                    }                                                                                     \
                else if (has_u)                                                                           \
                    for (int_t i = 0; i < kc; i++) {                                                      \
                        double sacc = 0;                                                                  \
                        for (int_t j = 0; j < kc; j++) sacc += (double)CtC[(size_t)i * kc + j] * (double)v[j]; \
                        out[i] += (resid) ? w_user * (ctu[i] - (real_t)sacc) : w_user * (real_t)sacc;     \
                    }                                                                                     \
            } while (0)
        if (has_u && !sparse_u)                                                  /* C^T u, :2500-2515 / :3040-3046 */
            for (int_t i = 0; i < kc; i++) {
                double sacc = 0;
                for (int_t l = 0; l < p; l++) sacc += (double)U[(size_t)ix * p + l] * (double)C[(size_t)l * kc + i];
                ctu[i] = (real_t)sacc;
            }
        BLOCK_MATVEC(a, r, 1);
        for (int_t f = 0; f < kt; f++) r[f] -= lam_i * a[f];                     /* diag(lam) */
        if (!implicit && lam_i != lam_last_i) r[kt - 1] -= (lam_last_i - lam_i) * a[kt - 1];
        for (int_t f = 0; f < lo; f++) r[f] = 0;
        real_t r_old, r_new;
        if (precondition_cg) {
            for (int_t f = 0; f < kt; f++) PC[f] = 0;
            for (size_t jx = st; jx < en; jx++) {
                const real_t *b = B + (size_t)k_item + (size_t)Xcsr_i[jx] * ldb;
                for (int_t f = 0; f < kb; f++) PC[k_user + f] += (implicit ? Xcsr[jx] : (real_t)1) * b[f] * b[f];
            }
            if (has_u && sparse_u)
                for (size_t jx = us; jx < ue; jx++) {
                    const real_t *cj = C + (size_t)Ucsr_i[jx] * kc;
                    for (int_t f = 0; f < kc; f++) PC[f] += cj[f] * cj[f];               /* unweighted, :2292-2298 */
                }
            else if (has_u) for (int_t f = 0; f < kc; f++) PC[f] += CtC[(size_t)f * kc + f];   /* unweighted, :2281-2286 */
            if (Bi != NULL) for (int_t f = 0; f < ki; f++) PC[k_user + f] += BiTBi[(size_t)f * ki + f];   /* unweighted, :2301-2304 */
            if (implicit) for (int_t f = 0; f < kb; f++) PC[k_user + f] += BtB[(size_t)f * kb + f];
            else {
                for (int_t f = 0; f < kt; f++) PC[f] += lam_i;
                if (lam_i != lam_last_i) PC[kt - 1] += (lam_last_i - lam_i);
            }
            for (int_t f = 0; f < kt; f++) PC[f] = (f >= lo) ? (real_t)1 / PC[f] : 0;
            r_old = 0;
            for (int_t f = 0; f < kt; f++) { z[f] = r[f] * PC[f]; pp[f] = z[f]; r_old += z[f] * r[f]; }
        } else {
            r_old = 0;
            for (int_t f = 0; f < kt; f++) { pp[f] = r[f]; r_old += r[f] * r[f]; }
            if (r_old <= (real_t)1e-12) continue;
        }
        for (int_t step = 0; step < max_cg_steps; step++) {
            BLOCK_MATVEC(pp, Ap, 0);
            for (int_t f = 0; f < kt; f++) Ap[f] += lam_i * pp[f];
            if (!implicit && lam_i != lam_last_i) Ap[kt - 1] += (lam_last_i - lam_i) * pp[kt - 1];
            for (int_t f = 0; f < lo; f++) Ap[f] = 0;
            real_t pAp = 0;
            for (int_t f = 0; f < kt; f++) pAp += pp[f] * Ap[f];
            const real_t alpha = r_old / pAp;
            for (int_t f = lo; f < kt; f++) { a[f] += alpha * pp[f]; r[f] -= alpha * Ap[f]; }
            r_new = 0;
            if (precondition_cg) {
                for (int_t f = 0; f < kt; f++) { z[f] = r[f] * PC[f]; r_new += z[f] * r[f]; }
                for (int_t f = 0; f < kt; f++) pp[f] = pp[f] * (r_new / r_old) + z[f];
            } else {
                for (int_t f = 0; f < kt; f++) r_new += r[f] * r[f];
                if (r_new <= (real_t)1e-8) break;
                for (int_t f = 0; f < kt; f++) pp[f] = pp[f] * (r_new / r_old) + r[f];
            }
            r_old = r_new;
        }
        #undef BLOCK_MATVEC
    }
    free(CtC); free(BtB); free(BiTBi);
}

void oracle_optimizeA_collective_cg(real_t *A, size_t lda, const real_t *B, size_t ldb, const real_t *C,
                                    int_t m, int_t m_u, int_t n, int_t p,
                                    int_t k, int_t k_main, int_t k_user, int_t k_item,
                                    const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                    const real_t *U, real_t lam, real_t w_user, real_t lam_last,
                                    bool scale_lam, bool scale_lam_sideinfo, bool implicit,
                                    int_t max_cg_steps, bool precondition_cg, int nthreads)
{
    collective_cg_impl(A, lda, B, ldb, C, m, m_u, n, p, k, k_main, k_user, k_item, Xcsr_p, Xcsr_i, Xcsr, U, NULL, NULL, NULL,
                       lam, w_user, lam_last, scale_lam, scale_lam_sideinfo, implicit, max_cg_steps, precondition_cg, nthreads);
}

/* the same with sparse side information (U as CSR over m_u rows) */
void oracle_optimizeA_collective_sparse_cg(real_t *A, size_t lda, const real_t *B, size_t ldb, const real_t *C,
                                           int_t m, int_t m_u, int_t n, int_t p,
                                           int_t k, int_t k_main, int_t k_user, int_t k_item,
                                           const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                           const size_t *Ucsr_p, const int_t *Ucsr_i, const real_t *Ucsr,
                                           real_t lam, real_t w_user, real_t lam_last,
                                           bool scale_lam, bool scale_lam_sideinfo, bool implicit,
                                           int_t max_cg_steps, bool precondition_cg, int nthreads)
{
    collective_cg_impl(A, lda, B, ldb, C, m, m_u, n, p, k, k_main, k_user, k_item, Xcsr_p, Xcsr_i, Xcsr, NULL, Ucsr_p, Ucsr_i,
                       Ucsr, lam, w_user, lam_last, scale_lam, scale_lam_sideinfo, implicit, max_cg_steps, precondition_cg,
                       nthreads);
}

/* optimizeA_collective_implicit (collective.c:5971-6244) + collective_closed_form_block_implicit
 * (:1849-2131): implicit-feedback X, dense full U without NaN, Cholesky.  m_u <= m: rows >= m_u go
 * through optimizeA_implicit on the k+k_main block at column offset k_user (:6037-6054), their k_user
 * coordinates stay 0 (A was zeroed, :6018-6019). */
void oracle_optimizeA_collective_implicit_chol(real_t *A, size_t lda, const real_t *B, size_t ldb,
                                               const real_t *C,
                                               int_t m, int_t m_u, int_t n, int_t p,
                                               int_t k, int_t k_main, int_t k_user, int_t k_item,
                                               const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                               const real_t *U, real_t lam, real_t w_user, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    int_t k_totA = k_user + k + k_main, k_totC = k_user + k, kb = k + k_main;
    for (int_t i = 0; i < m; i++) memset(A + (size_t)i * lda, 0, (size_t)k_totA * sizeof(real_t));   /* :6018-6019 */
    /* BtB + lam I  (:6056-6061) */
    real_t *BtB = (real_t *)calloc((size_t)kb * kb + 1, sizeof(real_t));
    oracle_gram(B + k_item, ldb, n, kb, BtB, nthreads);
    for (int_t i = 0; i < kb; i++) BtB[(size_t)i * kb + i] += lam;
    /* BeTBe: LR = BtB + lam I, first k_user diagonal entries lam, UL += w C^T C  (:6121-6160) */
    real_t *BeTBe = (real_t *)calloc((size_t)k_totA * k_totA + 1, sizeof(real_t));
    for (int_t i = 0; i < kb; i++)
        for (int_t j = 0; j < kb; j++) BeTBe[(size_t)(k_user + i) * k_totA + (k_user + j)] = BtB[(size_t)i * kb + j];
    for (int_t i = 0; i < k_user; i++) BeTBe[(size_t)i * k_totA + i] += lam;
    real_t *CtC = (real_t *)calloc((size_t)k_totC * k_totC + 1, sizeof(real_t));
    oracle_gram(C, (size_t)k_totC, p, k_totC, CtC, nthreads);
    for (int_t i = 0; i < k_totC; i++)
        for (int_t j = 0; j < k_totC; j++) BeTBe[(size_t)i * k_totA + j] += w_user * CtC[(size_t)i * k_totC + j];
    /* A[:m_u, :k_totC] = w U C  (:6163-6168) */
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int_t i = 0; i < m_u; i++) {
        real_t *a = A + (size_t)i * lda;
        for (int_t c = 0; c < k_totC; c++) {
            double s = 0;
            for (int_t j = 0; j < p; j++) s += (double)U[(size_t)i * p + j] * (double)C[(size_t)j * k_totC + c];
            a[c] = (real_t)((double)w_user * s);
        }
    }
    size_t szbuf = (size_t)k_totA * k_totA;
    real_t *bufs = (real_t *)malloc(szbuf * (size_t)nthreads * sizeof(real_t));
    #pragma omp parallel for schedule(dynamic) num_threads(nthreads)
    for (int_t ix = 0; ix < m; ix++) {
        t_l1_mult = 1;
        size_t st = Xcsr_p[ix], en = Xcsr_p[(size_t)ix + 1];
        real_t *a = A + (size_t)ix * lda;
        real_t *M = bufs + szbuf * (size_t)omp_get_thread_num();
        if (ix < m_u) {                                                        /* :1849-2131, few_NAs branch */
            memcpy(M, BeTBe, szbuf * sizeof(real_t));
            for (size_t jx = st; jx < en; jx++)                                /* :2097-2101 */
                axpy_(kb, Xcsr[jx] + (real_t)1, B + (size_t)k_item + (size_t)Xcsr_i[jx] * ldb, a + k_user);
            real_t *Mlr = M + (size_t)k_user + (size_t)k_user * k_totA;
            for (size_t jx = st; jx < en; jx++)                                /* :2103-2108 */
                syr_upper_(kb, Xcsr[jx], B + (size_t)k_item + (size_t)Xcsr_i[jx] * ldb, Mlr, k_totA);
            solve_sym_(k_totA, M, k_totA, a);
        } else if (en > st) {                                                  /* optimizeA_implicit, common.c:2063-2126 */
            for (int_t i = 0; i < kb; i++) memcpy(M + (size_t)i * kb, BtB + (size_t)i * kb, (size_t)kb * sizeof(real_t));
            for (size_t jx = st; jx < en; jx++)
                axpy_(kb, Xcsr[jx] + (real_t)1, B + (size_t)k_item + (size_t)Xcsr_i[jx] * ldb, a + k_user);
            for (size_t jx = st; jx < en; jx++)
                syr_upper_(kb, Xcsr[jx], B + (size_t)k_item + (size_t)Xcsr_i[jx] * ldb, M, kb);
            solve_sym_(kb, M, kb, a + k_user);
        }
    }
    free(bufs); free(BtB); free(BeTBe); free(CtC);
}

/* ------------------------------------------------------------------------------------------ */
real_t oracle_calc_mean_and_center(real_t *X, size_t nnz, int nthreads)
{
    double xsum = 0;
    real_t glob_mean;
    if (nthreads >= 8) {                                                       /* common.c:3497-3505 */
        #pragma omp parallel for schedule(static) num_threads(nthreads) reduction(+:xsum)
        for (size_t ix = 0; ix < nnz; ix++) xsum += X[ix];
        glob_mean = (real_t)(xsum / (double)nnz);
    } else {                                                                   /* :3510-3512 */
        size_t cnt = 0;
        for (size_t ix = 0; ix < nnz; ix++) xsum += (X[ix] - xsum) / (double)(++cnt);
        glob_mean = (real_t)xsum;
    }
    if (fabs_t(glob_mean) < sqrt_t(EPSILON_T)) glob_mean = 0;                  /* :3603 */
    if (glob_mean != 0)
        for (size_t ix = 0; ix < nnz; ix++) X[ix] -= glob_mean;                /* :3642-3643 */
    return glob_mean;
}

/* weighted mean of the entries: running form below 8 threads (common.c:3574-3584); with 8 threads or more the reference divides
 * the UNWEIGHTED sum of the entries by the sum of the weights (:3561-3571) -- not a mean, but what a caller with nthreads >= 8
 * receives, so it is restated as it stands (sums in entry order; the reference's OpenMP reduction adds per-thread partial
 * sums in an unspecified order: equal to rounding); then centring as above */
real_t oracle_calc_mean_and_center_weighted(real_t *X, const real_t *weight, size_t nnz, int nthreads)
{
    double xsum = 0, wsum = DBL_EPSILON;
    real_t glob_mean;
    if (nthreads >= 8) {                                                       /* common.c:3561-3571 */
        wsum = 0;
        for (size_t ix = 0; ix < nnz; ix++) { xsum += X[ix]; wsum += weight[ix]; }
        glob_mean = (real_t)(xsum / wsum);
    } else {
        for (size_t ix = 0; ix < nnz; ix++)
            xsum += ((X[ix] - xsum) * weight[ix]) / (wsum += weight[ix]);
        glob_mean = (real_t)xsum;
    }
    if (g_nn_AB) glob_mean = (glob_mean > 0) ? glob_mean : (real_t)0;          /* :3604-3605 */
    if (fabs_t(glob_mean) < sqrt_t(EPSILON_T)) glob_mean = 0;
    if (glob_mean != 0)
        for (size_t ix = 0; ix < nnz; ix++) X[ix] -= glob_mean;
    return glob_mean;
}

/* initialize_biases_twosided, weighted branches without NA_as_zero (common.c:4672-4692 items, :4826-4847 users); wsumA / wsumB:
 * the driver's lambda multipliers under scale_lam (NULL without scale_lam: the penalty is not scaled) */
void oracle_initialize_biases_twosided_weighted(int_t m, int_t n,
                                                const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr, const real_t *weightR,
                                                const size_t *Xcsc_p, const int_t *Xcsc_i, const real_t *Xcsc, const real_t *weightC,
                                                real_t lam_user, real_t lam_item, const real_t *wsumA, const real_t *wsumB,
                                                real_t *biasA, real_t *biasB)
{
    if (fabs_t(lam_user) < EPSILON_T) lam_user = EPSILON_T;
    if (fabs_t(lam_item) < EPSILON_T) lam_item = EPSILON_T;
    memset(biasA, 0, (size_t)m * sizeof(real_t));
    memset(biasB, 0, (size_t)n * sizeof(real_t));
    for (int iter = 0; iter < 5; iter++) {
        for (int_t col = 0; col < n; col++) {
            double bmean = 0, wsum = DBL_EPSILON;
            for (size_t ix = Xcsc_p[col]; ix < Xcsc_p[(size_t)col + 1]; ix++)
                bmean += (weightC[ix] * (Xcsc[ix] - biasA[Xcsc_i[ix]] - bmean)) / (wsum += weightC[ix]);
            if (Xcsc_p[(size_t)col + 1] > Xcsc_p[col])
                bmean *= wsum / (wsum + lam_item * ((wsumB != NULL) ? wsumB[col] : 1.));
            biasB[col] = (real_t)bmean;
        }
        for (int_t row = 0; row < m; row++) {
            double bmean = 0, wsum = DBL_EPSILON;
            for (size_t ix = Xcsr_p[row]; ix < Xcsr_p[(size_t)row + 1]; ix++)
                bmean += (weightR[ix] * (Xcsr[ix] - biasB[Xcsr_i[ix]] - bmean)) / (wsum += weightR[ix]);
            if (Xcsr_p[(size_t)row + 1] > Xcsr_p[row])
                bmean *= wsum / (wsum + lam_user * ((wsumA != NULL) ? wsumA[row] : 1.));
            biasA[row] = (real_t)bmean;
        }
    }
}

/* NA_as_zero_X of the next oracle_fit_explicit_als call (sparse X whose absent entries are zeros; model without side
 * information, without weights; cleared by the call): the mean over all m x n cells (common.c:3517-3523), no centring of the
 * stored values, bias start values of the missing-as-zero branches, every half-step through optimizeA Case 3 (one shared
 * matrix, closed form whatever use_cg says) with the bias / mean correction of the right-hand side. */
static bool g_fit_naz = false;
void oracle_set_fit_NA_as_zero_X(bool on) { g_fit_naz = on; }

/* initialize_biases_onesided, missing-as-zero without weights (common.c:4207-4237); mult: wsumA[row] = n under scale_lam
 * (collective.c:8033-8036), else 1 */
static void naz_biases_onesided(int_t m, int_t n, const size_t *p, const real_t *v, real_t glob_mean, real_t lam, bool scale_lam,
                                real_t *bias)
{
    if (fabs_t(lam) < EPSILON_T) lam = EPSILON_T;
    const double mult = scale_lam ? (double)n : 1.;
    for (int_t row = 0; row < m; row++) {
        double bmean = 0;
        const size_t st = p[row], en = p[(size_t)row + 1];
        for (size_t ix = st; ix < en; ix++) bmean += (v[ix] - bmean) / (double)(ix - st + 1);
        bmean -= glob_mean / ((double)(en - st) / (double)n);
        bmean *= (double)(en - st) / ((double)n + lam * mult);
        bias[row] = (en > st) ? (real_t)bmean : (real_t)(-glob_mean / ((double)n / ((double)n + lam * mult)));
    }
}

/* initialize_biases_twosided, missing-as-zero without weights (common.c:4453-4476, :4693-4710, :4849-4868).  The item sweep
 * averages biasA over `row < n` (:4697-4698: the bound of the OTHER dimension); restated as written for n <= m, over the m
 * users for n > m (where the reference reads past the array). */
static void naz_biases_twosided(int_t m, int_t n, const size_t *pr, const real_t *vr, const size_t *pc, const real_t *vc,
                                real_t glob_mean, real_t lam_user, real_t lam_item, bool scale_lam, bool nonneg,
                                real_t *biasA, real_t *biasB)
{
    if (fabs_t(lam_user) < EPSILON_T) lam_user = EPSILON_T;
    if (fabs_t(lam_item) < EPSILON_T) lam_item = EPSILON_T;
    double *meanA = (double *)malloc((size_t)m * sizeof(double)), *meanB = (double *)malloc((size_t)n * sizeof(double));
    for (int_t row = 0; row < m; row++) {
        double xmean = 0;
        for (size_t ix = pr[row]; ix < pr[(size_t)row + 1]; ix++) xmean += (vr[ix] - xmean) / (double)(ix - pr[row] + 1);
        meanA[row] = xmean * ((double)(pr[(size_t)row + 1] - pr[row]) / (double)n);
    }
    for (int_t col = 0; col < n; col++) {
        double xmean = 0;
        for (size_t ix = pc[col]; ix < pc[(size_t)col + 1]; ix++) xmean += (vc[ix] - xmean) / (double)(ix - pc[col] + 1);
        meanB[col] = xmean * ((double)(pc[(size_t)col + 1] - pc[col]) / (double)m);
    }
    memset(biasA, 0, (size_t)m * sizeof(real_t));
    memset(biasB, 0, (size_t)n * sizeof(real_t));
    const int niter = nonneg ? 15 : 5;
    for (int iter = 0; iter < niter; iter++) {
        double bmean = 0;
        if (iter > 0) for (int_t row = 0; row < ((n < m) ? n : m); row++) bmean += (biasA[row] - bmean) / (double)(row + 1);
        for (int_t col = 0; col < n; col++) {
            biasB[col] = (real_t)((meanB[col] - bmean - glob_mean) * ((double)m / ((double)m + lam_item * (scale_lam ? (double)m : 1.))));
            if (nonneg && !(biasB[col] >= 0)) biasB[col] = 0;
        }
        bmean = 0;
        if (iter > 0) for (int_t col = 0; col < n; col++) bmean += (biasB[col] - bmean) / (double)(col + 1);
        for (int_t row = 0; row < m; row++) {
            biasA[row] = (real_t)((meanA[row] - bmean - glob_mean) * ((double)n / ((double)n + lam_user * (scale_lam ? (double)n : 1.))));
            if (nonneg && !(biasA[row] >= 0)) biasA[row] = 0;
        }
    }
    free(meanA); free(meanB);
}

/* observation weights of the next oracle_fit_explicit_als call (COO order; cleared by the call) */
static const real_t *g_fit_weight = NULL;
void oracle_set_fit_weights(const real_t *weight) { g_fit_weight = weight; }

/* rows < g_init_rows_u / columns < g_init_cols_i count g_init_p / g_init_q attributes on top of their entries: wsumA /
 * wsumB under scale_lam_sideinfo with dense side information (collective.c:8071-8104), set by the fit around its call */
static int_t g_init_p = 0, g_init_rows_u = 0, g_init_q = 0, g_init_cols_i = 0;
void oracle_initialize_biases_twosided(int_t m, int_t n,
                                       const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                       const size_t *Xcsc_p, const int_t *Xcsc_i, const real_t *Xcsc,
                                       real_t lam_user, real_t lam_item, bool scale_lam,
                                       real_t *biasA, real_t *biasB)
{
    if (fabs_t(lam_user) < EPSILON_T) lam_user = EPSILON_T;                    /* common.c:4434-4437 */
    if (fabs_t(lam_item) < EPSILON_T) lam_item = EPSILON_T;
    memset(biasA, 0, (size_t)m * sizeof(real_t));
    memset(biasB, 0, (size_t)n * sizeof(real_t));
    for (int iter = 0; iter < 5; iter++) {
        for (int_t col = 0; col < n; col++) {                                  /* :4643-4669 */
            double bmean = 0;
            size_t st = Xcsc_p[col], en = Xcsc_p[(size_t)col + 1];
            for (size_t ix = st; ix < en; ix++)
                bmean += (Xcsc[ix] - biasA[Xcsc_i[ix]] - bmean) / (double)(ix - st + 1);
            size_t cnt = en - st;
            bmean *= (double)cnt / ((double)cnt + lam_item * (scale_lam ? (double)(cnt > 1 ? cnt : 1)
                                                                          + (double)(col < g_init_cols_i ? g_init_q : 0) : 1.));
            biasB[col] = (real_t)bmean;
        }
        for (int_t row = 0; row < m; row++) {                                  /* :4799-4825 */
            double bmean = 0;
            size_t st = Xcsr_p[row], en = Xcsr_p[(size_t)row + 1];
            for (size_t ix = st; ix < en; ix++)
                bmean += (Xcsr[ix] - biasB[Xcsr_i[ix]] - bmean) / (double)(ix - st + 1);
            size_t cnt = en - st;
            if (cnt > 0)
                bmean *= (double)cnt / ((double)cnt + lam_user * (scale_lam ? (double)cnt
                                                                          + (double)(row < g_init_rows_u ? g_init_p : 0) : 1.));
            biasA[row] = (real_t)bmean;
        }
    }
}

/* NA_as_zero_U / NA_as_zero_I (sparse side information whose absent entries are zeros): restated as the dense route on the
 * zero-filled matrix -- equal to the reference's sparse branches to 1e-15 (tests/test_oracle_vs_ref.py) -- with ONE exception the
 * reference makes: a row with neither an entry of X nor an entry of U is not solved but set to zero, bias included
 * (collective_closed_form_block's first check, collective.c:1262-1271; _implicit: :1876-1884).  The rows / columns this applies to
 * are handed to the next fit call here (consumed by it). */
static const int_t *g_zero_rows_A = NULL, *g_zero_rows_B = NULL;
static int_t g_n_zero_rows_A = 0, g_n_zero_rows_B = 0;
void oracle_set_zero_rows(const int_t *rowsA, int_t nA, const int_t *rowsB, int_t nB)
{
    g_zero_rows_A = rowsA; g_n_zero_rows_A = rowsA ? nA : 0; g_zero_rows_B = rowsB; g_n_zero_rows_B = rowsB ? nB : 0;
}
static void zero_rows_(real_t *M, size_t ld, int_t ncols, const int_t *rows, int_t cnt)
{
    for (int_t e = 0; e < cnt; e++) memset(M + (size_t)rows[e] * ld, 0, (size_t)ncols * sizeof(real_t));
}

/* ------------------------------------------------------------------------------------------ */
/* optimizeA_collective with the main matrix missing-as-zero and dense, complete side information, closed form
 * (collective.c:5566-5968 with bufferBeTBeChol, :5607-5617): every row with side information shares ONE matrix
 *     blockdiag(0, B^T B) + w C^T C (upper-left block) + lam mult I,   mult = n + p | n | 1   (:4787-4799, :5700-5716)
 * factorised once (:5715); right-hand sides  X B (tgemm_sp_dense, :5753-5762)  +  w U C (gemm with beta 1, :5764-5770)
 * +  bias_BtX on the k + k_main columns behind k_user (:5815-5821); rows go through the tpotrs branch of
 * collective_closed_form_block (:1364-1460).  Rows of X beyond the side information (m > m_u, :4832-4908): optimizeA Case 3
 * on the columns behind k_user, lam x n under scale_lam; their first k_user entries stay zero (:4817-4822). */
/* Observation weights of the collective_naz_chol call that follows (cleared by the call): with them a row that has entries is no
 * longer served by the shared factorisation (collective.c:1367-1372: weight == NULL || nnz == 0) but by the general branch --
 * the shared matrix plus (w_j - 1) B_j B_j^T over its entries (:1659-1665), right-hand side sum_j [w_j x_j - (w_j - 1)(mean +
 * bias_j)] B_j (:1741-1753), lambda multiplier wsum_i = sum of the row's weights + number of its absent entries (+ p under
 * scale_lam_sideinfo, :1305-1346).  w in the order of Xcsr. */
static const real_t *g_cn_w = NULL, *g_cn_wsum = NULL, *g_cn_biasX = NULL;
static real_t g_cn_glob = 0;
static void collective_naz_chol(real_t *A, size_t lda, const real_t *B, size_t ldb, const real_t *C,
                                int_t m, int_t m_u, int_t n, int_t p, int_t k, int_t k_main, int_t k_user, int_t k_item,
                                const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr, const real_t *U,
                                real_t lam, real_t w_user, real_t lam_last, bool scale_lam, bool scale_lam_sideinfo,
                                const real_t *bias_BtX, int nthreads)
{
    const real_t *wts = g_cn_w, *wsum = g_cn_wsum, *biasX = g_cn_biasX;
    const real_t globX = g_cn_glob;
    g_cn_w = NULL; g_cn_wsum = NULL; g_cn_biasX = NULL; g_cn_glob = 0;
    if (nthreads < 1) nthreads = 1;
    if (m_u > m) m_u = m;
    const int_t kt = k_user + k + k_main, kc = k_user + k, kb = k + k_main;
    for (int_t r = 0; r < m; r++) memset(A + (size_t)r * lda, 0, (size_t)kt * sizeof(real_t));   /* :4817-4822 */
    const real_t mult = scale_lam_sideinfo ? (real_t)(n + p) : (scale_lam ? (real_t)n : (real_t)1);
    real_t *M = (real_t *)calloc((size_t)kt * kt, sizeof(real_t));
    real_t *BtB = (real_t *)malloc((size_t)kb * kb * sizeof(real_t)), *CtC = (real_t *)malloc((size_t)kc * kc * sizeof(real_t));
    oracle_gram(B + k_item, ldb, n, kb, BtB, nthreads);
    oracle_gram(C, (size_t)kc, p, kc, CtC, nthreads);
    for (int_t i = 0; i < kc; i++) for (int_t j = 0; j < kc; j++) M[(size_t)i * kt + j] = w_user * CtC[(size_t)i * kc + j];
    for (int_t i = 0; i < kb; i++) for (int_t j = 0; j < kb; j++) M[(size_t)(k_user + i) * kt + (k_user + j)] += BtB[(size_t)i * kb + j];
    real_t *M0 = NULL;                                                         /* the shared matrix without its diagonal term */
    if (wts != NULL) { M0 = (real_t *)malloc((size_t)kt * kt * sizeof(real_t)); memcpy(M0, M, (size_t)kt * kt * sizeof(real_t)); }
    for (int_t i = 0; i < kt - 1; i++) M[(size_t)i * kt + i] += lam * mult;
    M[(size_t)(kt - 1) * kt + (kt - 1)] += lam_last * mult;
    const int bad = chol_upper_(kt, M, kt);
    #pragma omp parallel for schedule(dynamic) num_threads(nthreads)
    for (int_t ix = 0; ix < m_u; ix++) {
        real_t *a = A + (size_t)ix * lda;
        const size_t st = Xcsr_p[ix], en = Xcsr_p[(size_t)ix + 1];
        if (wts != NULL && en > st) {                                          /* general branch, :1534-1846 */
            real_t *Mi = (real_t *)malloc((size_t)kt * kt * sizeof(real_t));
            memcpy(Mi, M0, (size_t)kt * kt * sizeof(real_t));
            real_t *Mlr = Mi + (size_t)k_user + (size_t)k_user * kt;
            double ws = 0;
            for (size_t jx = st; jx < en; jx++) {
                const real_t *b = B + k_item + (size_t)Xcsr_i[jx] * ldb;
                const real_t w = wts[jx];
                syr_upper_(kb, w - (real_t)1, b, Mlr, kt);                     /* :1659-1665 */
                axpy_(kb, (w * Xcsr[jx]) - (w - (real_t)1) * (globX + ((biasX == NULL) ? (real_t)0 : biasX[Xcsr_i[jx]])), b, a + k_user);
                ws += (double)w;
            }
            for (int_t c = 0; c < kc; c++) {
                double acc = 0;
                for (int_t j = 0; j < p; j++) acc += (double)U[(size_t)ix * p + j] * (double)C[(size_t)j * kc + c];
                a[c] += w_user * (real_t)acc;
            }
            if (bias_BtX != NULL) axpy_(kb, (real_t)1, bias_BtX, a + k_user);
            real_t mi = (real_t)1;
            if (scale_lam || scale_lam_sideinfo) {                             /* :1285-1355 */
                mi = (wsum != NULL && wsum[ix] > 0) ? wsum[ix] : (real_t)ws + (real_t)(n - (int_t)(en - st));
                if (scale_lam_sideinfo) mi += (real_t)p;
            }
            for (int_t i = 0; i < kt - 1; i++) Mi[(size_t)i * kt + i] += lam * mi;
            Mi[(size_t)(kt - 1) * kt + (kt - 1)] += lam_last * mi;
            for (int_t i = 0; i < kt; i++) for (int_t j = 0; j < i; j++) Mi[(size_t)i * kt + j] = Mi[(size_t)j * kt + i];
            solve_sym_(kt, Mi, kt, a);
            free(Mi);
            continue;
        }
        for (size_t jx = Xcsr_p[ix]; jx < Xcsr_p[(size_t)ix + 1]; jx++)
            axpy_(kb, Xcsr[jx], B + k_item + (size_t)Xcsr_i[jx] * ldb, a + k_user);
        for (int_t c = 0; c < kc; c++) {
            double acc = 0;
            for (int_t j = 0; j < p; j++) acc += (double)U[(size_t)ix * p + j] * (double)C[(size_t)j * kc + c];
            a[c] += w_user * (real_t)acc;
        }
        if (bias_BtX != NULL) axpy_(kb, (real_t)1, bias_BtX, a + k_user);
        if (!bad) chol_solve_upper_(kt, M, kt, a);
        else for (int_t c = 0; c < kt; c++) a[c] = NAN;
    }
    free(M); free(M0); free(BtB); free(CtC);
    if (m > m_u) {
        oracle_set_naz_bias_BtX(bias_BtX);
        oracle_optimizeA_naz(A + k_user + (size_t)m_u * lda, lda, B + k_item, ldb, m - m_u, n, kb, Xcsr_p + m_u, Xcsr_i, Xcsr,
                             lam, lam_last, scale_lam || scale_lam_sideinfo, nthreads);
    }
}

/* column means + centering of a dense side-info matrix, common.c:4911-4997 (dense, no NaN) */
static real_t *center_by_cols_dense(const real_t *U, int_t m_u, int_t p, real_t *colmeans)
{
    real_t *Uc = (real_t *)malloc((size_t)m_u * p * sizeof(real_t));
    for (int_t c = 0; c < p; c++) colmeans[c] = 0;
    for (int_t r = 0; r < m_u; r++)
        for (int_t c = 0; c < p; c++) colmeans[c] += U[(size_t)r * p + c];
    for (int_t c = 0; c < p; c++) colmeans[c] /= (double)m_u;
    for (int_t r = 0; r < m_u; r++)
        for (int_t c = 0; c < p; c++) Uc[(size_t)r * p + c] = U[(size_t)r * p + c] - colmeans[c];
    return Uc;
}

/* ------------------------------------------------------------------------------------------ */
/* fit_collective_implicit_als, collective.c:9375-10207, for: sparse X, optional dense full U / II without
 * NaN (then Cholesky only), m_u <= m, n_i <= n, reset_values = false.  A[m, k_user+k+k_main],
 * B[n, k_item+k+k_main], C[p, k_user+k], D[q, k_item+k]. */
int oracle_fit_implicit_als_sideinfo(real_t *A, real_t *B, real_t *C, real_t *D,
                                     real_t *U_colmeans, real_t *I_colmeans,
                                     int_t m, int_t n, int_t k,
                                     const int_t *ixA, const int_t *ixB, const real_t *X, size_t nnz,
                                     const real_t *U, int_t m_u, int_t p, const real_t *II, int_t n_i, int_t q,
                                     int_t k_main, int_t k_user, int_t k_item,
                                     real_t w_main, real_t w_user, real_t w_item,
                                     real_t lam, real_t alpha, bool apply_log_transf,
                                     int_t niter, int nthreads,
                                     bool use_cg, int_t max_cg_steps, bool precondition_cg, bool finalize_chol)
{
    const real_t w_main_orig = w_main;
    if (U == NULL) { m_u = 0; p = 0; }
    if (II == NULL) { n_i = 0; q = 0; }
    /* side information may cover more users / items than X: A, B have max(m, m_u) / max(n, n_i) rows
     * (collective.c:9437-9440); X is padded with empty rows / columns, the Gramians keep X's own shape */
    const int_t *zrA = g_zero_rows_A, *zrB = g_zero_rows_B;                    /* see oracle_set_zero_rows */
    const int_t nzrA = g_n_zero_rows_A, nzrB = g_n_zero_rows_B;
    g_zero_rows_A = g_zero_rows_B = NULL; g_n_zero_rows_A = g_n_zero_rows_B = 0;
    const int_t m_x = m, n_x = n;
    if (m_u > m) m = m_u;
    if (n_i > n) n = n_i;
    int_t k_totA = k_user + k + k_main, k_totB = k_item + k + k_main;
    real_t *Xc = (real_t *)malloc(nnz * sizeof(real_t));
    memcpy(Xc, X, nnz * sizeof(real_t));
    if (apply_log_transf) for (size_t i = 0; i < nnz; i++) Xc[i] = log_t(Xc[i]);  /* collective.c:9578-9587 */
    if (alpha != (real_t)1.) for (size_t i = 0; i < nnz; i++) Xc[i] *= alpha;     /* :9588-9599 */
    size_t *csr_p = (size_t *)malloc(((size_t)m + 1) * sizeof(size_t));
    size_t *csc_p = (size_t *)malloc(((size_t)n + 1) * sizeof(size_t));
    int_t *csr_i = (int_t *)malloc(nnz * sizeof(int_t)), *csc_i = (int_t *)malloc(nnz * sizeof(int_t));
    real_t *csr_v = (real_t *)malloc(nnz * sizeof(real_t)), *csc_v = (real_t *)malloc(nnz * sizeof(real_t));
    oracle_coo_to_csr_and_csc(ixA, ixB, Xc, m, n, nnz, csr_p, csr_i, csr_v, csc_p, csc_i, csc_v);
    free(Xc);
    real_t *Uc = NULL, *Ic = NULL;
    if (U != NULL) Uc = center_by_cols_dense(U, m_u, p, U_colmeans);              /* :9640ff preprocess_sideinfo_matrix */
    if (II != NULL) Ic = center_by_cols_dense(II, n_i, q, I_colmeans);
    if (w_main != (real_t)1.) { lam /= w_main; w_user /= w_main; w_item /= w_main; }   /* :9786-9811 */
    const real_t l1f = g_l1_base / ((w_main_orig != (real_t)1.) ? w_main_orig : (real_t)1.);   /* :9789 */
    if (g_nn_AB || g_nn_C || g_nn_D || l1f != 0 || g_has_l16) use_cg = false;     /* :9568-9571: any of them */
    const real_t wdiv = (w_main_orig != (real_t)1.) ? w_main_orig : (real_t)1.;   /* :9793-9809: entries 2..5 */
    const real_t lamA = g_has_lam6 ? g_lam6[2] / wdiv : lam, lamB = g_has_lam6 ? g_lam6[3] / wdiv : lam;
    const real_t lamC = g_has_lam6 ? g_lam6[4] / wdiv : lam, lamD = g_has_lam6 ? g_lam6[5] / wdiv : lam;
    const real_t l1A = g_has_l16 ? g_l16[2] / wdiv : l1f, l1B = g_has_l16 ? g_l16[3] / wdiv : l1f;
    const real_t l1C = g_has_l16 ? g_l16[4] / wdiv : l1f, l1D = g_has_l16 ? g_l16[5] / wdiv : l1f;
    if (!use_cg) finalize_chol = false;                                           /* :9518 */
    for (int_t iter = 0; iter < niter; iter++) {                                  /* :9827-10045 */
        if (iter == niter - 1 && use_cg && finalize_chol) use_cg = false;
        g_nonneg = g_nn_C; g_l1 = l1C / w_user;
        if (U != NULL)                                                            /* :9834-9873 */
            oracle_optimizeA_dense_full(C, (size_t)(k_user + k), A, (size_t)k_totA, p, m_u, k_user + k,
                                        Uc, (size_t)p, true, lamC / w_user, lamC / w_user, false, nthreads);
        g_nonneg = g_nn_D; g_l1 = l1D / w_item;
        if (II != NULL)                                                           /* :9877-9917 */
            oracle_optimizeA_dense_full(D, (size_t)(k_item + k), B, (size_t)k_totB, q, n_i, k_item + k,
                                        Ic, (size_t)q, true, lamD / w_item, lamD / w_item, false, nthreads);
        g_nonneg = g_nn_AB; g_l1 = l1B;
        if (II != NULL && use_cg)                                                 /* :9924-9963 */
            oracle_optimizeA_collective_cg(B, (size_t)k_totB, A, (size_t)k_totA, D, n, n_i, m_x, q, k, k_main, k_item, k_user,
                                           csc_p, csc_i, csc_v, Ic, lamB, w_item, lamB, false, false, true,
                                           max_cg_steps, precondition_cg, nthreads);
        else if (II != NULL)
            oracle_optimizeA_collective_implicit_chol(B, (size_t)k_totB, A, (size_t)k_totA, D, n, n_i, m_x, q,
                                                      k, k_main, k_item, k_user, csc_p, csc_i, csc_v,
                                                      Ic, lamB, w_item, nthreads);
        else                                                                      /* :9965-9981 */
            oracle_optimizeA_implicit(B + k_item, (size_t)k_totB, A + k_user, (size_t)k_totA, n, m_x, k + k_main,
                                      csc_p, csc_i, csc_v, lamB, nthreads, use_cg, precondition_cg, max_cg_steps, NULL);
        zero_rows_(B, (size_t)k_totB, k_totB, zrB, nzrB);
        g_l1 = l1A;
        if (U != NULL && use_cg)                                                  /* :9985-10022 */
            oracle_optimizeA_collective_cg(A, (size_t)k_totA, B, (size_t)k_totB, C, m, m_u, n_x, p, k, k_main, k_user, k_item,
                                           csr_p, csr_i, csr_v, Uc, lamA, w_user, lamA, false, false, true,
                                           max_cg_steps, precondition_cg, nthreads);
        else if (U != NULL)
            oracle_optimizeA_collective_implicit_chol(A, (size_t)k_totA, B, (size_t)k_totB, C, m, m_u, n_x, p,
                                                      k, k_main, k_user, k_item, csr_p, csr_i, csr_v,
                                                      Uc, lamA, w_user, nthreads);
        else
            oracle_optimizeA_implicit(A + k_user, (size_t)k_totA, B + k_item, (size_t)k_totB, m, n_x, k + k_main,
                                      csr_p, csr_i, csr_v, lamA, nthreads, use_cg, precondition_cg, max_cg_steps, NULL);
        zero_rows_(A, (size_t)k_totA, k_totA, zrA, nzrA);
    }
    g_nonneg = false; g_l1 = 0; t_l1_mult = 1;
    free(Uc); free(Ic);
    free(csr_p); free(csc_p); free(csr_i); free(csc_i); free(csr_v); free(csc_v);
    return 0;
}

int oracle_fit_implicit_als(real_t *A, real_t *B, int_t m, int_t n, int_t k,
                            const int_t *ixA, const int_t *ixB, const real_t *X, size_t nnz,
                            real_t lam, real_t alpha, bool apply_log_transf,
                            int_t niter, int nthreads,
                            bool use_cg, int_t max_cg_steps, bool precondition_cg, bool finalize_chol)
{
    return oracle_fit_implicit_als_sideinfo(A, B, NULL, NULL, NULL, NULL, m, n, k, ixA, ixB, X, nnz,
                                            NULL, 0, 0, NULL, 0, 0, 0, 0, 0, (real_t)1, (real_t)1, (real_t)1,
                                            lam, alpha, apply_log_transf, niter, nthreads,
                                            use_cg, max_cg_steps, precondition_cg, finalize_chol);
}


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
                            bool init_biases)
{
    return oracle_fit_explicit_als_implicit_features(biasA, biasB, A, B, C, D, NULL, NULL, (real_t)1, glob_mean, U_colmeans,
                                                     I_colmeans, m, n, k, ixA, ixB, X, nnz, user_bias, item_bias, center, lam,
                                                     scale_lam, scale_lam_sideinfo, U, m_u, p, II, n_i, q, k_main, k_user,
                                                     k_item, w_user, w_item, niter, nthreads, use_cg, max_cg_steps,
                                                     precondition_cg, finalize_chol, init_biases);
}

/* Ai, Bi != NULL: add_implicit_features (collective.c:8448-8534 and the extra term of the A / B updates); closed-form
 * solves only, side information inside the shape of X.  Iteration order C, D, Bi, Ai, B, A. */
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
                            bool init_biases)
{
    const real_t *weight = g_fit_weight;
    g_fit_weight = NULL;
    const bool naz = g_fit_naz;
    g_fit_naz = false;
    const unsigned char *cfA = g_cf_rows_A, *cfB = g_cf_rows_B;                 /* see oracle_set_closed_form_rows */
    g_cf_rows_A = g_cf_rows_B = NULL;
    const real_t *lmA = g_lam_mult_A, *lmB = g_lam_mult_B;                      /* see oracle_set_lambda_multipliers */
    g_lam_mult_A = g_lam_mult_B = NULL;
    const int_t *zrA = g_zero_rows_A, *zrB = g_zero_rows_B;                    /* see oracle_set_zero_rows */
    const int_t nzrA = g_n_zero_rows_A, nzrB = g_n_zero_rows_B;
    g_zero_rows_A = g_zero_rows_B = NULL; g_n_zero_rows_A = g_n_zero_rows_B = 0;
    /* missing-as-zero with side information: dense complete U / I, closed form, side information on exactly the rows / columns of X
     * (with fewer the reference's own build corrupts its heap, so nothing pins the m > m_u branch restated in collective_naz_chol) */
    if (naz && g_scale_bias_const) return 2;
    /* NA_as_zero_X with implicit features: the model without side information and weights, closed form (the half-steps then are
     * optimizeA_collective's general branch on a matrix all rows share; the block CG with NA_as_zero_X is not restated) */
    if (naz && (Ai != NULL && Bi != NULL) && (U != NULL || II != NULL || weight != NULL || use_cg)) return 2;
    /* missing-as-zero WITH weights: the model without side information, start values given (the reference's weighted bias start
     * values under NA_as_zero index biasB by row inside its item sweep, common.c:4727-4731 -- nothing to restate) */
    if (naz && weight != NULL && (init_biases || ((U != NULL || II != NULL) && use_cg))) return 2;
    /* (use_cg changes nothing there: the factorised block matrix is taken before the solver is looked at, collective.c:1364-1460) */
    if (naz && (U != NULL || II != NULL) && (g_nn_AB || g_l1_base != 0 || g_has_l16 || (U != NULL && m_u != m) || (II != NULL && n_i != n))) return 2;
    if (U == NULL) { m_u = 0; p = 0; }
    if (II == NULL) { n_i = 0; q = 0; }
    if ((k_user && U == NULL) || (k_item && II == NULL)) return 2;             /* collective.c:7308-7318 */
    /* weights: restated for the model without side information (the row solvers of common.c); the collective solvers with
     * weights are checked against the reference build itself (tests/test_gpu_fit.py) */
    if (weight != NULL && (((U != NULL || II != NULL) && !naz) || (Ai != NULL && Bi != NULL) || g_scale_bias_const)) return 2;
    /* side information may cover more users / items than X: A, B have max(m, m_u) / max(n, n_i) rows
     * (collective.c:7332-7335); X is padded with empty rows / columns.  The rows beyond X are fitted to their
     * side information alone by a separate dense solve (optimizeA Case 1, collective.c:4967-5101). */
    const int_t m_x = m, n_x = n;
    if (m_u > m) m = m_u;
    if (n_i > n) n = n_i;
    if (init_biases && (user_bias != item_bias) && !naz) return 2;
    const bool imp = (Ai != NULL && Bi != NULL);
    if (imp && (m_u > m_x || n_i > n_x)) return 2;
    scale_lam = scale_lam || scale_lam_sideinfo;                               /* :7465 */
    const real_t l1f = g_l1_base;
    if (g_nn_AB || l1f != 0 || g_has_l16) use_cg = false;                      /* :7474-7479 */
    /* lam_unique / l1_lam_unique: [2] / [3] for A / B with [0] / [1] on a fitted bias (:8649-8654, :8820-8825), [4] / [5] for
     * C / D (:8367, :8418), [3] / [2] for Bi / Ai (:8469, :8510) */
    const real_t lamA = g_has_lam6 ? g_lam6[2] : lam, lamB = g_has_lam6 ? g_lam6[3] : lam;
    const real_t lamC = g_has_lam6 ? g_lam6[4] : lam, lamD = g_has_lam6 ? g_lam6[5] : lam;
    real_t lamAl = (g_has_lam6 && user_bias) ? g_lam6[0] : lamA, lamBl = (g_has_lam6 && item_bias) ? g_lam6[1] : lamB;
    const real_t l1A = g_has_l16 ? g_l16[2] : l1f, l1B = g_has_l16 ? g_l16[3] : l1f;
    const real_t l1C = g_has_l16 ? g_l16[4] : l1f, l1D = g_has_l16 ? g_l16[5] : l1f;
    real_t l1Al = (g_has_l16 && user_bias) ? g_l16[0] : l1A, l1Bl = (g_has_l16 && item_bias) ? g_l16[1] : l1B;
    bool sbc = g_scale_bias_const && scale_lam && (user_bias || item_bias);    /* :7555-7556 */
    if (sbc && (imp || (item_bias && !user_bias && !use_cg))) return 2;        /* not restated / scaling_biasB unset in the reference */
    if (!use_cg) finalize_chol = false;                                        /* :7481 */
    int_t has_bias = (user_bias || item_bias) ? 1 : 0;
    int_t k_totA = k_user + k + k_main, k_totB = k_item + k + k_main;
    size_t ldA = (size_t)(k_totA + has_bias), ldB = (size_t)(k_totB + has_bias);

    real_t *Xc = (real_t *)malloc(nnz * sizeof(real_t));
    memcpy(Xc, X, nnz * sizeof(real_t));
    if (naz) {                                                                 /* common.c:3494-3523, :3600-3607: X stays as it is */
        *glob_mean = 0;
        if (center && weight != NULL) {                                        /* common.c:3558-3596 */
            double xsum = 0, wsum = DBL_EPSILON;
            if (nthreads >= 8) {
                wsum = 0;
                for (size_t ix = 0; ix < nnz; ix++) { xsum += Xc[ix]; wsum += weight[ix]; }
                *glob_mean = (real_t)(xsum / wsum);
            } else {
                for (size_t ix = 0; ix < nnz; ix++) xsum += ((Xc[ix] - xsum) * weight[ix]) / (wsum += weight[ix]);
                *glob_mean = (real_t)xsum;
            }
            double err = 0, res = 0;                                           /* compensated_sum, helpers.c:1691-1707 */
            for (size_t ix = 0; ix < nnz; ix++) { const double diff = (double)weight[ix] - err, temp = res + diff; err = (temp - res) - diff; res = temp; }
            const long double wsum_l = (long double)res;
            /* (the mean is DIVIDED by the weights' share of all cells, as the reference does, :3590-3594) */
            *glob_mean = (real_t)((long double)(*glob_mean) / (wsum_l / (wsum_l + ((long double)m * (long double)n - (long double)nnz))));
            if (g_nn_AB) *glob_mean = (*glob_mean > 0) ? *glob_mean : (real_t)0;
            if (fabs_t(*glob_mean) < sqrt_t(EPSILON_T)) *glob_mean = 0;
        }
        else if (center) {
            double xsum = 0;
            if (nthreads >= 8) { for (size_t ix = 0; ix < nnz; ix++) xsum += Xc[ix]; *glob_mean = (real_t)(xsum / (double)nnz); }
            else { size_t cnt = 0; for (size_t ix = 0; ix < nnz; ix++) xsum += (Xc[ix] - xsum) / (double)(++cnt); *glob_mean = (real_t)xsum; }
            *glob_mean = (real_t)((long double)(*glob_mean) * ((long double)nnz / ((long double)m * (long double)n)));
            if (g_nn_AB) *glob_mean = (*glob_mean > 0) ? *glob_mean : (real_t)0;
            if (fabs_t(*glob_mean) < sqrt_t(EPSILON_T)) *glob_mean = 0;
        }
    }
    else if (weight != NULL) *glob_mean = center ? oracle_calc_mean_and_center_weighted(Xc, weight, nnz, nthreads) : (real_t)0;
    else
    *glob_mean = center ? oracle_calc_mean_and_center(Xc, nnz, nthreads) : (real_t)0;  /* :7552-7568 */
    size_t *csr_p = (size_t *)malloc(((size_t)m + 1) * sizeof(size_t));
    size_t *csc_p = (size_t *)malloc(((size_t)n + 1) * sizeof(size_t));
    int_t *csr_i = (int_t *)malloc(nnz * sizeof(int_t)), *csc_i = (int_t *)malloc(nnz * sizeof(int_t));
    real_t *csr_v = (real_t *)malloc(nnz * sizeof(real_t)), *csc_v = (real_t *)malloc(nnz * sizeof(real_t));
    oracle_coo_to_csr_and_csc(ixA, ixB, Xc, m, n, nnz, csr_p, csr_i, csr_v, csc_p, csc_i, csc_v);
    free(Xc);
    /* weights in CSR / CSC order (weightR / weightC, helpers.c:1419-1446) and, under scale_lam, the lambda multipliers
     * wsumA / wsumB: the row's sum of weights in double, 1 for a row without entries (collective.c:7978-8008) */
    real_t *weightR = NULL, *weightC = NULL, *wsumA = NULL, *wsumB = NULL;
    if (weight != NULL) {
        weightR = (real_t *)malloc(nnz * sizeof(real_t)); weightC = (real_t *)malloc(nnz * sizeof(real_t));
        size_t *tp_r = (size_t *)malloc(((size_t)m + 1) * sizeof(size_t)), *tp_c = (size_t *)malloc(((size_t)n + 1) * sizeof(size_t));
        int_t *ti_r = (int_t *)malloc(nnz * sizeof(int_t)), *ti_c = (int_t *)malloc(nnz * sizeof(int_t));
        oracle_coo_to_csr_and_csc(ixA, ixB, weight, m, n, nnz, tp_r, ti_r, weightR, tp_c, ti_c, weightC);
        free(tp_r); free(tp_c); free(ti_r); free(ti_c);
        if (scale_lam) {
            wsumA = (real_t *)malloc((size_t)m * sizeof(real_t)); wsumB = (real_t *)malloc((size_t)n * sizeof(real_t));
            for (int_t r = 0; r < m; r++) {
                double ws = 0;
                for (size_t ix = csr_p[r]; ix < csr_p[(size_t)r + 1]; ix++) ws += weightR[ix];
                wsumA[r] = (naz || csr_p[(size_t)r + 1] > csr_p[r]) ? (real_t)ws : (real_t)1;
                if (naz) wsumA[r] += (real_t)(n - (int_t)(csr_p[(size_t)r + 1] - csr_p[r]));      /* collective.c:8014-8022 */
            }
            for (int_t c = 0; c < n; c++) {
                double ws = 0;
                for (size_t ix = csc_p[c]; ix < csc_p[(size_t)c + 1]; ix++) ws += weightC[ix];
                wsumB[c] = (naz || csc_p[(size_t)c + 1] > csc_p[c]) ? (real_t)ws : (real_t)1;
                if (naz) wsumB[c] += (real_t)(m - (int_t)(csc_p[(size_t)c + 1] - csc_p[c]));
            }
        }
    }
    real_t *Uc = NULL, *Ic = NULL;
    if (U != NULL) Uc = center_by_cols_dense(U, m_u, p, U_colmeans);           /* :7850ff preprocess_sideinfo_matrix */
    if (II != NULL) Ic = center_by_cols_dense(II, n_i, q, I_colmeans);

    real_t *A_bias = A, *B_bias = B, *csr_orig = NULL, *csc_orig = NULL;
    if (has_bias) {                                                            /* :7651-7677 */
        A_bias = (real_t *)malloc((size_t)m * ldA * sizeof(real_t));
        B_bias = (real_t *)malloc((size_t)n * ldB * sizeof(real_t));
        if (item_bias) { csr_orig = (real_t *)malloc(nnz * sizeof(real_t)); memcpy(csr_orig, csr_v, nnz * sizeof(real_t)); }
        if (user_bias) { csc_orig = (real_t *)malloc(nnz * sizeof(real_t)); memcpy(csc_orig, csc_v, nnz * sizeof(real_t)); }
    }
    if (sbc) {                                                                 /* :8026-8048, :8071-8160 */
        if (user_bias) {
            double wmean = 0;
            for (int_t r = 0; r < m_x; r++) {
                size_t cnt = csr_p[(size_t)r + 1] - csr_p[r];
                real_t w = (real_t)(cnt + (cnt == 0)) + (real_t)((scale_lam_sideinfo && U != NULL && r < m_u) ? p : 0);
                wmean += ((double)w - wmean) / (double)(r + 1);
            }
            lamAl *= (real_t)wmean; l1Al *= (real_t)wmean;
        }
        if (item_bias) {
            double wmean = 0;
            for (int_t c = 0; c < n_x; c++) {
                size_t cnt = csc_p[(size_t)c + 1] - csc_p[c];
                real_t w = (real_t)(cnt + (cnt == 0)) + (real_t)((scale_lam_sideinfo && II != NULL && c < n_i) ? q : 0);
                wmean += ((double)w - wmean) / (double)(c + 1);
            }
            lamBl *= (real_t)wmean; l1Bl *= (real_t)wmean;
        }
    }
    if (scale_lam_sideinfo) { g_init_p = (U != NULL) ? p : 0; g_init_rows_u = m_u; g_init_q = (II != NULL) ? q : 0; g_init_cols_i = n_i; }
    if (has_bias && init_biases && naz) {                                      /* :8164-8226, missing-as-zero branches */
        if (user_bias && item_bias)
            naz_biases_twosided(m, n, csr_p, csr_v, csc_p, csc_v, *glob_mean, lamAl, lamBl, scale_lam, g_nn_AB, biasA, biasB);
        else if (user_bias) naz_biases_onesided(m, n, csr_p, csr_v, *glob_mean, lamAl, scale_lam, biasA);
        else if (use_cg) naz_biases_onesided(n, m, csc_p, csc_v, *glob_mean, lamBl, scale_lam, biasB);   /* :8187 only with use_cg_B */
        if (g_nn_AB && (user_bias != item_bias)) {
            if (user_bias) for (int_t r = 0; r < m; r++) biasA[r] = (biasA[r] >= 0) ? biasA[r] : 0;
            else for (int_t c = 0; c < n; c++) biasB[c] = (biasB[c] >= 0) ? biasB[c] : 0;
        }
    }
    else
    if (has_bias && init_biases && weight != NULL)
        oracle_initialize_biases_twosided_weighted(m, n, csr_p, csr_i, csr_v, weightR, csc_p, csc_i, csc_v, weightC,
                                                   user_bias ? lamAl : lam, item_bias ? lamBl : lam, wsumA, wsumB, biasA, biasB);
    else
    if (has_bias && init_biases)                                               /* :8164-8226 */
        oracle_initialize_biases_twosided(m, n, csr_p, csr_i, csr_v, csc_p, csc_i, csc_v,
                                          user_bias ? lamAl : lam, item_bias ? lamBl : lam, scale_lam, biasA, biasB);
    g_init_p = g_init_q = 0; g_init_rows_u = g_init_cols_i = 0;
    if (has_bias) {                                                            /* :8283-8317 */
        for (int_t r = 0; r < m; r++) {
            memcpy(A_bias + (size_t)r * ldA, A + (size_t)r * k_totA, (size_t)k_totA * sizeof(real_t));
            A_bias[(size_t)r * ldA + k_totA] = user_bias ? biasA[r] : (real_t)1;
        }
        for (int_t c = 0; c < n; c++) {
            memcpy(B_bias + (size_t)c * ldB, B + (size_t)c * k_totB, (size_t)k_totB * sizeof(real_t));
            B_bias[(size_t)c * ldB + k_totB] = item_bias ? biasB[c] : (real_t)1;
        }
    }

    if (wsumA != NULL && lmA != NULL) memcpy(wsumA, lmA, (size_t)m * sizeof(real_t));
    if (wsumB != NULL && lmB != NULL) memcpy(wsumB, lmB, (size_t)n * sizeof(real_t));
    for (int_t iter = 0; iter < niter; iter++) {                               /* :8334-8898 */
        if (iter == niter - 1 && use_cg && finalize_chol) use_cg = false;
        g_nonneg = g_nn_C; g_l1 = l1C / w_user; g_l1_last_set = false;
        if (U != NULL)                                                         /* :8358-8387 */
            oracle_optimizeA_dense_full(C, (size_t)(k_user + k), A_bias, ldA, p, m_u, k_user + k,
                                        Uc, (size_t)p, true, lamC / w_user, lamC / w_user, scale_lam, nthreads);
        g_nonneg = g_nn_D; g_l1 = l1D / w_item;
        if (II != NULL)                                                        /* :8409-8441 */
            oracle_optimizeA_dense_full(D, (size_t)(k_item + k), B_bias, ldB, q, n_i, k_item + k,
                                        Ic, (size_t)q, true, lamD / w_item, lamD / w_item, scale_lam, nthreads);
        g_nonneg = g_nn_AB; g_l1 = l1B / w_implicit;
        if (imp) {                                                             /* :8448-8534 */
            oracle_optimizeA_naz(Bi, (size_t)(k + k_main), A_bias + k_user, ldA, n, m, k + k_main, csc_p, csc_i, NULL,
                                 lamB / w_implicit, lamB / w_implicit, scale_lam, nthreads);
            g_l1 = l1A / w_implicit;
            oracle_optimizeA_naz(Ai, (size_t)(k + k_main), B_bias + k_item, ldB, m, n, k + k_main, csr_p, csr_i, NULL,
                                 lamA / w_implicit, lamA / w_implicit, scale_lam, nthreads);
        }
        g_nonneg = g_nn_AB; g_l1 = l1B; g_l1_last = l1Bl; g_l1_last_set = true;
        if (item_bias)                                                         /* :8538-8543 */
            for (int_t r = 0; r < m; r++) A_bias[(size_t)r * ldA + k_totA] = 1;
        if (user_bias && !naz)                                                 /* :8566-8570 */
            for (size_t ix = 0; ix < nnz; ix++) csc_v[ix] = csc_orig[ix] - biasA[csc_i[ix]];
        g_cg_Bi = imp ? Ai : NULL; g_cg_ki = k + k_main; g_cg_wimp = w_implicit;
        if ((II != NULL || imp) && use_cg && !naz)                             /* :8634-8678 */
            oracle_optimizeA_collective_cg(B_bias, ldB, A_bias, ldA, D, n_x, (n_i < n_x) ? n_i : n_x, m, q, k, k_main + (int_t)item_bias, k_item, k_user,
                                           csc_p, csc_i, csc_v, Ic, lamB, w_item, lamBl, scale_lam, scale_lam_sideinfo, false,
                                           max_cg_steps, precondition_cg, nthreads);
        else if (II != NULL && naz) {                                          /* :8573-8600 + :8612 with NA_as_zero_X */
            const int_t ks = k + k_main + (int_t)item_bias;
            real_t *btx = NULL;
            if (user_bias || center) {
                btx = (real_t *)calloc((size_t)ks, sizeof(real_t));
                for (int_t r = 0; r < m; r++)
                    axpy_(ks, -((user_bias ? biasA[r] : (real_t)0) + (center ? *glob_mean : (real_t)0)), A_bias + k_user + (size_t)r * ldA, btx);
            }
            if (weight != NULL) { g_cn_w = weightC; g_cn_wsum = wsumB; g_cn_biasX = (btx != NULL && user_bias) ? biasA : NULL; g_cn_glob = *glob_mean; }
            collective_naz_chol(B_bias, ldB, A_bias, ldA, D, n_x, n_i, m, q, k, k_main + (int_t)item_bias, k_item, k_user,
                                csc_p, csc_i, csc_v, Ic, lamB, w_item, lamBl, scale_lam, scale_lam_sideinfo, btx, nthreads);
            free(btx);
        }
        else if (II != NULL || imp) {                                          /* :8612 */
            real_t *btx = NULL;
            if (naz) {                                                         /* (imp without side information, see above) :8573-8600 */
                const int_t ks = k + k_main + (int_t)item_bias;
                if (user_bias || center) {
                    btx = (real_t *)calloc((size_t)ks, sizeof(real_t));
                    for (int_t r = 0; r < m; r++)
                        axpy_(ks, -((user_bias ? biasA[r] : (real_t)0) + (center ? *glob_mean : (real_t)0)), A_bias + k_user + (size_t)r * ldA, btx);
                }
                g_cc_naz = true; g_cc_btx = btx;
            }
            collective_chol_impl(B_bias, ldB, A_bias, ldA, D, n_x, (n_i < n_x) ? n_i : n_x, m, q,
                                             k, k_main + (int_t)item_bias, k_item, k_user,
                                             csc_p, csc_i, csc_v, Ic, lamB, w_item, lamBl,
                                             scale_lam, scale_lam_sideinfo, nthreads, imp ? Ai : NULL, k_main, w_implicit);
            free(btx);
        }
        else if (naz) {                                                        /* :8573-8600 + optimizeA Case 3 */
            const int_t ks = k + k_main + (int_t)item_bias;
            real_t *btx = NULL;
            if (user_bias || center) {
                btx = (real_t *)calloc((size_t)ks, sizeof(real_t));
                for (int_t r = 0; r < m; r++)
                    axpy_(ks, -((user_bias ? biasA[r] : (real_t)0) + (center ? *glob_mean : (real_t)0)), A_bias + k_user + (size_t)r * ldA, btx);
            }
            if (weight != NULL)                                                /* optimizeA Case 4, NA_as_zero + weights (:8680-8717) */
                oracle_optimizeA_naz_weighted(B_bias + k_item, ldB, A_bias + k_user, ldA, n, m, ks, csc_p, csc_i, csc_v, weightC, wsumB, btx,
                                              (btx != NULL && user_bias) ? biasA : NULL, *glob_mean, lamB, lamBl, scale_lam, sbc,
                                              use_cg, precondition_cg, max_cg_steps, nthreads);
            else {
            oracle_set_naz_bias_BtX(btx);
            oracle_optimizeA_naz(B_bias + k_item, ldB, A_bias + k_user, ldA, n, m, ks, csc_p, csc_i, csc_v, lamB, lamBl, scale_lam, nthreads);
            }
            free(btx);
        }
        else {                                                                 /* :8680-8717 */
            oracle_set_row_weights(weightC, wsumB);
            g_cf_now = use_cg ? cfB : NULL;
            oracle_optimizeA_explicit(B_bias + k_item, ldB, A_bias + k_user, ldA, n, m,
                                      k + k_main + (int_t)item_bias, csc_p, csc_i, csc_v,
                                      lamB, lamBl, scale_lam, sbc, nthreads,
                                      use_cg, precondition_cg, max_cg_steps);
        }
        if (II != NULL) {
        if (n_i > n_x) {                                            /* rows known from side information only */
            if (!use_cg)
                for (int_t r = n_x; r < n_i; r++) memset(B_bias + (size_t)r * ldB, 0, (size_t)(k_totB + has_bias) * sizeof(real_t));
            oracle_optimizeA_dense_full(B_bias + (size_t)n_x * ldB, ldB, D, (size_t)(k_item + k), n_i - n_x, q,
                                        k_item + k, Ic + (size_t)n_x * q, (size_t)q, false, lamB / w_item, lamB / w_item,
                                        scale_lam, nthreads);
        }
        }
        zero_rows_(B_bias, ldB, k_totB + (int_t)item_bias, zrB, nzrB);
        if (item_bias)                                                         /* :8723-8725 */
            for (int_t c = 0; c < n; c++) biasB[c] = B_bias[(size_t)c * ldB + k_totB];
        if (user_bias)                                                         /* :8728-8732 */
            for (int_t c = 0; c < n; c++) B_bias[(size_t)c * ldB + k_totB] = 1;
        g_l1 = l1A; g_l1_last = l1Al;
        if (item_bias && !naz)                                                 /* :8750-8754 */
            for (size_t ix = 0; ix < nnz; ix++) csr_v[ix] = csr_orig[ix] - biasB[csr_i[ix]];
        g_cg_Bi = imp ? Bi : NULL;
        if ((U != NULL || imp) && use_cg && !naz)                              /* :8805-8845 */
            oracle_optimizeA_collective_cg(A_bias, ldA, B_bias, ldB, C, m_x, (m_u < m_x) ? m_u : m_x, n, p, k, k_main + (int_t)user_bias, k_user, k_item,
                                           csr_p, csr_i, csr_v, Uc, lamA, w_user, lamAl, scale_lam, scale_lam_sideinfo, false,
                                           max_cg_steps, precondition_cg, nthreads);
        else if (U != NULL && naz) {                                           /* :8756-8787 + :8783 with NA_as_zero_X */
            const int_t ks = k + k_main + (int_t)user_bias;
            real_t *btx = NULL;
            if (item_bias || center) {
                btx = (real_t *)calloc((size_t)ks, sizeof(real_t));
                for (int_t c = 0; c < n; c++)
                    axpy_(ks, -((item_bias ? biasB[c] : (real_t)0) + (center ? *glob_mean : (real_t)0)), B_bias + k_item + (size_t)c * ldB, btx);
            }
            if (weight != NULL) { g_cn_w = weightR; g_cn_wsum = wsumA; g_cn_biasX = (btx != NULL && item_bias) ? biasB : NULL; g_cn_glob = *glob_mean; }
            collective_naz_chol(A_bias, ldA, B_bias, ldB, C, m_x, m_u, n, p, k, k_main + (int_t)user_bias, k_user, k_item,
                                csr_p, csr_i, csr_v, Uc, lamA, w_user, lamAl, scale_lam, scale_lam_sideinfo, btx, nthreads);
            free(btx);
        }
        else if (U != NULL || imp) {                                           /* :8783 */
            real_t *btx = NULL;
            if (naz) {                                                         /* :8756-8787 */
                const int_t ks = k + k_main + (int_t)user_bias;
                if (item_bias || center) {
                    btx = (real_t *)calloc((size_t)ks, sizeof(real_t));
                    for (int_t c = 0; c < n; c++)
                        axpy_(ks, -((item_bias ? biasB[c] : (real_t)0) + (center ? *glob_mean : (real_t)0)), B_bias + k_item + (size_t)c * ldB, btx);
                }
                g_cc_naz = true; g_cc_btx = btx;
            }
            collective_chol_impl(A_bias, ldA, B_bias, ldB, C, m_x, (m_u < m_x) ? m_u : m_x, n, p,
                                             k, k_main + (int_t)user_bias, k_user, k_item,
                                             csr_p, csr_i, csr_v, Uc, lamA, w_user, lamAl,
                                             scale_lam, scale_lam_sideinfo, nthreads, imp ? Bi : NULL, k_main, w_implicit);
            free(btx);
        }
        else if (naz) {                                                        /* :8756-8787 + optimizeA Case 3 */
            const int_t ks = k + k_main + (int_t)user_bias;
            real_t *btx = NULL;
            if (item_bias || center) {
                btx = (real_t *)calloc((size_t)ks, sizeof(real_t));
                for (int_t c = 0; c < n; c++)
                    axpy_(ks, -((item_bias ? biasB[c] : (real_t)0) + (center ? *glob_mean : (real_t)0)), B_bias + k_item + (size_t)c * ldB, btx);
            }
            if (weight != NULL)                                                /* :8847-8876 */
                oracle_optimizeA_naz_weighted(A_bias + k_user, ldA, B_bias + k_item, ldB, m, n, ks, csr_p, csr_i, csr_v, weightR, wsumA, btx,
                                              (btx != NULL && item_bias) ? biasB : NULL, *glob_mean, lamA, lamAl, scale_lam, sbc,
                                              use_cg, precondition_cg, max_cg_steps, nthreads);
            else {
            oracle_set_naz_bias_BtX(btx);
            oracle_optimizeA_naz(A_bias + k_user, ldA, B_bias + k_item, ldB, m, n, ks, csr_p, csr_i, csr_v, lamA, lamAl, scale_lam, nthreads);
            }
            free(btx);
        }
        else {                                                                 /* :8847-8876 */
            oracle_set_row_weights(weightR, wsumA);
            g_cf_now = use_cg ? cfA : NULL;
            oracle_optimizeA_explicit(A_bias + k_user, ldA, B_bias + k_item, ldB, m, n,
                                      k + k_main + (int_t)user_bias, csr_p, csr_i, csr_v,
                                      lamA, lamAl, scale_lam, sbc, nthreads,
                                      use_cg, precondition_cg, max_cg_steps);
        }
        if (U != NULL) {
        if (m_u > m_x) {                                            /* rows known from side information only */
            if (!use_cg)
                for (int_t r = m_x; r < m_u; r++) memset(A_bias + (size_t)r * ldA, 0, (size_t)(k_totA + has_bias) * sizeof(real_t));
            oracle_optimizeA_dense_full(A_bias + (size_t)m_x * ldA, ldA, C, (size_t)(k_user + k), m_u - m_x, p,
                                        k_user + k, Uc + (size_t)m_x * p, (size_t)p, false, lamA / w_user, lamA / w_user,
                                        scale_lam, nthreads);
        }
        }
        g_cg_Bi = NULL;
        zero_rows_(A_bias, ldA, k_totA + (int_t)user_bias, zrA, nzrA);
        if (user_bias)                                                         /* :8882-8884 */
            for (int_t r = 0; r < m; r++) biasA[r] = A_bias[(size_t)r * ldA + k_totA];
    }
    if (has_bias) {                                                            /* :8908-8920 */
        for (int_t r = 0; r < m; r++)
            memcpy(A + (size_t)r * k_totA, A_bias + (size_t)r * ldA, (size_t)k_totA * sizeof(real_t));
        for (int_t c = 0; c < n; c++)
            memcpy(B + (size_t)c * k_totB, B_bias + (size_t)c * ldB, (size_t)k_totB * sizeof(real_t));
        free(A_bias); free(B_bias);
    }
    if (user_bias) for (int_t r = m_x; r < m; r++) biasA[r] = 0;                /* :8296, :8923: no bias beyond X */
    if (item_bias) for (int_t c = n_x; c < n; c++) biasB[c] = 0;                /* :8308, :8925 */
    g_nonneg = false; g_l1 = 0; t_l1_mult = 1; g_l1_last_set = false;
    free(csr_orig); free(csc_orig); free(Uc); free(Ic);
    free(csr_p); free(csc_p); free(csr_i); free(csc_i); free(csr_v); free(csc_v);
    free(weightR); free(weightC); free(wsumA); free(wsumB);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Factors of new rows. */
void oracle_factors_explicit_multiple(real_t *A, real_t *biasA, int_t m,
                                      const real_t *U, int_t m_u, int_t p, const real_t *C,
                                      real_t glob_mean, const real_t *biasB, const real_t *U_colmeans,
                                      const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                      int_t n, const real_t *B,
                                      int_t k, int_t k_user, int_t k_item, int_t k_main,
                                      real_t lam, real_t lam_bias,
                                      bool scale_lam, bool scale_lam_sideinfo, bool scale_bias_const, real_t scaling_biasA,
                                      real_t w_main, real_t w_user, const real_t *TransCtCinvCt, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (U == NULL || p <= 0) { m_u = 0; p = 0; }
    const int_t m_max = m > m_u ? m : m_u;
    const int_t ub = biasA != NULL;
    const int_t k_totA = k_user + k + k_main, k_totC = k_user + k, kb = k + k_main + ub, kt = k_totA + ub;
    const size_t ldb0 = (size_t)(k_item + k + k_main), ldb = ldb0 + (size_t)ub;
    /* factors_collective_explicit_single :10611-10630 */
    if (!ub) scale_bias_const = false;
    if ((scale_lam || scale_lam_sideinfo) && scale_bias_const) lam_bias *= scaling_biasA;
    /* append_ones_last_col, :10927-10937 */
    real_t *Bp = (real_t *)malloc((size_t)n * ldb * sizeof(real_t));
    for (int_t j = 0; j < n; j++) {
        memcpy(Bp + (size_t)j * ldb, B + (size_t)j * ldb0, ldb0 * sizeof(real_t));
        if (ub) Bp[(size_t)j * ldb + ldb0] = 1;
    }
    /* cold rows: lam / w_user on C^T C (collective_factors_cold :3353-3357, :3397-3411) */
    const real_t lam_cold = lam / w_user;
    /* collective_factors_warm :3694-3702 */
    real_t lam_w = lam, lam_bias_w = lam_bias, w_user_w = w_user;
    if (w_main != 1) { w_user_w /= w_main; lam_w /= w_main; lam_bias_w /= w_main; }
    real_t *CtC = NULL;
    if (p > 0) {
        CtC = (real_t *)calloc((size_t)k_totC * k_totC + 1, sizeof(real_t));
        oracle_gram(C, (size_t)k_totC, p, k_totC, CtC, nthreads);
    }
    size_t maxnnz = 1;
    for (int_t i = 0; i < m; i++) if (Xcsr_p[i + 1] - Xcsr_p[i] > maxnnz) maxnnz = Xcsr_p[i + 1] - Xcsr_p[i];
    const size_t szbuf = (size_t)kt * kt + (size_t)kt + maxnnz + (size_t)(p > 0 ? p : 1);
    real_t *bufs = (real_t *)malloc(szbuf * (size_t)nthreads * sizeof(real_t));
    const bool sl = scale_lam || scale_lam_sideinfo;                           /* :3631 */
    #pragma omp parallel for schedule(dynamic) num_threads(nthreads)
    for (int_t ix = 0; ix < m_max; ix++) {
        real_t *buf = bufs + szbuf * (size_t)omp_get_thread_num();
        real_t *M = buf, *sol = buf + (size_t)kt * kt, *x = sol + kt, *u = x + maxnnz;
        real_t *a = A + (size_t)ix * k_totA;
        const size_t st = ix < m ? Xcsr_p[ix] : 0, en = ix < m ? Xcsr_p[(size_t)ix + 1] : 0, nnz = en - st;
        const bool has_u = ix < m_u;
        memset(a, 0, (size_t)k_totA * sizeof(real_t));
        if (ub) biasA[ix] = 0;
        if (nnz == 0 && !has_u) continue;                                      /* :3634-3650 */
        for (size_t jx = 0; jx < nnz; jx++)                                    /* preprocess_vec :6337-6388 */
            x[jx] = Xcsr[st + jx] - (biasB ? (biasB[Xcsr_i[st + jx]] + glob_mean) : glob_mean);
        if (has_u) for (int_t c = 0; c < p; c++) u[c] = U[(size_t)ix * p + c] - (U_colmeans ? U_colmeans[c] : 0);
        if (nnz == 0) {                                                        /* cold, :3309-3440 */
            if (TransCtCinvCt != NULL) {
                for (int_t c = 0; c < k_totC; c++) {
                    double s = 0;
                    for (int_t j = 0; j < p; j++) s += (double)u[j] * (double)TransCtCinvCt[(size_t)j * k_totC + c];
                    a[c] = (real_t)s;
                }
                continue;
            }
            for (int_t c = 0; c < k_totC; c++) {
                double s = 0;
                for (int_t j = 0; j < p; j++) s += (double)u[j] * (double)C[(size_t)j * k_totC + c];
                sol[c] = (real_t)s;
            }
            for (int_t i = 0; i < k_totC; i++)
                for (int_t j = 0; j < k_totC; j++) M[(size_t)i * k_totC + j] = CtC[(size_t)i * k_totC + j];
            /* factors_closed_form with scale_lam = scale_bias_const = scale_lam_sideinfo: lam * p on all but the last */
            for (int_t i = 0; i < k_totC; i++)
                M[(size_t)i * k_totC + i] += (scale_lam_sideinfo && i < k_totC - 1) ? lam_cold * (real_t)p : lam_cold;
            solve_sym_(k_totC, M, k_totC, sol);
            memcpy(a, sol, (size_t)k_totC * sizeof(real_t));
            continue;
        }
        if (!has_u) {                                                          /* factors_closed_form, :3772-3815 */
            /* without a bias the call passes scale_lam in the place of scale_bias_const (:3789-3799), so the last
             * factor keeps the unscaled lam */
            real_t lam_i = lam_w, lam_last_i = ub ? lam_bias_w : lam_w;
            if (sl) {
                lam_i *= (real_t)nnz;
                if (ub && !scale_bias_const) lam_last_i *= (real_t)nnz;
            }
            explicit_chol_row(sol, kb, Bp + k_item, ldb, x, Xcsr_i + st, nnz, NULL, lam_i, lam_last_i, M);
            memcpy(a + k_user, sol, (size_t)(k + k_main) * sizeof(real_t));
            if (ub) biasA[ix] = sol[k + k_main];
            continue;
        }
        /* collective_closed_form_block on [B | 1], :3866-3930 -> :1223-1847 */
        real_t mult = 1;
        if (sl) { mult = (real_t)nnz; if (scale_lam_sideinfo) mult += (real_t)p; }
        memset(M, 0, (size_t)kt * kt * sizeof(real_t));
        memset(sol, 0, (size_t)kt * sizeof(real_t));
        for (int_t i = 0; i < k_totC; i++)
            for (int_t j = 0; j < k_totC; j++) M[(size_t)i * kt + j] = w_user_w * CtC[(size_t)i * k_totC + j];
        for (int_t c = 0; c < k_totC; c++) {
            double s = 0;
            for (int_t j = 0; j < p; j++) s += (double)u[j] * (double)C[(size_t)j * k_totC + c];
            sol[c] = (real_t)((double)w_user_w * s);
        }
        real_t *Mlr = M + (size_t)k_user + (size_t)k_user * kt;
        for (size_t jx = 0; jx < nnz; jx++)
            syr_upper_(kb, (real_t)1, Bp + (size_t)k_item + (size_t)Xcsr_i[st + jx] * ldb, Mlr, kt);
        for (size_t jx = 0; jx < nnz; jx++)
            axpy_(kb, x[jx], Bp + (size_t)k_item + (size_t)Xcsr_i[st + jx] * ldb, sol + k_user);
        for (int_t i = 0; i < kt - 1; i++) M[(size_t)i * kt + i] += lam_w * mult;
        M[(size_t)(kt - 1) * kt + (kt - 1)] += (ub ? lam_bias_w : lam_w) * mult;
        solve_sym_(kt, M, kt, sol);
        memcpy(a, sol, (size_t)k_totA * sizeof(real_t));
        if (ub) biasA[ix] = sol[k_totA];
    }
    free(bufs); free(Bp); free(CtC);
}

void oracle_factors_implicit_multiple(real_t *A, int_t m,
                                      const real_t *U, int_t m_u, int_t p, const real_t *C, const real_t *U_colmeans,
                                      const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                      int_t n, const real_t *B,
                                      int_t k, int_t k_user, int_t k_item, int_t k_main,
                                      real_t lam, real_t alpha, real_t w_main, real_t w_user, real_t w_main_multiplier,
                                      bool apply_log_transf, const real_t *BtB_in, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (U == NULL || p <= 0) { m_u = 0; p = 0; }
    const int_t m_max = m > m_u ? m : m_u;
    const int_t k_totA = k_user + k + k_main, k_totC = k_user + k, kb = k + k_main;
    const size_t ldb = (size_t)(k_item + k + k_main);
    /* :11270-11280: BtB + lam I with the lam of the call */
    real_t *BtB = (real_t *)calloc((size_t)kb * kb + 1, sizeof(real_t));
    if (BtB_in) memcpy(BtB, BtB_in, (size_t)kb * kb * sizeof(real_t));
    else {
        oracle_gram(B + k_item, ldb, n, kb, BtB, nthreads);
        for (int_t i = 0; i < kb; i++) BtB[(size_t)i * kb + i] += lam;
    }
    /* collective_factors_warm_implicit :4000-4004 (and _cold_implicit :3490-3497) */
    real_t wm = w_main * w_main_multiplier;
    if (wm != 1) { lam /= wm; w_user /= wm; }
    real_t *CtC = NULL;
    if (p > 0) {
        CtC = (real_t *)calloc((size_t)k_totC * k_totC + 1, sizeof(real_t));
        oracle_gram(C, (size_t)k_totC, p, k_totC, CtC, nthreads);
    }
    size_t maxnnz = 1;
    for (int_t i = 0; i < m; i++) if (Xcsr_p[i + 1] - Xcsr_p[i] > maxnnz) maxnnz = Xcsr_p[i + 1] - Xcsr_p[i];
    const size_t szbuf = (size_t)k_totA * k_totA + maxnnz + (size_t)(p > 0 ? p : 1);
    real_t *bufs = (real_t *)malloc(szbuf * (size_t)nthreads * sizeof(real_t));
    #pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int_t ix = 0; ix < m_max; ix++) {
        real_t *buf = bufs + szbuf * (size_t)omp_get_thread_num();
        real_t *M = buf, *x = buf + (size_t)k_totA * k_totA, *u = x + maxnnz;
        real_t *a = A + (size_t)ix * k_totA;
        const size_t st = ix < m ? Xcsr_p[ix] : 0, en = ix < m ? Xcsr_p[(size_t)ix + 1] : 0, nnz = en - st;
        const bool has_u = ix < m_u;
        memset(a, 0, (size_t)k_totA * sizeof(real_t));
        if (nnz == 0 && !has_u) continue;
        for (size_t jx = 0; jx < nnz; jx++) {
            real_t v = Xcsr[st + jx];
            if (apply_log_transf) v = (real_t)log((double)v);                  /* :10802-10810 */
            x[jx] = (alpha != 1) ? v * alpha : v;                              /* :4006-4016 */
        }
        if (!has_u) {                                                          /* factors_implicit_chol, :4057-4067 */
            for (int_t i = 0; i < kb; i++) memcpy(M + (size_t)i * kb, BtB + (size_t)i * kb, (size_t)kb * sizeof(real_t));
            for (size_t jx = 0; jx < nnz; jx++)
                axpy_(kb, x[jx] + (real_t)1, B + (size_t)k_item + (size_t)Xcsr_i[st + jx] * ldb, a + k_user);
            for (size_t jx = 0; jx < nnz; jx++)
                syr_upper_(kb, x[jx], B + (size_t)k_item + (size_t)Xcsr_i[st + jx] * ldb, M, kb);
            solve_sym_(kb, M, kb, a + k_user);
            continue;
        }
        /* collective_closed_form_block_implicit, few_NAs branch with a precomputed BtB (:1966-1990) */
        for (int_t c = 0; c < p; c++) u[c] = U[(size_t)ix * p + c] - (U_colmeans ? U_colmeans[c] : 0);
        memset(M, 0, (size_t)k_totA * k_totA * sizeof(real_t));
        for (int_t i = 0; i < k_totC; i++)
            for (int_t j = 0; j < k_totC; j++) M[(size_t)i * k_totA + j] = w_user * CtC[(size_t)i * k_totC + j];
        for (int_t i = 0; i < kb; i++)
            for (int_t j = 0; j < kb; j++) M[(size_t)(k_user + i) * k_totA + (k_user + j)] += BtB[(size_t)i * kb + j];
        for (int_t i = 0; i < k_user; i++) M[(size_t)i * k_totA + i] += lam;
        for (int_t c = 0; c < k_totC; c++) {
            double s = 0;
            for (int_t j = 0; j < p; j++) s += (double)u[j] * (double)C[(size_t)j * k_totC + c];
            a[c] = (real_t)((double)w_user * s);
        }
        for (size_t jx = 0; jx < nnz; jx++)
            axpy_(kb, x[jx] + (real_t)1, B + (size_t)k_item + (size_t)Xcsr_i[st + jx] * ldb, a + k_user);
        real_t *Mlr = M + (size_t)k_user + (size_t)k_user * k_totA;
        for (size_t jx = 0; jx < nnz; jx++)
            syr_upper_(kb, x[jx], B + (size_t)k_item + (size_t)Xcsr_i[st + jx] * ldb, Mlr, k_totA);
        solve_sym_(k_totA, M, k_totA, a);
    }
    free(bufs); free(BtB); free(CtC);
}

/* ------------------------------------------------------------------------------------------ */
/* optimizeA_collective (explicit, collective.c:5566-5968 -> collective_closed_form_block :1223-1847) and
 * optimizeA_collective_implicit (:5971-6244 -> collective_closed_form_block_implicit :1849-2131) with SPARSE side
 * information: u_vec == NULL, u_vec_sp != NULL, !NA_as_zero_U ("add_C" branches :1636-1653 / :2003-2011 and the
 * tgemv_dense_sp right-hand sides :1719-1731 / :2013-2021); Cholesky; U_csr has m_u rows. */
/* NA_as_zero_X of the oracle_optimizeA_collective_sparse_chol call that follows (explicit model; cleared by the call): the main
 * matrix's absent entries are zeros, so every row's lower-right block is the whole B^T B (collective_closed_form_block part 2,
 * collective.c:1631-1640, prefer_BtB), its right-hand side sum_j x_j B_j + bias_BtX (:1738-1742, :1774-1775), its lambda
 * multiplier n (+ the row's attributes under scale_lam_sideinfo, :1300-1346), and a row without entries is solved like the others
 * unless it has neither side information nor a constant (:1258-1268). */
static bool g_csp_naz = false;
static const real_t *g_csp_btx = NULL;
void oracle_set_collective_sparse_naz(bool on, const real_t *bias_BtX) { g_csp_naz = on; g_csp_btx = bias_BtX; }

void oracle_optimizeA_collective_sparse_chol(real_t *A, size_t lda, const real_t *B, size_t ldb, const real_t *C,
                                             int_t m, int_t m_u, int_t n, int_t p,
                                             int_t k, int_t k_main, int_t k_user, int_t k_item,
                                             const size_t *Xcsr_p, const int_t *Xcsr_i, const real_t *Xcsr,
                                             const size_t *Ucsr_p, const int_t *Ucsr_i, const real_t *Ucsr,
                                             real_t lam, real_t w_user, real_t lam_last,
                                             bool scale_lam, bool scale_lam_sideinfo, bool implicit, int nthreads)
{
    (void)p;
    const bool naz = g_csp_naz && !implicit;
    const real_t *btx = naz ? g_csp_btx : NULL;
    g_csp_naz = false; g_csp_btx = NULL;
    if (nthreads < 1) nthreads = 1;
    const int_t k_totA = k_user + k + k_main, k_totC = k_user + k, kb = k + k_main;
    for (int_t i = 0; i < m; i++) memset(A + (size_t)i * lda, 0, (size_t)k_totA * sizeof(real_t));  /* :4817-4822, :6018-6019 */
    real_t *BtB = NULL;
    if (implicit || naz) {                                                     /* :6056-6061; :5650-5668 (build_BtB_CtC) */
        BtB = (real_t *)calloc((size_t)kb * kb + 1, sizeof(real_t));
        oracle_gram(B + k_item, ldb, n, kb, BtB, nthreads);
        if (implicit) for (int_t i = 0; i < kb; i++) BtB[(size_t)i * kb + i] += lam;
    }
    const size_t szbuf = (size_t)k_totA * k_totA;
    real_t *bufs = (real_t *)malloc(szbuf * (size_t)nthreads * sizeof(real_t));
    #pragma omp parallel for schedule(dynamic) num_threads(nthreads)
    for (int_t ix = 0; ix < m; ix++) {
        const size_t st = Xcsr_p[ix], en = Xcsr_p[(size_t)ix + 1], nnz = en - st;
        const size_t us = ix < m_u ? Ucsr_p[ix] : 0, ue = ix < m_u ? Ucsr_p[(size_t)ix + 1] : 0, nnz_u = ue - us;
        real_t *a = A + (size_t)ix * lda;
        if (nnz == 0 && nnz_u == 0 && !(naz && btx != NULL)) continue;         /* :1258-1268, :1876-1885: zeros */
        real_t *M = bufs + szbuf * (size_t)omp_get_thread_num();
        memset(M, 0, szbuf * sizeof(real_t));
        real_t lam_i = lam, lam_last_i = lam_last;
        if (!implicit && (scale_lam || scale_lam_sideinfo)) {                  /* :1285-1355 */
            real_t mult = naz ? (real_t)n : (nnz ? (real_t)nnz : (real_t)1);   /* :1300-1301 */
            if (scale_lam_sideinfo) mult += (real_t)nnz_u;                     /* :1338-1346 */
            lam_i *= mult; lam_last_i *= mult;
            t_l1_mult = mult;
        } else t_l1_mult = 1;
        for (size_t jx = us; jx < ue; jx++)                                    /* :1636-1653 / :2003-2011 */
            syr_upper_(k_totC, w_user, C + (size_t)Ucsr_i[jx] * k_totC, M, k_totA);
        for (size_t jx = us; jx < ue; jx++)                                    /* :1719-1731 / :2013-2021 */
            axpy_(k_totC, w_user * Ucsr[jx], C + (size_t)Ucsr_i[jx] * k_totC, a);
        real_t *Mlr = M + (size_t)k_user + (size_t)k_user * k_totA;
        if (implicit) {
            for (int_t i = 0; i < kb; i++)
                for (int_t j = i; j < kb; j++) Mlr[(size_t)i * k_totA + j] += BtB[(size_t)i * kb + j];
            for (int_t i = 0; i < k_user; i++) M[(size_t)i * k_totA + i] += lam;
            for (size_t jx = st; jx < en; jx++)
                axpy_(kb, Xcsr[jx] + (real_t)1, B + (size_t)k_item + (size_t)Xcsr_i[jx] * ldb, a + k_user);
            for (size_t jx = st; jx < en; jx++)
                syr_upper_(kb, Xcsr[jx], B + (size_t)k_item + (size_t)Xcsr_i[jx] * ldb, Mlr, k_totA);
        } else {
            if (naz) {                                                         /* :1631-1640 */
                for (int_t i = 0; i < kb; i++)
                    for (int_t j = i; j < kb; j++) Mlr[(size_t)i * k_totA + j] += BtB[(size_t)i * kb + j];
            } else
            for (size_t jx = st; jx < en; jx++)
                syr_upper_(kb, (real_t)1, B + (size_t)k_item + (size_t)Xcsr_i[jx] * ldb, Mlr, k_totA);
            for (size_t jx = st; jx < en; jx++)
                axpy_(kb, Xcsr[jx], B + (size_t)k_item + (size_t)Xcsr_i[jx] * ldb, a + k_user);
            if (btx != NULL) axpy_(kb, (real_t)1, btx, a + k_user);           /* :1774-1775 */
            for (int_t i = 0; i < k_totA - 1; i++) M[(size_t)i * k_totA + i] += lam_i;
            M[(size_t)(k_totA - 1) * k_totA + (k_totA - 1)] += lam_last_i;
        }
        solve_sym_(k_totA, M, k_totA, a);
    }
    free(bufs); free(BtB);
}

/* ------------------------------------------------------------------------------------------ */
/* fit_collective_explicit_als (collective.c:7263-9370) / fit_collective_implicit_als (:9375-10207) with SPARSE side
 * information on either side (COO triplets, missing = absent, no centring of it), Cholesky updates (use_cg = false),
 * m_u <= m, n_i <= n, reset_values = false.  A side without side information is a plain optimizeA / optimizeA_implicit
 * step.  C / D: optimizeA Case 4 on the CSC of U / I with lam / w (:8354-8441, :9834-9917).  Explicit: optional
 * biases (start values passed in) and centring; implicit: alpha scaling, w_main folded into lam / w_user / w_item. */
/* NA_as_zero_X of the oracle_fit_als_sparse_sideinfo call that follows (explicit model, closed form, sparse side information on
 * exactly the rows / columns of X where there is any; cleared by the call): the mean over all cells (common.c:3494-3523), the
 * stored values left as they are, the bias / mean constant on every right-hand side (collective.c:8573-8600, :8756-8787); a side
 * without side information is optimizeA Case 3. */
static bool g_spfit_naz = false;
void oracle_set_sparse_fit_NA_as_zero_X(bool on) { g_spfit_naz = on; }
/* DENSE side information with NaN runs as this fit on its centred present entries; what the dense C / D update (optimizeA Cases
 * 1-2 on the transposed matrix, common.c:2793-3116) does differently for an attribute that misses only a few of its m_u values
 * arrives as per-attribute rules of the oracle_fit_als_sparse_sideinfo call that follows (cleared by the call), the ones of the
 * dense main matrix above (oracle_set_closed_form_rows / oracle_set_lambda_multipliers):
 *   cf*:   1 = solved in closed form whatever use_cg says (fewer than 2 (k_side + k) missing values: the precomputed Gramian
 *          minus the missing rows, factors_closed_form :759-790, which comes before the CG branch at :884; complete attributes of
 *          a matrix with >= 75 % complete ones: Case 1's shared factorisation); 2 = CG from zero with k_side + k steps (the other
 *          attributes of such a matrix: Case 1's fix-up loop, :2953-2985);
 *   mult*: the lambda multiplier under scale_lam (such an attribute keeps the m_u lam of a complete one, :3031-3032 / :2832, the
 *          others lam times their present values). */
static const unsigned char *g_side_cf_C = NULL, *g_side_cf_D = NULL;
static const real_t *g_side_mult_C = NULL, *g_side_mult_D = NULL;
void oracle_set_sideinfo_dense_rules(const unsigned char *cfC, const real_t *multC, const unsigned char *cfD, const real_t *multD)
{
    g_side_cf_C = cfC; g_side_mult_C = multC; g_side_cf_D = cfD; g_side_mult_D = multD;
}

int oracle_fit_als_sparse_sideinfo(bool implicit, real_t *biasA, real_t *biasB, real_t *A, real_t *B, real_t *C, real_t *D,
                                   real_t *glob_mean, int_t m, int_t n, int_t k,
                                   const int_t *ixA, const int_t *ixB, const real_t *X, size_t nnz,
                                   bool user_bias, bool item_bias, bool center,
                                   real_t lam, real_t alpha, bool scale_lam, bool scale_lam_sideinfo,
                                   const int_t *U_row, const int_t *U_col, const real_t *U_sp, size_t nnz_U, int_t m_u, int_t p,
                                   const int_t *I_row, const int_t *I_col, const real_t *I_sp, size_t nnz_I, int_t n_i, int_t q,
                                   int_t k_main, int_t k_user, int_t k_item,
                                   real_t w_main, real_t w_user, real_t w_item, int_t niter, int nthreads,
                                   bool use_cg, int_t max_cg_steps, bool precondition_cg, bool finalize_chol)
{
    const bool naz = g_spfit_naz && !implicit;
    g_spfit_naz = false;
    const unsigned char *cfC = g_side_cf_C, *cfD = g_side_cf_D;
    const real_t *multC = g_side_mult_C, *multD = g_side_mult_D;
    g_side_cf_C = g_side_cf_D = NULL; g_side_mult_C = g_side_mult_D = NULL;
    if (nnz_U == 0) { m_u = 0; p = 0; }
    if (nnz_I == 0) { n_i = 0; q = 0; }
    if (m_u > m || n_i > n || (k_user && !p) || (k_item && !q)) return 2;
    if (naz && (use_cg || (p && m_u != m) || (q && n_i != n) || g_nn_AB || g_l1_base != 0)) return 2;    /* (not restated) */
    real_t *onesU = NULL, *onesI = NULL;                                       /* unit weights: the multipliers ride on the weighted solvers */
    if (multC != NULL && nnz_U) { onesU = (real_t *)malloc(nnz_U * sizeof(real_t)); for (size_t e = 0; e < nnz_U; e++) onesU[e] = 1; }
    if (multD != NULL && nnz_I) { onesI = (real_t *)malloc(nnz_I * sizeof(real_t)); for (size_t e = 0; e < nnz_I; e++) onesI[e] = 1; }
    if (implicit) { user_bias = item_bias = center = false; scale_lam = scale_lam_sideinfo = false; }
    scale_lam = scale_lam || scale_lam_sideinfo;                               /* :7465 */
    const real_t l1f = g_l1_base / ((w_main != (real_t)1.) ? w_main : (real_t)1.);
    if (w_main != (real_t)1.) { lam /= w_main; w_user /= w_main; w_item /= w_main; }   /* :7497-7521, :9786-9811 */
    const int_t has_bias = (user_bias || item_bias) ? 1 : 0;
    const int_t k_totA = k_user + k + k_main, k_totB = k_item + k + k_main, kcu = k_user + k, kci = k_item + k;
    const size_t ldA = (size_t)(k_totA + has_bias), ldB = (size_t)(k_totB + has_bias);
    real_t *Xc = (real_t *)malloc((nnz + 1) * sizeof(real_t));
    memcpy(Xc, X, nnz * sizeof(real_t));
    if (implicit && alpha != (real_t)1.) for (size_t i = 0; i < nnz; i++) Xc[i] *= alpha;
    real_t gmean = 0;
    if (naz) {                                                                 /* common.c:3494-3523, :3600-3607: X stays as it is */
        if (center) {
            double xsum = 0;
            if (nthreads >= 8) { for (size_t ix = 0; ix < nnz; ix++) xsum += Xc[ix]; gmean = (real_t)(xsum / (double)nnz); }
            else { size_t cnt = 0; for (size_t ix = 0; ix < nnz; ix++) xsum += (Xc[ix] - xsum) / (double)(++cnt); gmean = (real_t)xsum; }
            gmean = (real_t)((long double)gmean * ((long double)nnz / ((long double)m * (long double)n)));
            if (fabs_t(gmean) < sqrt_t(EPSILON_T)) gmean = 0;
        }
        if (glob_mean) *glob_mean = gmean;
    } else
    if (glob_mean) *glob_mean = center ? oracle_calc_mean_and_center(Xc, nnz, nthreads) : (real_t)0;
    size_t *csr_p = (size_t *)malloc(((size_t)m + 1) * sizeof(size_t)), *csc_p = (size_t *)malloc(((size_t)n + 1) * sizeof(size_t));
    int_t *csr_i = (int_t *)malloc((nnz + 1) * sizeof(int_t)), *csc_i = (int_t *)malloc((nnz + 1) * sizeof(int_t));
    real_t *csr_v = (real_t *)malloc((nnz + 1) * sizeof(real_t)), *csc_v = (real_t *)malloc((nnz + 1) * sizeof(real_t));
    oracle_coo_to_csr_and_csc(ixA, ixB, Xc, m, n, nnz, csr_p, csr_i, csr_v, csc_p, csc_i, csc_v);
    free(Xc);
    /* side information: CSR by row, CSC by attribute (convert_sparse_X on U / I, collective.c:6452) */
    size_t *Ur_p = NULL, *Uc_p = NULL, *Ir_p = NULL, *Ic_p = NULL;
    int_t *Ur_i = NULL, *Uc_i = NULL, *Ir_i = NULL, *Ic_i = NULL;
    real_t *Ur_v = NULL, *Uc_v = NULL, *Ir_v = NULL, *Ic_v = NULL;
    if (p) {
        Ur_p = (size_t *)malloc(((size_t)m_u + 1) * sizeof(size_t)); Uc_p = (size_t *)malloc(((size_t)p + 1) * sizeof(size_t));
        Ur_i = (int_t *)malloc(nnz_U * sizeof(int_t)); Uc_i = (int_t *)malloc(nnz_U * sizeof(int_t));
        Ur_v = (real_t *)malloc(nnz_U * sizeof(real_t)); Uc_v = (real_t *)malloc(nnz_U * sizeof(real_t));
        oracle_coo_to_csr_and_csc(U_row, U_col, U_sp, m_u, p, nnz_U, Ur_p, Ur_i, Ur_v, Uc_p, Uc_i, Uc_v);
    }
    if (q) {
        Ir_p = (size_t *)malloc(((size_t)n_i + 1) * sizeof(size_t)); Ic_p = (size_t *)malloc(((size_t)q + 1) * sizeof(size_t));
        Ir_i = (int_t *)malloc(nnz_I * sizeof(int_t)); Ic_i = (int_t *)malloc(nnz_I * sizeof(int_t));
        Ir_v = (real_t *)malloc(nnz_I * sizeof(real_t)); Ic_v = (real_t *)malloc(nnz_I * sizeof(real_t));
        oracle_coo_to_csr_and_csc(I_row, I_col, I_sp, n_i, q, nnz_I, Ir_p, Ir_i, Ir_v, Ic_p, Ic_i, Ic_v);
    }
    real_t *A_b = A, *B_b = B, *csr_orig = NULL, *csc_orig = NULL;
    if (has_bias) {                                                            /* :7651-7677, :8283-8317 */
        A_b = (real_t *)malloc((size_t)m * ldA * sizeof(real_t));
        B_b = (real_t *)malloc((size_t)n * ldB * sizeof(real_t));
        if (item_bias) { csr_orig = (real_t *)malloc(nnz * sizeof(real_t)); memcpy(csr_orig, csr_v, nnz * sizeof(real_t)); }
        if (user_bias) { csc_orig = (real_t *)malloc(nnz * sizeof(real_t)); memcpy(csc_orig, csc_v, nnz * sizeof(real_t)); }
        for (int_t r = 0; r < m; r++) {
            memcpy(A_b + (size_t)r * ldA, A + (size_t)r * k_totA, (size_t)k_totA * sizeof(real_t));
            A_b[(size_t)r * ldA + k_totA] = user_bias ? biasA[r] : (real_t)1;
        }
        for (int_t c = 0; c < n; c++) {
            memcpy(B_b + (size_t)c * ldB, B + (size_t)c * k_totB, (size_t)k_totB * sizeof(real_t));
            B_b[(size_t)c * ldB + k_totB] = item_bias ? biasB[c] : (real_t)1;
        }
    }
    if (g_nn_AB || l1f != 0 || (implicit && (g_nn_C || g_nn_D))) use_cg = false;   /* :7474-7479, :9568-9571 */
    if (!use_cg) finalize_chol = false;
    for (int_t iter = 0; iter < niter; iter++) {
        if (iter == niter - 1 && use_cg && finalize_chol) use_cg = false;          /* :8336-8340, :9829-9830 */
        g_nonneg = g_nn_C; g_l1 = l1f / w_user;
        if (p && onesU != NULL && scale_lam) oracle_set_row_weights(onesU, multC);
        if (p && use_cg) g_cf_now = cfC;
        if (p) oracle_optimizeA_explicit(C, (size_t)kcu, A_b, ldA, p, m_u, kcu, Uc_p, Uc_i, Uc_v, lam / w_user, lam / w_user,
                                         scale_lam, false, nthreads, use_cg, precondition_cg, max_cg_steps);
        g_nonneg = g_nn_D; g_l1 = l1f / w_item;
        if (q && onesI != NULL && scale_lam) oracle_set_row_weights(onesI, multD);
        if (q && use_cg) g_cf_now = cfD;
        if (q) oracle_optimizeA_explicit(D, (size_t)kci, B_b, ldB, q, n_i, kci, Ic_p, Ic_i, Ic_v, lam / w_item, lam / w_item,
                                         scale_lam, false, nthreads, use_cg, precondition_cg, max_cg_steps);
        g_nonneg = g_nn_AB; g_l1 = l1f;
        if (item_bias) for (int_t r = 0; r < m; r++) A_b[(size_t)r * ldA + k_totA] = 1;
        if (user_bias && !naz) for (size_t ix = 0; ix < nnz; ix++) csc_v[ix] = csc_orig[ix] - biasA[csc_i[ix]];
        real_t *btxB = NULL;
        if (naz && (user_bias || center)) {                                    /* collective.c:8573-8600 */
            const int_t ks = k + k_main + (int_t)item_bias;
            btxB = (real_t *)calloc((size_t)ks, sizeof(real_t));
            for (int_t r = 0; r < m; r++) axpy_(ks, -((user_bias ? biasA[r] : (real_t)0) + gmean), A_b + k_user + (size_t)r * ldA, btxB);
        }
        if (naz && q) {
            oracle_set_collective_sparse_naz(true, btxB);
            oracle_optimizeA_collective_sparse_chol(B_b, ldB, A_b, ldA, D, n, n_i, m, q, k, k_main + (int_t)item_bias, k_item, k_user,
                                                    csc_p, csc_i, csc_v, Ir_p, Ir_i, Ir_v, lam, w_item, lam, scale_lam,
                                                    scale_lam_sideinfo, false, nthreads);
        } else if (naz) {
            oracle_set_naz_bias_BtX(btxB);
            oracle_optimizeA_naz(B_b + k_item, ldB, A_b + k_user, ldA, n, m, k + k_main + (int_t)item_bias, csc_p, csc_i, csc_v, lam, lam,
                                 scale_lam, nthreads);
        } else
        if (q && use_cg)
            oracle_optimizeA_collective_sparse_cg(B_b, ldB, A_b, ldA, D, n, n_i, m, q, k, k_main + (int_t)item_bias, k_item, k_user,
                                                  csc_p, csc_i, csc_v, Ir_p, Ir_i, Ir_v, lam, w_item, lam, scale_lam,
                                                  scale_lam_sideinfo, implicit, max_cg_steps, precondition_cg, nthreads);
        else if (q)
            oracle_optimizeA_collective_sparse_chol(B_b, ldB, A_b, ldA, D, n, n_i, m, q, k, k_main + (int_t)item_bias, k_item, k_user,
                                                    csc_p, csc_i, csc_v, Ir_p, Ir_i, Ir_v, lam, w_item, lam, scale_lam,
                                                    scale_lam_sideinfo, implicit, nthreads);
        else if (implicit)
            oracle_optimizeA_implicit(B_b + k_item, ldB, A_b + k_user, ldA, n, m, k + k_main, csc_p, csc_i, csc_v, lam, nthreads,
                                      use_cg, precondition_cg, max_cg_steps, NULL);
        else
            oracle_optimizeA_explicit(B_b + k_item, ldB, A_b + k_user, ldA, n, m, k + k_main + (int_t)item_bias, csc_p, csc_i, csc_v,
                                      lam, lam, scale_lam, false, nthreads, use_cg, precondition_cg, max_cg_steps);
        if (item_bias) for (int_t c = 0; c < n; c++) biasB[c] = B_b[(size_t)c * ldB + k_totB];
        if (user_bias) for (int_t c = 0; c < n; c++) B_b[(size_t)c * ldB + k_totB] = 1;
        free(btxB);
        if (item_bias && !naz) for (size_t ix = 0; ix < nnz; ix++) csr_v[ix] = csr_orig[ix] - biasB[csr_i[ix]];
        real_t *btxA = NULL;
        if (naz && (item_bias || center)) {                                    /* collective.c:8756-8787 */
            const int_t ks = k + k_main + (int_t)user_bias;
            btxA = (real_t *)calloc((size_t)ks, sizeof(real_t));
            for (int_t c = 0; c < n; c++) axpy_(ks, -((item_bias ? biasB[c] : (real_t)0) + gmean), B_b + k_item + (size_t)c * ldB, btxA);
        }
        if (naz && p) {
            oracle_set_collective_sparse_naz(true, btxA);
            oracle_optimizeA_collective_sparse_chol(A_b, ldA, B_b, ldB, C, m, m_u, n, p, k, k_main + (int_t)user_bias, k_user, k_item,
                                                    csr_p, csr_i, csr_v, Ur_p, Ur_i, Ur_v, lam, w_user, lam, scale_lam,
                                                    scale_lam_sideinfo, false, nthreads);
        } else if (naz) {
            oracle_set_naz_bias_BtX(btxA);
            oracle_optimizeA_naz(A_b + k_user, ldA, B_b + k_item, ldB, m, n, k + k_main + (int_t)user_bias, csr_p, csr_i, csr_v, lam, lam,
                                 scale_lam, nthreads);
        } else
        if (p && use_cg)
            oracle_optimizeA_collective_sparse_cg(A_b, ldA, B_b, ldB, C, m, m_u, n, p, k, k_main + (int_t)user_bias, k_user, k_item,
                                                  csr_p, csr_i, csr_v, Ur_p, Ur_i, Ur_v, lam, w_user, lam, scale_lam,
                                                  scale_lam_sideinfo, implicit, max_cg_steps, precondition_cg, nthreads);
        else if (p)
            oracle_optimizeA_collective_sparse_chol(A_b, ldA, B_b, ldB, C, m, m_u, n, p, k, k_main + (int_t)user_bias, k_user, k_item,
                                                    csr_p, csr_i, csr_v, Ur_p, Ur_i, Ur_v, lam, w_user, lam, scale_lam,
                                                    scale_lam_sideinfo, implicit, nthreads);
        else if (implicit)
            oracle_optimizeA_implicit(A_b + k_user, ldA, B_b + k_item, ldB, m, n, k + k_main, csr_p, csr_i, csr_v, lam, nthreads,
                                      use_cg, precondition_cg, max_cg_steps, NULL);
        else
            oracle_optimizeA_explicit(A_b + k_user, ldA, B_b + k_item, ldB, m, n, k + k_main + (int_t)user_bias, csr_p, csr_i, csr_v,
                                      lam, lam, scale_lam, false, nthreads, use_cg, precondition_cg, max_cg_steps);
        free(btxA);
        if (user_bias) for (int_t r = 0; r < m; r++) biasA[r] = A_b[(size_t)r * ldA + k_totA];
    }
    if (has_bias) {
        for (int_t r = 0; r < m; r++) memcpy(A + (size_t)r * k_totA, A_b + (size_t)r * ldA, (size_t)k_totA * sizeof(real_t));
        for (int_t c = 0; c < n; c++) memcpy(B + (size_t)c * k_totB, B_b + (size_t)c * ldB, (size_t)k_totB * sizeof(real_t));
        free(A_b); free(B_b);
    }
    g_nonneg = false; g_l1 = 0; t_l1_mult = 1;
    free(csr_orig); free(csc_orig);
    free(csr_p); free(csc_p); free(csr_i); free(csc_i); free(csr_v); free(csc_v);
    free(Ur_p); free(Uc_p); free(Ir_p); free(Ic_p); free(Ur_i); free(Uc_i); free(Ir_i); free(Ic_i);
    free(Ur_v); free(Uc_v); free(Ir_v); free(Ic_v); free(onesU); free(onesI);
    return 0;
}
