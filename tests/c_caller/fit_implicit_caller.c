/*
 * A plain C99 user of the drop-in entry point -- what /root/reference/example/c_example.c:99-140 is to the reference:
 * it includes include/cmfrec_hip.h, links libcmfrec_hip_double.so and calls fit_collective_implicit_als with the
 * reference's positional argument list (src/cmfrec.h:1893-1921).  tests/test_gpu_c_caller.py compiles it with
 * `gcc -std=c99 -pedantic -Wall -Werror`, feeds it the g5 fixture through a flat binary file and compares what it writes.
 *
 * Input file (little endian):  int32 m, n, k, niter, use_cg, finalize_chol;  int64 nnz;  double lam, alpha;
 *                              int32 row[nnz], col[nnz];  double val[nnz];  double A0[m*k];
 * Output file:                 int32 return code;  double A[m*k];  double B[n*k]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cmfrec_hip.h"

static int read_all(void *dst, size_t size, size_t count, FILE *f)
{
    return fread(dst, size, count, f) == count ? 0 : 1;
}

int main(int argc, char **argv)
{
    int32_t hdr[6];
    int64_t nnz64;
    double scal[2];
    int_t *row = NULL, *col = NULL;
    real_t *val = NULL, *A = NULL, *B = NULL;
    real_t w_main_multiplier = 0;
    FILE *f;
    size_t nnz, szA, szB;
    int_t ret;
    int32_t ret32;

    if (argc != 3) { fprintf(stderr, "usage: %s <input.bin> <output.bin>\n", argv[0]); return 64; }
    f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 65; }
    if (read_all(hdr, sizeof(int32_t), 6, f) || read_all(&nnz64, sizeof(int64_t), 1, f) || read_all(scal, sizeof(double), 2, f)) return 66;
    nnz = (size_t)nnz64;
    szA = (size_t)hdr[0] * (size_t)hdr[2];
    szB = (size_t)hdr[1] * (size_t)hdr[2];
    row = (int_t *)malloc(nnz * sizeof(int_t));
    col = (int_t *)malloc(nnz * sizeof(int_t));
    val = (real_t *)malloc(nnz * sizeof(real_t));
    A = (real_t *)malloc(szA * sizeof(real_t));
    B = (real_t *)calloc(szB, sizeof(real_t));             /* B starts at zero, as in the fixture */
    if (!row || !col || !val || !A || !B) return 67;
    if (read_all(row, sizeof(int_t), nnz, f) || read_all(col, sizeof(int_t), nnz, f) || read_all(val, sizeof(real_t), nnz, f) ||
        read_all(A, sizeof(real_t), szA, f))
        return 66;
    fclose(f);

    /* the caller owns every output; optional pointers are NULL, optional sizes 0 (include/cmfrec.h.in:238-241) */
    ret = fit_collective_implicit_als(
        A, B, NULL, NULL,
        false, 1,                                   /* reset_values, seed: start values are the caller's */
        NULL, NULL,
        hdr[0], hdr[1], hdr[2],
        row, col, val, nnz,
        (real_t)scal[0], NULL, (real_t)0, NULL,     /* lam, lam_unique, l1_lam, l1_lam_unique */
        NULL, 0, 0, NULL, 0, 0,                     /* U, m_u, p, II, n_i, q */
        NULL, NULL, NULL, 0, NULL, NULL, NULL, 0,   /* sparse side information */
        false, false,
        0, 0, 0,                                    /* k_main, k_user, k_item */
        (real_t)1, (real_t)1, (real_t)1, &w_main_multiplier,
        (real_t)scal[1], false, false,              /* alpha, adjust_weight, apply_log_transf */
        hdr[3], 1,                                  /* niter, nthreads */
        false, false,                               /* verbose, handle_interrupt */
        hdr[4] != 0, 3, false, hdr[5] != 0,         /* use_cg, max_cg_steps, precondition_cg, finalize_chol */
        false, 100, false, false,
        false, NULL, NULL, NULL, NULL);

    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 65; }
    ret32 = (int32_t)ret;
    fwrite(&ret32, sizeof ret32, 1, f);
    fwrite(A, sizeof(real_t), szA, f);
    fwrite(B, sizeof(real_t), szB, f);
    fclose(f);
    free(row); free(col); free(val); free(A); free(B);
    return ret == 0 ? 0 : 1;
}
