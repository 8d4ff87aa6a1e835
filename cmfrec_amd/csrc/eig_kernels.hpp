// eig_kernels.hpp -- symmetric eigen-decomposition of the small shared matrix of the low-rank row path (gfx950), round 6.
//
// The low-rank closed form (lowrank_kernels.hpp) needs  w C^T C = Q L Q^T  once per half-step (k_c x k_c, 64 .. 320; config 5:
// 256 / 257).  Rounds 3-5 bound rocSOLVER's dsyevd at run time: a chain of ~4000 small launches (its tridiagonalisation is
// launch-bound: 6 ms on an idle device, 33 ms beside the persistent batches of the long rows -- profiles/r05/r05_zg_*), and the
// one-workgroup Jacobi kernel (91 ms) as the fallback.  This file is the library's own decomposition in TWO launches:
//
//   eig_tridiag_kernel   one workgroup of 16 wavefronts: Householder tridiagonalisation  A = Q1 T Q1^T  (the scheme of LAPACK's
//                        dsytd2, lower variant, on a full symmetric copy in double precision that stays in L2: per step one
//                        matrix-vector product and one rank-2 update of the trailing square, three barriers).
//   eig_ql_rows_kernel   ceil(n / 64) workgroups of one wavefront: every LANE owns one ROW of the eigenvector matrix (n doubles
//                        in LDS, lane-interleaved: conflict-free) -- it first applies the n - 2 reflectors to its unit row
//                        (row r of Q1), then runs the implicit QL iteration on (d, e) (the EISPACK tql2 recurrences; every
//                        wavefront repeats the scalar recurrences, they are a chain of latencies and cost nothing beside it)
//                        and applies each plane rotation to its own row as it is generated.  Rows are independent under
//                        column rotations, so there is no communication at all.
//
// The sequential part is the ~0.75 n^2 plane rotations of the QL sweeps (n = 257: ~50 k, each a dependent chain of ~12 double
// precision operations: one v_rsq_f64 gives both r = x rsq(x) and 1 / r).  Results are deterministic (fixed order everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lanes.hpp"

namespace cmfhip {

constexpr int EIG_MAX_N = 320;

// A [n, n] row-major symmetric (any precision T) -> W [n, n] doubles: the reflector of step i in W[i][i+2 ..] (v[0] = 1 implied at
// column i + 1), the trailing squares as they were consumed; d [n], e [n] the tridiagonal matrix (e[i] couples i and i + 1), tau [n].
template <typename T>
__global__ void __launch_bounds__(1024)
eig_tridiag_kernel(const T *__restrict__ A, int n, double *W, double *__restrict__ d, double *__restrict__ e, double *__restrict__ tau)
{
    __shared__ double s_v[EIG_MAX_N], s_p[EIG_MAX_N];
    __shared__ double s_tau;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int x = tid; x < n * n; x += 1024) W[x] = (double)A[x];
    __syncthreads();
    for (int i = 0; i < n - 1; i++) {
        const int L = n - i - 1;                     // x = W[i][i+1 .. n-1], the part of row (= column) i beside the diagonal
        double *row = W + (size_t)i * n + i + 1;
        if (wave == 0) {
            double xv[EIG_MAX_N / 64];
            double ss = 0.0;
#pragma unroll
            for (int q = 0; q < EIG_MAX_N / 64; q++) {
                const int c = lane + 64 * q;
                xv[q] = (c < L) ? row[c] : 0.0;
                if (c >= 1) ss += xv[q] * xv[q];
            }
            ss = lanes::wave_sum(ss);
            const double alpha = __shfl(xv[0], 0);
            double t = 0.0, beta = alpha, sc = 0.0;     // dlarfg: H = I - t v v^T, H x = (beta, 0, ..), v[0] = 1
            if (ss > 0.0) {
                beta = -copysign(sqrt(alpha * alpha + ss), alpha);
                t = (beta - alpha) / beta;
                sc = 1.0 / (alpha - beta);
            }
#pragma unroll
            for (int q = 0; q < EIG_MAX_N / 64; q++) {
                const int c = lane + 64 * q;
                if (c < L) {
                    const double v = (c == 0) ? 1.0 : xv[q] * sc;
                    s_v[c] = v;
                    if (c >= 1) row[c] = v;
                }
            }
            if (lane == 0) { e[i] = beta; tau[i] = t; d[i] = W[(size_t)i * n + i]; s_tau = t; }
        }
        __syncthreads();
        const double t = s_tau;
        if (t != 0.0) {                                 // (uniform over the workgroup)
            for (int r = wave; r < L; r += 16) {        // p = t A22 v
                const double *ar = W + (size_t)(i + 1 + r) * n + i + 1;
                double acc = 0.0;
                for (int c = lane; c < L; c += 64) acc += ar[c] * s_v[c];
                acc = lanes::wave_sum(acc);
                if (lane == 0) s_p[r] = t * acc;
            }
            __syncthreads();
            double dot = 0.0;                           // w = p - (t / 2) (p . v) v ;  A22 -= v w^T + w v^T
            for (int c = lane; c < L; c += 64) dot += s_p[c] * s_v[c];
            dot = lanes::wave_sum(dot);
            const double a2 = -0.5 * t * dot;
            for (int r = wave; r < L; r += 16) {
                double *ar = W + (size_t)(i + 1 + r) * n + i + 1;
                const double vr = s_v[r], wr = s_p[r] + a2 * vr;
                for (int c = lane; c < L; c += 64) {
                    const double vc = s_v[c], wc = s_p[c] + a2 * vc;
                    ar[c] -= vr * wc + wr * vc;
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) { d[n - 1] = W[(size_t)(n - 1) * n + n - 1]; e[n - 1] = 0.0; tau[n - 1] = 0.0; }
}

// r = sqrt(x) and 1 / r from ONE reciprocal square root (x > 0): v_rsq_f64 + two Newton steps
__device__ __forceinline__ void eig_sqrt_inv(double x, double &r, double &inv)
{
    double y = __builtin_amdgcn_rsq(x);
    double h = 0.5 * x;
    y = y * __builtin_fma(-h * y, y, 1.5);
    y = y * __builtin_fma(-h * y, y, 1.5);
    double s = x * y;                                   // one correction of the root itself
    s = __builtin_fma(__builtin_fma(-s, s, x), 0.5 * y, s);
    r = s; inv = y;
}

// Q [n, ldq] row-major: Q[i][c] = component i of eigenvector c;  Qt its transpose;  lam [n] (clamped at zero: the matrix is
// positive semi-definite up to rounding).  status[0] |= 1 when a QL iteration did not converge in 60 sweeps.
// Dynamic LDS: (n ROWS + 2 n) doubles.
template <typename T, int ROWS>
__global__ void __launch_bounds__(64)
eig_ql_rows_kernel(int n, const double *__restrict__ W, const double *__restrict__ d_in, const double *__restrict__ e_in,
                   const double *__restrict__ tau, T *__restrict__ Q, T *__restrict__ Qt, size_t ldq, T *__restrict__ lam,
                   int *__restrict__ status)
{
    extern __shared__ double eig_sm[];
    double *Z = eig_sm;                     // [n][ROWS]: element c of the lane's row at Z[c ROWS + lane]
    double *d = Z + (size_t)n * ROWS;
    double *e = d + n;
    const int lane = threadIdx.x;
    const int zl = (ROWS == 64) ? lane : (lane & (ROWS - 1));       // (ROWS = 32: the upper half-wavefront mirrors the lower one)
    const int r = blockIdx.x * ROWS + zl;
    const bool writer = lane < ROWS;
    if (writer)
        for (int c = 0; c < n; c++) Z[c * ROWS + zl] = (c == r) ? 1.0 : 0.0;
    for (int c = lane; c < n; c += 64) { d[c] = d_in[c]; e[c] = e_in[c]; }
    __syncthreads();
    // ---- row r of Q1 = H(0) H(1) .. H(n-3): the unit row times the reflectors in order ----
    for (int i = 0; i + 2 < n; i++) {
        const double t = tau[i];
        if (t == 0.0) continue;
        const double *v = W + (size_t)i * n + i + 1;    // v[0] = 1 implied
        const int L = n - i - 1;
        double *z = Z + (size_t)(i + 1) * ROWS + zl;
        double a0 = z[0], a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int c = 1;
        for (; c + 3 < L; c += 4) {
            a0 = __builtin_fma(z[(size_t)c * ROWS], v[c], a0);
            a1 = __builtin_fma(z[(size_t)(c + 1) * ROWS], v[c + 1], a1);
            a2 = __builtin_fma(z[(size_t)(c + 2) * ROWS], v[c + 2], a2);
            a3 = __builtin_fma(z[(size_t)(c + 3) * ROWS], v[c + 3], a3);
        }
        for (; c < L; c++) a0 = __builtin_fma(z[(size_t)c * ROWS], v[c], a0);
        const double dot = t * ((a0 + a1) + (a2 + a3));
        if (writer) {
            z[0] -= dot;
            for (c = 1; c < L; c++) z[(size_t)c * ROWS] = __builtin_fma(-dot, v[c], z[(size_t)c * ROWS]);
        }
    }
    __syncthreads();
    // ---- implicit QL with shifts on (d, e), every rotation applied to the lane's row (EISPACK tql2) ----
    const double eps = 2.220446049250313e-16;
    double f = 0.0, tst1 = 0.0;
    bool failed = false;
    for (int l = 0; l < n; l++) {
        tst1 = fmax(tst1, fabs(d[l]) + fabs(e[l]));
        int m = l;
        while (m < n - 1 && fabs(e[m]) > eps * tst1) m++;
        if (m > l) {
            int iter = 0;
            do {
                iter++;
                double g = d[l];
                const double el = e[l];
                double p = (d[l + 1] - g) / (2.0 * el);
                double rr = sqrt(__builtin_fma(p, p, 1.0));
                if (p < 0.0) rr = -rr;
                const double dl = el / (p + rr);
                const double dl1 = el * (p + rr);
                const double h0 = g - dl;
                __syncthreads();
                if (lane == 0) { d[l] = dl; d[l + 1] = dl1; }
                for (int i = l + 2 + lane; i < n; i += 64) d[i] -= h0;
                __syncthreads();
                f += h0;
                p = d[m];
                double c = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0;
                const double el1 = e[l + 1];
                double e_i = e[m - 1], d_i = d[m - 1];
                double *zp = Z + zl;
                double z_hi = zp[(size_t)m * ROWS];          // element i + 1 of the row, carried from rotation to rotation
                for (int i = m - 1; i >= l; i--) {
                    const double e_n = (i > l) ? e[i - 1] : 0.0, d_n = (i > l) ? d[i - 1] : 0.0;     // the next trip's operands, early
                    const double z_lo = zp[(size_t)i * ROWS];
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e_i;
                    const double h = c * p;
                    const double x = __builtin_fma(p, p, e_i * e_i);
                    double rt = 0.0, inv = 0.0;
                    if (x > 0.0) eig_sqrt_inv(x, rt, inv);
                    const double e_out = s * rt;
                    s = (x > 0.0) ? e_i * inv : 0.0;
                    c = (x > 0.0) ? p * inv : 1.0;
                    p = __builtin_fma(c, d_i, -s * g);
                    const double d_out = __builtin_fma(s, __builtin_fma(c, g, s * d_i), h);
                    if (lane == 0) { e[i + 1] = e_out; d[i + 1] = d_out; }
                    if (writer) zp[(size_t)(i + 1) * ROWS] = __builtin_fma(s, z_lo, c * z_hi);
                    z_hi = __builtin_fma(c, z_lo, -s * z_hi);
                    e_i = e_n; d_i = d_n;
                }
                if (writer) zp[(size_t)l * ROWS] = z_hi;
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                __syncthreads();
                if (lane == 0) { e[l] = s * p; d[l] = c * p; }
                __syncthreads();
                if (iter >= 60) { failed = true; break; }
            } while (fabs(e[l]) > eps * tst1);
        }
        __syncthreads();
        if (lane == 0) { d[l] = d[l] + f; e[l] = 0.0; }
        __syncthreads();
    }
    if (failed && lane == 0 && blockIdx.x == 0) atomicOr(status, 1);
    if (writer && r < n)
        for (int c = 0; c < n; c++) {
            const T val = (T)Z[(size_t)c * ROWS + zl];
            Qt[(size_t)c * ldq + r] = val;
            Q[(size_t)r * ldq + c] = val;
        }
    if (blockIdx.x == 0)
        for (int c = lane; c < n; c += 64) lam[c] = (T)fmax(d[c], 0.0);
}

}  // namespace cmfhip
