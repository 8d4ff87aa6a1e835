// eig_kernels.hpp -- symmetric eigen-decomposition of the small shared matrix of the low-rank row path (gfx950), round 6.
//
// The low-rank closed form (lowrank_kernels.hpp) needs  w C^T C = Q L Q^T  once per half-step (k_c x k_c, 64 .. 320; config 5:
// 256 / 257).  Rounds 3-5 bound rocSOLVER's dsyevd at run time: a chain of ~4000 small launches (its tridiagonalisation is
// launch-bound: 6 ms on an idle device, 33 ms beside the persistent batches of the long rows -- profiles/r05/r05_zg_*), and the
// one-workgroup Jacobi kernel (91 ms) as the fallback.  This file is the library's own decomposition in TWO launches:
//
//   eig_tridiag_kernel   one workgroup of 8 wavefronts: Householder tridiagonalisation  A = Q1 T Q1^T  (the scheme of LAPACK's
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
constexpr int EIG_TRIDIAG_THREADS = 512;          // 8 wavefronts, two per SIMD: 256 registers each for the rows a trip keeps in flight
template <typename T>
__global__ void __launch_bounds__(EIG_TRIDIAG_THREADS)
eig_tridiag_kernel(const T *__restrict__ A, int n, double *W, double *__restrict__ d, double *__restrict__ e, double *__restrict__ tau)
{
    __shared__ double s_v[EIG_MAX_N], s_p[EIG_MAX_N];
    __shared__ double s_tau;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NQ = EIG_MAX_N / 64, RB = 12, NW = EIG_TRIDIAG_THREADS / 64;
    for (int x = tid; x < n * n; x += EIG_TRIDIAG_THREADS) W[x] = (double)A[x];
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
            // A wavefront takes the rows wave, wave + NW, ..; RB of them per trip with all their loads in flight before the first
            // product (the matrix lives in L2: a trip is one memory latency, not RB of them)
            double vv[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) { const int c = lane + 64 * q; vv[q] = (c < L) ? s_v[c] : 0.0; }
            for (int r0 = wave; r0 < L; r0 += NW * RB) {        // p = t A22 v
                double a[RB][NQ];
#pragma unroll
                for (int b = 0; b < RB; b++) {
                    const int r = r0 + NW * b;
                    const double *ar = W + (size_t)(i + 1 + min(r, L - 1)) * n + i + 1;
#pragma unroll
                    for (int q = 0; q < NQ; q++) { const int c = lane + 64 * q; a[b][q] = (r < L && c < L) ? ar[c] : 0.0; }
                }
#pragma unroll
                for (int b = 0; b < RB; b++) {
                    double acc = 0.0;
#pragma unroll
                    for (int q = 0; q < NQ; q++) acc = __builtin_fma(a[b][q], vv[q], acc);
                    acc = lanes::wave_sum(acc);
                    if (lane == 0 && r0 + NW * b < L) s_p[r0 + NW * b] = t * acc;
                }
            }
            __syncthreads();
            double dot = 0.0;                           // w = p - (t / 2) (p . v) v ;  A22 -= v w^T + w v^T
            double wc[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) { const int c = lane + 64 * q; wc[q] = (c < L) ? s_p[c] : 0.0; dot = __builtin_fma(wc[q], vv[q], dot); }
            dot = lanes::wave_sum(dot);
            const double a2 = -0.5 * t * dot;
#pragma unroll
            for (int q = 0; q < NQ; q++) wc[q] = __builtin_fma(a2, vv[q], wc[q]);
            for (int r0 = wave; r0 < L; r0 += NW * RB) {
                double a[RB][NQ];
#pragma unroll
                for (int b = 0; b < RB; b++) {
                    const int r = r0 + NW * b;
                    const double *ar = W + (size_t)(i + 1 + min(r, L - 1)) * n + i + 1;
#pragma unroll
                    for (int q = 0; q < NQ; q++) { const int c = lane + 64 * q; a[b][q] = (r < L && c < L) ? ar[c] : 0.0; }
                }
#pragma unroll
                for (int b = 0; b < RB; b++) {
                    const int r = r0 + NW * b;
                    if (r < L) {
                        double *ar = W + (size_t)(i + 1 + r) * n + i + 1;
                        const double vr = s_v[r], wr = __builtin_fma(a2, vr, s_p[r]);
#pragma unroll
                        for (int q = 0; q < NQ; q++) {
                            const int c = lane + 64 * q;
                            if (c < L) ar[c] = a[b][q] - (vr * wc[q] + wr * vv[q]);
                        }
                    }
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
// Dynamic LDS: (n ROWS + 4 n) doubles.
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
    double *sv = e + n;                     // [2][n]: the reflector of the step in hand (broadcast reads) and the next one
    const int lane = threadIdx.x;
    const int zl = (ROWS == 64) ? lane : (lane & (ROWS - 1));       // (ROWS = 32: the upper half-wavefront mirrors the lower one)
    const int r = blockIdx.x * ROWS + zl;
    const bool writer = lane < ROWS;
    if (writer)
        for (int c = 0; c < n; c++) Z[c * ROWS + zl] = (c == r) ? 1.0 : 0.0;
    for (int c = lane; c < n; c += 64) { d[c] = d_in[c]; e[c] = e_in[c]; }
    __syncthreads();
    // ---- row r of Q1 = H(0) H(1) .. H(n-3): the unit row times the reflectors in order ----
    // (the reflector of step i + 1 is on its way from L2 while step i is applied: two buffers in LDS)
    constexpr int NQ = EIG_MAX_N / 64;
    double vpre[NQ];
    auto fetch = [&](int i) {
        const double *v = W + (size_t)i * n + i + 1;
        const int L = n - i - 1;
#pragma unroll
        for (int q = 0; q < NQ; q++) { const int c = 1 + lane + 64 * q; vpre[q] = (i + 2 < n && c < L) ? v[c] : 0.0; }
    };
    auto stash = [&](int i, double *buf) {
        const int L = n - i - 1;
#pragma unroll
        for (int q = 0; q < NQ; q++) { const int c = 1 + lane + 64 * q; if (c < L) buf[c] = vpre[q]; }
    };
    fetch(0);
    stash(0, sv);
    __syncthreads();
    for (int i = 0; i + 2 < n; i++) {
        const double *svi = sv + (size_t)(i & 1) * n;
        fetch(i + 1);
        const double t = tau[i];
        if (t != 0.0) {                                 // (uniform)
            const int L = n - i - 1;
            double *z = Z + (size_t)(i + 1) * ROWS + zl;
            double a0 = z[0], a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int c = 1;
            for (; c + 3 < L; c += 4) {
                a0 = __builtin_fma(z[(size_t)c * ROWS], svi[c], a0);
                a1 = __builtin_fma(z[(size_t)(c + 1) * ROWS], svi[c + 1], a1);
                a2 = __builtin_fma(z[(size_t)(c + 2) * ROWS], svi[c + 2], a2);
                a3 = __builtin_fma(z[(size_t)(c + 3) * ROWS], svi[c + 3], a3);
            }
            for (; c < L; c++) a0 = __builtin_fma(z[(size_t)c * ROWS], svi[c], a0);
            const double dot = t * ((a0 + a1) + (a2 + a3));
            if (writer) {
                z[0] -= dot;
                for (c = 1; c < L; c++) z[(size_t)c * ROWS] = __builtin_fma(-dot, svi[c], z[(size_t)c * ROWS]);
            }
        }
        stash(i + 1, sv + (size_t)((i + 1) & 1) * n);
        __syncthreads();
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
                // Software pipeline: the operands of trip i - 1 (e, d and the row's element i - 1) are read from LDS at the top of
                // trip i and waited for at its end, behind the ~12 dependent double-precision operations of the rotation; the
                // stores are unconditional (every lane writes the same e / d: no exec-mask juggling in the loop).
                double *zp = Z + zl;
                double e_i = e[m - 1], d_i = d[m - 1], z_lo = zp[(size_t)(m - 1) * ROWS];
                double z_hi = zp[(size_t)m * ROWS];          // element i + 1 of the row, carried from rotation to rotation
                for (int i = m - 1; i >= l; i--) {
                    const int ip = max(i - 1, 0);            // (i == l: loaded and not used)
                    const double e_n = e[ip], d_n = d[ip], z_n = zp[(size_t)ip * ROWS];
                    __builtin_amdgcn_sched_barrier(0);       // (the three reads first: their latency belongs behind the whole rotation)
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e_i;
                    const double h = c * p;
                    // r = hypot(p, e_i) and 1 / r from one v_rsq_f64 + two Newton steps.  (x > 0: inside an unreduced block e_i is
                    //  above eps x the matrix's scale; the clamp only keeps a denormal-scale matrix free of NaN)
                    const double x = fmax(__builtin_fma(p, p, e_i * e_i), 1e-290);
                    double inv = __builtin_amdgcn_rsq(x);
                    const double hx = 0.5 * x;
                    inv = inv * __builtin_fma(-hx * inv, inv, 1.5);
                    inv = inv * __builtin_fma(-hx * inv, inv, 1.5);
                    const double e_out = s * (x * inv);
                    s = e_i * inv;
                    c = p * inv;
                    p = __builtin_fma(c, d_i, -s * g);
                    const double d_out = __builtin_fma(s, __builtin_fma(c, g, s * d_i), h);
                    e[i + 1] = e_out; d[i + 1] = d_out;
                    const double z_new = __builtin_fma(s, z_lo, c * z_hi);
                    if (ROWS == 64 || writer) zp[(size_t)(i + 1) * ROWS] = z_new;
                    z_hi = __builtin_fma(c, z_lo, -s * z_hi);
                    __builtin_amdgcn_sched_barrier(0);
                    e_i = e_n; d_i = d_n; z_lo = z_n;
                }
                if (writer) zp[(size_t)l * ROWS] = z_hi;
                p = -s * s2 * c3 * el1 * el / dl1;
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
