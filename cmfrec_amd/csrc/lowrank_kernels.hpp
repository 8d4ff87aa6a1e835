// lowrank_kernels.hpp -- closed-form ALS row updates for rows with FEW entries against MANY unknowns (gfx950).
//
// The reference solves every row of the collective model by a k_t x k_t Cholesky factorisation
// (collective_closed_form_block, /root/reference/src/collective.c:1534-1846: k_t^3/3 flops per row whatever the row
// holds).  BASELINE config 5 has 20 entries per user against k_t = 257 unknowns: the row's matrix is
//        M_i = w C^T C (+) 0  +  diag(lam_i, .., lam_i, lam_last_i)  +  sum_{j in row} v_j v_j^T ,   v_j = [0; B_j]
// i.e. a matrix shared by all rows, a row-dependent multiple of the identity, and a rank-s update with s << k_t.
// With the eigen-decomposition  w C^T C = Q L Q^T  (once per half-step, jacobi_eig_kernel) and the unknowns rotated by
// Q~ = blockdiag(Q, I), the shared part is DIAGONAL for every row,  D_i = L (+) 0 + diag(lam_i .. lam_last_i),  and
// Woodbury's identity leaves an s x s system per row:
//        t = D^-1 r~ ,   G = I + V~ D^-1 V~^T  (s x s) ,   z = G^-1 (V~ t) ,   x~ = t - D^-1 V~^T z ,   x = Q~ x~
// with V~ = the rows of  B~ = [B(:, :k) Q(k_user:, :) | B(:, k:)]  selected by the row's entries and
// r~ = w U (C Q) + sum_j x_j v~_j.  Cost per row: O(s k_t) bytes gathered twice and (k_t / 4) s (s+16) / 512 MFMAs
// instead of k_t^3 / 3 flops: 1.56 M users of config 5's shard in ~30 ms instead of 1.5-3.9 s.
// Same solution as the reference's up to rounding (parity is tolerance-based: tests/test_gpu_config_widths.py).
// Without side information (plain factors_closed_form rows, common.c:978-1070) Q = I and nothing is rotated.
//
// lowrank_rows_kernel: one wavefront per row (s <= 16 NB entries), G in MFMA accumulator tiles, factorised in the
// wave's registers with the blocked scheme of chol_wave_kernels.hpp.
#pragma once
#include "chol_wave_kernels.hpp"

namespace cmfhip {

// ------------------------------------------------------------------------------------------------------------------
// Eigen-decomposition of a symmetric positive semi-definite n x n matrix by one-sided (Hestenes) Jacobi rotations in
// double precision: W = A V is driven to orthogonal columns, then A v_c = lambda_c v_c.  One workgroup of 16
// wavefronts; a tournament round pairs every column exactly once, a wavefront rotates its pairs (columns are
// contiguous: W, V are column-major), one barrier per round.  n = 256: ~7 sweeps x 255 rounds to tol = 1e-9 (what a
// single-precision system can see; 1e-13 for double: two more sweeps, the convergence is quadratic).  The launch runs on
// the auxiliary stream beside the kernels that do not need its result (session.hip).
// A [n, n] row-major symmetric (any precision T); W, V: n x n doubles of workspace each;
// Q [n, ldq] row-major: Q[i][c] = component i of eigenvector c;  Qt its transpose;  lam [n].
template <typename T>
__global__ void __launch_bounds__(1024)
jacobi_eig_kernel(const T *__restrict__ A, int n, double *W, double *V, T *__restrict__ Q,
                  T *__restrict__ Qt, size_t ldq, T *__restrict__ lam, int max_sweeps, double tol)
{
    __shared__ double s_off[16];
    __shared__ double s_nmax[16];
    __shared__ int s_done;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NE = 5;                       // column elements per lane: n <= 320
    double dmax = 0.0;
    for (int e = tid; e < n * n; e += 1024) {
        const int c = e / n, i = e % n;
        const double a = (double)A[(size_t)i * n + c];
        W[e] = a;
        V[e] = (i == c) ? 1.0 : 0.0;
        if (i == c) dmax = fmax(dmax, fabs(a));
    }
    dmax = lanes::wave_sum(dmax);               // an upper bound of the largest diagonal entry is all that is needed
    if (lane == 0) s_nmax[wave] = dmax;
    if (tid == 0) s_done = 0;
    __syncthreads();
    double scale = 0.0;
    for (int w = 0; w < 16; w++) scale += s_nmax[w];
    // columns of W = A V whose norm is below this are the null space of A (rank-deficient C^T C when p < k): their
    // mutual angles are rounding noise, rotating them never converges and changes nothing that is used
    const double null2 = (scale * 1e-14) * (scale * 1e-14);
    const int m = (n + 1) & ~1;                 // players of the tournament (a dummy one when n is odd)
    const int npairs = m / 2;
    for (int sweep = 0; sweep < max_sweeps; sweep++) {
        double off = 0.0;
        for (int round = 0; round < m - 1; round++) {
            // two pairs per trip: the loads of the second are in flight while the first is rotated
            for (int pi0 = wave; pi0 < npairs; pi0 += 32) {
                int pp[2], qq[2];
                bool live[2];
                double wp[2][NE], wq[2][NE], vp[2][NE], vq[2][NE];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int pi = pi0 + 16 * u;
                    int p, q;
                    if (pi == 0) { p = m - 1; q = round; }
                    else { p = (round + pi) % (m - 1); q = (round - pi + (m - 1)) % (m - 1); }
                    live[u] = (pi < npairs) && p < n && q < n;
                    if (p > q) { const int t2 = p; p = q; q = t2; }
                    pp[u] = live[u] ? p : 0; qq[u] = live[u] ? q : 0;
#pragma unroll
                    for (int e = 0; e < NE; e++) {
                        const int i = lane + 64 * e;
                        const bool ok = live[u] && i < n;
                        wp[u][e] = ok ? W[(size_t)pp[u] * n + i] : 0.0; wq[u][e] = ok ? W[(size_t)qq[u] * n + i] : 0.0;
                        vp[u][e] = ok ? V[(size_t)pp[u] * n + i] : 0.0; vq[u][e] = ok ? V[(size_t)qq[u] * n + i] : 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    if (!live[u]) continue;
                    double a = 0.0, b = 0.0, g = 0.0;
#pragma unroll
                    for (int e = 0; e < NE; e++) { a += wp[u][e] * wp[u][e]; b += wq[u][e] * wq[u][e]; g += wp[u][e] * wq[u][e]; }
                    a = lanes::wave_sum(a); b = lanes::wave_sum(b); g = lanes::wave_sum(g);
                    if (!(a > null2) || !(b > null2)) continue;              // a null-space column: leave it
                    const double rel = fabs(g) / sqrt(a * b);
                    off = fmax(off, rel);
                    if (rel < 0.01 * tol) continue;
                    const double zeta = (b - a) / (2.0 * g);
                    const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                    const int p = pp[u], q = qq[u];
#pragma unroll
                    for (int e = 0; e < NE; e++) {
                        const int i = lane + 64 * e;
                        if (i < n) {
                            W[(size_t)p * n + i] = c * wp[u][e] - s * wq[u][e]; W[(size_t)q * n + i] = s * wp[u][e] + c * wq[u][e];
                            V[(size_t)p * n + i] = c * vp[u][e] - s * vq[u][e]; V[(size_t)q * n + i] = s * vp[u][e] + c * vq[u][e];
                        }
                    }
                }
            }
            __syncthreads();
        }
        if (lane == 0) s_off[wave] = off;
        __syncthreads();
        if (tid == 0) {
            double mx = 0.0;
            for (int w = 0; w < 16; w++) mx = fmax(mx, s_off[w]);
            s_done = (mx < tol) ? 1 : 0;
        }
        __syncthreads();
        if (s_done) break;
    }
    // lambda_c = v_c . (A v_c) = v_c . W_c ;  Q[i][c] = V[c][i]
    for (int c = wave; c < n; c += 16) {
        double d = 0.0;
        for (int i = lane; i < n; i += 64) d += V[(size_t)c * n + i] * W[(size_t)c * n + i];
        d = lanes::wave_sum(d);
        if (lane == 0) lam[c] = (T)fmax(d, 0.0);
        for (int i = lane; i < n; i += 64) {
            Q[(size_t)i * ldq + c] = (T)V[(size_t)c * n + i];
            Qt[(size_t)c * ldq + i] = (T)V[(size_t)c * n + i];
        }
    }
}

// dst[r, c0 + c] = src[r, s0 + c]  (the un-rotated columns of the opposing factors: k_main, bias)
template <typename T>
__global__ void copy_cols_kernel(T *__restrict__ dst, size_t ldd, int c0, const T *__restrict__ src, size_t lds, int s0, int ncols, size_t rows)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * (size_t)ncols) return;
    const size_t r = e / ncols; const int c = (int)(e % ncols);
    dst[r * ldd + c0 + c] = src[r * lds + s0 + c];
}

// A[order[first + r], :ncols] = src[r, :ncols]  (the back-rotated unknowns of the low-rank rows return to their rows)
template <typename T>
__global__ void scatter_rows_kernel(T *__restrict__ A, size_t lda, const RowDesc *__restrict__ desc, int first, const T *__restrict__ src,
                                    size_t lds, int ncols, size_t rows)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * (size_t)ncols) return;
    const size_t r = e / ncols; const int c = (int)(e % ncols);
    A[(size_t)desc[first + r].row * lda + c] = src[r * lds + c];
}

template <typename T>
struct LrParams {
    T *A; size_t lda;                 // out: x~[kc, kt) of the row (or all of x when !rotated)
    const T *pre; size_t ldpre;       // prefilled right-hand side (rotated basis), row r at pre + r * ldpre, [0, kc); or null
    T *Tc; size_t ldt;                // rotated: x~[0, kc) of the row at position rix goes to Tc[(rix - pos0) * ldt ..]
    int pos0;                         // first position of the processing order that owns a row of Tc
    const T *Bt; size_t ldbt;         // opposing factors in the rotated basis, [*, kt] (ldbt a multiple of 4); !rotated: columns [koff, kt)
    const T *lam_eig;                 // eigenvalues of w C^T C [kc], or null (no side information: D = lam)
    int kt, kc, koff;
    int rotated;                      // 1: rows live in the rotated basis (Bt has kt columns); 0: Bt = B, unknown u <-> column u - koff
    const size_t *indptr; const int *indices; const T *values; const T *bias_sub;
    T lam, lam_last;
    int scale_lam, scale_lam_sideinfo, scale_bias_const, p_side;
    int collective;                   // lambda scaling rules of collective_closed_form_block (else factors_closed_form)
    int row_first, nrows;             // positions [row_first, nrows) of the processing order
    int *counter;
};

// NB: 16-blocks of the s x s system (s <= 16 NB entries per row); WPS: wavefronts per SIMD.
template <typename T, int NB, int WPS>
__global__ void __launch_bounds__(256, WPS)
lowrank_rows_kernel(const LrParams<T> P, const RowDesc *__restrict__ desc)
{
    using Mf = CholMfma<T>;
    using vec = typename Mf::vec;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    constexpr int NT = NB * (NB + 1) / 2;
    constexpr int LDR = Mf::LDR, RSZ = 16 * LDR;
    constexpr int NV = 16 * NB + 16;
    constexpr int KTMAX = 320;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lm = lane & 15, g = lane >> 4;
    constexpr size_t PERW = (size_t)NB * RSZ + 2 * NV + 2 * KTMAX + 16 * NB * 2;
    T *wbase = reinterpret_cast<T *>(smem_raw) + (size_t)wave * PERW;
    T *rinv = wbase;                         // [NB][16][LDR]
    T *yv0 = rinv + (size_t)NB * RSZ;        // [NV]  q -> y -> back-substituted
    T *xall = yv0 + NV;                      // [NV]  z
    T *tvec = xall + NV;                     // [KTMAX] t = D^-1 r~
    T *dtab = tvec + KTMAX;                  // [KTMAX] 1 / D of the row (0 beyond k_t)
    int *idxs = reinterpret_cast<int *>(dtab + KTMAX);   // [16 NB] entry -> opposing row
    T *xent = reinterpret_cast<T *>(idxs + 16 * NB);     // [16 NB] entry values (x_j - bias_j)

    const int kt = P.kt, kc = P.kc;
    const int ngroups = (kt + 15) >> 4;      // 16-column groups
    const int nwaves = gridDim.x * 4;
    int rix = P.row_first + blockIdx.x * 4 + wave;
    for (;;) {
        if (rix >= P.nrows) break;
        int claim = 0;
        if (lane == 0) claim = atomicAdd(P.counter, 1);
        RowDesc d = desc[rix];
        const int row = __builtin_amdgcn_readfirstlane(d.row);
        const int s = __builtin_amdgcn_readfirstlane(d.nnz);
        const size_t st = ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(d.st >> 32)) << 32) |
                          (unsigned)__builtin_amdgcn_readfirstlane((int)(d.st & 0xffffffffu));
        const int nb = (s + 15) >> 4;
        T *arow = P.A + (size_t)row * P.lda;
        T lam = P.lam, lam_last = P.lam_last;
        if (!P.collective) {
            if (P.scale_lam) {                                           // common.c:679-723
                lam *= (T)s;
                if (!P.scale_bias_const) lam_last *= (T)s;
            }
        } else if (P.scale_lam || P.scale_lam_sideinfo) {                // collective.c:1285-1355 (rows with side information)
            T mult = (s > 0) ? (T)s : T(1);
            if (P.scale_lam_sideinfo) mult += (T)P.p_side;
            lam *= mult; lam_last *= mult;
        }
        // entries of the row: opposing row ids and values, through LDS (two lane layouts read them)
        for (int a = lane; a < 16 * NB; a += 64) {
            const bool ok = a < s;
            const int ix = ok ? P.indices[st + a] : 0;
            T x = ok ? P.values[st + a] : T(0);
            if (ok && P.bias_sub != nullptr) x -= P.bias_sub[ix];
            idxs[a] = ix; xent[a] = x;
        }
        // 1 / D_u once per row (round 2 divided in every 16-column group of both sweeps: 136 IEEE divisions per lane and row, a
        // third of the kernel's instructions)
        for (int u = lane; u < 16 * ngroups; u += 64) {
            T dd = (u == kt - 1) ? lam_last : lam;
            if (P.lam_eig != nullptr && u < kc) dd += P.lam_eig[u];
            dtab[u] = (u < kt) ? T(1) / dd : T(0);
        }
        CMF_LDS_FENCE();
        int my_idx[NB]; T xl[NB]; unsigned amask = 0;
#pragma unroll
        for (int b = 0; b < NB; b++) {
            my_idx[b] = idxs[16 * b + lm]; xl[b] = xent[16 * b + lm];
            amask |= (16 * b + lm < s) ? (1u << b) : 0u;
        }
        vec acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = vec{0, 0, 0, 0};
        static_for<0, NB>([&](auto bic) {                    // G starts as the identity
            constexpr int bi = decltype(bic)::value;
#pragma unroll
            for (int r = 0; r < 4; r++) acc[wtix(bi, bi, NB)][r] = (Mf::row_of(lane, r) == lm) ? T(1) : T(0);
        });
        T qp[NB];                                            // q = V~ t: partial over the lane groups for entry 16 b + lm
#pragma unroll
        for (int b = 0; b < NB; b++) qp[b] = T(0);
        // One sweep over the row's gathered data in 16-column groups: lane (lm, g) reads columns 16 q + 4 g .. + 3 of the
        // opposing rows of the entries 16 b + lm (16-byte loads, 64-byte segments per row).
        //   pass 0:  tot_c = sum_a x_a V~[a][c] ;  t_c = (pre_c + tot_c) / D_c ;  G += MFMA ;  q_a += V~[a][c] t_c
        //   pass 1:  tot_c = sum_a z_a V~[a][c] ;  x~_c = t_c - tot_c / D_c
        // FULL: the group lies inside [0, min(k_t, k_c)) of the rotated basis -- every group but the last one or two of a row:
        // no column masks, the prefilled right-hand side, t and x~ move as 16-byte vectors
        typedef T vec4u __attribute__((ext_vector_type(4), aligned(sizeof(T))));
        // (only in the build for the shortest rows, which is most rows: with both forms of the group in one kernel the wider
        // builds spill)
        const int full_lim = (NB <= 2 && P.rotated) ? min(kt, kc) : 0;
        auto group = [&](auto pass_tag, auto full_tag, int qg, const T (&wl)[NB]) {
            constexpr int PASS = decltype(pass_tag)::value;
            constexpr bool FULL = decltype(full_tag)::value;
            const int c0 = 16 * qg + 4 * g;              // this lane's four columns (unknowns)
            vec4 val[NB];
#pragma unroll
            for (int b = 0; b < NB; b++) {
                // (the last group may hang over the row's end: the load stays inside the padded leading dimension)
                const bool rot = FULL || P.rotated;
                const T *src = P.Bt + (size_t)my_idx[b] * P.ldbt + (rot ? c0 : c0 - P.koff);
                if (rot) val[b] = *reinterpret_cast<const vec4 *>(src);
                else {
#pragma unroll
                    for (int r = 0; r < 4; r++) { const int u = c0 + r; val[b][r] = (u >= P.koff && u < kt) ? src[r] : T(0); }
                }
            }
            T dinv[4], pre[4];
            const vec4 dq = *reinterpret_cast<const vec4 *>(dtab + c0);
            if (FULL && PASS == 0 && P.pre != nullptr) {
                const vec4 pq = *reinterpret_cast<const vec4u *>(P.pre + (size_t)row * P.ldpre + c0);
#pragma unroll
                for (int r = 0; r < 4; r++) pre[r] = pq[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int u = c0 + r;
                    pre[r] = (!FULL && PASS == 0 && P.pre != nullptr && u < kc) ? P.pre[(size_t)row * P.ldpre + min(u, kc - 1)] : T(0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) dinv[r] = dq[r];
            T o[NB][4], part[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    o[b][r] = (((amask >> b) & 1u) && (FULL || c0 + r < kt)) ? val[b][r] : T(0);
                    part[r] += o[b][r] * wl[b];
                }
            // sum over the 16 lanes of the group (all 16 end with the total)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                T v = part[r];
                v += lanes::xor1(v); v += lanes::xor2(v); v += lanes::qxor4(v); v += lanes::xor8(v);
                part[r] = v;
            }
            if (PASS == 0) {
                vec4 tt;
#pragma unroll
                for (int r = 0; r < 4; r++) tt[r] = (pre[r] + part[r]) * dinv[r];
                if (lm == 0) {
                    if (FULL) *reinterpret_cast<vec4 *>(tvec + c0) = tt;
                    else {
#pragma unroll
                        for (int r = 0; r < 4; r++) if (c0 + r < kt) tvec[c0 + r] = tt[r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    T ao[NB];
#pragma unroll
                    for (int b = 0; b < NB; b++) { ao[b] = o[b][r] * dinv[r]; qp[b] += o[b][r] * tt[r]; }
                    static_for<0, NB>([&](auto bic) {
                        constexpr int bi = decltype(bic)::value;
                        static_for<bi, NB>([&](auto bjc) {
                            constexpr int bj = decltype(bjc)::value;
                            acc[wtix(bi, bj, NB)] = Mf::mma(ao[bi], o[bj][r], acc[wtix(bi, bj, NB)]);
                        });
                    });
                }
            } else if (lm == 0) {
                if (FULL) {
                    const vec4 tv = *reinterpret_cast<const vec4 *>(tvec + c0);
                    vec4 xv;
#pragma unroll
                    for (int r = 0; r < 4; r++) xv[r] = tv[r] - part[r] * dinv[r];
                    *reinterpret_cast<vec4u *>(P.Tc + (size_t)(rix - P.pos0) * P.ldt + c0) = xv;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int u = c0 + r;
                        if (u < kt) {
                            const T xv = tvec[u] - part[r] * dinv[r];
                            if (P.rotated && u < kc) P.Tc[(size_t)(rix - P.pos0) * P.ldt + u] = xv;
                            else arow[u] = xv;
                        }
                    }
                }
            }
        };
        auto sweep = [&](auto pass_tag, const T (&wl)[NB]) {
            for (int qg = 0; qg < ngroups; qg++) {
                if constexpr (NB <= 2) {
                    if (16 * qg + 16 <= full_lim) { group(pass_tag, std::true_type{}, qg, wl); continue; }
                }
                group(pass_tag, std::false_type{}, qg, wl);
            }
        };
        sweep(std::integral_constant<int, 0>{}, xl);
        CMF_LDS_FENCE();
        // ---- G z = q: blocked Cholesky in this wave's registers (chol_wave_kernels.hpp, steps 3 and 4; the right-hand
        //      side is forward-substituted on the vector ALU) ----
        for (int kb = 0; kb < nb; kb++) {
            T *rslot = rinv + (size_t)kb * RSZ;
            vec dd = vec{0, 0, 0, 0};
            static_for<0, NB>([&](auto kc_) {
                constexpr int KB = decltype(kc_)::value;
                if (kb == KB) dd = acc[wtix(KB, KB, NB)];
            });
            chol_diag_block<T>(dd, rslot, lane, 16);
            CMF_LDS_FENCE();
            T ainv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) ainv[r] = rslot[Mf::row_of(lane, r) * LDR + lm];
            T vk = T(0);
            static_for<0, NB>([&](auto kc_) {
                constexpr int KB = decltype(kc_)::value;
                if (kb == KB) vk = qp[KB];
            });
            vk = lanes::tswap16_add(vk, vk); vk = lanes::tswap32_add(vk, vk);
            if (lane < 16) yv0[16 * kb + lane] = vk;
            CMF_LDS_FENCE();
            T y0 = T(0);
#pragma unroll
            for (int c = 0; c < 16; c++) y0 += rslot[c * LDR + lm] * yv0[16 * kb + c];
            CMF_LDS_FENCE();
            if (lane < 16) yv0[16 * kb + lane] = y0;
            CMF_LDS_FENCE();
            T yq[4];
#pragma unroll
            for (int r = 0; r < 4; r++) yq[r] = yv0[16 * kb + Mf::row_of(lane, r)];
            static_for<0, NB>([&](auto kc_) {
                constexpr int KB = decltype(kc_)::value;
                if (kb == KB) {
                    static_for<KB + 1, NB>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        vec x = Mf::mma(ainv[0], acc[wtix(KB, j, NB)][0], vec{0, 0, 0, 0});
                        x = Mf::mma(ainv[1], acc[wtix(KB, j, NB)][1], x);
                        x = Mf::mma(ainv[2], acc[wtix(KB, j, NB)][2], x);
                        x = Mf::mma(ainv[3], acc[wtix(KB, j, NB)][3], x);
                        acc[wtix(KB, j, NB)] = x;
                        T s0 = T(0);
#pragma unroll
                        for (int r = 0; r < 4; r++) s0 += x[r] * yq[r];
                        qp[j] -= s0;
                    });
                    static_for<KB + 1, NB>([&](auto bic) {
                        constexpr int bi = decltype(bic)::value;
                        T na[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) na[r] = -acc[wtix(KB, bi, NB)][r];
                        static_for<bi, NB>([&](auto bjc) {
                            constexpr int bj = decltype(bjc)::value;
#pragma unroll
                            for (int r = 0; r < 4; r++)
                                acc[wtix(bi, bj, NB)] = Mf::mma(na[r], acc[wtix(KB, bj, NB)][r], acc[wtix(bi, bj, NB)]);
                        });
                    });
                }
            });
        }
        for (int bjk = nb - 1; bjk >= 0; bjk--) {
            const T *rslot = rinv + (size_t)bjk * RSZ;
            T xm = T(0);
{   // (round 6: the block of the right-hand side once, its elements broadcast inside the 16-lane row by DPP instead of sixteen more LDS reads)
                const T yb = yv0[16 * bjk + lm];
                static_for<0, 16>([&](auto nc) {
                    constexpr int n2 = decltype(nc)::value;
                    xm += rslot[lm * LDR + n2] * lanes::row_bcast16<n2>(yb);
                });
            }
            if (lane < 16) xall[16 * bjk + lane] = xm;
            CMF_LDS_FENCE();
            static_for<0, NB>([&](auto jc) {
                constexpr int BJ = decltype(jc)::value;
                if (bjk == BJ) {
                    static_for<0, BJ>([&](auto ic) {
                        constexpr int bi = decltype(ic)::value;
                        const vec tl = acc[wtix(bi, BJ, NB)];
                        const T p0 = tl[0] * xm, p1 = tl[1] * xm, p2 = tl[2] * xm, p3 = tl[3] * xm;
                        const bool o1 = (lm & 1) != 0, o2 = (lm & 2) != 0;
                        const T s01 = (o1 ? p1 : p0) + lanes::xor1(o1 ? p0 : p1);
                        const T s23 = (o1 ? p3 : p2) + lanes::xor1(o1 ? p2 : p3);
                        T sr = (o2 ? s23 : s01) + lanes::xor2(o2 ? s01 : s23);
                        sr += lanes::xor4(sr);
                        sr += lanes::xor8(sr);
                        if (lm < 4) yv0[16 * bi + Mf::row_of(lane, lm)] -= sr;
                    });
                }
            });
            CMF_LDS_FENCE();
        }
        T zl[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) zl[b] = (b < nb) ? xall[16 * b + lm] : T(0);
        sweep(std::integral_constant<int, 1>{}, zl);
        CMF_LDS_FENCE();
        rix = P.row_first + nwaves + __builtin_amdgcn_readfirstlane(claim);
    }
}

template <typename T>
__host__ __device__ constexpr size_t lowrank_lds_elems(int NB)
{
    return (size_t)NB * 16 * CholMfma<T>::LDR + 2 * (16 * (size_t)NB + 16) + 2 * 320 + 16 * (size_t)NB * 2;
}

}  // namespace cmfhip
