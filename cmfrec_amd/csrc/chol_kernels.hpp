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
// One workgroup (256 threads, 4 wavefronts) per row, persistent over rows; the k_t x k_t normal matrix
// never leaves the MFMA accumulators: its upper triangle is cut into 16x16 tiles, tile t of the packed
// order is owned by wave t & 3, in the C/D register layout of v_mfma_*_16x16x4.
//   1. rank-k update  G = sum_j w_j B_j B_j^T  (+ right-hand side) from the gathered rows, staged through
//      a two-slot LDS ring; indices are fetched two chunks ahead and rows one chunk ahead.
//   2. the initial matrix (lam*I | BtB+lam*I | w*CtC block) is added in the same layout.
//   3. blocked right-looking Cholesky  M = R^T R, 16 columns per step:
//        a. the owner wave of the diagonal tile factorises it and inverts the factor (column per lane,
//           v_readlane broadcasts), publishes inv(R_kk) in LDS;
//        b. panel tiles  X = inv(R_kk)^T * tile  and  c. trailing tiles  tile -= X_i^T X_j  are MFMAs whose
//           operands are exactly the C/D-layout registers of the tiles (the k index of a 16x16x4 step is
//           free to be permuted), exchanged between waves through LDS in lane-linear order;
//        the forward substitution rides along on the VALU.
//   4. blocked backward substitution: every wave applies its own tiles (DPP row reductions).
// Only the upper triangle is referenced, as in the reference (tposv_ 'L' on the column-major view
// == upper of the row-major one).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "lanes.hpp"

namespace cmfhip {

enum CholMode { CHOL_EXPLICIT = 0, CHOL_IMPLICIT = 1, CHOL_COLLECTIVE = 2,
                CHOL_COLLECTIVE_IMPLICIT = 4 /* implicit-feedback weights + side information:
                                      M = Mfull[kt,kt] (BeTBe without the C^T C block, lam included)
                                        + Minit[kc,kc] (w C^T C) for rows with side information,
                                      rhs = the prefilled w U C row + sum (x+1) B  (collective.c:1849-2131) */,
                CHOL_PREFILLED = 3 /* M = Minit[kt,kt] (diag included), rhs = the row itself, no gather:
                                      the multi-RHS posv of the C / D update, common.c:2872-2875 */,
                CHOL_NAZ = 5 /* missing-as-zero, unweighted (optimizeA Case 3, common.c:3100-3205): every row shares
                                M = Minit[kt,kt] (B^T B + diag), rhs = sum_j x_j B_j over the row's entries
                                (tgemm_sp_dense); rows without entries are left to the caller (zero) */,
                CHOL_NAZ_W = 6 /* missing-as-zero WITH observation weights (optimizeA Case 4 with NA_as_zero && weight,
                                  common.c:3209-3302; factors_closed_form :846-907): absent entries are zeros of weight one, so
                                  M_i = Minit[kt,kt] (B^T B, no diagonal) + sum_j (w_j - 1) B_j B_j^T + diag(lam_i .. lam_last_i),
                                  rhs_i = the prefilled row (bias_BtX or zero) + sum_j [w_j x_j - (w_j - 1)(mean + bias_j)] B_j.
                                  The caller hands over `weights` = w - 1 and `values` = the bracket, per entry (session.hip,
                                  naz_entry_transform_kernel); lam_i = lam x wsum[row] under scale_lam (the driver's sum of the
                                  row's weights + the number of its absent entries, collective.c:7991-8022).  TWO_SRC build. */ };

template <typename T>
struct CholParams {
    T *A; size_t lda;          // row r -> A + r*lda : the k_t unknowns (in/out)
    const T *B; size_t ldb;    // opposing factors, first used column; kb = kt - koff columns are used
    int kt;                    // unknowns per row
    int koff;                  // offset of the X-block inside the unknowns (k_user), 0 otherwise
    const size_t *indptr; const int *indices; const T *values;
    const T *bias_sub;         // x_j := x_j - bias_sub[idx_j], or null
    // explicit model with observation weights (common.c:985-1012: tgemv_dense_sp_weighted, syr with weight[ix]): one weight
    // per entry of X in the order of `values`, and the row's lambda multiplier under scale_lam (the driver's wsumA / wsumB,
    // collective.c:7978-8008).  Workgroup-per-row kernel only (launch_chol keeps weighted calls off the wave kernel).
    const T *weights = nullptr;
    const T *wsum = nullptr;
    const int *order; int nrows;
    const T *Minit;            // implicit: BtB + lam*I [kt,kt];  collective (both): w*CtC [kc,kc];  explicit: null
    const T *Mfull;            // collective implicit: [kt,kt] matrix every solved row starts from;  else null
    int kc;                    // collective: size of the side-info block (k_user + k), else 0
    // optional second gather source (sparse side information: the row's entries of U select rows of C):
    // after the nnz entries of X the row has indptr2[row+1] - indptr2[row] more entries (indices2, values2) that gather
    // rows of B2[*, ldb2] into the unknowns [0, kc2), weighted w2 (rank-1 terms) and w2 * value (right-hand side)
    const size_t *indptr2 = nullptr;
    const int *indices2 = nullptr;
    const T *values2 = nullptr;
    const T *B2 = nullptr;
    size_t ldb2 = 0;
    int kc2 = 0;
    T w2 = 0;
    // by default the second source lands on the unknowns [0, kc2) with rank-1 weight w2 (side information).  The
    // implicit-features term of the explicit model (collective.c:1704-1707, :1757-1771) uses it differently: rows of Bi
    // at the X block's offset, right-hand side only (w2_syr = 0: such chunks skip the matrix cores).
    int koff2 = 0;
    int w2_syr_zero = 0;
    int rows_src2 = -1;                   // rows that have second-source entries (default: rows_with_u)
    int rhs_prefilled_all = 0;            // every row's right-hand side starts from what the caller left in A (collective modes:
                                          // not only the rows with side information) -- the implicit-features term
    int rhs_only = 0;                     // CHOL_NAZ: store the gathered right-hand side and stop (the shared matrix is
                                          // factorised once by the caller, the solve is one triangular-solve pair)
    const T *values_override = nullptr;   // read the entries' values from here instead of `values` (all-ones indicator)
    int entry_pairs = 0;                  // `weights` / `values` hold the entry's rank-1 weight and right-hand-side weight as they are
                                          // (NA_as_zero_X with observation weights in the collective modes: w - 1 and the bracket)
    int x_rhs_only = 0;                   // the entries of X add to the right-hand side only (NA_as_zero_X in the collective modes: their
                                          // Gramian is the shared B^T B inside Mfull, collective.c:1631-1640).  TWO_SRC build.
    // non-negative factors: the assembled system is solved by the reference's cyclic coordinate descent instead of the
    // Cholesky factorisation (solve_nonneg, common.c:2131-2179), at most max_cd_steps sweeps
    int nonneg = 0;
    int max_cd_steps = 100;
    // L1 penalty (elastic net): l1 on every unknown, l1_last on the last one; scaled per row exactly like lam / lam_last.
    // With nonneg it shifts the right-hand side of solve_nonneg, without it the system goes through solve_elasticnet
    // (common.c:2228-2294: the same descent on a positive and a negative part, a = a+ - a-).
    T l1 = 0, l1_last = 0;
    int rows_with_u;           // collective: rows < rows_with_u carry side information
    int p_side;                // collective: number of side-info columns (scale_lam_sideinfo)
    T lam, lam_last;
    int scale_lam, scale_lam_sideinfo, scale_bias_const;
    int mode;
    int *counter;              // zero-initialised: rows are handed out to the workgroups in `order` (heaviest first)
#ifdef CMF_CHOL_DEBUG
    int dbg = 0;               // timing experiments (results are wrong): 1 gather + rank-k update, 2 rank-k MFMAs only, 4 factorisation,
                               // 8 backward substitution, 16 initial matrix
    unsigned long long *tstamp = nullptr;   // [8]: shader-clock ticks summed over the rows of the launch -- 0 row claim + set-up,
                                            // 1 rank-k loop, 2 initial matrix + diagonal, 3 factorisation, 4 backward substitution
                                            // + store, 5 rows, 6 non-zeros
#endif
    int row_first;             // positions row_first .. nrows-1 of the processing order are handled
    // gramk_consumer_kernel (gramk_kernels.hpp; 17-block rows in single precision) only: the rank-k update was done beforehand
    // by gramk_producer_kernel, the row's tiles and right-hand side are the sum of its work items' partials
    // [item - gk_base][gk_stride] -- the split rows (positions < gk_n_heavy) own the items gk_row_off[pos] ..
    // gk_row_off[pos + 1], position pos >= gk_n_heavy the item gk_n_slices + pos - gk_n_heavy.
    const T *gk_part = nullptr;
    const int *gk_row_off = nullptr;
    int gk_n_heavy = 0, gk_n_slices = 0, gk_base = 0;
    size_t gk_stride = 0;
    // ... and the launch's initial matrices in the same tile-linear layout (tile_pack_lane_kernel): gk_init1 for every row (Mfull),
    // gk_init2 for the rows with side information (Minit = w C^T C); null: none
    const T *gk_init1 = nullptr, *gk_init2 = nullptr;
};

__host__ __device__ inline int chol_tiles(int kt) { return (kt + 15) / 16; }

template <typename T> struct CholMfma;
template <> struct CholMfma<double> {
    typedef double vec __attribute__((ext_vector_type(4)));
    static constexpr int LDR = 17;      // leading dimension of inv(R_kk) in LDS: conflict-free row writes and operand reads
    static __device__ __forceinline__ vec mma(double a, double b, vec c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row_of(int lane, int r) { return (lane >> 4) + 4 * r; }
    // position of tile element (i, j) in the lane-linear C/D layout [r][lane]
    static __device__ __forceinline__ int cidx(int i, int j) { return (i >> 2) * 64 + (i & 3) * 16 + j; }
};
template <> struct CholMfma<float> {
    typedef float vec __attribute__((ext_vector_type(4)));
    static constexpr int LDR = 20;
    static __device__ __forceinline__ vec mma(float a, float b, vec c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row_of(int lane, int r) { return (lane >> 4) * 4 + r; }
    static __device__ __forceinline__ int cidx(int i, int j) { return (i & 3) * 64 + (i >> 2) * 16 + j; }
};

// LDS: ring of 2 x [CHUNK][ldc] staged rows (later: the panel tiles X, NTT x 256), inv(R_kk) for every
// block, right-hand side, solution, chunk weights
template <typename T>
__host__ __device__ inline size_t chol_lds_elems(int NTT, int CH)
{
    const size_t ldc = 16 * NTT + ((NTT % 2 == 0) ? 16 : 0);
    return 2 * (size_t)CH * ldc + (size_t)NTT * 16 * CholMfma<T>::LDR + 2 * 16 * (size_t)NTT + 4 * CH + 8;
}

// (static_for lives in lanes.hpp)
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

// wave-uniform broadcast of one lane's value (v_readlane_b32, no LDS round trip)
__device__ __forceinline__ float bcast_lane(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ double bcast_lane(double v, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// value of lane `src` for every lane through the LDS crossbar (ds_bpermute_b32): unlike v_readlane the
// result stays in vector registers, so a row of broadcasts can be in flight at once
__device__ __forceinline__ float perm_lane(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src * 4, __float_as_int(v)));
}
__device__ __forceinline__ double perm_lane(double v, int src)
{
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(src * 4, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(src * 4, __double2loint(v)));
}

// 1/sqrt(x) in four short steps (hardware estimate + one third-order correction, relative error
// ~1e-16 in double) so that the chain of the next pivot can be interleaved with the updates of the
// current one.
template <typename T> struct RsqChain {
    T x, y, e, p, q;
    __device__ __forceinline__ void s0(T xin) { x = xin; y = hw_rsq(x); }
    __device__ __forceinline__ void s1() { e = fma_(-(x * y), y, T(1)); }
    __device__ __forceinline__ void s2() { p = fma_(e, T(0.375), T(0.5)); q = y * e; }
    __device__ __forceinline__ T s3() { return fma_(q, p, y); }
    static __device__ __forceinline__ double hw_rsq(double v) { return __builtin_amdgcn_rsq(v); }
    static __device__ __forceinline__ float hw_rsq(float v) { return __builtin_amdgcn_rsqf(v); }
    static __device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static __device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
};

// Diagonal block: d = the 16x16 tile D (C/D layout) of one wave, nact = rows of it that hold unknowns
// (the rest is identity padding).  Computes R (R^T R = D, upper) and inv(R), one column per lane:
// lanes 0-15 carry the columns of D -> R, lanes 16-31 the columns of I -> inv(R)^T (forward substitution
// of R^T V = I, right-looking), so that the one row R[c][*] broadcast by v_readlane at pivot c and the one
// FMA per (pivot, row) serve the elimination and the inversion at once (lanes 32-63 mirror them).
// slot <- inv(R) row-major [16][LDR]; it doubles as the transposition scratch (>= 256 elements).
template <typename T>
__device__ __forceinline__ void chol_diag_block(typename CholMfma<T>::vec d, T *slot, int lane, int nact)
{
    using Mf = CholMfma<T>;
#pragma unroll
    for (int r = 0; r < 4; r++) slot[r * 64 + lane] = d[r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int j = lane & 15;
    const bool inv_half = (lane & 16) != 0;
    T u[16];
    T one = T(1);
    asm volatile("" : "+v"(one));          // opaque: keeps the identity columns from being hoisted out of the row loop
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const T dv = slot[Mf::cidx(i, j)];
        u[i] = inv_half ? ((i == j) ? one : T(0)) : dv;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    T *wrow = slot + j * Mf::LDR;
    RsqChain<T> ch;
    ch.s0(bcast_lane(u[0], 0)); ch.s1(); ch.s2();
    T rs = ch.s3();
    static_for<0, 16>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if (c < nact) {
            const T v = u[c] * rs;                  // lanes 0-15: R[c][j] (j >= c);  lanes 16-31: inv(R)[j][c]
            if (lane >= 16 && lane < 32) wrow[c] = v;
            const bool more = (c + 1 < nact);
            static_for<c + 1, 16>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                u[i] -= perm_lane(v, i) * v;        // R[c][i] from lane i
                // 1/sqrt of the next pivot, one step per update so its latency hides behind them
                if (more) {
                    if (i == c + 1) ch.s0(bcast_lane(u[c + 1], c + 1));
                    if (i == c + 2) ch.s1();
                    if (i == c + 3) ch.s2();
                    if (i == c + 4) rs = ch.s3();
                }
            });
            if (more) {                             // late pivots: fewer than four updates to hide behind
                if (c + 1 >= 15) ch.s1();
                if (c + 1 >= 14) ch.s2();
                if (c + 1 >= 13) rs = ch.s3();
            }
        } else {
            if (lane >= 16 && lane < 32) wrow[c] = u[c];   // identity padding: R = inv(R) = I there
        }
    });
}

// NTT = tiles per dimension of the compiled grid (k_t <= 16 NTT);  NW = wavefronts per workgroup
// (wave w owns tiles t = w, w + NW, ... of the packed upper triangle and stages CHOL_CHUNK / NW of the
// CHOL_CHUNK gathered rows of a round);  WGS = workgroups per CU the register budget is set for.
// TWO_SRC compiles the second gather source in (sparse side information); the single-source build carries none of its
// selects (they cost 4-19 % on the plain Cholesky configurations when they were unconditional).
#ifdef CMF_CHOL_DEBUG
#define CMF_CDBG(P, bit) (((P).dbg & (bit)) != 0)
#define CMF_CTICK(P, slot)                                                                   \
    do {                                                                                     \
        if ((P).tstamp != nullptr && tid == 0) {                                             \
            const unsigned long long now_ = __builtin_readcyclecounter();                    \
            atomicAdd(&(P).tstamp[slot], now_ - tick_);                                      \
            tick_ = now_;                                                                    \
        }                                                                                    \
    } while (0)
#else
#define CMF_CDBG(P, bit) false
#define CMF_CTICK(P, slot) do { } while (0)
#endif
template <typename T, int NTT, int NW, int CHOL_CHUNK, int WGS, bool TWO_SRC = false>
__global__ void __launch_bounds__(64 * NW, WGS)
chol_rows_kernel(const CholParams<T> P)
{
    using Mf = CholMfma<T>;
    using vec = typename Mf::vec;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int kt = P.kt, koff = P.koff;
    const int nb = chol_tiles(kt);                  // 16-blocks that hold unknowns (the rest of the grid is padding)
    constexpr int ldc = 16 * NTT + ((NTT % 2 == 0) ? 16 : 0);   // == 16 (mod 32): conflict-free MFMA operand reads
    constexpr int NTALL = NTT * (NTT + 1) / 2;
    constexpr int LDR = Mf::LDR, RSZ = 16 * LDR;
    constexpr int TPW = (NTALL + NW - 1) / NW;      // tile slots per wave
    constexpr int NTH = 64 * NW;
    constexpr int RPW = CHOL_CHUNK / NW;            // staged rows per wave and chunk
    constexpr int NCJ = (16 * NTT + 63) / 64;       // 64-column groups of a staged row
    static_assert(CHOL_CHUNK % NW == 0 && NTH >= 16 * NTT, "workgroup shape");
    static_assert(RSZ >= 256, "inv(R) slot doubles as a tile scratch");
    T *ring = reinterpret_cast<T *>(smem_raw);             // 2 x [CHUNK][ldc];  after the gather: X tiles [NTT][256]
    T *rinv = ring + 2 * (size_t)CHOL_CHUNK * ldc;         // [NTT][16][LDR]
    T *rhs = rinv + (size_t)NTT * RSZ;                     // [16 NTT]  right-hand side -> y (in place)
    T *xall = rhs + 16 * NTT;                              // [16 NTT]  solution
    T *wsc = xall + 16 * NTT;                              // 2 x [CHUNK] rank-1 weights
    T *wrh = wsc + 2 * CHOL_CHUNK;                         // 2 x [CHUNK] rhs weights
    T *Xt = ring;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lm = lane & 15;

    // this wave's tiles: t = wave, wave+NW, ...  kept as the operand column offsets (16 bi, 16 bj) of the
    // staged rows; idle slots are marked bi > bj.  Wave-uniform values, deliberately held in vector
    // registers: as scalars they would be spilled and restored around every use.
    int offa[TPW], offb[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; tt++) {
        const int t = wave + NW * tt;
        const int bi = tile_bi(min(t, NTALL - 1), NTT), bj = tile_bj(min(t, NTALL - 1), NTT);
        const bool real = (t < NTALL) && (bj < nb);          // padding tiles are never referenced
        offa[tt] = real ? 16 * bi : 16;
        offb[tt] = real ? 16 * bj : 0;
    }
#define T_BI(tt) (offa[tt] >> 4)
#define T_BJ(tt) (offb[tt] >> 4)
#define T_REAL(tt) (offa[tt] <= offb[tt])
    // the same offsets in LDS for the rank-k loop of the widest builds, where the register copies above end up in scratch
    // (17 tiles / 16 waves: 26 scratch loads per round of the hottest loop): one broadcast ds_read per tile and k-step
    constexpr bool OFF_LDS = (NTT >= 16);
    __shared__ unsigned s_otab[OFF_LDS ? NW * TPW : 1];
    if (OFF_LDS) {
#pragma unroll
        for (int tt = 0; tt < TPW; tt++)
            if (lane == 0) s_otab[wave * TPW + tt] = (unsigned)offa[tt] | ((unsigned)offb[tt] << 16);
        __syncthreads();
    }

    // staging: wave w gathers rows RPW*w .. RPW*w + RPW-1 of a chunk, lane l the unknowns l, l+64, ...
    int scol[NCJ];                             // column of B to read (clamped); valid <=> svalid bit
    unsigned svalid = 0;
#pragma unroll
    for (int j = 0; j < NCJ; j++) {
        const int c = lane + 64 * j;
        const bool ok = (c >= koff && c < kt);
        scol[j] = ok ? c - koff : 0;
        svalid |= ok ? (1u << j) : 0u;
    }

    const bool two_src = TWO_SRC && P.indptr2 != nullptr;
    int scol2[NCJ];
    unsigned svalid2 = 0;
#pragma unroll
    for (int j = 0; j < NCJ; j++) {
        const int c = lane + 64 * j;
        const bool ok = two_src && (c >= P.koff2) && (c < P.koff2 + P.kc2);
        scol2[j] = ok ? c - P.koff2 : 0;
        svalid2 |= ok ? (1u << j) : 0u;
    }

    __shared__ int s_rix;
#ifdef CMF_CHOL_DEBUG
    unsigned long long tick_ = __builtin_readcyclecounter();
#endif
    for (;;) {
        if (tid == 0) s_rix = P.row_first + atomicAdd(P.counter, 1);
        __syncthreads();
        const int rix = s_rix;
        __syncthreads();                            // s_rix may be rewritten
        if (rix >= P.nrows) break;
        const int row = (P.order != nullptr) ? P.order[rix] : rix;
        const size_t st = (P.mode == CHOL_PREFILLED) ? 0 : P.indptr[row];
        // the right-hand-side-only variants (CHOL_NAZ, w2_syr_zero) live in the TWO_SRC build: the plain build keeps its
        // straight-line chunk loop (a branch around the MFMAs cost the k = 128 double-precision step 16 %)
        const T *xvals = (TWO_SRC && P.values_override != nullptr) ? P.values_override : P.values;
        const int nnz1 = (P.mode == CHOL_PREFILLED) ? 0 : (int)(P.indptr[row + 1] - st);
        const bool in2 = two_src && row < (P.rows_src2 >= 0 ? P.rows_src2 : P.rows_with_u);
        const size_t st2 = in2 ? P.indptr2[row] : 0;
        const int nnz2 = in2 ? (int)(P.indptr2[row + 1] - st2) : 0;
        const int nnz = nnz1 + nnz2;              // entries to gather; nnz1 of them are observations of X
        T *arow = P.A + (size_t)row * P.lda;
        const bool coll = (P.mode == CHOL_COLLECTIVE || P.mode == CHOL_COLLECTIVE_IMPLICIT);
        const bool impl_w = (P.mode == CHOL_IMPLICIT || P.mode == CHOL_COLLECTIVE_IMPLICIT);
        const bool naz = TWO_SRC && (P.mode == CHOL_NAZ);
        const bool has_u = (P.mode == CHOL_PREFILLED) || (coll && row < P.rows_with_u);
        if (coll && nnz == 0 && !has_u) {                               // collective.c:1258-1268, :1876-1885
            for (int e = tid; e < kt; e += NTH) arow[e] = T(0);
            continue;
        }
        T lam = P.lam, lam_last = P.lam_last;
        T l1 = P.l1, l1_last = P.l1_last;
        if (P.mode == CHOL_EXPLICIT || P.mode == CHOL_NAZ_W) {
            if (P.scale_lam) {                                           // common.c:679-723
                const T mult = (P.wsum != nullptr) ? P.wsum[row] : (T)nnz1;
                lam *= mult; l1 *= mult;
                if (!P.scale_bias_const) { lam_last *= mult; l1_last *= mult; }
            }
        } else if (P.mode == CHOL_COLLECTIVE) {
            if (P.scale_lam || P.scale_lam_sideinfo) {                   // collective.c:1285-1355
                T mult = (P.wsum != nullptr) ? P.wsum[row] : ((nnz1 > 0) ? (T)nnz1 : T(1));
                if (P.scale_lam_sideinfo && has_u) mult += (two_src && !P.w2_syr_zero) ? (T)nnz2 : (T)P.p_side;   // :1338-1346
                lam *= mult;
                // rows without side information are plain factors_closed_form rows when new rows are fitted
                // (collective.c:3772-3815): there scale_bias_const keeps the bias' lambda (common.c:679-723)
                if (has_u || !P.scale_bias_const) lam_last *= mult;
                if (!P.scale_bias_const) { l1 *= mult; l1_last *= mult; }      // collective.c:1349-1354
            }
        }
        // ---- 1. rank-k update on the matrix cores: G[koff:, koff:] = sum_j w_j B_j B_j^T ----
        vec acc[TPW];
#pragma unroll
        for (int tt = 0; tt < TPW; tt++) acc[tt] = vec{0, 0, 0, 0};
        // right-hand side: thread t owns unknown t
        T racc = ((has_u || P.rhs_prefilled_all) && tid < kt) ? arow[tid] : T(0);   // w*U*C prefilled (collective.c:5768-5773)
        // software pipeline over chunks of CHOL_CHUNK gathered rows:
        //   indices (+ x values) of chunk c+2  ->  rows (+ bias of x) of chunk c+1  ->  LDS slot / MFMAs of chunk c
        // loads are unconditional on clamped addresses, padding is selected to zero afterwards
        T pre[RPW][NCJ];
        int idn[RPW];
        int widx = 0; T wx = T(0), wg = T(1);
        T pre_wsyr = T(0), pre_wrhs = T(0);
        int nr_rows = 0, nr_idx = 0;
        unsigned src_idx = 0, src_rows = 0;       // bit i: staged row i of this wave comes from the second source
        bool wsrc2 = false;
        // entry e of the row: e < nnz1 -> observation e of X, else side-information entry e - nnz1
        auto load_idx = [&](int c0) {
            nr_idx = min(CHOL_CHUNK, nnz - c0);
            src_idx = 0;
#pragma unroll
            for (int i = 0; i < RPW; i++) {
                const int e = c0 + min(RPW * wave + i, nr_idx - 1);
                const bool s2 = e >= nnz1;
                idn[i] = s2 ? P.indices2[st2 + (e - nnz1)] : P.indices[st + e];
                src_idx |= s2 ? (1u << i) : 0u;
            }
            const int e = c0 + min(tid, nr_idx - 1);
            wsrc2 = e >= nnz1;
            widx = wsrc2 ? 0 : P.indices[st + e];
            wx = wsrc2 ? P.values2[st2 + (e - nnz1)] : xvals[st + e];
            wg = (!wsrc2 && P.weights != nullptr) ? P.weights[st + e] : T(1);
        };
        auto load_rows = [&]() {
            nr_rows = nr_idx;
            src_rows = src_idx;
#pragma unroll
            for (int i = 0; i < RPW; i++) {
                const bool s2 = (src_idx >> i) & 1u;
                const T *base = s2 ? P.B2 + (size_t)idn[i] * P.ldb2 : P.B + (size_t)idn[i] * P.ldb;
#pragma unroll
                for (int j = 0; j < NCJ; j++) pre[i][j] = base[s2 ? scol2[j] : scol[j]];
            }
            T x = wx;
            if (P.bias_sub != nullptr && !wsrc2) x -= P.bias_sub[widx];
            pre_wsyr = impl_w ? x : wg;             // common.c:2091-2095, collective.c:2103-2108 vs common.c:1007-1012
            pre_wrhs = impl_w ? x + T(1) : x * wg;  // common.c:2082-2085, collective.c:2097-2101 vs common.c:985-996
            if (naz) pre_wsyr = T(0);               // the matrix is shared (common.c:3130-3140)
            if (TWO_SRC && (P.mode == CHOL_NAZ_W || P.entry_pairs)) { pre_wsyr = wg; pre_wrhs = x; }     // common.c:866-885 (w - 1 and the bracket arrive per entry)
            if (TWO_SRC && P.x_rhs_only) pre_wsyr = T(0);
            if (wsrc2) { pre_wsyr = P.w2_syr_zero ? T(0) : P.w2; pre_wrhs = P.w2 * wx; }   // collective.c:1636-1653, :1719-1731
        };
        if (nnz > 0) { load_idx(0); load_rows(); }
        if (nnz > CHOL_CHUNK) load_idx(CHOL_CHUNK);
        __syncthreads();          // previous row's LDS readers (backward substitution) are done
        CMF_CTICK(P, 0);
#ifdef CMF_CHOL_DEBUG
        if (P.tstamp != nullptr && tid == 0) { atomicAdd(&P.tstamp[5], 1ull); atomicAdd(&P.tstamp[6], (unsigned long long)nnz); }
#endif
        for (int c0 = 0, slot = 0; c0 < (CMF_CDBG(P, 1) ? 0 : nnz); c0 += CHOL_CHUNK, slot ^= 1) {
            T *Bs = ring + (size_t)slot * CHOL_CHUNK * ldc;
#pragma unroll
            for (int i = 0; i < RPW; i++)
#pragma unroll
                for (int j = 0; j < NCJ; j++)
                    if (lane + 64 * j < ldc)
                        Bs[(RPW * wave + i) * ldc + lane + 64 * j] =
                            (RPW * wave + i < nr_rows && (((((src_rows >> i) & 1u) ? svalid2 : svalid) >> j) & 1u)) ? pre[i][j] : T(0);
            if (tid < CHOL_CHUNK) {
                wsc[slot * CHOL_CHUNK + tid] = (tid < nr_rows) ? pre_wsyr : T(0);
                wrh[slot * CHOL_CHUNK + tid] = (tid < nr_rows) ? pre_wrhs : T(0);
            }
            __syncthreads();                  // slot visible; the other slot may still be read by slower waves
            if (c0 + CHOL_CHUNK < nnz) load_rows();                           // chunk c+1: in flight during the MFMAs
            if (c0 + 2 * CHOL_CHUNK < nnz) load_idx(c0 + 2 * CHOL_CHUNK);     // chunk c+2
            if ((tid >= koff || two_src) && tid < kt) {                        // rhs[e] += sum_r wrhs_r B_r[e]
#pragma unroll
                for (int r = 0; r < CHOL_CHUNK; r++) racc += wrh[slot * CHOL_CHUNK + r] * Bs[r * ldc + tid];   // padded rows: weight 0, row 0
            }
            // chunks whose rank-1 weights are all zero contribute to the right-hand side only
            const bool skip_mma = (TWO_SRC && (naz || (two_src && P.w2_syr_zero && c0 >= nnz1))) || CMF_CDBG(P, 2);
            // straight-line: operand reads of every slot, weights, MFMAs (idle slots run on tile 0 and are ignored)
            if (!skip_mma) {
                if constexpr (OFF_LDS) {
                    // element addresses of this lane's operand column in slab row (lane >> 4), once per round; the k-step
                    // (4 q rows further down) is an immediate offset of the ds_read
                    const T *pa[TPW], *pb[TPW];
#pragma unroll
                    for (int tt = 0; tt < TPW; tt++) {
                        const unsigned o = s_otab[wave * TPW + tt];
                        pa[tt] = Bs + (lane >> 4) * ldc + lm + (o & 0xffffu);
                        pb[tt] = Bs + (lane >> 4) * ldc + lm + (o >> 16);
                    }
#pragma unroll
                    for (int q = 0; q < CHOL_CHUNK / 4; q++) {
                        const T w = wsc[slot * CHOL_CHUNK + 4 * q + (lane >> 4)];
                        T opa[TPW], opb[TPW];
#pragma unroll
                        for (int tt = 0; tt < TPW; tt++) { opa[tt] = pa[tt][4 * q * ldc]; opb[tt] = pb[tt][4 * q * ldc]; }
#pragma unroll
                        for (int tt = 0; tt < TPW; tt++) acc[tt] = Mf::mma(opa[tt] * w, opb[tt], acc[tt]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                for (int q = 0; q < CHOL_CHUNK / 4; q++) {
                    const int rr = 4 * q + (lane >> 4);
                    const T *brow = Bs + rr * ldc + lm;
                    const T w = wsc[slot * CHOL_CHUNK + rr];       // 1 in the explicit models
                    T opa[TPW], opb[TPW];
#pragma unroll
                    for (int tt = 0; tt < TPW; tt++) { opa[tt] = brow[offa[tt]]; opb[tt] = brow[offb[tt]]; }
#pragma unroll
                    for (int tt = 0; tt < TPW; tt++) acc[tt] = Mf::mma(opa[tt] * w, opb[tt], acc[tt]);
                    __builtin_amdgcn_sched_barrier(0);       // one k-step of operands in registers at a time
                }
                }
            }
        }
        CMF_CTICK(P, 1);
        if (TWO_SRC && P.rhs_only) {
            if (tid < kt) arow[tid] = racc;
            continue;                     // the loop head's barrier orders the staging ring against the next row
        }
        // ---- 2. the initial matrix, in the accumulator layout (padding: identity) ----
        {
            const bool full = (P.mode == CHOL_IMPLICIT || P.mode == CHOL_PREFILLED || P.mode == CHOL_NAZ || P.mode == CHOL_NAZ_W);
            const T *M1 = full ? P.Minit : P.Mfull;                    // [kt, kt], every row
            const T *M2 = (!full && has_u) ? P.Minit : nullptr;        // [kc, kc], rows with side information
#pragma unroll 1
            for (int pass = 0; pass < 2; pass++) {                     // unconditional loads on clamped addresses
                const T *Mi = pass ? M2 : M1;
                const int lim = pass ? P.kc : kt;
                if (Mi == nullptr || lim <= 0 || CMF_CDBG(P, 16)) continue;
#pragma unroll
                for (int tt = 0; tt < TPW; tt++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int gi = offa[tt] + Mf::row_of(lane, r), gj = offb[tt] + lm;
                        const int lo = min(gi, gj), hi = max(gi, gj);
                        const T v = Mi[(size_t)min(lo, lim - 1) * lim + min(hi, lim - 1)];   // collective.c:1566-1571
                        acc[tt][r] += (hi < lim) ? v : T(0);
                    }
                }
            }
            const bool add_lam = (P.mode == CHOL_EXPLICIT || P.mode == CHOL_COLLECTIVE || P.mode == CHOL_NAZ_W);
#pragma unroll
            for (int tt = 0; tt < TPW; tt++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int gi = offa[tt] + Mf::row_of(lane, r), gj = offb[tt] + lm;
                    T dv = T(0);
                    if (gi == gj) dv = (gi >= kt) ? T(1) : (!add_lam ? T(0) : ((gi == kt - 1) ? lam_last : lam));   // add_to_diag2: common.c:1060-1062, collective.c:1819
                    acc[tt][r] += dv;
                }
            }
        }
        if (P.nonneg || P.l1 != T(0) || P.l1_last != T(0)) {
            // ---- 3'. non-negative solution by cyclic coordinate descent (solve_nonneg, common.c:2131-2179):
            //   a = 0, g = rhs;  sweep ix = 0..kt-1:  new = max(a_ix + g_ix / M_ix,ix, 0);  if |new - a_ix| > 1e-8:
            //   g -= (new - a_ix) M[ix, :], a_ix = new;  stop after a sweep that moved less than 1e-8 in total.
            // The matrix leaves the accumulators for LDS (full, both triangles, [kt][kt] over the staging area -- the
            // launch reserves kt^2 + 2 kt elements), one wavefront sweeps with lane <-> unknown.
            __syncthreads();                  // ring fully consumed
            T *Mn = ring;
            T *gn = ring + (size_t)kt * kt;
#pragma unroll
            for (int tt = 0; tt < TPW; tt++) {
                if (!T_REAL(tt)) continue;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int gi = offa[tt] + Mf::row_of(lane, r), gj = offb[tt] + lm;
                    if (gi < kt && gj < kt) {
                        Mn[(size_t)gi * kt + gj] = acc[tt][r];
                        Mn[(size_t)gj * kt + gi] = acc[tt][r];           // fill_lower_triangle
                    }
                }
            }
            if (tid < kt) gn[tid] = racc;
            __syncthreads();
            if (wave == 0 && !P.nonneg) {
                // solve_elasticnet (common.c:2228-2294): g+ = rhs - l1, g- = -rhs - l1; per sweep the descent step on a+
                // (g+ -= step M_ix, g- += step M_ix), then on a- (the other way round); a = a+ - a-
                constexpr int NFN = (16 * NTT + 63) / 64;
                T ap[NFN], am[NFN], gp[NFN], gm[NFN];
#pragma unroll
                for (int c = 0; c < NFN; c++) {
                    const int f = lane + 64 * c;
                    const T b = (f < kt) ? gn[f] : T(0);
                    const T l = (f == kt - 1) ? l1_last : l1;
                    ap[c] = T(0); am[c] = T(0);
                    gp[c] = (f < kt) ? b - l : T(0);
                    gm[c] = (f < kt) ? -b - l : T(0);
                }
                const int sweeps = (P.max_cd_steps > 0) ? P.max_cd_steps : 0x7fffffff;
                for (int it = 0; it < sweeps; it++) {
                    T moved = T(0);
#pragma unroll 1
                    for (int side = 0; side < 2; side++) {
                        for (int ix = 0; ix < kt; ix++) {
                            const int cq = ix >> 6, lq = ix & 63;
                            T a_ix = T(0), g_ix = T(0);
#pragma unroll
                            for (int c = 0; c < NFN; c++)
                                if (c == cq) {
                                    a_ix = bcast_lane(side ? am[c] : ap[c], lq);
                                    g_ix = bcast_lane(side ? gm[c] : gp[c], lq);
                                }
                            T nv = a_ix + g_ix / Mn[(size_t)ix * kt + ix];
                            nv = (nv > T(0)) ? nv : T(0);
                            const T dv = nv - a_ix;
                            if (fabs(dv) > T(1e-8)) {
                                moved += fabs(dv);
#pragma unroll
                                for (int c = 0; c < NFN; c++) {
                                    const int f = lane + 64 * c;
                                    const T stepv = (f < kt) ? dv * Mn[(size_t)ix * kt + f] : T(0);
                                    if (side) { gp[c] += stepv; gm[c] -= stepv; }
                                    else      { gm[c] += stepv; gp[c] -= stepv; }
                                    if (c == cq && lane == lq) { if (side) am[c] = nv; else ap[c] = nv; }
                                }
                            }
                        }
                    }
                    if (!(moved >= T(1e-8)) || isinf(moved)) break;
                }
#pragma unroll
                for (int c = 0; c < NFN; c++) if (lane + 64 * c < kt) arow[lane + 64 * c] = ap[c] - am[c];
            } else if (wave == 0) {
                constexpr int NFN = (16 * NTT + 63) / 64;
                T an[NFN], gg[NFN];
#pragma unroll
                for (int c = 0; c < NFN; c++) {
                    const int f = lane + 64 * c;
                    an[c] = T(0);
                    gg[c] = (f < kt) ? gn[f] - ((f == kt - 1) ? l1_last : l1) : T(0);          // common.c:2148-2154
                }
                const int sweeps = (P.max_cd_steps > 0) ? P.max_cd_steps : 0x7fffffff;
                for (int it = 0; it < sweeps; it++) {
                    T moved = T(0);
                    for (int ix = 0; ix < kt; ix++) {
                        const int cq = ix >> 6, lq = ix & 63;
                        T a_ix = T(0), g_ix = T(0);
#pragma unroll
                        for (int c = 0; c < NFN; c++)
                            if (c == cq) { a_ix = bcast_lane(an[c], lq); g_ix = bcast_lane(gg[c], lq); }
                        T nv = a_ix + g_ix / Mn[(size_t)ix * kt + ix];
                        nv = (nv > T(0)) ? nv : T(0);                       // max2(newval, 0.): NaN -> 0 like the reference's macro
                        const T dv = nv - a_ix;
                        if (fabs(dv) > T(1e-8)) {
                            moved += fabs(dv);
#pragma unroll
                            for (int c = 0; c < NFN; c++) {
                                const int f = lane + 64 * c;
                                if (f < kt) gg[c] -= dv * Mn[(size_t)ix * kt + f];
                                if (c == cq && lane == lq) an[c] = nv;
                            }
                        }
                    }
                    if (!(moved >= T(1e-8)) || isinf(moved)) break;          // isnan / !isfinite / < 1e-8
                }
#pragma unroll
                for (int c = 0; c < NFN; c++) if (lane + 64 * c < kt) arow[lane + 64 * c] = an[c];
            }
            continue;                         // the next row starts with a barrier
        }
        if (tid < 16 * NTT) rhs[tid] = (tid < kt) ? racc : T(0);
        __syncthreads();                      // ring fully consumed (X tiles alias it), rhs visible
        CMF_CTICK(P, 2);
        // ---- 3. blocked Cholesky  M = R^T R ----
        for (int kbk = 0; kbk < (CMF_CDBG(P, 4) ? 0 : nb); kbk++) {
            T *rslot = rinv + (size_t)kbk * RSZ;
            {   // a. diagonal block, by its owner wave
                vec d = vec{0, 0, 0, 0};
                bool mine = false;
#pragma unroll
                for (int tt = 0; tt < TPW; tt++)
                    if (T_BI(tt) == kbk && T_BJ(tt) == kbk) { d = acc[tt]; mine = true; }
                if (mine) chol_diag_block<T>(d, rslot, lane, min(16, kt - 16 * kbk));
            }
            __syncthreads();
            CMF_CTICK(P, 3);                  // (debug builds: slot 3 = diagonal blocks + the wait for them)
            {   // b. panel tiles of block row kbk:  X = inv(R_kk)^T * tile
                T ainv[4];
#pragma unroll
                for (int r = 0; r < 4; r++) ainv[r] = rslot[Mf::row_of(lane, r) * LDR + lm];
                if (wave == 0) {              // y_k = inv(R_kk)^T rhs_k, in place
                    T yv = T(0);
#pragma unroll
                    for (int l = 0; l < 16; l++) yv += rslot[l * LDR + lm] * rhs[16 * kbk + l];
                    if (lane < 16) rhs[16 * kbk + lane] = yv;
                }
#pragma unroll
                for (int tt = 0; tt < TPW; tt++) {
                    if (T_BI(tt) == kbk && T_BJ(tt) > kbk) {
                        vec x = Mf::mma(ainv[0], acc[tt][0], vec{0, 0, 0, 0});          // two independent chains
                        vec x2 = Mf::mma(ainv[2], acc[tt][2], vec{0, 0, 0, 0});
                        x = Mf::mma(ainv[1], acc[tt][1], x);
                        x2 = Mf::mma(ainv[3], acc[tt][3], x2);
                        x += x2;
                        acc[tt] = x;
#pragma unroll
                        for (int r = 0; r < 4; r++) Xt[T_BJ(tt) * 256 + r * 64 + lane] = x[r];
                    }
                }
            }
            __syncthreads();
            CMF_CTICK(P, 7);                  // slot 7 = panels
            // c. trailing tiles:  tile(bi, bj) -= X_bi^T X_bj ;  forward substitution of the later blocks.
            //    Operands of every slot are read unconditionally (two k-steps at a time); the MFMAs of one
            //    k-step are independent of each other.
            // A wave's slots are in tile order, i.e. by block row, so the live tiles (block row > kbk) are the suffix
            // [tt0, TPW) of them: a scalar compare per MFMA.  (The per-slot tile coordinates live in vector registers,
            // see above; predicates on them compile to exec-mask juggling around every MFMA.)  Padding tiles inside
            // the suffix are updated too: nothing reads them.
            const int ft1 = (kbk + 1) * NTT - ((kbk + 1) * kbk) / 2;            // first tile of block row kbk + 1
            const int tt0 = max(0, (ft1 - __builtin_amdgcn_readfirstlane(wave) + NW - 1) / NW);
#pragma unroll
            for (int half = 0; half < 2; half++) {
                T xa[TPW][2], xb[TPW][2];
#pragma unroll
                for (int tt = 0; tt < TPW; tt++)
#pragma unroll
                    for (int r2 = 0; r2 < 2; r2++) {
                        xa[tt][r2] = -Xt[offa[tt] * 16 + (2 * half + r2) * 64 + lane];
                        xb[tt][r2] = Xt[offb[tt] * 16 + (2 * half + r2) * 64 + lane];
                    }
#pragma unroll
                for (int r2 = 0; r2 < 2; r2++)
#pragma unroll
                    for (int tt = 0; tt < TPW; tt++)
                        if (tt >= tt0) acc[tt] = Mf::mma(xa[tt][r2], xb[tt][r2], acc[tt]);
            }
            {
                const int jg = tid;
                if (jg >= 16 * (kbk + 1) && jg < 16 * nb) {
                    T sacc = rhs[jg];
                    const T *xt = Xt + (jg >> 4) * 256;
#pragma unroll
                    for (int k2 = 0; k2 < 16; k2++) sacc -= xt[Mf::cidx(k2, jg & 15)] * rhs[16 * kbk + k2];
                    rhs[jg] = sacc;
                }
            }
            CMF_CTICK(P, 1);                  // slot 1 = trailing updates (wave 0's share)
        }
        __syncthreads();
        CMF_CTICK(P, 3);
        // ---- 4. backward substitution R x = y, one block column per step ----
        for (int bjk = (CMF_CDBG(P, 8) ? -1 : nb - 1); bjk >= 0; bjk--) {
            const T *rslot = rinv + (size_t)bjk * RSZ;
            T xm = T(0);                          // x[16 bjk + lm], computed redundantly by every 16-lane group
{   // (round 6: the block of the right-hand side once, its elements broadcast inside the 16-lane row by DPP instead of sixteen more LDS reads)
                const T yb = rhs[16 * bjk + lm];
                static_for<0, 16>([&](auto nc) {
                    constexpr int n2 = decltype(nc)::value;
                    xm += rslot[lm * LDR + n2] * lanes::row_bcast16<n2>(yb);
                });
            }
            if (wave == 0 && lane < 16) xall[16 * bjk + lane] = xm;
#pragma unroll
            for (int tt = 0; tt < TPW; tt++) {
                if (T_BJ(tt) == bjk && T_BI(tt) < bjk) {
                    // sum over the 16 lanes of a row for the four registers at once: after two
                    // select-and-exchange steps lane l carries register (l & 3), then two plain butterflies
                    const T p0 = acc[tt][0] * xm, p1 = acc[tt][1] * xm, p2 = acc[tt][2] * xm, p3 = acc[tt][3] * xm;
                    const bool o1 = (lm & 1) != 0, o2 = (lm & 2) != 0;
                    const T s01 = (o1 ? p1 : p0) + lanes::xor1(o1 ? p0 : p1);
                    const T s23 = (o1 ? p3 : p2) + lanes::xor1(o1 ? p2 : p3);
                    T sr = (o2 ? s23 : s01) + lanes::xor2(o2 ? s01 : s23);
                    sr += lanes::xor4(sr);
                    sr += lanes::xor8(sr);
                    if (lm < 4) rhs[offa[tt] + Mf::row_of(lane, lm)] -= sr;
                }
            }
            __syncthreads();
        }
        if (wave == 0)
            for (int e = lane; e < kt; e += 64) arow[e] = xall[e];
        CMF_CTICK(P, 4);
    }
}

#undef T_BI
#undef T_BJ
#undef T_REAL

}  // namespace cmfhip
