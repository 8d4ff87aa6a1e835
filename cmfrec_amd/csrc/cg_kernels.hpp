// cg_kernels.hpp -- conjugate-gradient ALS row updates for gfx950 (MI355X, wave64).
//
// Device-side replacement of the reference's per-row CG solvers and of the OpenMP row loops
// that call them:
//   implicit: factors_implicit_cg   /root/reference/src/common.c:1914-1986, loop :3349-3368
//   explicit: factors_explicit_cg   /root/reference/src/common.c:1098-1188, loop :3259-3299
//             (per-row lambda scaling of factors_closed_form, common.c:679-723)
//
// Mapping (see DESIGN.md "CG row kernel"): one wavefront (or W cooperating wavefronts) per row.
// The 64 lanes form an 8 x 8 grid  lane = jj*8 + ll :
//     jj = lane>>3  owns 8 consecutive non-zeros of a 64-nnz tile  (j = jj*8 + t, t<8)
//     ll = lane&7   owns factor columns f = ll + 8*s, s<S           (S = ceil(k/8))
// so every lane keeps an 8 x S register block of the gathered opposing-factor rows.  Each row of
// B is read from HBM/L2 as 8-lane x 8-byte (64 B) contiguous segments, once; for rows whose
// tiles fit the cooperating waves' registers the block is reused by all (steps+1) CG passes, so
// the (s+1) gather passes of the reference become one.  Per pass and tile:
//     phase 1  c_j  = B_j . v            8*S FMAs / lane + 3-stage transposed reduction over ll
//     phase 2  out += w_j B_j            8*S FMAs / lane, reduced over jj once per pass
// Vectors of length k live either "distributed" (lane f holds element f) or "replicated over
// jj" (lane (jj,ll) holds elements ll+8s).  No LDS is used for the gather; LDS only holds the
// k x k Gramian BtB of the implicit model (one copy per workgroup) and the cross-wave partials.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "lanes.hpp"

namespace cmfhip {

constexpr int WAVE = 64;
constexpr int TILE = 64;          // non-zeros per register tile
constexpr int MAX_W = 8;          // max cooperating waves per row
#ifndef CMF_CG_WAVES_PER_SIMD
#define CMF_CG_WAVES_PER_SIMD 2   // register budget of the row kernels: 2 -> 256 VGPRs, 3 -> 168 (spills)
#endif
// Single precision, teams of up to four wavefronts on one resident tile: three wavefronts per SIMD (168 VGPRs).  Round 4's build fitted
// that by itself (167); with the row body compiled per tile size (round 5) the allocator drifted to 193-195 registers and the
// kernels lost a third of their wavefronts -- config 4's 65..256-entry bins ran 15-20 % slower until the budget was stated
// (same-box A/B against the round-4 build, profiles/r05/r05_q_*).
#ifndef CMF_CG_WAVES_PER_SIMD_F32
#define CMF_CG_WAVES_PER_SIMD_F32 3
#endif
// Double precision (round 5, NTSEL): the register need of a row follows its tile -- k = 50: 154-163 VGPRs with 5 entries per lane
// group, 168 with 6, 184-196 with 7, 230-240 with 8 -- so a build that only holds the bodies for 5 and 6 fits three wavefronts per
// SIMD where the build for all four sizes has two.  The host launches a length bin as two ranges of its processing order (rows are
// sorted by length): NTSEL = 2 for the rows of more than 48 W entries (7 or 8 per lane group), NTSEL = 1 for the others; 0 = one
// launch with every body (single precision, eight-wave teams, k < 25, k > 56).  A row's arithmetic does not depend on the launch it is in.
#ifndef CMF_CG_WAVES_PER_SIMD_LOW
#define CMF_CG_WAVES_PER_SIMD_LOW 3
#endif
// (k <= 56: with S = 8 columns per lane the bodies for 5 and 6 entries need 30-55 registers more than the budget)
constexpr int CG_NTSEL_MAX_S = 7;
// Single precision likewise: the body for 6 entries per lane group alone needs 126-136 registers and fits FOUR wavefronts per SIMD (128,
// no spills), the body for 8 alone is the round-4 kernel (160-167: three).
template <typename T, int W, int NRES_, int NTSEL = 0> constexpr int cg_waves_per_simd()
{
    if (sizeof(T) == 8 && NTSEL == 1 && W <= 8) return CMF_CG_WAVES_PER_SIMD_LOW;
    if (sizeof(T) == 4 && NTSEL == 1 && W <= 4 && NRES_ == 0) return CMF_CG_WAVES_PER_SIMD_F32 + 1;
    return (sizeof(T) == 4 && W <= 4 && NRES_ == 0) ? CMF_CG_WAVES_PER_SIMD_F32 : CMF_CG_WAVES_PER_SIMD;
}
// entries per lane group below which a row belongs to the NTSEL = 1 launch: nnz <= CG_NT_LOW * 8 * W
constexpr int CG_NT_LOW = 6;
#ifndef CMF_CG_NT_F32
#define CMF_CG_NT_F32 1           // single precision takes part in the tiles by row length (0: the 64-entry tile, A/B build)
#endif
#ifndef CMF_CG_NT_MIN_S
#define CMF_CG_NT_MIN_S 4         // tiles by row length (cg_rows_kernel, round 5) for k > 8 (S - 1): below, the tile products are a small
                                  // part of a pass and the extra pass bodies only cost build time.  9 = the 64-entry tile everywhere
#endif

// One entry per row in processing order: replaces the order[] -> indptr[] pointer chase in the
// persistent kernels (one 16-byte load gives the row id, its length and its CSR offset).
struct RowDesc {
    int row;
    int nnz;
    unsigned long long st;
};

constexpr int CG_NCOUNTERS = 64;        // work counters per launch of a tiled CG kernel ...
constexpr int CG_COUNTER_STRIDE = 32;    // ... each on its own 128-byte line

// The two quotients of a CG step (alpha = r.r / p.Ap, beta = r'.r' / r.r), every lane computing the same wave-uniform number.
// Single precision: numerator x v_rcp_f32(denominator) -- 1 ulp for the reciprocal, ~2 ulp (1.2e-7) for the quotient against the
// twelve-instruction IEEE sequence (v_div_scale x 2, v_rcp, four FMAs, v_div_fmas, v_div_fixup); 0 / 0 and x / 0 come out as in
// the IEEE sequence (NaN, inf).  Double precision (round 5): v_rcp_f64, two Newton steps, one correction of the quotient -- the
// arithmetic core of the IEEE sequence without its scaling and fix-up instructions (eight issue slots instead of twelve beside the
// quarter-rate reciprocal; the operands are sums of squares between 1e-12 and what a row's data allows, far from the exponent
// range's ends), <= 1 ulp from the IEEE quotient.  -DCMF_CG_IEEE_DIV: IEEE in both.
template <typename T>
__device__ __forceinline__ T cg_div(T num, T den)
{
#ifndef CMF_CG_IEEE_DIV
    if constexpr (sizeof(T) == 4) return num * __builtin_amdgcn_rcpf(den);
    else {
        double y = __builtin_amdgcn_rcp(den);
        double e = __builtin_fma(-den, y, 1.0);
        y = __builtin_fma(y, e, y);
        e = __builtin_fma(-den, y, 1.0);
        y = __builtin_fma(y, e, y);
        double q = num * y;
        const double r = __builtin_fma(-den, q, num);
        return __builtin_fma(r, y, q);
    }
#else
    return num / den;
#endif
}

template <typename V>
__device__ __forceinline__ V lds_g(const V *p) { return *p; }

template <typename T>
struct CgParams {
    T *A;                 // [nrows_total, lda] matrix being updated, first solved column
    size_t lda;
    const T *B;           // opposing factor matrix, first used column
    size_t ldb;
    int k;                // columns solved (<= 8*S)
    const size_t *indptr; // CSR of the local rows
    const int *indices;
    const T *values;
    const T *bias_sub;    // explicit: x_j := x_j - bias_sub[idx_j] (fused "X - bias" sweep), or null
    // explicit model with observation weights (factors_explicit_cg weighted branches, common.c:1126-1135, :1162-1171): one
    // weight per entry in the order of `values`, and the row's lambda multiplier under scale_lam -- the driver's wsumA /
    // wsumB (collective.c:7978-8008: the sum of the row's weights, 1 for a row without entries) instead of its length
    const T *weights = nullptr;
    const T *wsum = nullptr;
    const int *order;     // row ids to process
    const RowDesc *desc;  // same rows, same order, with length and CSR offset
    int nrows;
    const T *BtB;         // implicit: k x k Gramian of B (row-major, ld = k, both triangles)
    T lam, lam_last;
    int scale_lam, scale_bias_const;
    int max_cg_steps;
    int precond;          // Jacobi-preconditioned CG (factors_*_pcg): generic kernel only
    // ---- block system with dense side information (collective_block_cg[_implicit], generic kernel only) ----
    // unknowns per row kt = koff + k: coordinates [koff, kt) couple to X (the k columns above), [0, kc) to U
    int koff = 0, kc = 0;
    const T *CtC = nullptr;   // [kc, kc] C^T C, NOT multiplied by w_side (collective.c:2156)
    const T *UC = nullptr;    // [rows_with_u, kc] U C of the local rows, NOT multiplied by w_side
    T w_side = 0;
    int rows_with_u = 0;      // local rows < rows_with_u carry side information; the others are plain rows
    int row_first = 0;        // generic kernel: first position of the processing order to handle
    // generic kernel, SPARSE side information instead of CtC / UC (collective_block_cg u_vec_sp branches,
    // collective.c:2292-2298, :2609-2621, :2847-2860): the row's attributes gather rows of C2[*, kc]
    const size_t *indptr2 = nullptr;
    const int *indices2 = nullptr;
    const T *values2 = nullptr;
    const T *C2 = nullptr;
    int *counter = nullptr;   // tiled kernels: work counter of this launch (zero at launch); a team's first row is its own
                              // index, the following ones are claimed in order (longest rows first) from here
    int p_side = 0, scale_lam_sideinfo = 0;
    // generic kernel, implicit features of the explicit model (collective_block_cg, collective.c:2301-2304, :2624-2643,
    // :2862-2868): Bi [*, ki] is gathered at the row's observed positions with unit values (right-hand side only) and
    // BiTBi [ki, ki] = Bi^T Bi (unweighted) acts on the unknowns [koff, koff + ki); both enter with weight w_imp
    const T *Bi = nullptr;
    const T *BiTBi = nullptr;
    int ki = 0;
    T w_imp = 0;
    // tiled kernels, explicit model with a block term (GRAMX builds): the weighted Gramian of the block system --
    // w C^T C and / or w_i Bi^T Bi, embedded in k x k -- arrives through BtB like the implicit model's, and the row's
    // constant -- w (U C)_row and / or w_i sum_{j observed} Bi_j -- is added to the first residual from rconst[row, ldr]
    const T *rconst = nullptr;
    size_t ldr = 0;
    // generic kernel, explicit model (round 6): the same pair -- a matrix every row shares (through BtB, acting on the unknowns
    // [koff, kt)) and a per-row constant in the first residual -- for the Jacobi-preconditioned solver of NA_as_zero_X with
    // observation weights (factors_explicit_pcg_NA_as_zero_weighted, common.c:1443-1613); rows without entries that the host put
    // into the launch are solved like the others (optimizeA's rule, common.c:3270-3271)
    int gx = 0;
#ifdef CMF_CG_DEBUG
    int dbg = 0;   // phase skipping for timing experiments (results are wrong): 1 gathers, 2 Gramian product, 4 tile products, 8 dot products
#endif
#ifdef CMF_CG_TICKS
    // Where a wavefront's time goes (a -DCMF_CG_TICKS build, tools/microbench/cg_ticks.py): per kernel family four 64-bit sums over
    // all wavefronts -- s_memtime ticks spent waiting for the row's gather (from the last load issued to s_waitcnt vmcnt(0)), ticks
    // in the CG passes, rows, total ticks inside the row loop.  Results are unchanged; the explicit wait moves a few instructions.
    unsigned long long *ticks = nullptr;
#endif
};
#ifdef CMF_CG_TICKS
#define CMF_TICK() __builtin_amdgcn_s_memtime()
__device__ __forceinline__ void cg_ticks_flush(unsigned long long *t, int slot, unsigned long long wait, unsigned long long pass, unsigned long long rows,
                                               unsigned long long total, int lane)
{
    if (t != nullptr && lane == 0) {
        atomicAdd(t + 4 * slot + 0, wait); atomicAdd(t + 4 * slot + 1, pass); atomicAdd(t + 4 * slot + 2, rows); atomicAdd(t + 4 * slot + 3, total);
    }
}
#endif
// Timing experiments (tools/gpu/r02_o.sh): a build with -DCMF_CG_DEBUG reads CMFREC_HIP_CG_SKIP and leaves phases out.
#ifdef CMF_CG_DEBUG
#define CMF_DBG(P, bit) (((P).dbg & (bit)) != 0)
#else
#define CMF_DBG(P, bit) false
#endif
template <int NT, int S, typename TILE, typename T>
__device__ __forceinline__ void dbg_fill_tile(TILE &tile, T val)
{
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int s = 0; s < S; s++) tile.set(t, s, val);
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) { return lanes::wave_sum(v); }

// Transposed butterfly over the lane bits 0..2 (the 8 lanes of one non-zero group): 8 values per
// lane in, the lane whose low bits are b ends with the 8-lane total of v[b].
// NT < 8 (round 5, tiles of NT entries per lane group): only v[0 .. NT-1] exist (NT >= 4).  A first-stage pair whose upper
// member v[i+4] does not exist is a plain sum into the lower half-group -- three instructions instead of the seven of a
// transposing step in double precision; the lanes b >= NT end with undefined bits nobody reads (their entry is not valid:
// pass_weight selects zero for it).
// (first stage for the pair (v[I], v[I + 4]); `if constexpr`: an ordinary `if` on the loop counter of the unrolled loop left the
//  single-precision kernels 12 registers heavier -- 179 instead of 167, a wavefront per SIMD less -- although it folds away)
template <typename T, int NT, int I>
__device__ __forceinline__ T treduce8_low_first(const T (&v)[8], bool h)
{
    if constexpr (I + 4 < NT) {
        T keep = h ? v[I + 4] : v[I];
        return keep + lanes::recv_xor4(v[I], v[I + 4]);
    } else return v[I] + lanes::xor4_lower(v[I]);          // (the lanes with bit 2 set stand for entry I + 4: none)
}
template <typename T, int NT = 8>
__device__ __forceinline__ T treduce8_low(const T (&v)[8], int lane)
{
    static_assert(NT >= 4 && NT <= 8, "entries per lane group");
    T u[4], q[2];
    bool h = (lane & 4) != 0;
    u[0] = treduce8_low_first<T, NT, 0>(v, h); u[1] = treduce8_low_first<T, NT, 1>(v, h);
    u[2] = treduce8_low_first<T, NT, 2>(v, h); u[3] = treduce8_low_first<T, NT, 3>(v, h);
    h = (lane & 2) != 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        T keep = h ? u[i + 2] : u[i];
        T send = h ? u[i] : u[i + 2];
        q[i] = keep + lanes::xor2(send);
    }
    h = (lane & 1) != 0;
    T keep = h ? q[1] : q[0];
    T send = h ? q[0] : q[1];
    return keep + lanes::xor1(send);
}

// Transposed butterfly over the lane bits 3..5 (the 8 non-zero groups): the lane whose bits 3..5
// are jj ends with the total of v[jj] over the 8 lanes that share its low bits.
template <typename T>
__device__ __forceinline__ T treduce8_high(const T (&v)[8], int lane)
{
    T u[4], q[2];
#pragma unroll
    for (int i = 0; i < 4; i++) u[i] = lanes::tswap32_add(v[i], v[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; i++) q[i] = lanes::tswap16_add(u[i], u[i + 2]);
    const bool h = (lane & 8) != 0;
    T keep = h ? q[1] : q[0];
    return keep + lanes::recv_xor8(q[0], q[1]);
}

// The vector of a pass (lane f holds element f) through the wavefront's own 64-element LDS buffer: one store, then the lane's S
// replicated elements ll + 8 s (what `replicate` fetches with 2 S ds_bpermute in double precision) and the eight Gramian weights
// jj 8 + t (what bcast8 builds with sixteen DPP moves) are plain LDS reads of addresses shared by the lanes of a group (broadcast
// reads, no conflicts) -- fewer vector AND fewer LDS instructions, and the round trip stands where the ds_bpermute one stood.
// Double precision (round 4): C2 3.62-3.65 -> 3.50-3.54 ms, tiny bin 0.44 -> 0.41.  Single precision measured slower (c4shard
// 7.31-7.38 -> 7.58-7.65 ms: its moves are single instructions and the register budget of three wavefronts per SIMD is full) and
// keeps the cross-lane form.  -DCMF_PASS_VECTOR_LDS=0 / =1: neither / both precisions.  profiles/r04/r04_zj_*.
#ifndef CMF_PASS_VECTOR_LDS
#define CMF_PASS_VECTOR_LDS 2
#endif
template <typename T> constexpr bool pass_vector_in_lds() { return CMF_PASS_VECTOR_LDS == 1 || (CMF_PASS_VECTOR_LDS == 2 && sizeof(T) == 8); }
template <typename T, int S>
__device__ __forceinline__ void replicate(T vdist, T (&vrep)[S], int lane);
// GR: rows of the staged Gramian per lane group (gram_rows below) -- the group jj multiplies the rows GR jj + t, t < GR
template <typename T, int S, int GR = 8>
__device__ __forceinline__ void pass_vector_lds(T vdist, T (&vrep)[S], T (&wts)[8], int lane, T *__restrict__ pv)
{
    const int jj = lane >> 3, ll = lane & 7;
    pv[lane] = vdist;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < S; s++) vrep[s] = pv[ll + 8 * s];
    const T *wp = pv + GR * jj;
#pragma unroll
    for (int t = 0; t < GR; t++) wts[t] = wp[t];
    __builtin_amdgcn_wave_barrier();
}

// LDS leading dimension for the staged Gramian: odd, so that the jj-groups of a lane group land on
// distinct banks both for ds_read_b64 (32 lanes, 64 banks) and ds_read2_b64 (16 lanes, 32 banks).
__host__ __device__ constexpr int gram_ld(int S) { return 8 * S + 1; }
// Round 5 -- the Gramian's rows follow k like its columns: the k x k matrix needs ceil(k / 8) = S rows per lane group, not 8 (k = 50:
// 56 padded rows instead of 64, 7 S instead of 8 S products per lane and pass).  Where the vector of a pass travels through LDS
// (pass_vector_in_lds: double precision) the group's weights are plain reads at any offset, so those kernels take GR = S rows per
// group; the cross-lane form (single precision: bcast8 moves inside an 8-lane group) keeps 8.  Leading dimension for GR = S: the
// smallest LD >= 8 S with S LD = 8 or 24 (mod 32), which puts the four lane groups of a half-wavefront 16 banks apart.
#ifndef CMF_GRAM_ROWS_BY_K
#define CMF_GRAM_ROWS_BY_K 1         // 0: eight rows per lane group everywhere (rounds 1-4; A/B build)
#endif
template <typename T> __host__ __device__ constexpr int gram_rows(int S) { return (CMF_GRAM_ROWS_BY_K && pass_vector_in_lds<T>()) ? S : 8; }
__host__ __device__ constexpr int gram_ld_rows(int S, int GR)
{
    if (GR == 8) return gram_ld(S);
    int ld = 8 * S;
    while ((S * ld) % 32 != 8 && (S * ld) % 32 != 24) ld++;
    return ld;
}
// Single precision keeps the Gramian's rows 2q and 2q+1 interleaved -- element (r, c) at 2 ((r >> 1) LD2 + c) + (r & 1),
// LD2 = 8 S + 2 -- so that one ds_read_b64 brings the register pair a v_pk_fma_f32 wants; 4 LD2 = 8 (mod 32) keeps the
// four jj-groups of a half-wave on distinct banks.
__host__ __device__ constexpr int gram_ld2(int S) { return 8 * S + 2; }
template <typename T> __host__ __device__ constexpr int gram_elems(int S) { return sizeof(T) == 4 ? 64 * gram_ld2(S) : 64 * gram_ld(S); }
template <typename T, int S, int GR = 8> __device__ __forceinline__ int gram_index(int r, int c)
{
    if constexpr (sizeof(T) == 4) return 2 * ((r >> 1) * gram_ld2(S) + c) + (r & 1);
    else return r * gram_ld_rows(S, GR) + c;
}
// stage the k x k Gramian (row-major, ld = k) of a launch in LDS, zero-padded to 8 GR x 8 S
template <typename T, int S, int GR = 8>
__device__ __forceinline__ void stage_gramian(T *__restrict__ G, const T *__restrict__ BtB, int k, int tid, int nthreads)
{
    static_assert(GR == 8 || sizeof(T) == 8, "rows by k: double precision");
    for (int e = tid; e < 8 * GR * 8 * S; e += nthreads) {
        const int r = e / (8 * S), c = e % (8 * S);
        G[gram_index<T, S, GR>(r, c)] = (r < k && c < k) ? BtB[(size_t)r * k + c] : T(0);
    }
}

template <typename T, int S>
struct RegTile {
    T v[8][S];
    __device__ __forceinline__ void set(int t, int s, T x) { v[t][s] = x; }
};
// Single precision: the entries 2q and 2q+1 of a lane's tile sit in one aligned register pair, so that both products of
// the tile pass (c_j = B_j . v with v broadcast; out += w_j B_j with one partial sum per entry parity) and the Gramian
// product issue as v_pk_fma_f32 -- two FMAs per lane and instruction, the f32 VALU peak of this part.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int S>
struct RegTile<float, S> {
    f32x2 v[4][S];
    __device__ __forceinline__ void set(int t, int s, float x) { v[t >> 1][s][t & 1] = x; }
};

// accumulators of one pass, out[s] for the factor columns ll + 8 s (float: two partial sums per column, added by close())
template <typename T>
struct PassAcc {
    T v[8];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int s = 0; s < 8; s++) v[s] = T(0);
    }
    __device__ __forceinline__ void close(T (&out)[8]) const
    {
#pragma unroll
        for (int s = 0; s < 8; s++) out[s] = v[s];
    }
};
template <>
struct PassAcc<float> {
    f32x2 v[8];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int s = 0; s < 8; s++) v[s] = f32x2{0.f, 0.f};
    }
    __device__ __forceinline__ void close(float (&out)[8]) const
    {
#pragma unroll
        for (int s = 0; s < 8; s++) out[s] = v[s][0] + v[s][1];
    }
};

// Branch-free (like load_tile4 below): slots past the end of the tile re-read the row of the tile's first entry (their
// weight w_j is forced to zero by `valid`), factor columns past k re-read column k-1 (their vrep / Gramian entries are
// zero and the result lanes >= k are cleared).  One v_mad_u64_u32 per gathered row forms its address.
// NT entries per lane group (round 5): the tile holds 8 NT entries, entry jj NT + t of the tile in the lane group jj; its index
// (my_idx) and value sit in lane (jj, t).  NT = 8 is the 64-entry tile of rounds 1-4.
template <typename T, int S, int NT = 8>
__device__ __forceinline__ void load_tile(RegTile<T, S> &tile, const T *__restrict__ Bm, size_t ldb,
                                          int k, int my_idx, int cnt, int lane)
{
    const int jj = lane >> 3, ll = lane & 7;
    int its[8];
    its[0] = lanes::bcast8<0>(my_idx); its[1] = lanes::bcast8<1>(my_idx); its[2] = lanes::bcast8<2>(my_idx);
    its[3] = lanes::bcast8<3>(my_idx);
    if (NT > 4) its[4] = lanes::bcast8<4>(my_idx);
    if (NT > 5) its[5] = lanes::bcast8<5>(my_idx);
    if (NT > 6) its[6] = lanes::bcast8<6>(my_idx);
    if (NT > 7) its[7] = lanes::bcast8<7>(my_idx);
    const int first_idx = __builtin_amdgcn_readfirstlane(my_idx);
    const int col_last = min(ll + 8 * (S - 1), k - 1) - ll;
    const char *base = reinterpret_cast<const char *>(Bm + ll);
    const unsigned ldb_bytes = (unsigned)(ldb * sizeof(T));
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const unsigned it = (unsigned)(((jj * NT + t) < cnt) ? its[t] : first_idx);
        const T *rp = reinterpret_cast<const T *>(base + (unsigned long long)it * ldb_bytes);
#pragma unroll
        for (int s = 0; s < S; s++) tile.set(t, s, rp[(s < S - 1) ? 8 * s : col_last]);
    }
}

// One tile contribution:  c_j = B_j . vrep ; w_j = f(c_j, x_j) ; out[s] += sum_t w_j B_j[s]
// MODE 0: residual pass, MODE 1: A*p pass.
// g: the entry's observation weight (explicit model; 1 without weights -- an exact multiplication)
template <typename T, bool IMPLICIT, int MODE>
__device__ __forceinline__ T pass_weight(T coef, T x, bool valid, T g = T(1))
{
    T w;
    if (IMPLICIT) {
        if (MODE == 0) w = -(coef - T(1)) * x - coef;     // common.c:1939
        else           w = coef * (x - T(1)) + coef;      // common.c:1965
    } else {
        if (MODE == 0) w = -((coef - x) * g);             // common.c:1121-1123, :1130-1133
        else           w = coef * g;                      // common.c:1158-1159, :1166-1169
    }
    return valid ? w : T(0);
}

// lambda multiplier of a row under scale_lam: its number of entries, or the sum of its weights (common.c:679-723)
template <typename T>
__device__ __forceinline__ T row_lam_mult(const CgParams<T> &P, int row, int nnz)
{
    return (P.wsum != nullptr) ? P.wsum[row] : (T)nnz;
}
// the weight of the entry at CSR position `pos`
template <typename T, bool IMPLICIT>
__device__ __forceinline__ T entry_weight(const CgParams<T> &P, size_t pos)
{
    if (IMPLICIT) return T(1);
    return (P.weights != nullptr) ? P.weights[pos] : T(1);
}

// INIT (round 6): the pass accumulators are DEFINED by this call (first products as multiplications) instead of cleared beforehand and
// accumulated into -- the S (double) / S (packed single) clearing moves per pass go; the unused slots S .. 7 are cleared here.
template <int S, bool IMPLICIT, int MODE, int NT = 8, bool INIT = false>
__device__ __forceinline__ void tile_pass_f32(const RegTile<float, S> &tile, const float (&vrep)[S], float x, bool valid,
                                          PassAcc<float> &out, int lane, float g = 1.f)
{
    static_assert(NT % 2 == 0, "single precision: the entries of a lane group travel in pairs");
    float c[8];
#pragma unroll
    for (int q = 0; q < NT / 2; q++) {
        f32x2 acc = f32x2{0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S; s++) acc += tile.v[q][s] * f32x2{vrep[s], vrep[s]};
        c[2 * q] = acc[0]; c[2 * q + 1] = acc[1];
    }
    float coef = treduce8_low<float, NT>(c, lane);
    const float w = pass_weight<float, IMPLICIT, MODE>(coef, x, valid, g);
    float wts[8];
    wts[0] = lanes::bcast8<0>(w); wts[1] = lanes::bcast8<1>(w); wts[2] = lanes::bcast8<2>(w); wts[3] = lanes::bcast8<3>(w);
    if (NT > 4) { wts[4] = lanes::bcast8<4>(w); wts[5] = lanes::bcast8<5>(w); }
    if (NT > 6) { wts[6] = lanes::bcast8<6>(w); wts[7] = lanes::bcast8<7>(w); }
#pragma unroll
    for (int q = 0; q < NT / 2; q++) {
        const f32x2 w2 = f32x2{wts[2 * q], wts[2 * q + 1]};
#pragma unroll
        for (int s = 0; s < S; s++) out.v[s] = (INIT && q == 0) ? w2 * tile.v[q][s] : out.v[s] + w2 * tile.v[q][s];
    }
    if constexpr (INIT) {
#pragma unroll
        for (int s = S; s < 8; s++) out.v[s] = f32x2{0.f, 0.f};
    }
}

template <typename T, int S, bool IMPLICIT, int MODE, int NT = 8, bool INIT = false>
__device__ __forceinline__ void tile_pass(const RegTile<T, S> &tile, const T (&vrep)[S], T x, bool valid,
                                          PassAcc<T> &out, int lane, T g = T(1))
{
    if constexpr (std::is_same<T, float>::value) {
        tile_pass_f32<S, IMPLICIT, MODE, NT, INIT>(tile, vrep, x, valid, out, lane, g);
    } else {
    T c[8];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        T acc = T(0);
#pragma unroll
        for (int s = 0; s < S; s++) acc += tile.v[t][s] * vrep[s];
        c[t] = acc;
    }
    T coef = treduce8_low<T, NT>(c, lane);       // lane (jj, t) now holds B_j . v of its entry
    const T w = pass_weight<T, IMPLICIT, MODE>(coef, x, valid, g);
    T wts[8];
    wts[0] = lanes::bcast8<0>(w); wts[1] = lanes::bcast8<1>(w); wts[2] = lanes::bcast8<2>(w); wts[3] = lanes::bcast8<3>(w);
    if (NT > 4) wts[4] = lanes::bcast8<4>(w);
    if (NT > 5) wts[5] = lanes::bcast8<5>(w);
    if (NT > 6) wts[6] = lanes::bcast8<6>(w);
    if (NT > 7) wts[7] = lanes::bcast8<7>(w);
#pragma unroll
    for (int t = 0; t < NT; t++) {
#pragma unroll
        for (int s = 0; s < S; s++) out.v[s] = (INIT && t == 0) ? wts[t] * tile.v[t][s] : out.v[s] + wts[t] * tile.v[t][s];
    }
    if constexpr (INIT) {
#pragma unroll
        for (int s = S; s < 8; s++) out.v[s] = T(0);
    }
    }
}

// out[s] += sum_j wdist_j * G[j][ll+8s]  with the Gramian staged in LDS (rows padded to 64).
// The 8 row-slices t are dealt to the W waves of the team (wave wr takes t = wr, wr+W, ...).
template <typename T, int S, int W>
__device__ __forceinline__ void gram_pass(const T *__restrict__ G, T wdist, PassAcc<T> &out, int lane, int wr)
{
    constexpr int LD = gram_ld(S);
    const int jj = lane >> 3, ll = lane & 7;
    T wts[8];
    wts[0] = lanes::bcast8<0>(wdist); wts[1] = lanes::bcast8<1>(wdist); wts[2] = lanes::bcast8<2>(wdist);
    wts[3] = lanes::bcast8<3>(wdist); wts[4] = lanes::bcast8<4>(wdist); wts[5] = lanes::bcast8<5>(wdist);
    wts[6] = lanes::bcast8<6>(wdist); wts[7] = lanes::bcast8<7>(wdist);
    if constexpr (std::is_same<T, float>::value && W <= 4) {
        // row pairs (2q, 2q+1) of the lane's eight Gramian rows, dealt to the waves; one ds_read_b64 fetches both
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (W > 1 && (q % W) != wr) continue;
            const f32x2 *g = reinterpret_cast<const f32x2 *>(G) + (jj * 4 + q) * gram_ld2(S) + ll;
            const f32x2 w2 = f32x2{wts[2 * q], wts[2 * q + 1]};
#pragma unroll
            for (int s = 0; s < S; s++) out.v[s] += w2 * lds_g(g + 8 * s);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            if (W > 1 && (t % W) != wr) continue;
#pragma unroll
            for (int s = 0; s < S; s++) {
                if constexpr (std::is_same<T, float>::value) out.v[s][0] += wts[t] * G[gram_index<T, S>(jj * 8 + t, ll + 8 * s)];
                else out.v[s] += wts[t] * lds_g(G + (jj * 8 + t) * LD + ll + 8 * s);
            }
        }
    }
}

// the same with the eight weights already in registers (level 2 of the LDS experiment); NEG: out -= ...
template <bool NEG, typename T, int S, int W, int GR = 8>
__device__ __forceinline__ void gram_pass_w(const T *__restrict__ G, const T (&wts)[8], PassAcc<T> &out, int lane, int wr)
{
    constexpr int LD = gram_ld_rows(S, GR);
    const int jj = lane >> 3, ll = lane & 7;
    if constexpr (std::is_same<T, float>::value && W <= 4) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (W > 1 && (q % W) != wr) continue;
            const f32x2 *g = reinterpret_cast<const f32x2 *>(G) + (jj * 4 + q) * gram_ld2(S) + ll;
            const f32x2 w2 = f32x2{wts[2 * q], wts[2 * q + 1]};
#pragma unroll
            for (int s = 0; s < S; s++) out.v[s] = NEG ? out.v[s] - w2 * lds_g(g + 8 * s) : out.v[s] + w2 * lds_g(g + 8 * s);
        }
    } else {
#pragma unroll
        for (int t = 0; t < GR; t++) {
            if (W > 1 && (t % W) != wr) continue;
#pragma unroll
            for (int s = 0; s < S; s++) {
                if constexpr (std::is_same<T, float>::value) {
                    const float g = G[gram_index<T, S>(jj * 8 + t, ll + 8 * s)];
                    out.v[s][0] = NEG ? out.v[s][0] - wts[t] * g : out.v[s][0] + wts[t] * g;
                } else {
                    const T g = lds_g(G + (jj * GR + t) * LD + ll + 8 * s);
                    out.v[s] = NEG ? out.v[s] - wts[t] * g : out.v[s] + wts[t] * g;
                }
            }
        }
    }
}

template <typename T, int S>
__device__ __forceinline__ void replicate(T vdist, T (&vrep)[S], int lane)
{
    const int ll = lane & 7;
#pragma unroll
    for (int s = 0; s < S; s++) vrep[s] = __shfl(vdist, s * 8 + ll);
}

// Persistent kernel: W waves cooperate on one row (W = waves per row, blockDim.x = 64*W*RPB where
// RPB rows are processed concurrently by one workgroup).
// GRAMX (explicit model only): block systems under CG (collective_block_cg with dense side information on every row and /
// or implicit features, src/collective.c:2134-2903, no k_user offset): out -= / += G v on top of the gathered part, G staged
// in LDS exactly like the implicit model's B^T B, and a per-row constant in the first residual.
// NRES_: resident tiles per wavefront (0: two for the 8-wave teams of the single-precision build, one otherwise).  Round 4: the
// single-precision rows of 257..512 entries run on FOUR waves with two tiles each instead of eight waves of which three to
// five only take part in the barriers (config 4: the 257..1024 bin ran at 0.43-0.46 of the HBM peak against 0.55-0.62 for the
// 4-wave bin below it).
template <typename T, int S, bool IMPLICIT, int W, int RPB, bool GRAMX = false, int NRES_ = 0, int NTSEL = 0>
__global__ void __launch_bounds__(64 * W * RPB, (cg_waves_per_simd<T, W, NRES_, NTSEL>()))
cg_rows_kernel(const CgParams<T> P)
{
    constexpr bool GRAM = IMPLICIT || GRAMX;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *G = reinterpret_cast<T *>(smem_raw);                                  // [64][LD] (implicit / block systems)
    T *red = G + (GRAM ? gram_elems<T>(S) : 0);                              // [RPB][2][W][64]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int grp = wave / W;      // which concurrent row of this workgroup
    const int wr = wave % W;       // wave index inside the row team
    const int k = P.k;

    const int nteams = gridDim.x * RPB;
    // Dynamic row scheduling: team t starts with position t of the processing order and claims every further
    // position from the launch's counter, always two rows ahead of the one it is solving (the descriptor / index
    // prefetch below needs them early).  Rows are sorted longest first, so this is longest-processing-time-first
    // list scheduling; above all a team that becomes resident late (other kernels on the CUs: a collective, the
    // neighbouring bin) simply takes fewer rows instead of dragging a full static share behind the launch.
    // Multi-wave teams: wave 0 claims, the position travels through LDS and is ordered by the barrier of the row's
    // first pass.
    // (Same-address atomics with return serialise at about 7 ns each on this part -- 2 ms for the 250 k short rows of
    // C2's users -- so a launch has CG_NCOUNTERS counters on separate cache lines; counter j hands out the positions
    // nteams + j + CG_NCOUNTERS * c, and a team uses the counter of its index modulo CG_NCOUNTERS.)
    __shared__ int s_claim[4];
    constexpr bool PV = pass_vector_in_lds<T>();
    __shared__ __attribute__((aligned(16))) T s_pv[PV ? W * RPB : 1][PV ? 64 : 1];
    const int cslot = (blockIdx.x * RPB + grp) % CG_NCOUNTERS;
    int *const my_counter = P.counter + cslot * CG_COUNTER_STRIDE;
    const int cbase = nteams + cslot;
    auto issue_claim = [&]() -> int {
        int v = 0;
        if (wr == 0 && lane == 0) v = atomicAdd(my_counter, 1);
        return v;
    };
    int rnxt, rnn;
    if (W == 1) {
        const int c1 = issue_claim(), c2 = issue_claim();
        rnxt = cbase + CG_NCOUNTERS * __builtin_amdgcn_readfirstlane(c1);
        rnn = cbase + CG_NCOUNTERS * __builtin_amdgcn_readfirstlane(c2);
    } else if (wr == 0 && lane == 0) {
        s_claim[2] = cbase + CG_NCOUNTERS * atomicAdd(my_counter, 1);
        s_claim[3] = cbase + CG_NCOUNTERS * atomicAdd(my_counter, 1);
    }
    constexpr int GR = gram_rows<T>(S);
    if (GRAM) stage_gramian<T, S, GR>(G, P.BtB, k, tid, blockDim.x);
    if (GRAM || W > 1) __syncthreads();
    if (W > 1) { rnxt = s_claim[2]; rnn = s_claim[3]; }
    T *myred = red + (size_t)grp * 2 * W * 64;

    int buf = 0;   // cross-wave exchange buffer parity; persists across rows (see docs/DESIGN_HISTORY.md 3.1)
    static_assert(W == 1 || RPB == 1, "multi-wave teams own their workgroup (barriers are per row)");

    // Software pipeline over the team's rows: while row i is being solved, the descriptor of row
    // i+2 and the first-tile indices / values / warm start of row i+1 are already in flight, so
    // the only exposed memory latency per row is the gather of its opposing-factor rows.
    // Single precision, 8-wave teams: a wavefront keeps TWO tiles (2 x 64 VGPRs), so rows up to 1024 entries -- the whole bin
    // of this launch -- are gathered once instead of once per pass (config 4: the bin ran at 2.9 TB/s against 4.5-5.2 for
    // the bins that already were resident).  The register budget goes from three to two wavefronts per SIMD.
    constexpr int NRES = NRES_ > 0 ? NRES_ : ((std::is_same<T, float>::value && W == 8) ? 2 : 1);
    static_assert(NRES == 1 || std::is_same<T, float>::value, "two resident tiles: single precision");
    // Round 5 -- the tile follows the row: a resident row of nnz entries runs on tiles of 8 NT entries, NT = entries per lane
    // group = ceil(nnz / (8 W NRES)) in [NT_MIN, 8] (single precision: even, the entries of a lane travel in pairs), instead of
    // always 8.  A launch holds rows of (W NRES 32, W NRES 64] entries, so with the 64-entry tile a third of all tile slots were
    // padding (C2: 24.4 M slots for 16.9 M entries; with NT by row 18.7 M) -- padding that costs the full 2 S FMAs per slot and
    // pass, its share of the reduction over the column lanes, a weight broadcast and S gather instructions.  NT is a function of
    // the row's length alone, so a row's arithmetic (and its bits) do not depend on the launch, the shard or the neighbours.
    // Rows that are not resident (more than W NRES 64 entries: launches outside the length bins) keep the 64-entry tile.
    static_assert(NTSEL == 0 || (S >= CMF_CG_NT_MIN_S && (sizeof(T) == 4 || S <= CG_NTSEL_MAX_S) && NRES_ == 0),
                  "launches by tile size: tiles by row length (double precision: k <= 56)");
    constexpr int NT_MIN = (S >= CMF_CG_NT_MIN_S && (CMF_CG_NT_F32 || !std::is_same<T, float>::value)) ? (std::is_same<T, float>::value ? 6 : 5) : 8;
    auto nt_of = [&](int nnz_) -> int {
        int nt = (nnz_ + 8 * W * NRES - 1) / (8 * W * NRES);
        if (std::is_same<T, float>::value) nt = (nt + 1) & ~1;
        return (nt > 8) ? 8 : (nt < NT_MIN ? NT_MIN : nt);
    };
    struct Pre { int idx; T x; T a; int idx2; T x2; T g; T g2; };
    auto load_desc = [&](int rix_) -> RowDesc {
        RowDesc d; d.row = 0; d.nnz = 0; d.st = 0;
        if (rix_ < P.nrows) d = P.desc[rix_];
        d.row = __builtin_amdgcn_readfirstlane(d.row);
        d.nnz = __builtin_amdgcn_readfirstlane(d.nnz);
        d.st = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(d.st >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)(d.st & 0xffffffffu));
        return d;
    };
    // the lane (jj, b) carries entry jj NT + b of its wavefront's tile (b < NT); NT = 8: entry = lane
    auto load_pre = [&](const RowDesc &d) -> Pre {
        Pre q; q.idx = 0; q.x = T(0); q.a = T(0); q.idx2 = 0; q.x2 = T(0); q.g = T(1); q.g2 = T(1);
        const int nt = nt_of(d.nnz), tl = 8 * nt;
        const int b = lane & 7;
        const int ent = (lane >> 3) * nt + b;
        const int cnt = min(tl, d.nnz - wr * tl);
        if (b < nt && ent < cnt) {
            const size_t pos = d.st + (size_t)wr * tl + ent;
            q.idx = P.indices[pos];
            q.x = P.values[pos];
            q.g = entry_weight<T, IMPLICIT>(P, pos);
            if (!IMPLICIT && P.bias_sub != nullptr) q.x -= P.bias_sub[q.idx];
        }
        if (NRES == 2) {
            const int cnt2 = min(tl, d.nnz - (wr + W) * tl);
            if (b < nt && ent < cnt2) {
                const size_t pos = d.st + (size_t)(wr + W) * tl + ent;
                q.idx2 = P.indices[pos];
                q.x2 = P.values[pos];
                q.g2 = entry_weight<T, IMPLICIT>(P, pos);
                if (!IMPLICIT && P.bias_sub != nullptr) q.x2 -= P.bias_sub[q.idx2];
            }
        }
        if (d.nnz > 0 && lane < k) q.a = P.A[(size_t)d.row * P.lda + lane];
        return q;
    };
    int rix = blockIdx.x * RPB + grp;
    RowDesc dcur = load_desc(rix);
    RowDesc dnxt = load_desc(rnxt);
    Pre pcur = load_pre(dcur);
    int pend = issue_claim();      // the position after rnn; lands while this row is solved
#ifdef CMF_CG_TICKS
    unsigned long long tk_wait = 0, tk_pass = 0, tk_rows = 0;
    const unsigned long long tk_begin = CMF_TICK();
#endif
    int it = 0;
    RowDesc dnn;
    Pre pnxt;
    // one row on tiles of 8 NT entries
    auto solve_row = [&](auto nt_tag) {
        constexpr int NT = decltype(nt_tag)::value;
        constexpr int TL = 8 * NT;
        const int row = dcur.row;
        const size_t st = dcur.st;
        const int nnz = dcur.nnz;
        const int ntiles = (nnz + TL - 1) / TL;
        const int my_ntiles = (ntiles > wr) ? (ntiles - wr + W - 1) / W : 0;
        const bool resident = NT < 8 || my_ntiles <= NRES;      // (a tile of fewer than 64 entries is only chosen for a row that fits)

        T lam = P.lam, lam_last = P.lam_last;
        if (GRAMX && P.kc > 0) {                              // rows of the block system: collective.c:1285-1355
            if (P.scale_lam || P.scale_lam_sideinfo) {
                T mult = row_lam_mult(P, row, nnz);
                if (P.scale_lam_sideinfo) mult += (T)P.p_side;
                lam *= mult; lam_last *= mult;
            }
        } else if (!IMPLICIT && P.scale_lam) {                // common.c:679-723
            const T mult = row_lam_mult(P, row, nnz);
            lam *= mult;
            if (!P.scale_bias_const) lam_last *= mult;
        }
        T *arow = P.A + (size_t)row * P.lda;
        T a_d = pcur.a;

        // first tile of this wave: gather now (critical path), then start the next row's loads
        RegTile<T, S> tile;
        const int ent = (lane >> 3) * NT + (lane & 7);     // this lane's entry of the tile (lanes with (lane & 7) >= NT: none)
        const bool lane_has = (NT == 8) || (lane & 7) < NT;
        const int cnt0 = min(TL, nnz - wr * TL);
        T x_res = pcur.x;
        const T g_res = pcur.g;
        bool valid_res = lane_has && ent < cnt0;
        if (CMF_DBG(P, 1)) dbg_fill_tile<8, S>(tile, (T)(pcur.idx & 3) * (T)0.001);
        else if (cnt0 > 0) load_tile<T, S, NT>(tile, P.B, P.ldb, k, pcur.idx, cnt0, lane);
        RegTile<T, (NRES == 2) ? S : 1> tile2;       // second resident tile (entries (wr + W) * TL ...)
        const int cnt1 = (NRES == 2) ? min(TL, nnz - (wr + W) * TL) : 0;
        const T x2_res = pcur.x2, g2_res = pcur.g2;
        const bool valid2_res = lane_has && ent < cnt1;
        if constexpr (NRES == 2) {
            if (cnt1 > 0 && !CMF_DBG(P, 1)) load_tile<T, S, NT>(tile2, P.B, P.ldb, k, pcur.idx2, cnt1, lane);
        }
        dnn = load_desc(rnn);
        pnxt = load_pre(dnxt);
#ifdef CMF_CG_TICKS
        const unsigned long long tk0 = CMF_TICK();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tk1 = CMF_TICK();
        tk_wait += tk1 - tk0;
#endif

        auto run_pass = [&](T vdist, auto mode_tag, bool first) -> T {
            constexpr int MODE = decltype(mode_tag)::value;
            // keep the staged Gramian in LDS: without this the compiler hoists its 8*S loads per
            // lane out of the pass / row loops and pins 16*S VGPRs (occupancy 2 -> 1 wave/SIMD)
            asm volatile("" ::: "memory");
            T vrep[S];
            T gw[8];
            if constexpr (PV) pass_vector_lds<T, S, GR>(vdist, vrep, gw, lane, &s_pv[wave][0]);
            else replicate<T, S>(vdist, vrep, lane);
            PassAcc<T> acc;
            acc.zero();
            if constexpr (NRES == 2) {
                // both tiles of the wave are resident: the launch holds rows of at most 2 * W * 64 = 1024 entries (the host
                // keeps the split-row boundary at or below that in single precision)
                if (cnt0 > 0 && !CMF_DBG(P, 4)) tile_pass<T, S, IMPLICIT, MODE, NT>(tile, vrep, x_res, valid_res, acc, lane, g_res);
                if (cnt1 > 0 && !CMF_DBG(P, 4)) tile_pass<T, S, IMPLICIT, MODE, NT>(tile2, vrep, x2_res, valid2_res, acc, lane, g2_res);
            } else if constexpr (NT < 8) {
                // (round 6: defining the accumulators by the first tile product instead of clearing them was measured here -- the
                //  guard for a wavefront without entries brings the clearing moves back, 3022 -> 3054 vector instructions; the
                //  tiny kernel, which has no such guard, keeps that form)
                if (cnt0 > 0 && !CMF_DBG(P, 4)) tile_pass<T, S, IMPLICIT, MODE, NT>(tile, vrep, x_res, valid_res, acc, lane, g_res);
            } else
            for (int tl = wr; tl < ntiles; tl += W) {
                T x, g; bool valid;
                const bool have = (tl == wr) && (resident || first);   // still in registers
                if (!have) {
                    // (rows beyond the team's registers: NT = 8 by construction, entry = lane)
                    const int cnt = min(TL, nnz - tl * TL);
                    valid = lane_has && ent < cnt;
                    const size_t pos = st + (size_t)tl * TL + ent;
                    int my_idx = valid ? P.indices[pos] : 0;
                    x = valid ? P.values[pos] : T(0);
                    g = valid ? entry_weight<T, IMPLICIT>(P, pos) : T(1);
                    if (!IMPLICIT && P.bias_sub != nullptr && valid) x -= P.bias_sub[my_idx];
                    if (!CMF_DBG(P, 1)) load_tile<T, S, NT>(tile, P.B, P.ldb, k, my_idx, cnt, lane);
                } else {
                    x = x_res; g = g_res; valid = valid_res;
                }
                if (!CMF_DBG(P, 4)) tile_pass<T, S, IMPLICIT, MODE, NT>(tile, vrep, x, valid, acc, lane, g);
            }
            if (GRAM && !CMF_DBG(P, 2)) {   // common.c:1932 / :1958; collective.c:2609-2643
                if constexpr (PV) gram_pass_w<MODE == 0, T, S, W, GR>(G, gw, acc, lane, wr);
                else gram_pass<T, S, W>(G, (MODE == 0) ? -vdist : vdist, acc, lane, wr);
            }
            T out[8];
            acc.close(out);
            T tot = treduce8_high<T>(out, lane);                              // lane f <- element f
            if (W > 1) {
                T *rb = myred + (size_t)buf * W * 64;
                rb[wr * 64 + lane] = tot;
                if (first && wr == 0 && lane == 0) s_claim[it & 1] = cbase + CG_NCOUNTERS * pend;
                __syncthreads();
                tot = T(0);
#pragma unroll
                for (int w = 0; w < W; w++) tot += rb[w * 64 + lane];
                buf ^= 1;
            }
            return tot;
        };

        // ---- residual (common.c:1932-1943 / :1112-1139) ----
        T r_d = run_pass(a_d, std::integral_constant<int, 0>{}, true);
        r_d -= lam * a_d;
        if (!IMPLICIT && lam != lam_last && lane == k - 1) r_d -= (lam_last - lam) * a_d;
        if (GRAMX && P.rconst != nullptr && lane < k) r_d += P.rconst[(size_t)row * P.ldr + lane];
        if (lane >= k) r_d = T(0);
        T p_d = r_d;
        T r_old = CMF_DBG(P, 8) ? T(1) : wave_sum(r_d * r_d);
        // r_old / r_new are bit-identical on every wave of the team (same LDS partials summed in
        // the same order), so the data-dependent exits below are uniform over the workgroup.
        bool done = (r_old <= (T)1e-12);            // common.c:1952 / :1147
        for (int step = 0; step < P.max_cg_steps && !done; step++) {
            T Ap_d = run_pass(p_d, std::integral_constant<int, 1>{}, false);
            Ap_d += lam * p_d;
            if (!IMPLICIT && lam != lam_last && lane == k - 1) Ap_d += (lam_last - lam) * p_d;
            if (lane >= k) Ap_d = T(0);
            T alpha = CMF_DBG(P, 8) ? T(0.001) : cg_div(r_old, wave_sum(Ap_d * p_d));
            a_d += alpha * p_d;
            r_d -= alpha * Ap_d;
            T r_new = CMF_DBG(P, 8) ? T(0.5) : wave_sum(r_d * r_d);
            if (r_new <= (T)1e-8) done = true;      // common.c:1979 / :1180
            else {
                p_d = p_d * cg_div(r_new, r_old) + r_d;
                r_old = r_new;
            }
        }
        if (wr == 0 && lane < k) arow[lane] = a_d;
#ifdef CMF_CG_TICKS
        tk_pass += CMF_TICK() - tk1; tk_rows += (wr == 0);
#endif
    };
    for (; rix < P.nrows; it++) {
        const int nt = nt_of(dcur.nnz);      // uniform over the team
        if constexpr (NT_MIN == 8) solve_row(std::integral_constant<int, 8>{});
        else if constexpr (std::is_same<T, float>::value) {
            if constexpr (NTSEL == 1) solve_row(std::integral_constant<int, 6>{});         // the launch holds rows of at most CG_NT_LOW * 8 W entries
            else if constexpr (NTSEL == 2) solve_row(std::integral_constant<int, 8>{});    // ... of more than that
            else if (nt == 6) solve_row(std::integral_constant<int, 6>{});
            else solve_row(std::integral_constant<int, 8>{});
        } else if constexpr (NTSEL == 1) {      // the launch holds rows of at most CG_NT_LOW * 8 W entries
            if (nt <= 5) solve_row(std::integral_constant<int, 5>{});
            else solve_row(std::integral_constant<int, 6>{});
        } else if constexpr (NTSEL == 2) {      // ... of more than that
            if (nt <= 7) solve_row(std::integral_constant<int, 7>{});
            else solve_row(std::integral_constant<int, 8>{});
        } else {
            switch (nt) {
                case 5: solve_row(std::integral_constant<int, 5>{}); break;
                case 6: solve_row(std::integral_constant<int, 6>{}); break;
                case 7: solve_row(std::integral_constant<int, 7>{}); break;
                default: solve_row(std::integral_constant<int, 8>{}); break;
            }
        }
        const int r3 = (W == 1) ? cbase + CG_NCOUNTERS * __builtin_amdgcn_readfirstlane(pend) : s_claim[it & 1];
        dcur = dnxt; dnxt = dnn; pcur = pnxt;
        rix = rnxt; rnxt = rnn; rnn = r3;
        pend = issue_claim();
    }
#ifdef CMF_CG_TICKS
    cg_ticks_flush(P.ticks, W == 1 ? 0 : W == 2 ? 1 : W == 4 ? 2 : 3, tk_wait, tk_pass, tk_rows, CMF_TICK() - tk_begin, lane);
#endif
}

// ------------------------------------------------------------------------------------------
// Tiny rows (<= 32 non-zeros): half-size register tiles (8 groups x 4 non-zeros), two of them per
// wavefront used as a double buffer: while the CG passes of row i run on one buffer, the gather of
// row i+1 lands in the other, so the gather latency of these short rows hides behind compute
// inside a single wavefront (the register file admits only two wavefronts per SIMD).
constexpr int TILE4 = 32;

template <typename T, int S>
struct RegTile4 {
    T v[4][S];
    __device__ __forceinline__ void set(int t, int s, T x) { v[t][s] = x; }
    __device__ __forceinline__ T get(int t, int s) const { return v[t][s]; }
};
template <int S>
struct RegTile4<float, S> {
    f32x2 v[2][S];
    __device__ __forceinline__ void set(int t, int s, float x) { v[t >> 1][s][t & 1] = x; }
    __device__ __forceinline__ f32x2 pair0(int s) const { return v[0][s]; }      // entries 0 and 1 of the lane group
};

// lanes 2t and 2t+1 of every 8-lane group carry non-zero jj*4+t (my_idx / x are loaded with
// position lane>>1).  Branch-free: slots past the end of the row re-read the row's first entry
// (their weight w_j is forced to zero), factor columns past k re-read column k-1 (their vrep /
// Gramian entries are zero and the result lanes >= k are cleared).
template <typename T, int S>
__device__ __forceinline__ void load_tile4(RegTile4<T, S> &tile, const T *__restrict__ Bm, size_t ldb,
                                           int k, int my_idx, int cnt, int lane)
{
    const int jj = lane >> 3, ll = lane & 7;
    int its[4];
    its[0] = lanes::bcast8<0>(my_idx); its[1] = lanes::bcast8<2>(my_idx);
    its[2] = lanes::bcast8<4>(my_idx); its[3] = lanes::bcast8<6>(my_idx);
    const int first_idx = __builtin_amdgcn_readfirstlane(my_idx);
    const int col_last = min(ll + 8 * (S - 1), k - 1) - ll;
    const char *base = reinterpret_cast<const char *>(Bm + ll);
    const unsigned ldb_bytes = (unsigned)(ldb * sizeof(T));
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const unsigned it = (unsigned)(((jj * 4 + t) < cnt) ? its[t] : first_idx);
        const T *rp = reinterpret_cast<const T *>(base + (unsigned long long)it * ldb_bytes);     // one v_mad_u64_u32
#pragma unroll
        for (int s = 0; s < S; s++) tile.set(t, s, rp[(s < S - 1) ? 8 * s : col_last]);
    }
}

template <typename T>
__device__ __forceinline__ T treduce4_low(const T (&v)[4], int lane)
{
    T u[2];
    bool h = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        T keep = h ? v[i + 2] : v[i];
        u[i] = keep + lanes::recv_xor4(v[i], v[i + 2]);
    }
    h = (lane & 2) != 0;
    T keep = h ? u[1] : u[0];
    T send = h ? u[0] : u[1];
    T q = keep + lanes::xor2(send);
    return q + lanes::xor1(q);          // lanes 2t, 2t+1 both hold the total of v[t]
}

// (TILE: a RegTile4, or a RegTile whose first four entries per lane group are used -- the mixed launch of the pair kernel)
template <int S, bool IMPLICIT, int MODE, typename TILE, bool INIT = false>
__device__ __forceinline__ void tile_pass4_f32(const TILE &tile, const float (&vrep)[S], float x, bool valid,
                                           PassAcc<float> &out, int lane, float g = 1.f)
{
    float c[4];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        f32x2 acc = f32x2{0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S; s++) acc += tile.v[q][s] * f32x2{vrep[s], vrep[s]};
        c[2 * q] = acc[0]; c[2 * q + 1] = acc[1];
    }
    float coef = treduce4_low<float>(c, lane);
    const float w = pass_weight<float, IMPLICIT, MODE>(coef, x, valid, g);
    float wts[4];
    wts[0] = lanes::bcast8<0>(w); wts[1] = lanes::bcast8<2>(w); wts[2] = lanes::bcast8<4>(w); wts[3] = lanes::bcast8<6>(w);
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const f32x2 w2 = f32x2{wts[2 * q], wts[2 * q + 1]};
#pragma unroll
        for (int s = 0; s < S; s++) out.v[s] = (INIT && q == 0) ? w2 * tile.v[q][s] : out.v[s] + w2 * tile.v[q][s];
    }
    if constexpr (INIT) {
#pragma unroll
        for (int s = S; s < 8; s++) out.v[s] = f32x2{0.f, 0.f};
    }
}

template <typename T, int S, bool IMPLICIT, int MODE, typename TILE, bool INIT = false>
__device__ __forceinline__ void tile_pass4(const TILE &tile, const T (&vrep)[S], T x, bool valid,
                                           PassAcc<T> &out, int lane, T g = T(1))
{
    if constexpr (std::is_same<T, float>::value) {
        tile_pass4_f32<S, IMPLICIT, MODE, TILE, INIT>(tile, vrep, x, valid, out, lane, g);
    } else {
    T c[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        T acc = T(0);
#pragma unroll
        for (int s = 0; s < S; s++) acc += tile.v[t][s] * vrep[s];
        c[t] = acc;
    }
    T coef = treduce4_low<T>(c, lane);
    const T w = pass_weight<T, IMPLICIT, MODE>(coef, x, valid, g);
    T wts[4];
    wts[0] = lanes::bcast8<0>(w); wts[1] = lanes::bcast8<2>(w); wts[2] = lanes::bcast8<4>(w); wts[3] = lanes::bcast8<6>(w);
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
        for (int s = 0; s < S; s++) out.v[s] = (INIT && t == 0) ? wts[t] * tile.v[t][s] : out.v[s] + wts[t] * tile.v[t][s];
    }
    if constexpr (INIT) {
#pragma unroll
        for (int s = S; s < 8; s++) out.v[s] = T(0);
    }
    }
}

// Rows of at most 16 entries: a 16-slot tile, TWO entries per lane group instead of four -- lanes 4t .. 4t+3 of every 8-lane
// group carry non-zero jj*2+t (my_idx / x are loaded with position lane>>2).  Half the tile products and one butterfly stage less
// than the 32-slot tile for the rows that fill at most half of it (more than half of a tiny bin, typically).
template <typename T, int S>
struct RegTile2 {
    T v[2][S];
    __device__ __forceinline__ void set(int t, int s, T x) { v[t][s] = x; }
    __device__ __forceinline__ T get(int t, int s) const { return v[t][s]; }
};
template <int S>
struct RegTile2<float, S> {
    f32x2 v[S];
    __device__ __forceinline__ void set(int t, int s, float x) { v[s][t] = x; }
    __device__ __forceinline__ f32x2 pair0(int s) const { return v[s]; }
};

// (TILE: RegTile2, or a RegTile4 whose first two entries per lane group are used -- the mixed launch of the tiny kernel)
template <typename T, int S, typename TILE>
__device__ __forceinline__ void load_tile2(TILE &tile, const T *__restrict__ Bm, size_t ldb,
                                           int k, int my_idx, int cnt, int lane)
{
    const int jj = lane >> 3, ll = lane & 7;
    int its[2];
    its[0] = lanes::bcast8<0>(my_idx); its[1] = lanes::bcast8<4>(my_idx);
    const int first_idx = __builtin_amdgcn_readfirstlane(my_idx);
    const int col_last = min(ll + 8 * (S - 1), k - 1) - ll;
    const char *base = reinterpret_cast<const char *>(Bm + ll);
    const unsigned ldb_bytes = (unsigned)(ldb * sizeof(T));
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const unsigned it = (unsigned)(((jj * 2 + t) < cnt) ? its[t] : first_idx);
        const T *rp = reinterpret_cast<const T *>(base + (unsigned long long)it * ldb_bytes);
#pragma unroll
        for (int s = 0; s < S; s++) tile.set(t, s, rp[(s < S - 1) ? 8 * s : col_last]);
    }
}

template <typename T>
__device__ __forceinline__ T treduce2_low(const T (&v)[2], int lane)
{
    const bool h = (lane & 4) != 0;
    T u = (h ? v[1] : v[0]) + lanes::recv_xor4(v[0], v[1]);
    u += lanes::xor2(u);
    return u + lanes::xor1(u);          // lanes 4t .. 4t+3 hold the total of v[t]
}

template <typename T, int S, bool IMPLICIT, int MODE, typename TILE, bool INIT = false>
__device__ __forceinline__ void tile_pass2(const TILE &tile, const T (&vrep)[S], T x, bool valid,
                                           PassAcc<T> &out, int lane, T g = T(1))
{
    T c[2];
    if constexpr (std::is_same<T, float>::value) {
        f32x2 acc = f32x2{0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S; s++) acc += tile.pair0(s) * f32x2{vrep[s], vrep[s]};
        c[0] = acc[0]; c[1] = acc[1];
    } else {
#pragma unroll
        for (int t = 0; t < 2; t++) {
            T acc = T(0);
#pragma unroll
            for (int s = 0; s < S; s++) acc += tile.get(t, s) * vrep[s];
            c[t] = acc;
        }
    }
    const T coef = treduce2_low<T>(c, lane);
    const T w = pass_weight<T, IMPLICIT, MODE>(coef, x, valid, g);
    const T w0 = lanes::bcast8<0>(w), w1 = lanes::bcast8<4>(w);
    if constexpr (std::is_same<T, float>::value) {
        const f32x2 w2 = f32x2{w0, w1};
#pragma unroll
        for (int s = 0; s < S; s++) out.v[s] = INIT ? w2 * tile.pair0(s) : out.v[s] + w2 * tile.pair0(s);
        if constexpr (INIT) {
#pragma unroll
            for (int s = S; s < 8; s++) out.v[s] = f32x2{0.f, 0.f};
        }
    } else {
#pragma unroll
        for (int s = 0; s < S; s++) out.v[s] = INIT ? w0 * tile.get(0, s) : out.v[s] + w0 * tile.get(0, s);
#pragma unroll
        for (int s = 0; s < S; s++) out.v[s] += w1 * tile.get(1, s);
        if constexpr (INIT) {
#pragma unroll
            for (int s = S; s < 8; s++) out.v[s] = T(0);
        }
    }
}

// the tiny kernel's tile by entries per lane group (NE = 4: 32 slots, NE = 2: 16 slots)
template <typename T, int S, int NE> struct TinyTile { using type = RegTile4<T, S>; };
template <typename T, int S> struct TinyTile<T, S, 2> { using type = RegTile2<T, S>; };

#ifndef CMF_TINY_WAVES_PER_SIMD
#define CMF_TINY_WAVES_PER_SIMD 4     // 4: single tile buffer in a 128-VGPR budget (measured 10 % faster than
                                      // 2: double-buffered tiles, 2 x 56 VGPRs, 2 waves/SIMD)
#endif
// Round 4: in double precision the lane's 8 x S Gramian elements of the tiny kernel live in REGISTERS for the whole launch instead
// of being read from LDS in every pass (28 ds_read2_b64 per pass and wavefront: with four wavefronts per SIMD the LDS pipe of the CU
// was ~70 % busy with them).  Costs 16 S registers, i.e. two wavefronts per SIMD instead of four for the kernels that carry a
// Gramian (C2: tiny bin 0.502 -> 0.476 ms users, 0.166 -> 0.148 ms items, iteration 3.70 -> 3.65 ms; profiles/r04).  The
// two-rows-per-wavefront kernel below would need 32 S registers and keeps its Gramian in LDS.
#ifndef CMF_TINY_GREG_F32
#define CMF_TINY_GREG_F32 1          // single precision: 8 S registers more per lane, three wavefronts per SIMD (c4shard 8.03-8.17 -> 7.91-7.92 ms; 0 = Gramian in LDS)
#endif
#ifndef CMF_TINY_GREG_F64
#define CMF_TINY_GREG_F64 1          // double precision: 0 = Gramian in LDS, four wavefronts per SIMD
#endif
template <typename T, bool GRAM> constexpr bool tiny_greg() { return GRAM && (sizeof(T) == 8 ? CMF_TINY_GREG_F64 != 0 : CMF_TINY_GREG_F32 != 0); }
template <typename T, bool GRAM> constexpr int tiny_waves_per_simd() { return tiny_greg<T, GRAM>() ? (sizeof(T) == 8 ? 2 : 3) : CMF_TINY_WAVES_PER_SIMD; }
template <typename T, int S, int GR = 8>
struct GramRegs {
    T v[GR][S];
    __device__ __forceinline__ void load(const T *__restrict__ G, int lane)
    {
        const int jj = lane >> 3, ll = lane & 7;
#pragma unroll
        for (int t = 0; t < GR; t++)
#pragma unroll
            for (int s = 0; s < S; s++) v[t][s] = G[gram_index<T, S, GR>(jj * GR + t, ll + 8 * s)];
    }
};
// single precision: the row pairs (2q, 2q+1) as the register pairs a v_pk_fma_f32 wants (the LDS layout of gram_index)
template <int S>
struct GramRegs<float, S, 8> {
    f32x2 v[4][S];
    __device__ __forceinline__ void load(const float *__restrict__ G, int lane)
    {
        const int jj = lane >> 3, ll = lane & 7;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x2 *g = reinterpret_cast<const f32x2 *>(G) + (jj * 4 + q) * gram_ld2(S) + ll;
#pragma unroll
            for (int s = 0; s < S; s++) v[q][s] = g[8 * s];
        }
    }
};
template <int S>
__device__ __forceinline__ void gram_pass_regs(const GramRegs<float, S> &R, float wdist, PassAcc<float> &out)
{
    float wts[8];
    wts[0] = lanes::bcast8<0>(wdist); wts[1] = lanes::bcast8<1>(wdist); wts[2] = lanes::bcast8<2>(wdist);
    wts[3] = lanes::bcast8<3>(wdist); wts[4] = lanes::bcast8<4>(wdist); wts[5] = lanes::bcast8<5>(wdist);
    wts[6] = lanes::bcast8<6>(wdist); wts[7] = lanes::bcast8<7>(wdist);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const f32x2 w2 = f32x2{wts[2 * q], wts[2 * q + 1]};
#pragma unroll
        for (int s = 0; s < S; s++) out.v[s] += w2 * R.v[q][s];
    }
}
template <typename T, int S>
__device__ __forceinline__ void gram_pass_regs(const GramRegs<T, S> &R, T wdist, PassAcc<T> &out)
{
    T wts[8];
    wts[0] = lanes::bcast8<0>(wdist); wts[1] = lanes::bcast8<1>(wdist); wts[2] = lanes::bcast8<2>(wdist);
    wts[3] = lanes::bcast8<3>(wdist); wts[4] = lanes::bcast8<4>(wdist); wts[5] = lanes::bcast8<5>(wdist);
    wts[6] = lanes::bcast8<6>(wdist); wts[7] = lanes::bcast8<7>(wdist);
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int s = 0; s < S; s++) out.v[s] += wts[t] * R.v[t][s];
}

template <bool NEG, int S>
__device__ __forceinline__ void gram_pass_regs_w(const GramRegs<float, S> &R, const float (&wts)[8], PassAcc<float> &out)
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const f32x2 w2 = f32x2{wts[2 * q], wts[2 * q + 1]};
#pragma unroll
        for (int s = 0; s < S; s++) out.v[s] = NEG ? out.v[s] - w2 * R.v[q][s] : out.v[s] + w2 * R.v[q][s];
    }
}
template <bool NEG, typename T, int S, int GR>
__device__ __forceinline__ void gram_pass_regs_w(const GramRegs<T, S, GR> &R, const T (&wts)[8], PassAcc<T> &out)
{
#pragma unroll
    for (int t = 0; t < GR; t++)
#pragma unroll
        for (int s = 0; s < S; s++) out.v[s] = NEG ? out.v[s] - wts[t] * R.v[t][s] : out.v[s] + wts[t] * R.v[t][s];
}

template <typename T, int S, bool IMPLICIT, bool GRAMX = false, int NE = 4>
__global__ void __launch_bounds__(256, (tiny_waves_per_simd<T, IMPLICIT || GRAMX>()))
cg_rows_tiny_kernel(const CgParams<T> P)
{
    // NE: entries per lane group -- 4 (32-slot tile), 2 (16-slot tile: rows of <= 16 entries only) or 0: by row, the 16-slot tile
    // for the rows of <= 16 entries and the 32-slot tile for the others in ONE launch (a second launch for the short rows costs
    // more in launch tails beside the other bins than the short tile saves)
    static_assert(NE == 4 || NE == 2 || NE == 0, "entries per lane group");
    constexpr bool MIX = NE == 0;
    static_assert(!MIX || CMF_TINY_WAVES_PER_SIMD >= 3, "the mixed launch lives in the dynamically scheduled loop");
    using Tile = typename TinyTile<T, S, (NE == 2) ? 2 : 4>::type;
    constexpr bool GRAM = IMPLICIT || GRAMX;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *G = reinterpret_cast<T *>(smem_raw);
    constexpr bool PV = pass_vector_in_lds<T>();
    __shared__ __attribute__((aligned(16))) T s_pv[PV ? 4 : 1][PV ? 64 : 1];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    auto short_row = [&](int nnz_) -> bool { return MIX && nnz_ <= 16; };
    auto entry_of = [&](int nnz_) -> int { return (NE == 2 || short_row(nnz_)) ? (lane >> 2) : (lane >> 1); };
    const int k = P.k;
    constexpr bool GREG = tiny_greg<T, GRAM>();
    constexpr int GR = GREG ? gram_rows<T>(S) : 8;          // (the LDS form of the product below reads eight rows per group)
    if (GRAM) {
        stage_gramian<T, S, GR>(G, P.BtB, k, tid, blockDim.x);
        __syncthreads();
    }
    GramRegs<T, GREG ? S : 1, GREG ? GR : 8> greg;
    if constexpr (GREG) greg.load(G, lane);
    const int nwaves = gridDim.x * 4;
    struct Pre { int idx; T x; T a; T g; };
    auto load_desc = [&](int rix_) -> RowDesc {
        RowDesc d; d.row = 0; d.nnz = 0; d.st = 0;
        if (rix_ < P.nrows) d = P.desc[rix_];
        d.row = __builtin_amdgcn_readfirstlane(d.row);
        d.nnz = __builtin_amdgcn_readfirstlane(d.nnz);
        d.st = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(d.st >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)(d.st & 0xffffffffu));
        return d;
    };
    auto load_pre = [&](const RowDesc &d) -> Pre {
        Pre q; q.idx = 0; q.x = T(0); q.a = T(0); q.g = T(1);
        const int ent = entry_of(d.nnz);
        if (ent < d.nnz) {
            const size_t pos = d.st + (size_t)ent;
            q.idx = P.indices[pos];
            q.x = P.values[pos];
            q.g = entry_weight<T, IMPLICIT>(P, pos);
            if (!IMPLICIT && P.bias_sub != nullptr) q.x -= P.bias_sub[q.idx];
        }
        if (d.nnz > 0 && lane < k) q.a = P.A[(size_t)d.row * P.lda + lane];
        return q;
    };
    auto load_tile_ne = [&](Tile &t_, const T *Bm_, size_t ldb_, int k_, int idx_, int cnt_, int lane_) {
        if constexpr (NE != 2) load_tile4<T, S>(t_, Bm_, ldb_, k_, idx_, cnt_, lane_);
        else load_tile2<T, S>(t_, Bm_, ldb_, k_, idx_, cnt_, lane_);
    };
    // CG on one register-resident 32-nnz tile (same arithmetic as cg_rows_kernel)
    auto solve = [&](const RowDesc &d, const Pre &pr, const Tile &tile, auto short_tag) {
        constexpr bool T2 = NE == 2 || decltype(short_tag)::value;      // 16-slot tile (the first two entries of a RegTile4 in the mixed launch)
        const int nnz = d.nnz;
        T lam = P.lam, lam_last = P.lam_last;
        if (GRAMX && P.kc > 0) {
            if (P.scale_lam || P.scale_lam_sideinfo) {
                T mult = row_lam_mult(P, d.row, nnz);
                if (P.scale_lam_sideinfo) mult += (T)P.p_side;
                lam *= mult; lam_last *= mult;
            }
        } else if (!IMPLICIT && P.scale_lam) {
            const T mult = row_lam_mult(P, d.row, nnz);
            lam *= mult;
            if (!P.scale_bias_const) lam_last *= mult;
        }
        const bool valid = (T2 ? (lane >> 2) : (lane >> 1)) < nnz;
        T a_d = pr.a;
        auto run_pass = [&](T vdist, auto mode_tag) -> T {
            constexpr int MODE = decltype(mode_tag)::value;
            asm volatile("" ::: "memory");
            T vrep[S];
            T gw[8];
            if constexpr (PV) pass_vector_lds<T, S, GR>(vdist, vrep, gw, lane, &s_pv[tid >> 6][0]);
            else replicate<T, S>(vdist, vrep, lane);
            PassAcc<T> acc;
            if (!CMF_DBG(P, 4)) {          // (the accumulators are defined by the tile pass itself: no clearing moves)
                if constexpr (!T2) tile_pass4<T, S, IMPLICIT, MODE, decltype(tile), true>(tile, vrep, pr.x, valid, acc, lane, pr.g);
                else tile_pass2<T, S, IMPLICIT, MODE, decltype(tile), true>(tile, vrep, pr.x, valid, acc, lane, pr.g);
            } else acc.zero();
            if constexpr (GREG) {
                if (!CMF_DBG(P, 2)) {
                    if constexpr (PV) gram_pass_regs_w<MODE == 0>(greg, gw, acc);
                    else gram_pass_regs(greg, (MODE == 0) ? -vdist : vdist, acc);
                }
            }
            else if (GRAM && !CMF_DBG(P, 2)) gram_pass<T, S, 1>(G, (MODE == 0) ? -vdist : vdist, acc, lane, 0);
            T out[8];
            acc.close(out);
            return treduce8_high<T>(out, lane);
        };
        T r_d = run_pass(a_d, std::integral_constant<int, 0>{});
        r_d -= lam * a_d;
        if (!IMPLICIT && lam != lam_last && lane == k - 1) r_d -= (lam_last - lam) * a_d;
        if (GRAMX && P.rconst != nullptr && lane < k) r_d += P.rconst[(size_t)d.row * P.ldr + lane];
        if (lane >= k) r_d = T(0);
        T p_d = r_d;
        T r_old = wave_sum(r_d * r_d);
        bool done = (r_old <= (T)1e-12);
        for (int step = 0; step < P.max_cg_steps && !done; step++) {
            T Ap_d = run_pass(p_d, std::integral_constant<int, 1>{});
            Ap_d += lam * p_d;
            if (!IMPLICIT && lam != lam_last && lane == k - 1) Ap_d += (lam_last - lam) * p_d;
            if (lane >= k) Ap_d = T(0);
            T alpha = cg_div(r_old, wave_sum(Ap_d * p_d));
            a_d += alpha * p_d;
            r_d -= alpha * Ap_d;
            T r_new = wave_sum(r_d * r_d);
            if (r_new <= (T)1e-8) done = true;
            else {
                p_d = p_d * cg_div(r_new, r_old) + r_d;
                r_old = r_new;
            }
        }
        if (lane < k) P.A[(size_t)d.row * P.lda + lane] = a_d;
    };

    int rix = blockIdx.x * 4 + (tid >> 6);
#if CMF_TINY_WAVES_PER_SIMD >= 3
    {   // single tile buffer, latency hidden by the other wavefronts of the SIMD; rows are claimed dynamically, two
        // ahead (see cg_rows_kernel)
        const int cslot = rix % CG_NCOUNTERS;
        int *const my_counter = P.counter + cslot * CG_COUNTER_STRIDE;
        const int cbase = nwaves + cslot;
        auto issue_claim = [&]() -> int {
            int v = 0;
            if (lane == 0) v = atomicAdd(my_counter, 1);
            return v;
        };
        const int c1 = issue_claim(), c2 = issue_claim();
        int rnxt = cbase + CG_NCOUNTERS * __builtin_amdgcn_readfirstlane(c1);
        int rnn = cbase + CG_NCOUNTERS * __builtin_amdgcn_readfirstlane(c2);
        RowDesc d0 = load_desc(rix), d1 = load_desc(rnxt);
        Pre p0 = load_pre(d0);
        Tile tA;
        int pend = issue_claim();
        while (rix < P.nrows) {
            const bool shrt = short_row(d0.nnz);
            if (CMF_DBG(P, 1)) {
                dbg_fill_tile<(NE == 2) ? 2 : 4, S>(tA, (T)(p0.idx & 3) * (T)0.001);
            } else if constexpr (MIX) {
                if (shrt) load_tile2<T, S>(tA, P.B, P.ldb, k, p0.idx, d0.nnz, lane);
                else load_tile_ne(tA, P.B, P.ldb, k, p0.idx, d0.nnz, lane);
            } else load_tile_ne(tA, P.B, P.ldb, k, p0.idx, d0.nnz, lane);
            RowDesc d2 = load_desc(rnn);
            Pre p1 = load_pre(d1);
            if constexpr (MIX) {
                if (shrt) solve(d0, p0, tA, std::true_type{});
                else solve(d0, p0, tA, std::false_type{});
            } else solve(d0, p0, tA, std::false_type{});
            const int r3 = cbase + CG_NCOUNTERS * __builtin_amdgcn_readfirstlane(pend);
            d0 = d1; p0 = p1; d1 = d2;
            rix = rnxt; rnxt = rnn; rnn = r3;
            pend = issue_claim();
        }
        return;
    }
#endif
    RowDesc d0 = load_desc(rix), d1 = load_desc(rix + nwaves), d2 = load_desc(rix + 2 * nwaves);
    Pre p0 = load_pre(d0);
    Tile tA, tB;
    if (d0.nnz > 0) load_tile_ne(tA, P.B, P.ldb, k, p0.idx, d0.nnz, lane);
    Pre p1 = load_pre(d1);
    while (rix < P.nrows) {
        // row i   : buffer A (gather in flight), row i+1: start its gather into buffer B now
        if (d1.nnz > 0) load_tile_ne(tB, P.B, P.ldb, k, p1.idx, d1.nnz, lane);
        RowDesc d3 = load_desc(rix + 3 * nwaves);
        Pre p2 = load_pre(d2);
        solve(d0, p0, tA, std::false_type{});
        rix += nwaves;
        if (rix >= P.nrows) break;
        // row i+1 : buffer B, row i+2: gather into buffer A
        if (d2.nnz > 0) load_tile_ne(tA, P.B, P.ldb, k, p2.idx, d2.nnz, lane);
        RowDesc d4 = load_desc(rix + 3 * nwaves);
        Pre p3 = load_pre(d3);
        solve(d1, p1, tB, std::false_type{});
        rix += nwaves;
        d0 = d2; p0 = p2; d1 = d3; p1 = p3; d2 = d4;
    }
}

// ------------------------------------------------------------------------------------------
// Very heavy rows (nnz > VH_MIN): one workgroup cannot own such a row without becoming the serial
// tail of the half-step (a single popular item can hold > 1e5 non-zeros), so every CG pass of these
// rows is split over many workgroups: vh_pass_kernel computes the partial  sum_j w_j B_j  of one
// 512-nnz chunk (8 waves x one 64-nnz tile, same register-tile arithmetic as cg_rows_kernel) and
// vh_update_kernel (one wavefront per row) adds the partials in chunk order (deterministic, no
// floating-point atomics), the Gramian term and does the CG vector update.  One launch pair per
// pass; kernel boundaries are the grid-wide synchronisation.
constexpr int VH_CHUNK_TILES = 4;

// One workgroup of the pass kernel, in launch order: everything it needs to find its non-zeros in ONE 32-byte (scalar)
// load -- the chain launch[] -> chunk_row[] -> order[] -> indptr[] -> indices[] -> gather was six dependent round trips
// per workgroup, and a pass is a few rounds of short-lived workgroups.
struct VhWork {
    int vi;                 // index of the very heavy row (position in `order`), -1: no chunk at this launch position
    int row;                // its row id
    int cnt;                // non-zeros of the chunk
    int chunk;              // chunk id (where its partial goes)
    unsigned long long st;  // CSR position of the chunk's first non-zero
    unsigned long long pad_;
};

template <typename T>
struct VhState {
    T *r, *p;               // [nvh][64] distributed CG vectors (a lives in the factor matrix itself)
    T *r_old;               // [nvh]
    int *done;              // [nvh]
    T *part;                // [nchunks][64]
    const int *chunk_row;   // [nchunks] index of the very-heavy row (position in `order`)
    const int *chunk_start; // [nchunks] first non-zero of the chunk inside its row
    const int *chunk_cnt;   // [nchunks] non-zeros of the chunk (<= 64 * VH_CHUNK_TILES)
    const int *chunk_off;   // [nvh+1] chunk range of every row
    const int *launch;      // [nlaunch] workgroup -> chunk (-1: none): chunks of one range of gathered rows share an XCD
    const VhWork *work;     // [nlaunch] the same map with the chunk's row, length and CSR position resolved
    int nvh, nchunks, nlaunch;
};

template <typename T, int S, bool IMPLICIT, int MODE>
__global__ void __launch_bounds__(64 * VH_CHUNK_TILES, 3)
vh_pass_kernel(const CgParams<T> P, const VhState<T> V)
{
    __shared__ T red[VH_CHUNK_TILES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const VhWork wk = V.work[blockIdx.x];
    const int vi = wk.vi;
    if (vi < 0) return;
    if (MODE == 1 && V.done[vi]) return;
    const int c = wk.chunk;
    const int row = wk.row;
    const size_t st = (size_t)wk.st;
    const int nnz = wk.cnt;                  // of this chunk
    const int tl = wave;
    const int k = P.k;
    T vdist;
    if (MODE == 0) vdist = (lane < k) ? P.A[(size_t)row * P.lda + lane] : T(0);
    else           vdist = V.p[(size_t)vi * 64 + lane];
    PassAcc<T> acc;
    acc.zero();
    if (tl * TILE < nnz) {
        const int cnt = min(TILE, nnz - tl * TILE);
        const bool valid = lane < cnt;
        const size_t pos = st + (size_t)tl * TILE + lane;
        int my_idx = valid ? P.indices[pos] : 0;
        T x = valid ? P.values[pos] : T(0);
        const T g = valid ? entry_weight<T, IMPLICIT>(P, pos) : T(1);
        if (!IMPLICIT && P.bias_sub != nullptr && valid) x -= P.bias_sub[my_idx];
        RegTile<T, S> tile;
        if (CMF_DBG(P, 1)) dbg_fill_tile<8, S>(tile, (T)(my_idx & 3) * (T)0.001);
        else load_tile<T, S>(tile, P.B, P.ldb, k, my_idx, cnt, lane);
        T vrep[S];
        replicate<T, S>(vdist, vrep, lane);
        if (!CMF_DBG(P, 4)) tile_pass<T, S, IMPLICIT, MODE>(tile, vrep, x, valid, acc, lane, g);
    }
    T out[8];
    acc.close(out);
    T tot = treduce8_high<T>(out, lane);
    red[wave][lane] = tot;
    __syncthreads();
    if (wave == 0) {
        T s = T(0);
#pragma unroll
        for (int w = 0; w < VH_CHUNK_TILES; w++) s += red[w][lane];
        V.part[(size_t)c * 64 + lane] = s;
    }
}

constexpr int VH_UPD_WAVES = 4;

template <typename T, bool IMPLICIT, int MODE, bool GRAMX = false>
__global__ void __launch_bounds__(64 * VH_UPD_WAVES)
vh_update_kernel(const CgParams<T> P, const VhState<T> V)
{
    __shared__ T psum[VH_UPD_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int vi = blockIdx.x;
    if (MODE == 1 && V.done[vi]) return;
    const int k = P.k;
    const int row = P.order[vi];
    const int nnz = (int)(P.indptr[row + 1] - P.indptr[row]);
    // chunk partials: wave w adds chunks c0+w, c0+w+4, ... (4 loads in flight), then the four wave
    // sums are added in wave order -- a fixed order for a given row length
    {
        const int c0 = V.chunk_off[vi], c1 = V.chunk_off[vi + 1];
        T acc = T(0);
        int c = c0 + wave;
        for (; c + 3 * VH_UPD_WAVES < c1; c += 4 * VH_UPD_WAVES) {
            T v0 = V.part[(size_t)c * 64 + lane];
            T v1 = V.part[(size_t)(c + VH_UPD_WAVES) * 64 + lane];
            T v2 = V.part[(size_t)(c + 2 * VH_UPD_WAVES) * 64 + lane];
            T v3 = V.part[(size_t)(c + 3 * VH_UPD_WAVES) * 64 + lane];
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; c < c1; c += VH_UPD_WAVES) acc += V.part[(size_t)c * 64 + lane];
        psum[wave][lane] = acc;
    }
    __syncthreads();
    if (wave != 0) return;
    T tot = T(0);
#pragma unroll
    for (int w = 0; w < VH_UPD_WAVES; w++) tot += psum[w][lane];
    T lam = P.lam, lam_last = P.lam_last;
    if (GRAMX && P.kc > 0) {
        if (P.scale_lam || P.scale_lam_sideinfo) {
            T mult = row_lam_mult(P, row, nnz);
            if (P.scale_lam_sideinfo) mult += (T)P.p_side;
            lam *= mult; lam_last *= mult;
        }
    } else if (!IMPLICIT && P.scale_lam) {
        const T mult = row_lam_mult(P, row, nnz);
        lam *= mult;
        if (!P.scale_bias_const) lam_last *= mult;
    }
    T *arow = P.A + (size_t)row * P.lda;
    T a_d = (lane < k) ? arow[lane] : T(0);
    T v = (MODE == 0) ? a_d : V.p[(size_t)vi * 64 + lane];
    if (IMPLICIT || GRAMX) {                           // + (+-) BtB v   (common.c:1932 / :1958; block systems: collective.c:2609-2643)
        T g = T(0);
        for (int j = 0; j < k; j++) {
            T vj = __shfl(v, j);
            if (lane < k) g += vj * P.BtB[(size_t)j * k + lane];
        }
        tot += (MODE == 0) ? -g : g;
    }
    if (MODE == 0) {
        T r_d = tot - lam * a_d;
        if (!IMPLICIT && lam != lam_last && lane == k - 1) r_d -= (lam_last - lam) * a_d;
        if (GRAMX && P.rconst != nullptr && lane < k) r_d += P.rconst[(size_t)row * P.ldr + lane];
        if (lane >= k) r_d = T(0);
        T r_old = wave_sum(r_d * r_d);
        V.r[(size_t)vi * 64 + lane] = r_d;
        V.p[(size_t)vi * 64 + lane] = r_d;
        if (lane == 0) { V.r_old[vi] = r_old; V.done[vi] = (r_old <= (T)1e-12) ? 1 : 0; }
    } else {
        T p_d = v;
        T r_d = V.r[(size_t)vi * 64 + lane];
        T r_old = V.r_old[vi];
        T Ap_d = tot + lam * p_d;
        if (!IMPLICIT && lam != lam_last && lane == k - 1) Ap_d += (lam_last - lam) * p_d;
        if (lane >= k) Ap_d = T(0);
        T alpha = cg_div(r_old, wave_sum(Ap_d * p_d));
        a_d += alpha * p_d;
        r_d -= alpha * Ap_d;
        T r_new = wave_sum(r_d * r_d);
        if (lane < k) arow[lane] = a_d;
        if (r_new <= (T)1e-8) {
            if (lane == 0) V.done[vi] = 1;
        } else {
            V.p[(size_t)vi * 64 + lane] = p_d * cg_div(r_new, r_old) + r_d;
            V.r[(size_t)vi * 64 + lane] = r_d;
            if (lane == 0) V.r_old[vi] = r_new;
        }
    }
}

// One wavefront per row, lane <-> unknown (any k_t <= 64 NF).  Serves k > 64, the Jacobi-preconditioned
// variants (factors_*_pcg) and the block systems with dense side information
// (collective_block_cg, src/collective.c:2134-2903; collective_block_cg_implicit, :2905-3303; dense full U
// without NaN, prefer_CtC branch): unknowns [koff, k_t) couple to X through the gathered rows (+ BtB in the
// implicit model), unknowns [0, kc) to U through w C^T C and the constant w (U C)_row.
// TEAM = 1: one wavefront per row (4 rows per workgroup).  TEAM = 4: the four wavefronts of a workgroup share a
// row -- its non-zeros are dealt round-robin, every product is summed across the waves through LDS in wave order, and
// all four carry identical copies of a, r, p (the scalars of the CG are then bit-identical, so the exits agree).
// P.row_first .. P.nrows of the processing order are handled.
template <typename T, int NF, bool IMPLICIT, int TEAM>
__global__ void __launch_bounds__(TEAM == 16 ? 1024 : 256)
cg_rows_generic_kernel(const CgParams<T> P)
{
    // TEAM 1: four independent wavefronts per workgroup, a row each; TEAM 4 / 16: the whole workgroup on one row
    static_assert(TEAM == 1 || TEAM == 4 || TEAM == 16, "team");
    __shared__ T red[TEAM][64 * NF];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int unit_global = (TEAM == 1) ? ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) : (int)blockIdx.x;
    const int nunits = (TEAM == 1) ? ((gridDim.x * blockDim.x) >> 6) : (int)gridDim.x;
    const int kx = P.k, koff = P.koff, kt = koff + kx, kc = P.kc;
    const bool coll = kc > 0;
    for (int rix = P.row_first + unit_global; rix < P.nrows; rix += nunits) {
        const int row = P.order[rix];
        const size_t st = P.indptr[row];
        const int nnz = (int)(P.indptr[row + 1] - st);
        const bool has_u = coll && row < P.rows_with_u;
        if (!IMPLICIT && nnz == 0 && !has_u && P.gx == 1 && P.Bi != nullptr) {
            // shared matrix without the bias / mean constant, implicit features, no side information on this side: the reference's
            // collective route zeroes a row without entries (collective.c:1258-1268)
            if (TEAM == 1 || wv == 0)
                for (int f = lane; f < kt; f += 64) P.A[(size_t)row * P.lda + f] = T(0);
            continue;
        }
        if (nnz == 0 && !has_u && !(P.gx != 0 && !IMPLICIT)) {   // plain rows without entries stay untouched (common.c:3270,3354)
            if (P.Bi != nullptr && (TEAM == 1 || wv == 0))    // ... but with implicit features the row runs through
                for (int f = lane; f < kt; f += 64) P.A[(size_t)row * P.lda + f] = T(0);   // optimizeA_collective: zeros (collective.c:1258-1268)
            continue;
        }
        const bool sparse_u = P.indptr2 != nullptr;
        const size_t st2 = (sparse_u && has_u) ? P.indptr2[row] : 0;
        const int nnz2 = (sparse_u && has_u) ? (int)(P.indptr2[row + 1] - st2) : 0;
        // neither observations nor attributes: zeros (collective.c:1258-1268) -- unless the main matrix is missing-as-zero and its
        // constant exists (gx == 2; :1260-1261): then the row is solved like the others
        if (sparse_u && has_u && nnz == 0 && nnz2 == 0 && P.gx != 2) {
            if (TEAM == 1 || wv == 0)
                for (int f = lane; f < kt; f += 64) P.A[(size_t)row * P.lda + f] = T(0);
            continue;
        }
        // rows without side information: the X block only (collective.c:4832-4900) -- but with implicit features the reference
        // sends them through the block solver again with every unknown in the system (:4906-4960): the unknowns in front of the X
        // block then see lambda alone and share the step lengths
        const int lo = (has_u || (!IMPLICIT && P.Bi != nullptr)) ? 0 : koff;
        T lam = P.lam, lam_last = P.lam_last;
        if (!IMPLICIT) {
            if (has_u) {
                if (P.scale_lam || P.scale_lam_sideinfo) {    // collective.c:1285-1355
                    T mult = (P.wsum != nullptr) ? P.wsum[row] : ((nnz > 0) ? (T)nnz : T(1));
                    if (P.scale_lam_sideinfo) mult += sparse_u ? (T)nnz2 : (T)P.p_side;
                    lam *= mult; lam_last *= mult;
                }
            } else if (P.scale_lam) {                         // common.c:679-723
                const T mult = row_lam_mult(P, row, nnz);
                lam *= mult;
                if (!P.scale_bias_const) lam_last *= mult;
            }
        }
        T *arow = P.A + (size_t)row * P.lda;
        const T *ucrow = (has_u && !sparse_u) ? P.UC + (size_t)row * kc : nullptr;
        T a[NF], r[NF], p[NF], Ap[NF];
#pragma unroll
        for (int c = 0; c < NF; c++) { int f = lane + 64 * c; a[c] = (f >= lo && f < kt) ? arow[f] : T(0); }

        auto bcast = [&](const T (&v)[NF], int j) {           // v[j] for every lane
            T vj = T(0);
#pragma unroll
            for (int c = 0; c < NF; c++) if (c == (j >> 6)) vj = __shfl(v[c], j & 63);
            return vj;
        };
        auto matvec = [&](const T (&v)[NF], T (&out)[NF], int mode) {
#pragma unroll
            for (int c = 0; c < NF; c++) out[c] = T(0);
            if (IMPLICIT || P.gx != 0) {
                for (int j = 0; j < kx; j++) {                 // out[koff:] = +-BtB v[koff:]
                    T vj = bcast(v, koff + j);
                    if (mode == 0) vj = -vj;
#pragma unroll
                    for (int c = 0; c < NF; c++) { int f = lane + 64 * c; if (f >= koff && f < kt) out[c] += vj * P.BtB[(size_t)j * kx + (f - koff)]; }
                }
            }
            if (has_u && !sparse_u) {                          // out[:kc] = w (UC_row - CtC v[:kc])  |  w CtC v[:kc]
                T acc[NF];
#pragma unroll
                for (int c = 0; c < NF; c++) acc[c] = T(0);
                for (int j = 0; j < kc; j++) {
                    const T vj = bcast(v, j);
#pragma unroll
                    for (int c = 0; c < NF; c++) { int f = lane + 64 * c; if (f < kc) acc[c] += vj * P.CtC[(size_t)j * kc + f]; }
                }
#pragma unroll
                for (int c = 0; c < NF; c++) {
                    int f = lane + 64 * c;
                    if (f < kc) out[c] += (mode == 0) ? P.w_side * (ucrow[f] - acc[c]) : P.w_side * acc[c];
                }
            }
            if (!IMPLICIT && P.Bi != nullptr) {                // out[koff : koff+ki] -+= w_i BiTBi v[koff:]  (collective.c:2626-2629, :2864-2867)
                T acc[NF];
#pragma unroll
                for (int c = 0; c < NF; c++) acc[c] = T(0);
                // four rows of the matrix in flight: the loads do not depend on v
                for (int j0 = 0; j0 < P.ki; j0 += 4) {
                    T mrow[4][NF];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int j = min(j0 + u, P.ki - 1);
#pragma unroll
                        for (int c = 0; c < NF; c++) {
                            int f = lane + 64 * c;
                            mrow[u][c] = (f >= koff && f < koff + P.ki) ? P.BiTBi[(size_t)j * P.ki + (f - koff)] : T(0);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (j0 + u >= P.ki) break;
                        const T vj = bcast(v, koff + j0 + u);
#pragma unroll
                        for (int c = 0; c < NF; c++) acc[c] += vj * mrow[u][c];
                    }
                }
#pragma unroll
                for (int c = 0; c < NF; c++) out[c] += (mode == 0) ? -P.w_imp * acc[c] : P.w_imp * acc[c];
            }
            T gat[NF];                                         // the gathered part, this wave's share of the non-zeros
#pragma unroll
            for (int c = 0; c < NF; c++) gat[c] = T(0);
            // four entries in flight per wave: their loads and dot products are independent, only the accumulation into
            // gat keeps the entry order (so the sums are the ones of the one-at-a-time loop)
            constexpr int UNR = 4;
            for (int j0 = (TEAM == 1 ? 0 : wv); j0 < nnz; j0 += TEAM * UNR) {
                int idx[UNR]; T x[UNR]; T gw[UNR]; T bv[UNR][NF]; T part[UNR];
#pragma unroll
                for (int u = 0; u < UNR; u++) {
                    const int j = j0 + u * TEAM;
                    const bool ok = j < nnz;
                    idx[u] = ok ? P.indices[st + j] : 0;
                    x[u] = ok ? P.values[st + j] : T(0);
                    gw[u] = ok ? entry_weight<T, IMPLICIT>(P, st + j) : T(1);
                }
#pragma unroll
                for (int u = 0; u < UNR; u++) {
                    const bool ok = (j0 + u * TEAM) < nnz;
                    if (!IMPLICIT && P.bias_sub != nullptr && ok) x[u] -= P.bias_sub[idx[u]];
                    const T *b = P.B + (size_t)idx[u] * P.ldb;
                    part[u] = T(0);
#pragma unroll
                    for (int c = 0; c < NF; c++) { int f = lane + 64 * c; bv[u][c] = (ok && f >= koff && f < kt) ? b[f - koff] : T(0); part[u] += bv[u][c] * v[c]; }
                }
#pragma unroll
                for (int u = 0; u < UNR; u++) part[u] = wave_sum(part[u]);
#pragma unroll
                for (int u = 0; u < UNR; u++) {
                    if ((j0 + u * TEAM) >= nnz) break;
                    const T coef = part[u];
                    T w;
                    if (IMPLICIT) w = (mode == 0) ? (-(coef - T(1)) * x[u] - coef) : (coef * (x[u] - T(1)) + coef);
                    else          w = (mode == 0) ? -((coef - x[u]) * gw[u]) : coef * gw[u];   // common.c:1121-1133, :1158-1169
#pragma unroll
                    for (int c = 0; c < NF; c++) gat[c] += w * bv[u][c];
                    if (!IMPLICIT && mode == 0 && P.Bi != nullptr) {              // + w_i Bi_j (tgemv_dense_sp on ones, collective.c:2638-2643)
                        const T *bi = P.Bi + (size_t)idx[u] * P.ki;
#pragma unroll
                        for (int c = 0; c < NF; c++) { int f = lane + 64 * c; if (f >= koff && f < koff + P.ki) gat[c] += P.w_imp * bi[f - koff]; }
                    }
                }
            }
            // sparse side information: every present attribute j adds  w (u_j - C_j.v) C_j  /  w (C_j.v) C_j  to [0, kc)
            for (int j = (TEAM == 1 ? 0 : wv); j < nnz2; j += TEAM) {
                const T *cj = P.C2 + (size_t)P.indices2[st2 + j] * kc;
                const T uj = P.values2[st2 + j];
                T cv[NF]; T part = T(0);
#pragma unroll
                for (int c = 0; c < NF; c++) { int f = lane + 64 * c; cv[c] = (f < kc) ? cj[f] : T(0); part += cv[c] * v[c]; }
                const T coef = wave_sum(part);
                const T w = (mode == 0) ? P.w_side * (-coef + uj) : P.w_side * coef;
#pragma unroll
                for (int c = 0; c < NF; c++) gat[c] += w * cv[c];
            }
            if (TEAM > 1) {                                    // sum over the waves, in wave order, identical everywhere
                __syncthreads();
#pragma unroll
                for (int c = 0; c < NF; c++) red[wv][lane + 64 * c] = gat[c];
                __syncthreads();
#pragma unroll
                for (int c = 0; c < NF; c++) {
                    T sacc = T(0);
#pragma unroll
                    for (int w2 = 0; w2 < TEAM; w2++) sacc += red[w2][lane + 64 * c];
                    gat[c] = sacc;
                }
            }
#pragma unroll
            for (int c = 0; c < NF; c++) out[c] += gat[c];
        };
        auto vdot = [&](const T (&u)[NF], const T (&v)[NF]) {
            T s = T(0);
#pragma unroll
            for (int c = 0; c < NF; c++) s += u[c] * v[c];
            return wave_sum(s);
        };
        auto live = [&](int f) { return f >= lo && f < kt; };
        matvec(a, r, 0);
#pragma unroll
        for (int c = 0; c < NF; c++) {
            int f = lane + 64 * c;
            r[c] -= lam * a[c];
            if (!IMPLICIT && lam != lam_last && f == kt - 1) r[c] -= (lam_last - lam) * a[c];
            if (!IMPLICIT && P.gx != 0 && P.rconst != nullptr && f >= koff && f < kt) r[c] += P.rconst[(size_t)row * P.ldr + (f - koff)];   // common.c:1525-1546
            if (!live(f)) r[c] = T(0);
            p[c] = r[c];
        }
        if (P.precond) {
            // factors_implicit_pcg common.c:1988-2061 / factors_explicit_pcg :1190-1291 / the block versions
            // collective.c:2180-2316, :2960-3010: Jacobi preconditioner, fixed number of steps, no early exits
            T PC[NF], z[NF];
#pragma unroll
            for (int c = 0; c < NF; c++) PC[c] = T(0);
            for (int j = (TEAM == 1 ? 0 : wv); j < nnz; j += TEAM) {
                const int idx = P.indices[st + j];
                T x = P.values[st + j];
                const T gwt = entry_weight<T, IMPLICIT>(P, st + j);
                const T *b = P.B + (size_t)idx * P.ldb;
#pragma unroll
                for (int c = 0; c < NF; c++) {
                    int f = lane + 64 * c;
                    T bv = (f >= koff && f < kt) ? b[f - koff] : T(0);
                    PC[c] += IMPLICIT ? x * (bv * bv) : gwt * (bv * bv);         // :2009-2014 / :1238-1254
                }
            }
            for (int j = (TEAM == 1 ? 0 : wv); j < nnz2; j += TEAM) {            // C_j^2, unweighted (collective.c:2292-2298, :2993-2999)
                const T *cj = P.C2 + (size_t)P.indices2[st2 + j] * kc;
#pragma unroll
                for (int c = 0; c < NF; c++) {
                    int f = lane + 64 * c;
                    T cvv = (f < kc) ? cj[f] : T(0);
                    PC[c] += cvv * cvv;
                }
            }
            if (TEAM > 1) {
                __syncthreads();
#pragma unroll
                for (int c = 0; c < NF; c++) red[wv][lane + 64 * c] = PC[c];
                __syncthreads();
#pragma unroll
                for (int c = 0; c < NF; c++) {
                    T sacc = T(0);
#pragma unroll
                    for (int w2 = 0; w2 < TEAM; w2++) sacc += red[w2][lane + 64 * c];
                    PC[c] = sacc;
                }
            }
#pragma unroll
            for (int c = 0; c < NF; c++) {
                int f = lane + 64 * c;
                if (has_u && !sparse_u && f < kc) PC[c] += P.CtC[(size_t)f * kc + f];   // sum_l C_l^2, unweighted (collective.c:2281-2286)
                if (!IMPLICIT && P.Bi != nullptr && f >= koff && f < koff + P.ki)
                    PC[c] += P.BiTBi[(size_t)(f - koff) * P.ki + (f - koff)];           // unweighted too (collective.c:2301-2304)
                if (IMPLICIT) PC[c] += (f >= koff && f < kt) ? P.BtB[(size_t)(f - koff) * kx + (f - koff)] : T(0);
                else {
                    if (P.gx != 0) PC[c] += (f >= koff && f < kt) ? P.BtB[(size_t)(f - koff) * kx + (f - koff)] : T(0);   // common.c:1486-1497
                    PC[c] += lam;
                    if (lam != lam_last && f == kt - 1) PC[c] += (lam_last - lam);
                }
                PC[c] = live(f) ? T(1) / PC[c] : T(0);
                z[c] = r[c] * PC[c];
                p[c] = z[c];
            }
            T r_old = vdot(z, r);
            for (int step = 0; step < P.max_cg_steps; step++) {
                matvec(p, Ap, 1);
#pragma unroll
                for (int c = 0; c < NF; c++) {
                    int f = lane + 64 * c;
                    Ap[c] += lam * p[c];
                    if (!IMPLICIT && lam != lam_last && f == kt - 1) Ap[c] += (lam_last - lam) * p[c];
                    if (!live(f)) Ap[c] = T(0);
                }
                T alpha = cg_div(r_old, vdot(Ap, p));
#pragma unroll
                for (int c = 0; c < NF; c++) {
                    a[c] += alpha * p[c]; r[c] -= alpha * Ap[c];
                    z[c] = r[c] * PC[c];
                }
                T r_new = vdot(z, r);
                T ratio = cg_div(r_new, r_old);
#pragma unroll
                for (int c = 0; c < NF; c++) p[c] = p[c] * ratio + z[c];
                r_old = r_new;
            }
        } else {
        T r_old = vdot(r, r);
        if (r_old > (T)1e-12) {
            for (int step = 0; step < P.max_cg_steps; step++) {
                matvec(p, Ap, 1);
#pragma unroll
                for (int c = 0; c < NF; c++) {
                    int f = lane + 64 * c;
                    Ap[c] += lam * p[c];
                    if (!IMPLICIT && lam != lam_last && f == kt - 1) Ap[c] += (lam_last - lam) * p[c];
                    if (!live(f)) Ap[c] = T(0);
                }
                T alpha = cg_div(r_old, vdot(Ap, p));
#pragma unroll
                for (int c = 0; c < NF; c++) { a[c] += alpha * p[c]; r[c] -= alpha * Ap[c]; }
                T r_new = vdot(r, r);
                if (r_new <= (T)1e-8) break;
                T ratio = cg_div(r_new, r_old);
#pragma unroll
                for (int c = 0; c < NF; c++) p[c] = p[c] * ratio + r[c];
                r_old = r_new;
            }
        }
        }
        if (TEAM == 1 || wv == 0) {
#pragma unroll
            for (int c = 0; c < NF; c++) { int f = lane + 64 * c; if (live(f)) arow[f] = a[c]; }
        }
    }
}

}  // namespace cmfhip
