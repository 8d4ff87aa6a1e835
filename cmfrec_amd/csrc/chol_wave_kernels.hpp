// chol_wave_kernels.hpp -- closed-form (Cholesky) ALS row updates, one WAVEFRONT per row (gfx950).
//
// Same arithmetic as chol_rows_kernel (chol_kernels.hpp) -- factors_closed_form /root/reference/src/common.c:978-1070,
// factors_implicit_chol :2063-2126, collective_closed_form_block /root/reference/src/collective.c:1534-1846,
// collective_closed_form_block_implicit :1849-2131 -- but a different mapping for rows of up to ~1k entries, where the
// workgroup-per-row kernel is bound by its per-row chain of barriers and LDS exchanges (one row in flight per CU):
//   * one wavefront owns a row from the gather to the solution; the four wavefronts of a workgroup sit on the four
//     SIMDs of a CU and never synchronise with each other (no s_barrier in the row loop), so 4 (k_t <= 129) to 8
//     (k_t <= 65) rows are in flight per CU;
//   * the whole upper triangle of the k_t x k_t normal matrix lives in this wave's registers as 16x16 tiles in the C/D
//     layout of v_mfma_{f64,f32}_16x16x4 (NB (NB+1)/2 tiles: 36 tiles = 288 registers in double for k_t = 128);
//   * gathered rows go straight from HBM / L2 into MFMA operand registers: lane (g, c) = (lane >> 4, lane & 15) reads
//     element 16 b + c of gathered row g of a 4-row step, which is exactly the A / B operand of the 16x16x4 step for
//     block b -- NB loads (+ one for a border column) feed NB (NB+1)/2 MFMAs; PD steps are in flight;
//   * the blocked factorisation needs no exchange at all: panel and trailing updates are MFMAs whose operands are the
//     tiles' own C/D registers (chol_kernels.hpp, step 3); right-hand side and border column ride along as one more
//     tile column E (two live columns), so the forward substitution is MFMA work too;
//   * k_t = 16 n + 1 (k + a bias column: 129, 257, 65) does not pay for a whole extra block row: the last unknown is a
//     border  [G g; g^T gamma]:  r = R^-T g comes out of the factorisation as the second column of E,
//     rho^2 = gamma - r.r, and the last unknown is solved first in the backward substitution.
// Rows beyond 1024 entries (a popular item holds tens of thousands) are cut into slices of <= 2048 entries
// (SparseShard::sl_*): the same kernel, PRODUCER build, runs the rank-k update of one slice per wavefront and leaves
// the raw tiles + right-hand-side partials in HBM (NT x 2 KB per slice in double); the row's owner then adds its
// slices in slice order (fixed summation order, no atomics) instead of gathering, and factorises as usual.
#pragma once
#include "chol_kernels.hpp"
#include "cg_kernels.hpp"

namespace cmfhip {

#define CMF_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// Sixteen values of a lane-linear tile array (4 tiles: element [t][r][lane], t = -2 .. 1 around `mid`, r = 0 .. 3) with all
// sixteen loads in flight and one wait (loads and their wait in one statement, early-clobber outputs: the compiler neither
// counts nor reorders what is inside).
__device__ __forceinline__ void wave_load16(const double *mid, double (&v)[16])
{
    asm volatile(
        "global_load_dwordx2 %0, %16, off offset:-4096\n\t"
        "global_load_dwordx2 %1, %16, off offset:-3584\n\t"
        "global_load_dwordx2 %2, %16, off offset:-3072\n\t"
        "global_load_dwordx2 %3, %16, off offset:-2560\n\t"
        "global_load_dwordx2 %4, %16, off offset:-2048\n\t"
        "global_load_dwordx2 %5, %16, off offset:-1536\n\t"
        "global_load_dwordx2 %6, %16, off offset:-1024\n\t"
        "global_load_dwordx2 %7, %16, off offset:-512\n\t"
        "global_load_dwordx2 %8, %16, off\n\t"
        "global_load_dwordx2 %9, %16, off offset:512\n\t"
        "global_load_dwordx2 %10, %16, off offset:1024\n\t"
        "global_load_dwordx2 %11, %16, off offset:1536\n\t"
        "global_load_dwordx2 %12, %16, off offset:2048\n\t"
        "global_load_dwordx2 %13, %16, off offset:2560\n\t"
        "global_load_dwordx2 %14, %16, off offset:3072\n\t"
        "global_load_dwordx2 %15, %16, off offset:3584\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]),
          "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
        : "v"(mid)
        : "memory");
}
__device__ __forceinline__ void wave_load16(const float *mid, float (&v)[16])
{
    asm volatile(
        "global_load_dword %0, %16, off offset:-2048\n\t"
        "global_load_dword %1, %16, off offset:-1792\n\t"
        "global_load_dword %2, %16, off offset:-1536\n\t"
        "global_load_dword %3, %16, off offset:-1280\n\t"
        "global_load_dword %4, %16, off offset:-1024\n\t"
        "global_load_dword %5, %16, off offset:-768\n\t"
        "global_load_dword %6, %16, off offset:-512\n\t"
        "global_load_dword %7, %16, off offset:-256\n\t"
        "global_load_dword %8, %16, off\n\t"
        "global_load_dword %9, %16, off offset:256\n\t"
        "global_load_dword %10, %16, off offset:512\n\t"
        "global_load_dword %11, %16, off offset:768\n\t"
        "global_load_dword %12, %16, off offset:1024\n\t"
        "global_load_dword %13, %16, off offset:1280\n\t"
        "global_load_dword %14, %16, off offset:1536\n\t"
        "global_load_dword %15, %16, off offset:1792\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]),
          "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
        : "v"(mid)
        : "memory");
}

// does any of the tiles [T0, T1) of the packed order touch block b?
__host__ __device__ constexpr bool wave_block_needed(int b, int T0, int T1, int NB)
{
    for (int t = T0; t < T1; t++)
        if (tile_bi(t, NB) == b || tile_bj(t, NB) == b) return true;
    return false;
}
// tile (bi, bj), bj >= bi, of the packed upper triangle of an NB x NB grid
__host__ __device__ constexpr int wtix(int bi, int bj, int NB) { return bi * NB - (bi * (bi - 1)) / 2 + (bj - bi); }

// LDS elements per wavefront: inv(R_kk) of every block, two vectors of 16 NB (+16), the solution
template <typename T>
__host__ __device__ constexpr size_t chol_wave_lds_elems(int NB)
{
    return (size_t)NB * 16 * CholMfma<T>::LDR + 2 * (16 * (size_t)NB + 16) + 16 * (size_t)NB + 16 + 2 * 64 * (size_t)NB;
}

__host__ __device__ constexpr size_t chol_wave_lds_elems_producer(int NB) { return 2 * (16 * (size_t)NB + 16); }

// slices of the rows that lead the processing order (positions < n_heavy): slice s covers entries
// [first[s], first[s] + count[s]) of the row at position vrow[s]; row_off[v] .. row_off[v + 1] are the slices of position v
template <typename T>
struct CholSlices {
    const int *vrow = nullptr, *first = nullptr, *count = nullptr, *row_off = nullptr;
    T *part = nullptr;               // partials of the work items [part_base, ..): item - part_base selects the slot
    int n_slices = 0, n_heavy = 0;
    // Work items of a producer launch: items [0, n_slices) are the slices above; item n_slices + i is the WHOLE row at
    // position n_heavy + i (two-kernel mode: every row's rank-k update is done by the producer build).
    int part_base = 0;
    // initial matrices of the launch in the tile-linear layout of the partials (tile_pack_kernel), added like one more
    // partial; null: read from the row-major matrices element by element
    const T *init1 = nullptr, *init2 = nullptr;
    int dbg_skip = 0;                // timing experiments only (CMFREC_HIP_WAVE_SKIP): 1 partial loads, 2 factorisation, 4 forward pass, 8 backward pass
};
// elements of one slice's partial: NT tiles in lane-linear order, right-hand side, border column, two scalars
__host__ __device__ constexpr size_t chol_wave_part_elems(int NB) { return (size_t)(NB * (NB + 1) / 2) * 256 + 2 * 16 * (size_t)NB + 64; }

// out[t][r][lane] = M[gi][gj] of the symmetric lim x lim matrix M (upper triangle referenced, zero beyond lim) for the tile
// grid of an NB-block wave kernel: the initial matrix of a launch, once, in the layout the row kernels add partials in
template <typename T>
__global__ void tile_pack_kernel(const T *__restrict__ M, int lim, int NB, T *__restrict__ out)
{
    const int NT = NB * (NB + 1) / 2;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NT * 256) return;
    const int t = e >> 8, r = (e >> 6) & 3, lane = e & 63;
    int bi = 0, rem = t;
    while (rem >= NB - bi) { rem -= NB - bi; bi++; }
    const int bj = bi + rem;
    const int gi = 16 * bi + CholMfma<T>::row_of(lane, r), gj = 16 * bj + (lane & 15);
    const int lo = min(gi, gj), hi = max(gi, gj);
    out[e] = (hi < lim) ? M[(size_t)lo * lim + hi] : T(0);
}

// NB: 16-blocks of the compiled tile grid; BORDER: the last unknown is kept outside the tiles; PD: 4-row gather steps
// in flight; WPS: wavefronts per SIMD the register budget is set for (workgroups per CU).
// NOV: at one wavefront per SIMD hipcc emits the accumulator-file form of every MFMA (destination and C operand in
// AGPRs), so at most 256 registers can be MFMA destinations -- 32 tiles in double, and k_t = 128 has 36 (+ 8 of E).
// The last NOV tiles of the packed order (and E, when NOV > 0) are therefore kept as ordinary values and updated as
// tile += mfma(a, b, 0): the product lands in a short-lived accumulator, the sum is a vector add (12 more vector
// instructions per update, hidden behind the 64-cycle MFMAs), and the register allocator is free to place them.
// WMODE 0: a row per work item.  WMODE 1 (producer): work items are slices, the kernel stops after the rank-k update and
// stores the partials.  WMODE 2: rows whose rank-k update is the sum of their slices' partials.  (Separate builds: the
// register allocation of the plain build must not pay for the other two.)
// EV: right-hand side and border column are forward-substituted on the vector ALU after the factorisation (their
// per-group partial sums stay in registers, 2 NB values) instead of riding as the tile column E (NB more tiles).
template <typename T, int NB, bool BORDER, int PD, int WPS, int NOV = 0, int WMODE = 0, bool EV = false>
__global__ void __launch_bounds__(256, WPS)
chol_wave_kernel(const CholParams<T> P, const RowDesc *__restrict__ desc, const CholSlices<T> SL)
{
    using Mf = CholMfma<T>;
    using vec = typename Mf::vec;
    constexpr int NT = NB * (NB + 1) / 2;
    constexpr int NRES = (WMODE == 1) ? NT : NT - NOV;   // tiles [0, NRES) are MFMA accumulators, [NRES, NT) overflow tiles
                                                         // (producer build: NOV = tiles per sweep instead)
    constexpr int LDR = Mf::LDR, RSZ = 16 * LDR;
    constexpr int NV = 16 * NB + 16;
    constexpr size_t PART = chol_wave_part_elems(NB);
    constexpr bool PRODUCER = (WMODE == 1);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lm = lane & 15, g = lane >> 4;
    // (the producer build only passes its right-hand side / border partials through LDS: 2 NV elements per wavefront, so that two
    //  workgroups of the two-wavefronts-per-SIMD build fit a CU)
    T *wbase = reinterpret_cast<T *>(smem_raw) + (size_t)wave * (PRODUCER ? chol_wave_lds_elems_producer(NB) : chol_wave_lds_elems<T>(NB));
    T *rinv = wbase;                         // [NB][16][LDR]
    T *yv0 = PRODUCER ? wbase : rinv + (size_t)NB * RSZ;        // [NV]  right-hand side -> y -> z
    T *yv1 = yv0 + NV;                       // [NV]  border column g -> R^-T g
    T *xall = yv1 + NV;                      // [NV]  solution
    T *stash = xall + NV;                    // [2][NB][64]  per-lane partials of right-hand side / border column (EV builds)

    const int kt = P.kt, koff = P.koff;
    const int kq = kt - (BORDER ? 1 : 0);    // unknowns inside the tiles
    const int nb = (kq + 15) >> 4;
    const int kbcols = kt - koff;            // columns of B that are used
    // this lane's column of B for block b (clamped) and whether the unknown 16 b + lm takes a gathered value
    int colb[NB];
    unsigned vmask = 0;
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int u = 16 * b + lm;
        const bool ok = (u >= koff) && (u < kq);
        colb[b] = min(max(u - koff, 0), kbcols - 1);
        vmask |= ok ? (1u << b) : 0u;
    }
    const int bcol = kbcols - 1;             // border column of B

    const bool coll = (P.mode == CHOL_COLLECTIVE || P.mode == CHOL_COLLECTIVE_IMPLICIT);
    const bool impl_w = (P.mode == CHOL_IMPLICIT || P.mode == CHOL_COLLECTIVE_IMPLICIT);
    const bool full = (P.mode == CHOL_IMPLICIT);
    const bool add_lam = (P.mode == CHOL_EXPLICIT || P.mode == CHOL_COLLECTIVE);

    const int nwaves = gridDim.x * 4;
    int rix = P.row_first + blockIdx.x * 4 + wave;
    for (;;) {
        if (rix >= P.nrows) break;
        // the next position is claimed now and read when this row is done
        int claim = 0;
        if (lane == 0) claim = atomicAdd(P.counter, 1);
        int ritem = rix;                      // position of the row in the processing order
        int sfirst = 0, scount = 0;
        if (PRODUCER) {
            if (rix < SL.n_slices) { ritem = SL.vrow[rix]; sfirst = SL.first[rix]; scount = SL.count[rix]; }
            else { ritem = SL.n_heavy + (rix - SL.n_slices); scount = -1; }
        }
        RowDesc d = desc[ritem];
        const int row = __builtin_amdgcn_readfirstlane(d.row);
        const int nnz_row = __builtin_amdgcn_readfirstlane(d.nnz);
        const size_t st_row = ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(d.st >> 32)) << 32) |
                              (unsigned)__builtin_amdgcn_readfirstlane((int)(d.st & 0xffffffffu));
        // what the rank-k loop walks over: the whole row, or this slice of it
        const int nnz = (PRODUCER && scount >= 0) ? __builtin_amdgcn_readfirstlane(scount) : nnz_row;
        const size_t st = PRODUCER ? st_row + (size_t)__builtin_amdgcn_readfirstlane(sfirst) : st_row;
        // rows whose rank-k update was done slice by slice: add the partials instead of gathering
        constexpr bool from_slices = (WMODE == 2);
        T *arow = P.A + (size_t)row * P.lda;
        const bool has_u = coll && row < P.rows_with_u;
        if (!PRODUCER && coll && nnz_row == 0 && !has_u) {              // collective.c:1258-1268, :1876-1885
            for (int e = lane; e < kt; e += 64) arow[e] = T(0);
            rix = P.row_first + nwaves + __builtin_amdgcn_readfirstlane(claim);
            continue;
        }
        T lam = P.lam, lam_last = P.lam_last;
        if (P.mode == CHOL_EXPLICIT) {
            if (P.scale_lam) {                                           // common.c:679-723
                lam *= (T)nnz_row;
                if (!P.scale_bias_const) lam_last *= (T)nnz_row;
            }
        } else if (P.mode == CHOL_COLLECTIVE) {
            if (P.scale_lam || P.scale_lam_sideinfo) {                   // collective.c:1285-1355
                T mult = (nnz_row > 0) ? (T)nnz_row : T(1);
                if (P.scale_lam_sideinfo && has_u) mult += (T)P.p_side;  // :1338-1346
                lam *= mult;
                if (has_u || !P.scale_bias_const) lam_last *= mult;
            }
        }
        // ---- the initial matrix in the accumulator layout (padding: identity) ----
        // (the lane coordinates are made opaque per row: otherwise every address of the initial matrices -- loop
        //  invariant over the rows -- is hoisted out of the row loop and pinned in registers / spilled)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int lm_o = lane_o & 15;
        const T *M1 = full ? P.Minit : P.Mfull;                    // [kt, kt], every row
        const T *M2 = (!full && has_u) ? P.Minit : nullptr;        // [kc, kc], rows with side information
        const int kc = P.kc;
        vec acc[NT];
        // (producer in sweeps: a sweep's tiles are cleared when the sweep begins -- cleared here, the tiles of the later sweeps would
        //  be live through the earlier ones and the two-wavefronts-per-SIMD build would spill them)
        if constexpr (!(PRODUCER && NOV > 0)) {
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = vec{0, 0, 0, 0};
        }
        // (four tiles of loads in flight at a time: left to itself the scheduler issues all NT x 4 loads first and the
        //  register allocation of the whole kernel pays for it)
        auto add_matrix = [&](const T *Mi, int lim) __attribute__((always_inline)) {              // collective.c:1566-1571
            static_for<0, (NT + 3) / 4>([&](auto qc) {
                constexpr int q4 = decltype(qc)::value;
                static_for<4 * q4, (4 * q4 + 4 < NT ? 4 * q4 + 4 : NT)>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    constexpr int bi = tile_bi(t, NB), bj = tile_bj(t, NB);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int gi = 16 * bi + Mf::row_of(lane_o, r), gj = 16 * bj + lm_o;
                        const int lo = min(gi, gj), hi = max(gi, gj);
                        const T q = Mi[(size_t)min(lo, lim - 1) * lim + min(hi, lim - 1)];
                        acc[t][r] += (hi < lim) ? q : T(0);
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        auto add_tiles = [&](const T *pp) __attribute__((always_inline)) {    // a tile-linear array, 16 values per round trip
            static_for<0, NT / 4>([&](auto qc) {
                constexpr int t0 = 4 * decltype(qc)::value;
                T v16[16];
                wave_load16(pp + (size_t)(t0 + 2) * 256 + lane, v16);
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[t0 + u][r] += v16[4 * u + r];
            });
            static_for<(NT / 4) * 4, NT>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
#pragma unroll
                for (int r = 0; r < 4; r++) acc[t][r] += pp[t * 256 + r * 64 + lane];
            });
        };
        auto apply_init = [&]() __attribute__((always_inline)) {
            if (M1 != nullptr) { if (WMODE == 2 && SL.init1 != nullptr) add_tiles(SL.init1); else add_matrix(M1, kt); }
            if (M2 != nullptr && kc > 0) { if (WMODE == 2 && SL.init2 != nullptr) add_tiles(SL.init2); else add_matrix(M2, kc); }
            static_for<0, NB>([&](auto bic) {
                constexpr int bi = decltype(bic)::value;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int gi = 16 * bi + Mf::row_of(lane, r), gj = 16 * bi + lm;
                    T dv = T(0);
                    if (gi == gj) dv = (gi >= kq) ? T(1) : (!add_lam ? T(0) : ((gi == kt - 1) ? lam_last : lam));   // common.c:1060-1062, collective.c:1819
                    acc[wtix(bi, bi, NB)][r] += dv;
                }
            });
        };
        if constexpr (!PRODUCER) apply_init();
        // element (gi, kt - 1) of the initial matrices: the border column
        auto border_init = [&](int gi) -> T {
            T v = T(0);
            if (M1 != nullptr) v += M1[(size_t)min(gi, kt - 1) * kt + (kt - 1)];
            if (M2 != nullptr && kc > 0) { const T q = M2[(size_t)min(gi, kc - 1) * kc + (kc - 1)]; v += (kt - 1 < kc) ? q : T(0); }
            return v;
        };
        // right-hand side, border column: per lane the partial sum of its 4-row group g for the unknown 16 b + lm
        // (summed over the groups when the block becomes the pivot); prefilled values enter through group 0
        const bool pre_rhs = !PRODUCER && (has_u || P.rhs_prefilled_all);   // w*U*C prefilled (collective.c:5768-5773)
        T rp[NB], gp[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int u = 16 * b + lm;
            rp[b] = (pre_rhs && g == 0 && u < kq) ? arow[min(u, kt - 1)] : T(0);
            gp[b] = T(0);
            if (BORDER && !PRODUCER) { const T q = border_init(16 * b + lm_o); gp[b] = (g == 0 && u < kq) ? q : T(0); }
        }
        T gam = T(0), rbs = T(0);            // border diagonal and border right-hand side (partial over g, equal over lm)
        if (BORDER && !PRODUCER) {
            if (g == 0) {
                gam = (add_lam ? lam_last : T(0));
                gam += border_init(kt - 1);
                rbs = pre_rhs ? arow[kt - 1] : T(0);
            }
        }
        // ---- 1. rank-k update: PD steps of 4 gathered rows in flight ----
        const int nsteps = (nnz + 3) >> 2;
        const int niter = from_slices ? 0 : (nsteps + PD - 1) / PD;
        if constexpr (from_slices) {
            // the rank-k update of this row was done slice by slice: add the partials in slice order
            const bool hv = rix < SL.n_heavy;
            const int s0 = hv ? SL.row_off[rix] : SL.n_slices + (rix - SL.n_heavy), s1 = hv ? SL.row_off[rix + 1] : s0 + 1;
            for (int sl = s0; sl < ((SL.dbg_skip & 1) ? s0 : s1); sl++) {
                const T *pp = SL.part + (size_t)(sl - SL.part_base) * PART;
                // 16 values (4 tiles) per round trip, through wave_load16: left to the compiler every value waits for its own
                // load (with the factorisation's registers live it finds one free pair and serialises 144 round trips per
                // partial: 8 of 24 ms at the config-3 shape)
                add_tiles(pp);
                const T *pv = pp + (size_t)NT * 256;
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    const T q = pv[16 * b + lm];
                    rp[b] += (g == 0) ? q : T(0);
                    if (BORDER) { const T q2 = pv[16 * NB + 16 * b + lm]; gp[b] += (g == 0) ? q2 : T(0); }
                }
                if (BORDER) {
                    const T q3 = pv[32 * NB], q4 = pv[32 * NB + 1];
                    gam += (g == 0) ? q3 : T(0);
                    rbs += (g == 0) ? q4 : T(0);
                }
            }
        } else {
            // one sweep over the row's (or slice's) entries for the tiles [T0, T1) of the packed order; VEC: the
            // right-hand side and border column partials ride along
            auto rank_pass = [&](auto t0c, auto t1c, auto vecc) __attribute__((always_inline)) {
                constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value;
                constexpr bool VEC = decltype(vecc)::value;
                T opb[PD][NB], bvb[PD], xraw[PD], bsv[PD];
                bool vld[PD];
                int idxn[PD]; T xn[PD]; bool vn[PD];
                auto load_entry = [&](int s, int step) {
                    const int e = 4 * step + g;
                    vn[s] = e < nnz;
                    const int ec = min(e, nnz - 1);
                    idxn[s] = P.indices[st + ec];
                    xn[s] = P.values[st + ec];
                };
                auto issue_rows = [&](int s) {
                    const T *rowp = P.B + (size_t)idxn[s] * P.ldb;
                    static_for<0, NB>([&](auto bc) {
                        constexpr int b = decltype(bc)::value;
                        if constexpr (VEC || wave_block_needed(b, T0, T1, NB)) opb[s][b] = rowp[colb[b]];
                        else opb[s][b] = T(0);
                    });
                    bvb[s] = (BORDER && VEC) ? rowp[bcol] : T(0);
                    bsv[s] = (P.bias_sub != nullptr) ? P.bias_sub[idxn[s]] : T(0);
                    xraw[s] = xn[s]; vld[s] = vn[s];
                };
                if (nnz > 0 && niter > 0) {
#pragma unroll
                    for (int s = 0; s < PD; s++) load_entry(s, s);
#pragma unroll
                    for (int s = 0; s < PD; s++) { issue_rows(s); load_entry(s, PD + s); }
                }
                for (int it = 0; it < niter; it++) {
#pragma unroll
                    for (int s = 0; s < PD; s++) {
                        const T x = xraw[s] - bsv[s];
                        T ws = impl_w ? x : T(1);               // common.c:2091-2095, collective.c:2103-2108 vs common.c:1007-1012
                        T xw = impl_w ? x + T(1) : x;           // common.c:2082-2085, collective.c:2097-2101 vs common.c:991-996
                        if (!vld[s]) { ws = T(0); xw = T(0); }
                        T o[NB], a[NB];
#pragma unroll
                        for (int b = 0; b < NB; b++) {
                            o[b] = ((vmask >> b) & 1u) ? opb[s][b] : T(0);
                            a[b] = o[b] * ws;
                        }
                        const T bv = bvb[s];
                        // the next use of this buffer: step (it + 1) PD + s
                        if (it + 1 < niter) { issue_rows(s); load_entry(s, (it + 2) * PD + s); }
                        static_for<0, NB>([&](auto bic) {
                            constexpr int bi = decltype(bic)::value;
                            static_for<bi, NB>([&](auto bjc) {
                                constexpr int bj = decltype(bjc)::value;
                                constexpr int t = wtix(bi, bj, NB);
                                if constexpr (t >= T0 && t < T1) acc[t] = Mf::mma(a[bi], o[bj], acc[t]);
                            });
                        });
                        if constexpr (VEC) {
#pragma unroll
                            for (int b = 0; b < NB; b++) rp[b] += xw * o[b];
                            if (BORDER) {
                                const T bw = ws * bv;
#pragma unroll
                                for (int b = 0; b < NB; b++) gp[b] += bw * o[b];
                                gam += bw * bv;
                                rbs += xw * bv;
                            }
                        }
                    }
                }
            };
            if constexpr (PRODUCER && NOV > 0) {
                // The tiles do not fit the 256 accumulator registers (36 x 8 in double at 8 blocks, 153 x 4 in single at 17):
                // NOV tiles per sweep, stored, then the next NOV tiles in another sweep over the same entries (a sweep reads
                // only the blocks its tiles touch; the first one carries the right-hand side and border partials -- the LAST one in the
                // builds for two wavefronts per SIMD, whose first sweep holds more tiles than the last and has no registers to spare).
                constexpr int TP = NOV, NP = (NT + TP - 1) / TP;
                T *pp = SL.part + (size_t)(rix - SL.part_base) * PART;
                static_for<0, NP>([&](auto pc) {
                    constexpr int T0 = decltype(pc)::value * TP, T1 = (T0 + TP < NT) ? T0 + TP : NT;
                    static_for<T0, T1>([&](auto tc) { acc[decltype(tc)::value] = vec{0, 0, 0, 0}; });
                    rank_pass(std::integral_constant<int, T0>{}, std::integral_constant<int, T1>{}, std::integral_constant<bool, (WPS >= 2) ? (T1 == NT) : (T0 == 0)>{});
                    static_for<T0, T1>([&](auto tc) {
                        constexpr int t = decltype(tc)::value;
#pragma unroll
                        for (int r = 0; r < 4; r++) pp[t * 256 + r * 64 + lane] = acc[t][r];
                    });
                });
            } else {
                rank_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, NT>{}, std::true_type{});
            }
        }
        // ---- 2. right-hand side and border column become the tile column E (columns 0 and 1) ----
        if (PRODUCER || !EV) {
#pragma unroll
            for (int b = 0; b < NB; b++) {
                T v = lanes::tswap16_add(rp[b], rp[b]);
                v = lanes::tswap32_add(v, v);
                if (lane < 16) yv0[16 * b + lane] = v;
                if (BORDER) {
                    T w = lanes::tswap16_add(gp[b], gp[b]);
                    w = lanes::tswap32_add(w, w);
                    if (lane < 16) yv1[16 * b + lane] = w;
                }
            }
        }
        if (BORDER) {
            gam = lanes::tswap16_add(gam, gam); gam = lanes::tswap32_add(gam, gam);
            rbs = lanes::tswap16_add(rbs, rbs); rbs = lanes::tswap32_add(rbs, rbs);
        }
        CMF_LDS_FENCE();
        if constexpr (PRODUCER) {
            T *pp = SL.part + (size_t)(rix - SL.part_base) * PART;
            if constexpr (NOV == 0) {
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) pp[t * 256 + r * 64 + lane] = acc[t][r];
            }
            T *pv = pp + (size_t)NT * 256;
            for (int u = lane; u < 16 * NB; u += 64) {
                pv[u] = yv0[u];
                if (BORDER) pv[16 * NB + u] = yv1[u];
            }
            if (BORDER && lane == 0) { pv[32 * NB] = gam; pv[32 * NB + 1] = rbs; }
            CMF_LDS_FENCE();
            rix = P.row_first + nwaves + __builtin_amdgcn_readfirstlane(claim);
            continue;
        }
        vec E[EV ? 1 : NB];
        if constexpr (!EV) {
#pragma unroll
            for (int b = 0; b < NB; b++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int u = 16 * b + Mf::row_of(lane, r);
                    const T v0 = yv0[u];
                    const T v1 = BORDER ? yv1[u] : T(0);
                    E[b][r] = (lm == 0) ? v0 : ((BORDER && lm == 1) ? v1 : T(0));
                }
            }
            CMF_LDS_FENCE();
        }
        if constexpr (EV) {
            // the partials are not touched by the factorisation: out of the registers until the forward substitution
#pragma unroll
            for (int b = 0; b < NB; b++) { stash[b * 64 + lane] = rp[b]; if (BORDER) stash[(NB + b) * 64 + lane] = gp[b]; }
        }
        // ---- 3. blocked Cholesky  M = R^T R, in this wave's registers ----
        for (int kb = 0; kb < ((WMODE == 2 && (SL.dbg_skip & 2)) ? 0 : nb); kb++) {
            T *rslot = rinv + (size_t)kb * RSZ;
            vec dd = vec{0, 0, 0, 0};
            static_for<0, NB>([&](auto kc_) {
                constexpr int KB = decltype(kc_)::value;
                if (kb == KB) dd = acc[wtix(KB, KB, NB)];
            });
            chol_diag_block<T>(dd, rslot, lane, min(16, kq - 16 * kb));
            CMF_LDS_FENCE();
            T ainv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) ainv[r] = rslot[Mf::row_of(lane, r) * LDR + lm];
            static_for<0, NB>([&](auto kc_) {
                constexpr int KB = decltype(kc_)::value;
                if (kb == KB) {
                    // panel:  X = inv(R_kk)^T * tile  (one chain per tile; the tiles of a block row interleave)
                    auto panel = [&](vec tIn) -> vec {
                        vec x = Mf::mma(ainv[0], tIn[0], vec{0, 0, 0, 0});
                        x = Mf::mma(ainv[1], tIn[1], x);
                        x = Mf::mma(ainv[2], tIn[2], x);
                        x = Mf::mma(ainv[3], tIn[3], x);
                        return x;
                    };
                    static_for<KB + 1, NB>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        acc[wtix(KB, j, NB)] = panel(acc[wtix(KB, j, NB)]);
                    });
                    if constexpr (!EV) E[KB] = panel(E[KB]);
                    // trailing:  tile(bi, bj) -= X_bi^T X_bj ;  E_bi -= X_bi^T E_k  (forward substitution)
                    static_for<KB + 1, NB>([&](auto bic) {
                        constexpr int bi = decltype(bic)::value;
                        T na[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) na[r] = -acc[wtix(KB, bi, NB)][r];
                        static_for<bi, NB>([&](auto bjc) {
                            constexpr int bj = decltype(bjc)::value;
                            constexpr int t = wtix(bi, bj, NB);
                            if constexpr (t < NRES) {
#pragma unroll
                                for (int r = 0; r < 4; r++) acc[t] = Mf::mma(na[r], acc[wtix(KB, bj, NB)][r], acc[t]);
                            } else {
                                vec tmp = Mf::mma(na[0], acc[wtix(KB, bj, NB)][0], vec{0, 0, 0, 0});
#pragma unroll
                                for (int r = 1; r < 4; r++) tmp = Mf::mma(na[r], acc[wtix(KB, bj, NB)][r], tmp);
                                acc[t] += tmp;
                            }
                        });
                        if constexpr (EV) {
                        } else if constexpr (NOV == 0) {
#pragma unroll
                            for (int r = 0; r < 4; r++) E[bi] = Mf::mma(na[r], E[KB][r], E[bi]);
                        } else {
                            vec tmp = Mf::mma(na[0], E[KB][0], vec{0, 0, 0, 0});
#pragma unroll
                            for (int r = 1; r < 4; r++) tmp = Mf::mma(na[r], E[KB][r], tmp);
                            E[bi] += tmp;
                        }
                    });
                }
            });
        }
        if constexpr (EV) {
            // forward substitution after the factorisation, on the vector ALU:  y_k = inv(R_kk)^T (v_k - sum_{i<k} R_ik^T y_i)
            // for the right-hand side and the border column at once.  rp / gp still hold the per-group partials of v; a
            // lane adds the products of its four rows of every tile (i, k), the groups are summed, one 16 x 16
            // matrix-vector product per block through LDS.
            for (int kb = 0; kb < ((WMODE == 2 && (SL.dbg_skip & 4)) ? 0 : nb); kb++) {
                const T *rslot = rinv + (size_t)kb * RSZ;
                T vk = T(0), wk = T(0);
                static_for<0, NB>([&](auto kc_) {
                    constexpr int KB = decltype(kc_)::value;
                    if (kb == KB) {
                        T s0 = T(0), s1 = T(0);
                        static_for<0, KB>([&](auto ic) {
                            constexpr int i = decltype(ic)::value;
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const T tv = acc[wtix(i, KB, NB)][r];
                                s0 += tv * yv0[16 * i + Mf::row_of(lane, r)];
                                if (BORDER) s1 += tv * yv1[16 * i + Mf::row_of(lane, r)];
                            }
                        });
                        vk = stash[KB * 64 + lane] - s0; wk = BORDER ? stash[(NB + KB) * 64 + lane] - s1 : T(0);
                    }
                });
                vk = lanes::tswap16_add(vk, vk); vk = lanes::tswap32_add(vk, vk);
                if (BORDER) { wk = lanes::tswap16_add(wk, wk); wk = lanes::tswap32_add(wk, wk); }
                if (lane < 16) { yv0[16 * kb + lane] = vk; if (BORDER) yv1[16 * kb + lane] = wk; }
                CMF_LDS_FENCE();
                T y0 = T(0), y1 = T(0);
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    const T ic = rslot[c * LDR + lm];
                    y0 += ic * yv0[16 * kb + c];
                    if (BORDER) y1 += ic * yv1[16 * kb + c];
                }
                CMF_LDS_FENCE();
                if (lane < 16) { yv0[16 * kb + lane] = y0; if (BORDER) yv1[16 * kb + lane] = y1; }
                CMF_LDS_FENCE();
            }
        }
        // y (column 0 of E) and R^-T g (column 1) back to LDS
        if constexpr (!EV) {
#pragma unroll
            for (int b = 0; b < NB; b++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int u = 16 * b + Mf::row_of(lane, r);
                    if (lm == 0) yv0[u] = E[b][r];
                    if (BORDER && lm == 1) yv1[u] = E[b][r];
                }
            }
            CMF_LDS_FENCE();
        }
        T xlast = T(0);
        if (BORDER) {
            // rho^2 = gamma - r.r ;  y_last = (rhs_last - r.y) / rho ;  x_last = y_last / rho ;  z = y - r x_last
            T s1 = T(0), s2 = T(0);
            for (int u = lane; u < 16 * nb; u += 64) { const T rv = yv1[u]; s1 += rv * rv; s2 += rv * yv0[u]; }
            s1 = lanes::wave_sum(s1); s2 = lanes::wave_sum(s2);
            const T rho2 = gam - s1;
            xlast = (rbs - s2) / rho2;
            CMF_LDS_FENCE();
            for (int u = lane; u < 16 * nb; u += 64) yv0[u] -= yv1[u] * xlast;
            CMF_LDS_FENCE();
        }
        // ---- 4. backward substitution R x = z, one block column per step ----
        for (int bjk = ((WMODE == 2 && (SL.dbg_skip & 8)) ? -1 : nb - 1); bjk >= 0; bjk--) {
            const T *rslot = rinv + (size_t)bjk * RSZ;
            T xm = T(0);                          // x[16 bjk + lm], computed redundantly by every 16-lane group
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
                        // sum over the 16 lanes of a row for the four registers at once: after two select-and-exchange
                        // steps lane l carries register (l & 3), then two plain butterflies
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
        for (int e = lane; e < kq; e += 64) arow[e] = xall[e];
        if (BORDER && lane == 0) arow[kt - 1] = xlast;
        CMF_LDS_FENCE();                     // this wave's LDS is reused by its next row
        rix = P.row_first + nwaves + __builtin_amdgcn_readfirstlane(claim);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 5: the rank-k update of the eight-block rows by TWO (or four) wavefronts per row -- the producer of the two-kernel closed
// form in double precision at k_t = 128 / 129 (config 3).  One wavefront per SIMD issues a v_mfma_f64_16x16x4 every ~130 cycles, two or
// more keep the pipe busy (35-38 against 75 TFLOP/s on the whole chip, profiles/r05/r05_za_mfma_f64_occupancy.txt) -- and the
// one-wavefront-per-row producer holds 32 tiles in 256 accumulator registers, one wavefront per SIMD.  Its tiles in sweeps of 16 fit
// two wavefronts per SIMD, but every sweep gathers the row again and for the item step those re-reads go to HBM (measured slower,
// r05_zb).  Here the NP wavefronts of a workgroup take the SAME row (or slice) at the same time, each a part of the 36 tiles --
// halves: the tile rows {0, 3, 4, 7} / {1, 2, 5, 6}, 18 tiles = 144 registers, 8 / 7 block loads per step of four entries;
// quarters: the rows {Q, 7 - Q}, nine tiles -- straight from the gather like gramk_producer_kernel: the second wavefront finds the
// lines in cache.  Two wavefronts per SIMD (halves: four workgroups of 128 threads per CU).  The right-hand side and border-column
// partials of a part's own blocks ride with it.  Measured on config 3 (same box, alternating, tools/experiments/
// r05_rank_k_two_waves_per_simd): quarters 15.08 -> 15.00 ms (the per-step bookkeeping is paid four times, on the datapath the matrix
// instructions use), halves 15.10 -> 14.75-14.84 ms (the kernel 1.64 -> 1.47 ms per launch): the halves are the default.  FULL: every
// unknown of the tiles takes a gathered value (the only build instantiated; other shapes keep the one-wavefront producer).
// Output: the partial of chol_wave_kernel's producer build (tiles in the accumulator layout, packed order; right-hand side; border
// column; two scalars), so the factorisation build (WMODE 2) and gram_cg_wide_kernel consume it unchanged.
// tile rows of part Q of NP: quarters {Q, 7 - Q} (9 tiles each), halves {0, 3, 4, 7} / {1, 2, 5, 6} (18 tiles each)
__host__ __device__ constexpr int cq_row(int NP, int Q, int i)
{
    if (NP == 4) return i == 0 ? Q : 7 - Q;
    return Q == 0 ? (i == 0 ? 0 : i == 1 ? 3 : i == 2 ? 4 : 7) : (i == 0 ? 1 : i == 1 ? 2 : i == 2 ? 5 : 6);
}
__host__ __device__ constexpr int cq_off(int NP, int Q, int i)      // first accumulator of the part's i-th tile row
{
    int o = 0;
    for (int j = 0; j < i; j++) o += 8 - cq_row(NP, Q, j);
    return o;
}
template <typename T, int NB, bool BORDER, int PD, int NP, int Q, bool FULL>
__device__ __forceinline__ void chol_part_rank_k(const CholParams<T> &P, size_t st, int nnz, T *__restrict__ pp, int lane)
{
    using Mf = CholMfma<T>;
    using vec = typename Mf::vec;
    static_assert(NB == 8 && (NP == 4 || NP == 2) && Q >= 0 && Q < NP, "parts of an eight-block grid");
    constexpr int NR = NB / NP;                       // tile rows of this part
    constexpr int R0 = cq_row(NP, Q, 0);              // its first row: the part loads the blocks R0 .. NB - 1
    constexpr int NTP = cq_off(NP, Q, NR);
    constexpr int NT = NB * (NB + 1) / 2;
    vec acc[NTP];
#pragma unroll
    for (int i = 0; i < NTP; i++) acc[i] = vec{0, 0, 0, 0};
    const int lm = lane & 15, g = lane >> 4;
    const int kt = P.kt, koff = P.koff;
    const int kq = kt - (BORDER ? 1 : 0);
    const int kbcols = kt - koff;
    // FULL: every unknown of the tiles takes a gathered value (koff = 0, kq = 16 NB: config 3) -- the loads of a row are one address
    // plus immediate offsets and no operand is masked; otherwise the clamped column per block and the mask of chol_wave_kernel
    int colb[FULL ? 1 : NB];
    unsigned vmask = 0;
    if constexpr (!FULL) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int u = 16 * b + lm;
            const bool ok = (u >= koff) && (u < kq);
            colb[b] = min(max(u - koff, 0), kbcols - 1);
            vmask |= ok ? (1u << b) : 0u;
        }
    }
    const int bcol = kbcols - 1;
    const bool impl_w = (P.mode == CHOL_IMPLICIT || P.mode == CHOL_COLLECTIVE_IMPLICIT);
    T rp[NR], gp[NR], gam = T(0), rbs = T(0);
#pragma unroll
    for (int i = 0; i < NR; i++) { rp[i] = T(0); gp[i] = T(0); }
    const int nsteps = (nnz + 3) >> 2;
    const int niter = (nsteps + PD - 1) / PD;
    T opb[PD][NB], bvb[PD], xraw[PD], bsv[PD];
    bool vld[PD];
    int idxn[PD]; T xn[PD]; bool vn[PD];
    auto load_entry = [&](int s, int step) {
        const int e = 4 * step + g;
        vn[s] = e < nnz;
        const int ec = max(min(e, nnz - 1), 0);
        idxn[s] = P.indices[st + ec];
        xn[s] = P.values[st + ec];
    };
    auto issue_rows = [&](int s) {
        const T *rowp = P.B + (size_t)idxn[s] * P.ldb;
        if constexpr (FULL) {
            const T *rl = rowp + lm;
            static_for<R0, NB>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                opb[s][b] = rl[16 * b];
            });
        } else {
            static_for<R0, NB>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                opb[s][b] = rowp[colb[b]];
            });
        }
        bvb[s] = BORDER ? rowp[bcol] : T(0);
        bsv[s] = (P.bias_sub != nullptr) ? P.bias_sub[idxn[s]] : T(0);
        xraw[s] = xn[s]; vld[s] = vn[s];
    };
    if (nnz > 0) {
#pragma unroll
        for (int s = 0; s < PD; s++) load_entry(s, s);
#pragma unroll
        for (int s = 0; s < PD; s++) { issue_rows(s); load_entry(s, PD + s); }
    }
    for (int it = 0; it < niter; it++) {
#pragma unroll
        for (int s = 0; s < PD; s++) {
            const T x = xraw[s] - bsv[s];
            T ws = impl_w ? x : T(1);               // common.c:2091-2095, collective.c:2103-2108 vs common.c:1007-1012
            T xw = impl_w ? x + T(1) : x;           // common.c:2082-2085, collective.c:2097-2101 vs common.c:991-996
            if (!vld[s]) { ws = T(0); xw = T(0); }
            T o[NB];
            static_for<R0, NB>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                o[b] = (FULL || ((vmask >> b) & 1u)) ? opb[s][b] : T(0);
            });
            const T bv = bvb[s];
            if (it + 1 < niter) { issue_rows(s); load_entry(s, (it + 2) * PD + s); }
            static_for<0, NR>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int R = cq_row(NP, Q, i), OFF = cq_off(NP, Q, i);
                const T a = o[R] * ws;
                static_for<R, NB>([&](auto bjc) {
                    constexpr int bj = decltype(bjc)::value;
                    acc[OFF + bj - R] = Mf::mma(a, o[bj], acc[OFF + bj - R]);
                });
                rp[i] += xw * o[R];
                if (BORDER) gp[i] += (ws * bv) * o[R];
            });
            if (BORDER && Q == 0) { gam += (ws * bv) * bv; rbs += xw * bv; }
        }
    }
    static_for<0, NR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int R = cq_row(NP, Q, i), OFF = cq_off(NP, Q, i);
#pragma unroll
        for (int j = 0; j < NB - R; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) pp[(size_t)wtix(R, R + j, NB) * 256 + r * 64 + lane] = acc[OFF + j][r];
    });
    T *pv = pp + (size_t)NT * 256;
    auto over_groups = [&](T v) -> T { v = lanes::tswap16_add(v, v); return lanes::tswap32_add(v, v); };
    static_for<0, NR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int R = cq_row(NP, Q, i);
        const T v = over_groups(rp[i]);
        if (lane < 16) pv[16 * R + lane] = v;
        if (BORDER) {
            const T w = over_groups(gp[i]);
            if (lane < 16) pv[16 * NB + 16 * R + lane] = w;
        }
    });
    if (BORDER && Q == 0) {
        gam = over_groups(gam); rbs = over_groups(rbs);
        if (lane == 0) { pv[32 * NB] = gam; pv[32 * NB + 1] = rbs; }
    }
}

// NP wavefronts per workgroup and work item (4: quarters, 2: halves); WPS wavefronts per SIMD = 4 WPS / NP workgroups per CU
template <typename T, int NB, bool BORDER, int PD, int WPS, bool FULL, int NP = 4>
__global__ void __launch_bounds__(64 * NP, WPS)
chol_parts_producer_kernel(const CholParams<T> P, const RowDesc *__restrict__ desc, const CholSlices<T> SL)
{
    constexpr size_t PART = chol_wave_part_elems(NB);
    __shared__ int s_next;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // work items are claimed per workgroup: position blockIdx.x first, then gridDim.x + counter
    int rix = P.row_first + blockIdx.x;
    while (rix < P.nrows) {
        if (threadIdx.x == 0) s_next = atomicAdd(P.counter, 1);
        int ritem, sfirst = 0, scount;
        if (rix < SL.n_slices) { ritem = SL.vrow[rix]; sfirst = SL.first[rix]; scount = SL.count[rix]; }
        else { ritem = SL.n_heavy + (rix - SL.n_slices); scount = -1; }
        const RowDesc d = desc[ritem];
        const int nnz_row = __builtin_amdgcn_readfirstlane(d.nnz);
        const size_t st_row = ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(d.st >> 32)) << 32) |
                              (unsigned)__builtin_amdgcn_readfirstlane((int)(d.st & 0xffffffffu));
        const int nnz = (scount >= 0) ? __builtin_amdgcn_readfirstlane(scount) : nnz_row;
        const size_t st = st_row + (size_t)__builtin_amdgcn_readfirstlane(sfirst);
        T *pp = SL.part + (size_t)(rix - SL.part_base) * PART;
        if constexpr (NP == 4) {
            switch (wave) {
                case 0: chol_part_rank_k<T, NB, BORDER, PD, 4, 0, FULL>(P, st, nnz, pp, lane); break;
                case 1: chol_part_rank_k<T, NB, BORDER, PD, 4, 1, FULL>(P, st, nnz, pp, lane); break;
                case 2: chol_part_rank_k<T, NB, BORDER, PD, 4, 2, FULL>(P, st, nnz, pp, lane); break;
                default: chol_part_rank_k<T, NB, BORDER, PD, 4, 3, FULL>(P, st, nnz, pp, lane); break;
            }
        } else {
            if (wave == 0) chol_part_rank_k<T, NB, BORDER, PD, 2, 0, FULL>(P, st, nnz, pp, lane);
            else chol_part_rank_k<T, NB, BORDER, PD, 2, 1, FULL>(P, st, nnz, pp, lane);
        }
        __syncthreads();
        rix = P.row_first + (int)gridDim.x + s_next;
        __syncthreads();
    }
}

}  // namespace cmfhip
