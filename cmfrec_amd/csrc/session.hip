// session.hip -- device-resident ALS session (level 3 of include/cmfrec_hip.h) and the
// operator-level entry points with host buffers (level 2).
//
// The session owns the state the reference's fit drivers keep on the host
// (/root/reference/src/collective.c:7651-7677, :9468-9473): the factor matrices in the
// "[rows, k_tot + bias]" layout, the bias vectors, CSR and CSC copies of X, side information and
// the small k x k work matrices -- all resident in HBM for the whole fit.
#include "device.hpp"
#include "coo_device.hpp"
#include "chol_wave_kernels.hpp"
#include "gramk_kernels.hpp"
#include "lowrank_kernels.hpp"
#include "eig_kernels.hpp"
#include "gram_cg_wide_kernels.hpp"
#include <dlfcn.h>
#include <functional>
#include <memory>
#include <new>

namespace cmfhip {

thread_local std::string g_last_error;
thread_local int g_last_rc = 0;      // return code that goes with g_last_error where the entry point returns a pointer

CgVariant cg_variant_from_env() { return switches().cg_generic ? CgVariant::Generic : CgVariant::Auto; }

inline dim3 grid1d(size_t n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }
// (chol_wg_tu.hip: the four-wavefront factorisation of the eight-block rows, double precision)
hipError_t launch_chol_wg8(int num_cus, bool border, int waves_per_row, hipStream_t st, const CholParams<real_t> &W, const RowDesc *desc, const CholSlices<real_t> &SL);
hipError_t launch_chol_parts_coop(int num_cus, bool border, int depth, hipStream_t st, const CholParams<real_t> &W, const RowDesc *desc, const CholSlices<real_t> &SL);

struct CholCall {
    real_t *A; size_t lda;
    const real_t *B; size_t ldb;
    int kt, koff;
    const real_t *bias_sub;
    const real_t *Minit; int kc; int rows_with_u; int p_side;
    real_t lam, lam_last;
    bool scale_lam, scale_lam_sideinfo, scale_bias_const;
    int mode;
    const real_t *Mfull = nullptr;
    // second gather source (sparse side information): rows of B2[*, ldb2] selected by X2's entries, unknowns [0, kc2)
    const SparseShard *X2 = nullptr;
    const real_t *B2 = nullptr;
    size_t ldb2 = 0;
    int kc2 = 0;
    real_t w2 = 0;
    // non-negative factors: coordinate descent on the assembled system instead of the Cholesky solve
    bool nonneg = false;
    int max_cd_steps = 100;
    // second source placed at unknown koff2 and used for the right-hand side only (implicit-features term)
    int koff2 = 0;
    bool w2_syr_zero = false;
    int rows2 = -1;
    const real_t *values2 = nullptr;         // values of the second source (default: X2's own)
    const real_t *values_override = nullptr; // values of the first source (default: X's own)
    bool rhs_only = false;                   // CHOL_NAZ: gather the right-hand sides only
    bool rhs_prefilled_all = false;          // every row starts from the right-hand side left in A
    const real_t *weights_override = nullptr; // CHOL_NAZ_W: the entries' rank-1 weights (w - 1), CSR order of X
    bool x_rhs_only = false;                 // the entries of X add to the right-hand sides only (their Gramian is inside Mfull)
    bool entry_pairs = false;                // weights_override / values_override are the entries' rank-1 and right-hand-side weights
    const real_t *mult_override = nullptr;   // per-row lambda multipliers of the collective modes instead of the rows' lengths
    bool all_rows = false;                   // CHOL_NAZ_W: rows without entries are solved too (from the prefilled right-hand side)
    int row_limit = -1;                      // only the first row_limit positions of the processing order (the others are solved elsewhere)
    // CG on the row's Gramian instead of the factorisation (gram_cg_wide_kernels.hpp): the producer build of the wave kernel runs
    // every row's rank-k update, gram_cg_wide_kernel takes the partials; the parameters of the CG (explicit model)
    const CgParams<real_t> *cg_wide = nullptr;
};

static int launch_chol_rows(const DeviceInfo &dev, const CholCall &c, const SparseShard *X, CholParams<real_t> P, bool two_src,
                            size_t smem_nonneg);

static int launch_plain_lowrank(const DeviceInfo &dev, const CholCall &c, const SparseShard &X);

#ifndef CMF_PARTS_NP
#define CMF_PARTS_NP 2             // wavefronts per row of chol_parts_producer_kernel: 2 = halves of 18 tiles (default), 4 = quarters of 9 (measured on par with the round-2 producer)
#define CMF_PARTS_PD 2             // steps of four gathered rows in flight per wavefront (quarters: 4)
#define CMF_PARTS_WPS 2            // wavefronts per SIMD
#endif
// X may be null for CHOL_PREFILLED (then nrows_prefilled rows are solved in natural order)
static int launch_chol(const DeviceInfo &dev, const CholCall &c, const SparseShard *X, int nrows_prefilled = 0)
{
    if (X != nullptr && c.mode == CHOL_EXPLICIT && c.row_limit < 0 && c.cg_wide == nullptr) {
        const int rc_lr = launch_plain_lowrank(dev, c, *X);      // rows with few entries against many unknowns (-1: not applicable)
        if (rc_lr >= 0) return rc_lr;
    }
    CholParams<real_t> P;
    P.A = c.A; P.lda = c.lda; P.B = c.B; P.ldb = c.ldb; P.kt = c.kt; P.koff = c.koff;
    P.indptr = X ? X->p.ptr : nullptr; P.indices = X ? X->i.ptr : nullptr; P.values = X ? X->v.ptr : nullptr;
    P.bias_sub = c.bias_sub;
    // observation weights of the explicit model ride on the shard (SparseShard::w / wsum)
    const bool weighted = X != nullptr && X->weighted() && (c.mode == CHOL_EXPLICIT || c.mode == CHOL_COLLECTIVE) && c.values_override == nullptr;
    if (weighted) { P.weights = X->w.ptr; P.wsum = X->wsum.ptr; }
    if (c.mode == CHOL_NAZ_W) { P.weights = c.weights_override; P.wsum = X->wsum_naz.ptr; }
    if (c.mult_override != nullptr) P.wsum = c.mult_override;
    P.x_rhs_only = c.x_rhs_only ? 1 : 0;
    if (c.entry_pairs) { P.entry_pairs = 1; P.weights = c.weights_override; }
    P.order = X ? X->order.ptr : nullptr;
    if (c.mode == CHOL_PREFILLED) P.nrows = nrows_prefilled;
    else if (c.mode == CHOL_COLLECTIVE || c.mode == CHOL_COLLECTIVE_IMPLICIT || (c.mode == CHOL_NAZ_W && c.all_rows)) P.nrows = X->nrows;   // empty rows too
    else P.nrows = X->n_nonempty;                                                // non-empty rows
    if (c.row_limit >= 0) P.nrows = std::min(P.nrows, c.row_limit);
    P.Minit = c.Minit; P.Mfull = c.Mfull; P.kc = c.kc; P.rows_with_u = c.rows_with_u; P.p_side = c.p_side;
    P.lam = c.lam; P.lam_last = c.lam_last;
    P.scale_lam = c.scale_lam; P.scale_lam_sideinfo = c.scale_lam_sideinfo; P.scale_bias_const = c.scale_bias_const;
    P.mode = c.mode;
    if (c.X2) {
        P.indptr2 = c.X2->p.ptr; P.indices2 = c.X2->i.ptr; P.values2 = c.values2 ? c.values2 : c.X2->v.ptr;
        P.B2 = c.B2; P.ldb2 = c.ldb2; P.kc2 = c.kc2; P.w2 = c.w2; P.koff2 = c.koff2; P.w2_syr_zero = c.w2_syr_zero ? 1 : 0; P.rows_src2 = c.rows2;
    }
    P.values_override = c.values_override; P.rhs_only = c.rhs_only ? 1 : 0; P.rhs_prefilled_all = c.rhs_prefilled_all ? 1 : 0;
    const bool l1on = dev.l1_now != (real_t)0 || dev.l1_last_now != (real_t)0;
    const bool nonneg = c.nonneg || dev.nonneg_now;
    P.nonneg = nonneg ? 1 : 0; P.max_cd_steps = (dev.nonneg_now || l1on) ? dev.max_cd_steps : c.max_cd_steps;
    // PREFILLED launches carry their lambda inside the prefilled matrix, scaled on the host (common.c:2832-2833): their L1
    // penalty takes the same factor (solve_elasticnet_batch / solve_nonneg_batch calls, :2876-2902)
    const real_t l1_mult = (c.mode == CHOL_PREFILLED || c.mode == CHOL_NAZ) ? dev.l1_scale : (real_t)1;
    P.l1 = dev.l1_now * l1_mult; P.l1_last = dev.l1_last_now * l1_mult;
    if (P.nrows <= 0) return 0;
    const int T = chol_tiles(c.kt);
    const size_t smem_nonneg = (nonneg || l1on) ? ((size_t)c.kt * c.kt + 2 * (size_t)c.kt + 64) * sizeof(real_t) : 0;
    if (smem_nonneg > 160 * 1024) {
        g_last_error = "cmfrec_hip: nonneg: the k_t x k_t system must fit the 160 KB of LDS (k_t <= 140 in double, 199 in single precision)";
        return 2;
    }
    if (T > 17 || (sizeof(real_t) == 8 && T > 16)) {
        g_last_error = "cmfrec_hip: Cholesky path: k_t too large for the register-resident normal matrix "
                       "(k_t <= 256 in double, <= 272 in single precision)";
        return 2;
    }
    if (dev.row_counter.n < ROW_COUNTER_INTS) const_cast<DeviceInfo &>(dev).row_counter.alloc(ROW_COUNTER_INTS);
    HIP_CHECK(hipMemsetAsync(dev.row_counter.ptr, 0, 4 * sizeof(int), dev.stream));   // [0, 4): first launches of this call
    P.counter = dev.row_counter.ptr;
    P.row_first = 0;
    const bool two_src = c.X2 != nullptr || c.mode == CHOL_NAZ || c.mode == CHOL_NAZ_W || c.x_rhs_only || c.entry_pairs;   // (that build only)
    // Rows of up to WAVE_ROW_MAX entries: one wavefront per row, the matrix in its registers (chol_wave_kernels.hpp).
    // The rows beyond (they lead the processing order) stay with the workgroup-per-row kernel below; the two launches
    // run side by side on two streams.  CMFREC_HIP_CHOL=rows keeps everything on the workgroup-per-row kernel (A/B
    // switch and on-device cross-check).
    {
        const int chol_sw = switches().chol;        // 1: rows, 2: noslices
        const bool border = (c.kt > 16) && ((c.kt - 1) % 16 == 0);
        const int nbw = (c.kt - (border ? 1 : 0) + 15) / 16;
        const bool wave_ok = X != nullptr && !two_src && !nonneg && !l1on && !c.rhs_only && nbw <= 8 && !weighted &&
                             (c.mode == CHOL_EXPLICIT || c.mode == CHOL_IMPLICIT || c.mode == CHOL_COLLECTIVE ||
                              c.mode == CHOL_COLLECTIVE_IMPLICIT) &&
                             chol_sw != 1;
        if (wave_ok) {
            // rows beyond 1024 entries: sliced (their slice tables are the ones of the split-row CG path, SparseShard::sl_*);
            // CMFREC_HIP_CHOL=noslices leaves them to the workgroup-per-row kernel (A/B switch)
            const bool sliced = chol_sw != 2;
            const int n_heavy = std::min(X->bin_rows[BIN_VHEAVY], P.nrows);
            const int total = P.nrows;
            DeviceInfo &d = const_cast<DeviceInfo &>(dev);
            // one launch of the wave kernel over the positions [first, last) of the processing order (or over work items)
            auto wave_launch = [&](int wmode, int first, int last, int counter, const CholSlices<real_t> &SL, hipStream_t st) {
                if (last <= first) return;
                CholParams<real_t> W = P;
                W.row_first = first; W.nrows = last; W.counter = dev.row_counter.ptr + counter;
                auto wlaunch = [&](auto kern, int nb_, int wps) {
                    poison_lds(st, dev.num_cus);      // test hook, device.hpp
                    const size_t smem = 4 * (wmode == 1 ? chol_wave_lds_elems_producer(nb_) : chol_wave_lds_elems<real_t>(nb_)) * sizeof(real_t);
                    const int grid = std::min((last - first + 3) / 4, dev.num_cus * wps);
                    if (smem > 48 * 1024)
                        HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, W, X->desc.ptr, SL);
                };
                // <16-blocks, border, gather steps in flight, wavefronts per SIMD, overflow tiles, mode>
#define WAVE_KERN_M(nb_, pd, wps, nov, wm) (border ? chol_wave_kernel<real_t, nb_, true, pd, wps, nov, wm, false> : chol_wave_kernel<real_t, nb_, false, pd, wps, nov, wm, false>)
#define WAVE_KERN(nb_, pd, wps, nov) \
    (wmode == 1 ? WAVE_KERN_M(nb_, pd, wps, nov, 1) : wmode == 2 ? WAVE_KERN_M(nb_, pd, wps, nov, 2) : WAVE_KERN_M(nb_, pd, wps, nov, 0))
#ifdef CMFREC_HIP_FLOAT
                if (nbw <= 2) wlaunch(WAVE_KERN(2, 4, 4, 0), 2, 4);
                else if (nbw <= 4) wlaunch(WAVE_KERN(4, 4, 2, 0), 4, 2);
                else if (nbw <= 6) wlaunch(WAVE_KERN(6, 3, 2, 0), 6, 2);
                else wlaunch(WAVE_KERN(8, 2, 1, 0), 8, 1);
#else
                if (nbw <= 2) wlaunch(WAVE_KERN(2, 4, 3, 0), 2, 3);
                else if (nbw <= 4) wlaunch(WAVE_KERN(4, 3, 2, 0), 4, 2);
                else if (nbw <= 6) wlaunch(WAVE_KERN(6, 2, 1, 0), 6, 1);
                // 7 and 8 blocks in double: two kernels, see below
                // Round 5: one wavefront per SIMD issues a double-precision MFMA every ~130 cycles, two keep the pipe busy (35-38 against
                // 75 TFLOP/s, profiles/r05/r05_za_mfma_f64_occupancy.txt).  Where every unknown of the tiles is gathered (koff = 0,
                // k_t = 128 / 129: config 3) the rank-k update is done by TWO wavefronts per row with 18 tiles each, four workgroups
                // of 128 threads per CU (chol_parts_producer_kernel; config 3 15.10 -> 14.75-14.84 ms, r05_ze); the other shapes and
                // -DCMF_WAVE_PROD_ONE_WAVE keep the round-2 producer (one wavefront per row and SIMD, 32 + 4 tiles in two sweeps).
                else if (wmode == 1) {
                    const bool fullq = c.koff == 0 && c.kt - (border ? 1 : 0) == 128;
#ifdef CMF_WAVE_PROD_ONE_WAVE
                    const bool parts = false;
#else
                    const bool parts = fullq;
#endif
                    if (!parts) wlaunch(WAVE_KERN_M(8, 3, 1, 32, 1), 8, 1);      // 32 tiles per sweep
                    else if (switches().parts_coop && CMF_PARTS_NP == 2) {
                        // Round 6: the row's two wavefronts share ONE gather through LDS (chol_parts_coop_kernels.hpp, chol_wg_tu.hip):
                        // config 3 12.97 -> 12.80 ms, the partials bit for bit the same (CMFREC_HIP_PARTS_COOP=0: round 5's kernel)
                        poison_lds(st, dev.num_cus);
                        HIP_CHECK(launch_chol_parts_coop(dev.num_cus, border, switches().parts_coop == 3 ? 3 : 4, st, W, X->desc.ptr, SL));
                    }
                    else {
                        poison_lds(st, dev.num_cus);
                        constexpr int NPQ = CMF_PARTS_NP;
                        auto kern = border ? chol_parts_producer_kernel<real_t, 8, true, CMF_PARTS_PD, CMF_PARTS_WPS, true, NPQ>
                                           : chol_parts_producer_kernel<real_t, 8, false, CMF_PARTS_PD, CMF_PARTS_WPS, true, NPQ>;
                        const int grid = std::min(last - first, dev.num_cus * CMF_PARTS_WPS * 4 / NPQ);
                        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NPQ), 0, st, W, X->desc.ptr, SL);
                    }
                }
                else if (switches().chol_wg) {
                    // Round 6: the factorisation by one workgroup of TWO wavefronts per row, the 36 tiles dealt to them, four rows per CU in
                    // flight (chol_wg_kernels.hpp) instead of one wavefront per row and SIMD with 263 spilled registers: config 3
                    // 14.2 -> 13.0 ms (profiles/r06/r06_j_*).  CMFREC_HIP_CHOL_WG=4: four wavefronts per row, two rows per CU (14.0);
                    // =0: the one-wavefront build below (A/B switch and on-device cross-check).
                    poison_lds(st, dev.num_cus);
                    HIP_CHECK(launch_chol_wg8(dev.num_cus, border, switches().chol_wg == 2 ? 2 : 4, st, W, X->desc.ptr, SL));       // chol_wg_tu.hip
                }
                else wlaunch((border ? chol_wave_kernel<real_t, 8, true, 1, 1, 6, 2, true> : chol_wave_kernel<real_t, 8, false, 1, 1, 6, 2, true>), 8, 1);
#endif
#undef WAVE_KERN
#undef WAVE_KERN_M
                HIP_CHECK(hipGetLastError());
            };
            const int nb_inst = nbw <= 2 ? 2 : nbw <= 4 ? 4 : nbw <= 6 ? 6 : 8;
            const size_t part_elems = chol_wave_part_elems(nb_inst);
            CholSlices<real_t> SLT;
            if (n_heavy > 0) {
                SLT.vrow = X->sl_vrow.ptr; SLT.first = X->sl_first.ptr; SLT.count = X->sl_count.ptr; SLT.row_off = X->row_sl_off.ptr;
                SLT.n_slices = X->n_slices;
            }
            SLT.n_heavy = n_heavy;
            SLT.dbg_skip = switches().debug_skip;
            const bool cg_wide = c.cg_wide != nullptr;
            if ((sizeof(real_t) == 8 && nbw > 6) || cg_wide) {
                // (cg_wide: the second kernel is the CG on the summed partials, gram_cg_wide_kernels.hpp -- every precision and width)
                // Two kernels (7 and 8 blocks in double: 28 / 36 tiles of 8 registers + the factorisation's temporaries exceed
                // what one kernel can keep in registers -- hipcc emits the accumulator-file form of the MFMAs at one wave per
                // SIMD, 256 registers for all MFMA destinations).  The producer build runs every row's rank-k update at full
                // MFMA rate (no spills, three gather steps in flight) and leaves the raw tiles in HBM, the second build adds the
                // initial matrix and factorises.  76 KB per work item each way; batches bound the scratch buffer.
                const int BATCH = 32768;
                const std::vector<int> &ho = X->h_row_sl_off;
                int max_row_items = 1;
                for (int r = 0; r < n_heavy; r++) max_row_items = std::max(max_row_items, ho[r + 1] - ho[r]);
                const int nsl = n_heavy > 0 ? X->n_slices : 0;
                const size_t cap_items = (size_t)std::min<long long>((long long)nsl + (total - n_heavy), std::max(BATCH, max_row_items));
                if (X->chol_part.n < cap_items * part_elems) X->chol_part.alloc(cap_items * part_elems);
                SLT.part = X->chol_part.ptr;
                // the launch's initial matrices once, in the tile layout of the partials (the row kernel then adds them like a
                // partial, 16 values per round trip, instead of 144 dependent loads per row)
                if (!cg_wide) {
                    const bool fullm = (c.mode == CHOL_IMPLICIT);
                    const real_t *M1 = fullm ? c.Minit : c.Mfull, *M2 = fullm ? nullptr : c.Minit;
                    const size_t tl = (size_t)36 * 256;
                    if (M1 != nullptr || (M2 != nullptr && c.kc > 0)) d.tile_init.alloc_at_least(2 * tl);
                    if (M1 != nullptr) {
                        hipLaunchKernelGGL(tile_pack_kernel<real_t>, dim3((unsigned)((tl + 255) / 256)), dim3(256), 0, dev.stream, M1, c.kt, 8, d.tile_init.ptr);
                        SLT.init1 = d.tile_init.ptr;
                    }
                    if (M2 != nullptr && c.kc > 0) {
                        hipLaunchKernelGGL(tile_pack_kernel<real_t>, dim3((unsigned)((tl + 255) / 256)), dim3(256), 0, dev.stream, M2, c.kc, 8, d.tile_init.ptr + tl);
                        SLT.init2 = d.tile_init.ptr + tl;
                    }
                }
                int ctr = 4;
                auto run_batch = [&](int item0, int item1, int row0, int row1) {
                    if (ctr + 2 > 60) ctr = 4;
                    HIP_CHECK(hipMemsetAsync(dev.row_counter.ptr + ctr, 0, 2 * sizeof(int), dev.stream));
                    CholSlices<real_t> SL = SLT;
                    SL.part_base = item0;
                    wave_launch(1, item0, item1, ctr, SL, dev.stream);
                    if (cg_wide) {
                        WideCgParams<real_t> Wp;
                        Wp.part = SL.part; Wp.part_base = item0; Wp.NB = nb_inst; Wp.border = border ? 1 : 0;
                        Wp.row_off = X->row_sl_off.ptr; Wp.n_heavy = n_heavy; Wp.n_slices = nsl;
                        Wp.row_first = row0; Wp.row_last = row1;
                        const size_t smem = gcw_lds_elems<real_t>(c.kt) * sizeof(real_t);
                        auto kern = gram_cg_wide_kernel<real_t>;
                        HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                        const int per_cu = std::max(1, (int)((size_t)160 * 1024 / smem));
                        poison_lds(dev.stream, dev.num_cus);
                        hipLaunchKernelGGL(kern, dim3(std::min(row1 - row0, dev.num_cus * per_cu)), dim3(64 * GCW_NW), smem, dev.stream, *c.cg_wide, Wp);
                        HIP_CHECK(hipGetLastError());
                    } else
                    wave_launch(2, row0, row1, ctr + 1, SL, dev.stream);
                    ctr += 2;
                };
                for (int r = 0; r < n_heavy;) {                 // heavy rows: whole rows per batch
                    int r1 = r + 1;
                    while (r1 < n_heavy && ho[r1 + 1] - ho[r] <= (int)cap_items) r1++;
                    run_batch(ho[r], ho[r1], r, r1);
                    r = r1;
                }
                for (int r = n_heavy; r < total; r += (int)cap_items) {   // the others: item n_slices + i <-> position n_heavy + i
                    const int r1 = (int)std::min<long long>(total, (long long)r + (long long)cap_items);
                    run_batch(nsl + (r - n_heavy), nsl + (r1 - n_heavy), r, r1);
                }
                return 0;
            }
            const bool two = n_heavy > 0 && total > n_heavy;
            if (two) {
                d.ensure_aux();
                HIP_CHECK(hipEventRecord(d.fork_ev, dev.stream));
                HIP_CHECK(hipStreamWaitEvent(d.aux_stream, d.fork_ev, 0));
            }
            hipStream_t heavy_stream = two ? d.aux_stream : dev.stream;
            CholSlices<real_t> none;
            int rc_heavy = 0;
            if (n_heavy > 0 && sliced) {
                CholSlices<real_t> SL = SLT;
                const size_t need = (size_t)X->n_slices * part_elems;
                if (X->chol_part.n < need) X->chol_part.alloc(need);
                SL.part = X->chol_part.ptr;
                wave_launch(1, 0, X->n_slices, 1, SL, heavy_stream);       // partial Gramians of the slices
                wave_launch(2, 0, n_heavy, 3, SL, heavy_stream);           // their rows: sum, factorise, solve
            }
            wave_launch(0, n_heavy, total, 2, none, dev.stream);          // everything up to 1024 entries
            if (n_heavy > 0 && !sliced) {
                CholParams<real_t> H = P;
                H.nrows = n_heavy;
                // (on the stream the call was issued on: the k_t <= 64 variant of the row kernel forks onto the auxiliary one itself)
                rc_heavy = launch_chol_rows(dev, c, X, H, two_src, smem_nonneg);
            } else if (two) {
                HIP_CHECK(hipEventRecord(d.join_ev, d.aux_stream));
                HIP_CHECK(hipStreamWaitEvent(dev.stream, d.join_ev, 0));
            }
            return rc_heavy;
        }
        if (c.cg_wide != nullptr) {
            // the wide CG (launch_cg_wide) exists only as the second kernel of the wave path above: falling through would hand
            // its rows a factorisation without a word -- other numbers than the reference's CG
            g_last_error = "cmfrec_hip: internal: the Gramian CG beyond 64 unknowns needs the wave-per-row producer (weights, nonneg, L1 or "
                           "CMFREC_HIP_CHOL=rows keep it off)";
            return 2;
        }
    }
#ifdef CMFREC_HIP_FLOAT
    {
        // 17-block rows (k_t = 257 .. 272, config 5's item step): the rank-k update by gramk_producer_kernel (four independent
        // wavefronts per row or slice, operands straight from the gather), the partial matrices through HBM, the factorisation
        // and the substitutions by gramk_consumer_kernel (four wavefronts per row, two rows per CU).  CMFREC_HIP_GRAMK=0 keeps
        // the 16-wavefront row kernel with its own LDS-staged rank-k loop (A/B switch and cross-check).
        const bool gk_off = switches().gramk == 0;
        const bool gk_ok = X != nullptr && T == 17 && !two_src && c.koff == 0 && !weighted && !c.rhs_only && c.values_override == nullptr &&
                           (c.mode == CHOL_EXPLICIT || c.mode == CHOL_COLLECTIVE) && !nonneg && !l1on && !gk_off;
        if (gk_ok) {
            DeviceInfo &d = const_cast<DeviceInfo &>(dev);
            const int total = P.nrows;
            const int n_heavy = std::min(X->bin_rows[BIN_VHEAVY], total);
            const int nsl = n_heavy > 0 ? X->n_slices : 0;
            CholSlices<real_t> SLT;
            if (n_heavy > 0) {
                SLT.vrow = X->sl_vrow.ptr; SLT.first = X->sl_first.ptr; SLT.count = X->sl_count.ptr; SLT.row_off = X->row_sl_off.ptr;
            }
            SLT.n_slices = nsl; SLT.n_heavy = n_heavy;
            const std::vector<int> &ho = X->h_row_sl_off;
            int max_row_items = 1;
            for (int r = 0; r < n_heavy; r++) max_row_items = std::max(max_row_items, ho[r + 1] - ho[r]);
            const int batch_env = switches().gramk_batch;
            const int BATCH = batch_env > 0 ? batch_env : 32768;                      // x 158 KB = 5.2 GB of partials
            const size_t cap_items = (size_t)std::min<long long>((long long)nsl + (total - n_heavy), std::max(BATCH, max_row_items));
            // (Tried: two half-size buffers with the consumer of a batch on the second stream beside the producer of the next
            // one -- c5 shard 146.3 against 145.8 ms / iteration in line, profiles/r03: not kept.)
            if (X->chol_part.n < cap_items * GK_PART) X->chol_part.alloc(cap_items * GK_PART);
            // the launch's initial matrices once, in the tile layout of the partials (the consumer adds them like a partial)
            const real_t *init1 = nullptr, *init2 = nullptr;
            {
                const size_t tl = (size_t)GK_NT * 256;
                if (c.Mfull != nullptr || (c.Minit != nullptr && c.kc > 0)) d.tile_init.alloc_at_least(2 * tl);
                if (c.Mfull != nullptr) {
                    hipLaunchKernelGGL(tile_pack_lane_kernel<real_t>, dim3((unsigned)((tl + 255) / 256)), dim3(256), 0, dev.stream, c.Mfull, c.kt, GK_NB,
                                       d.tile_init.ptr);
                    init1 = d.tile_init.ptr;
                }
                if (c.Minit != nullptr && c.kc > 0) {
                    hipLaunchKernelGGL(tile_pack_lane_kernel<real_t>, dim3((unsigned)((tl + 255) / 256)), dim3(256), 0, dev.stream, c.Minit, c.kc, GK_NB,
                                       d.tile_init.ptr + tl);
                    init2 = d.tile_init.ptr + tl;
                }
            }
            int ctr = 4, rc_all = 0;
            // two persistent workgroups per CU fill the register file; a few slots stay open so that the small launches of
            // another stream (the eigen-decomposition chain of the low-rank rows, EigCache) are dispatched beside a batch
            // instead of behind it
            const int gk_open = GK_OPEN_SLOTS;
            const int gk_grid = std::max(2 * dev.num_cus - gk_open, dev.num_cus / 2);
            auto run_batch = [&](int item0, int item1, int row0, int row1) {
                if (ctr + 2 > 60) ctr = 4;
                real_t *part = X->chol_part.ptr;
                HIP_CHECK(hipMemsetAsync(dev.row_counter.ptr + ctr, 0, 2 * sizeof(int), dev.stream));
                CholSlices<real_t> SL = SLT;
                SL.part = part;
                SL.part_base = item0;
                CholParams<real_t> W = P;
                W.row_first = item0; W.nrows = item1; W.counter = dev.row_counter.ptr + ctr;
                if (item1 > item0) {
                    poison_lds(dev.stream, dev.num_cus);
                    hipLaunchKernelGGL(gramk_producer_kernel<real_t>, dim3(std::min(item1 - item0, gk_grid)), dim3(256), 0, dev.stream, W,
                                       X->desc.ptr, SL);
                    HIP_CHECK(hipGetLastError());
                }
                CholParams<real_t> H = P;
                H.row_first = row0; H.nrows = row1; H.counter = dev.row_counter.ptr + ctr + 1;
                H.gk_part = part; H.gk_row_off = X->row_sl_off.ptr; H.gk_n_heavy = n_heavy; H.gk_n_slices = nsl; H.gk_base = item0;
                H.gk_stride = (size_t)GK_PART; H.gk_init1 = init1; H.gk_init2 = init2;
                if (row1 > row0) {
                    poison_lds(dev.stream, dev.num_cus);
                    hipLaunchKernelGGL(gramk_consumer_kernel<real_t>, dim3(std::min(row1 - row0, gk_grid)), dim3(256), 0, dev.stream, H);
                    HIP_CHECK(hipGetLastError());
                }
                ctr += 2;
            };
            for (int r = 0; r < n_heavy;) {                 // split rows: whole rows per batch
                int r1 = r + 1;
                while (r1 < n_heavy && ho[r1 + 1] - ho[r] <= (int)cap_items) r1++;
                run_batch(ho[r], ho[r1], r, r1);
                r = r1;
            }
            for (int r = n_heavy; r < total; r += (int)cap_items) {   // the others: item n_slices + i <-> position n_heavy + i
                const int r1 = (int)std::min<long long>(total, (long long)r + (long long)cap_items);
                run_batch(nsl + (r - n_heavy), nsl + (r1 - n_heavy), r, r1);
            }
            return rc_all;
        }
    }
#endif
    return launch_chol_rows(dev, c, X, P, two_src, smem_nonneg);
}

// the workgroup-per-row kernel over the positions [P.row_first, P.nrows) of the processing order
static int launch_chol_rows(const DeviceInfo &dev, const CholCall &c, const SparseShard *X, CholParams<real_t> P, bool two_src,
                            size_t smem_nonneg)
{
    const int T = chol_tiles(c.kt);
#ifdef CMF_CHOL_DEBUG
    P.dbg = switches().debug_skip;
    if (switches().debug_ticks && T >= 16) {
        // phase timers of the widest kernel: printed (and reset) by the launch that follows, i.e. per half-step
        static unsigned long long *d_ticks = nullptr;
        unsigned long long h[8];
        if (d_ticks == nullptr) { HIP_CHECK(hipMalloc((void **)&d_ticks, sizeof(h))); HIP_CHECK(hipMemset(d_ticks, 0, sizeof(h))); }
        else {
            HIP_CHECK(hipDeviceSynchronize());
            HIP_CHECK(hipMemcpy(h, d_ticks, sizeof(h), hipMemcpyDeviceToHost));
            if (h[5] > 0)
                fprintf(stderr, "chol_rows ticks/row: setup %.0f rank-k|trailing %.0f init %.0f diag+wait %.0f panel %.0f back %.0f  (rows %llu, nnz/row %.0f)\n",
                        (double)h[0] / h[5], (double)h[1] / h[5], (double)h[2] / h[5], (double)h[3] / h[5], (double)h[7] / h[5], (double)h[4] / h[5], h[5],
                        (double)h[6] / h[5]);
            HIP_CHECK(hipMemset(d_ticks, 0, sizeof(h)));
        }
        P.tstamp = d_ticks;
    }
#endif
    // the two-source build (sparse side information) only where it is asked for
#define CHOL_KERN(a, b, c_, d) (two_src ? chol_rows_kernel<real_t, a, b, c_, d, true> : chol_rows_kernel<real_t, a, b, c_, d, false>)
    hipStream_t run_on = dev.stream;
    auto launch = [&](auto kern, int ntt, int nw, int ch, int wgs) {
        size_t smem = std::max(chol_lds_elems<real_t>(ntt, ch) * sizeof(real_t), smem_nonneg);
        int grid = std::min(P.nrows - P.row_first, dev.num_cus * wgs);
        if (grid <= 0) return;
        if (smem > 48 * 1024)
            HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        poison_lds(run_on, dev.num_cus);          // test hook, device.hpp
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), smem, run_on, P);
    };
    // <16-blocks per dimension, wavefronts per workgroup, gathered rows per round, waves per SIMD>
    if (T <= 4) {
        // k_t <= 64: the whole system is 10 tiles.  Rows of 129 non-zeros and more (they lead the processing order) get
        // a four-wave workgroup, the others a two-wave one: with few gathered rows the per-row latency chain dominates
        // and more rows in flight per CU win (C2 A-step 20.9 -> 12.8 ms).  The two launches overlap on two streams, so
        // the light rows fill the CUs while the heaviest rows finish.
        const int total = P.nrows;
        const int nheavy = (X != nullptr) ? std::min(total, X->bin_first[BIN_MED2]) : total;
        DeviceInfo &d = const_cast<DeviceInfo &>(dev);
        const bool two = nheavy > 0 && total > nheavy;
        if (two) {
            d.ensure_aux();
            HIP_CHECK(hipEventRecord(d.fork_ev, dev.stream));
            HIP_CHECK(hipStreamWaitEvent(d.aux_stream, d.fork_ev, 0));
        }
        if (nheavy > 0) {
            P.nrows = nheavy;
            launch(CHOL_KERN(4, 4, 16, 2), 4, 4, 16, 2);
        }
        if (total > nheavy) {
            P.row_first = nheavy; P.nrows = total; P.counter = dev.row_counter.ptr + 1;
            if (two) run_on = d.aux_stream;
            launch(CHOL_KERN(4, 2, 16, 2), 4, 2, 16, 4);
            if (two) {
                HIP_CHECK(hipEventRecord(d.join_ev, d.aux_stream));
                HIP_CHECK(hipStreamWaitEvent(dev.stream, d.join_ev, 0));
            }
        }
    }
    else if (T <= 6) launch(CHOL_KERN(6, 8, 32, 1), 6, 8, 32, 1);
    else if (T <= 9) launch(CHOL_KERN(9, 8, 32, 1), 9, 8, 32, 1);
#ifdef CMFREC_HIP_FLOAT
    else if (T <= 12) launch(CHOL_KERN(12, 8, 32, 1), 12, 8, 32, 1);
    else if (T <= 16) launch(CHOL_KERN(16, 16, 16, 1), 16, 16, 16, 1);
    // k = 256 + bias (BASELINE config 5).  Two workgroups per CU do not pay here: at the 128 VGPRs that allows the
    // kernel spills (c5 share 1.6 -> 4.0 s per A-step; the same on 9 tiles in double: 22.6 -> 25.6 ms).
    else {
        // The rows the producer / consumer pair of launch_chol does not take (weights, non-negativity, L1, a second gather
        // source): 16 wavefronts per row (10 tile slots per wave; the 8-wave build spilled ~500 registers: c5 shard B-step
        // 377 -> 274 ms), 32 gathered rows per round (half the barriers and staging rounds of the rank-k update: 272 -> 219 ms).
        // The build is register-starved either way (128 VGPRs at 16 waves, ~1 KB of scratch per lane).
        launch(CHOL_KERN(17, 16, 32, 1), 17, 16, 32, 1);
    }
#else
    else if (T <= 12) launch(CHOL_KERN(12, 8, 16, 1), 12, 8, 16, 1);
    else launch(CHOL_KERN(16, 8, 16, 1), 16, 8, 16, 1);
#endif
#undef CHOL_KERN
    HIP_CHECK(hipGetLastError());
    return 0;
}

// (launch_gemm: device.hpp)

// X := X (R^T R)^-1 for the row-major [rows, k] block X (ld = ldx) and the row-major upper Cholesky factor R [k, k] of a
// shared matrix: the multi-right-hand-side posv of optimizeA Case 3 (common.c:3171-3175).  Round 4: no library call -- the
// inverse of the shared matrix once (L^-1 = R^-T by one workgroup, column per thread; M^-1 = L^-T L^-1 by the library's own
// GEMM), then the rows times it as one tall GEMM on the matrix cores (rounds 1-3: two rocBLAS trsm over all rows).
static void launch_potrs_rows(const DeviceInfo &dev, int rows, int k, const real_t *R, real_t *X, size_t ldx)
{
    if (rows <= 0 || k <= 0) return;
    DeviceInfo &d = const_cast<DeviceInfo &>(dev);
    d.potrs_inv.alloc_at_least((size_t)2 * k * k);
    d.potrs_tmp.alloc_at_least((size_t)rows * k);
    real_t *Linv = d.potrs_inv.ptr, *Minv = d.potrs_inv.ptr + (size_t)k * k;
    const size_t tr_smem = (size_t)2 * k * (k + 1) * sizeof(real_t);
    if (tr_smem <= 96 * 1024) {
        auto kern = trtri_from_upper_kernel<real_t, true>;
        if (tr_smem > 48 * 1024) HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tr_smem));
        hipLaunchKernelGGL(kern, dim3(1), dim3(256), tr_smem, dev.stream, R, k, Linv);
    } else {
        hipLaunchKernelGGL((trtri_from_upper_kernel<real_t, false>), dim3(1), dim3(256), 0, dev.stream, R, k, Linv);
    }
    HIP_CHECK(hipGetLastError());
    launch_gemm<true>(dev, k, k, k, (real_t)1, Linv, (size_t)k, Linv, (size_t)k, Minv, (size_t)k);
    launch_gemm<false>(dev, rows, k, k, (real_t)1, X, ldx, Minv, (size_t)k, d.potrs_tmp.ptr, (size_t)k);
    if (sizeof(real_t) == 4) {
        // Single precision: one step of iterative refinement with the matrix itself.  The product with an explicit inverse carries
        // an error of cond(M) eps where the two triangular solves of the reference's posv are backward stable; x1 = x0 +
        // (b - x0 M) M^-1 brings the residual back to the level of the solves (the right-hand sides are still in X here).
        // M = R^T R from the factor; three tall products instead of one, on rows x k x k flops -- nothing beside the half-step's gather.
        d.potrs_ref.alloc_at_least((size_t)k * k + (size_t)rows * k);
        real_t *Mfull = d.potrs_ref.ptr, *Res = d.potrs_ref.ptr + (size_t)k * k;
        hipLaunchKernelGGL(upper_only_kernel<real_t>, grid1d((size_t)k * k), dim3(256), 0, dev.stream, R, k, Linv);       // (Linv is free again)
        launch_gemm<true>(dev, k, k, k, (real_t)1, Linv, (size_t)k, Linv, (size_t)k, Mfull, (size_t)k);                    // R^T R
        launch_gemm<false>(dev, rows, k, k, (real_t)1, d.potrs_tmp.ptr, (size_t)k, Mfull, (size_t)k, Res, (size_t)k);      // x0 M
        hipLaunchKernelGGL(residual_rows_kernel<real_t>, grid1d((size_t)rows * k), dim3(256), 0, dev.stream, Res, (size_t)k, X, ldx, (size_t)rows, k);   // b - x0 M
        launch_gemm<false>(dev, rows, k, k, (real_t)1, Res, (size_t)k, Minv, (size_t)k, X, ldx);                           // (b - x0 M) M^-1 -> X
        hipLaunchKernelGGL(add_rows_kernel<real_t>, grid1d((size_t)rows * k), dim3(256), 0, dev.stream, X, ldx, d.potrs_tmp.ptr, (size_t)k, (size_t)rows, k);
        HIP_CHECK(hipGetLastError());
        return;
    }
    HIP_CHECK(hipMemcpy2DAsync(X, ldx * sizeof(real_t), d.potrs_tmp.ptr, (size_t)k * sizeof(real_t), (size_t)k * sizeof(real_t), (size_t)rows,
                               hipMemcpyDeviceToDevice, dev.stream));
}

// CG of the explicit model beyond 64 unknowns (k_t <= 129): through the row's Gramian (gram_cg_wide_kernels.hpp) instead of the
// lane <-> unknown kernel.  -1: not applicable (the caller takes launch_cg).  CMFREC_HIP_CG_KERNEL=generic keeps the generic kernel
// (A/B switch and on-device cross-check).
static int launch_cg_wide(const DeviceInfo &dev, const CgCall &c, const SparseShard &X)
{
    const bool border = (c.k > 16) && ((c.k - 1) % 16 == 0);
    const int nbw = (c.k - (border ? 1 : 0) + 15) / 16;
    if (c.implicit || c.k <= 64 || nbw > 8 || c.precond || c.X2 != nullptr || c.Bi != nullptr || c.koff != 0 || X.weighted() ||
        dev.nonneg_now || dev.l1_now != (real_t)0 || dev.l1_last_now != (real_t)0 ||      // (launch_chol's wave path is off then)
        cg_variant_from_env() == CgVariant::Generic || switches().chol == 1 ||
        (c.kc > 0 && (c.CtC == nullptr || c.UC == nullptr || c.kc > c.k)) || gcw_lds_elems<real_t>(c.k) * sizeof(real_t) > (size_t)160 * 1024)
        return -1;
    CgParams<real_t> P;
    P.A = c.A; P.lda = c.lda; P.B = c.B; P.ldb = c.ldb; P.k = c.k;
    P.indptr = X.p.ptr; P.indices = X.i.ptr; P.values = X.v.ptr;
    P.bias_sub = c.bias_sub; P.order = X.order.ptr; P.desc = X.desc.ptr; P.nrows = 0; P.BtB = nullptr;
    P.lam = c.lam; P.lam_last = c.lam_last;
    P.scale_lam = c.scale_lam; P.scale_bias_const = c.scale_bias_const;
    P.max_cg_steps = c.max_cg_steps; P.precond = 0;
    P.koff = 0; P.kc = c.kc; P.CtC = c.CtC; P.UC = c.UC; P.w_side = c.w_side;
    P.rows_with_u = c.rows_with_u; P.p_side = c.p_side; P.scale_lam_sideinfo = c.scale_lam_sideinfo ? 1 : 0;
    // The Gramian costs k_t / 8 times the products of the CG's four passes per entry and saves three of the four gathers and the
    // generic kernel's wave-wide reduction per gathered row: it pays on the long rows (config 3's items under CG: 18.1 -> 6.8 ms)
    // and loses on the short ones (its users, 143 entries per row on average: 8.6 -> 15.8 ms).  So the rows beyond 256 entries --
    // they lead the processing order -- take it, the others stay on the lane <-> unknown kernel (CgCall::skip_first).
    const int n_wide = X.rows_longer_than(256, X.n_nonempty);
    if (n_wide <= 0) return -1;
    CholCall cc{c.A, c.lda, c.B, c.ldb, c.k, 0, c.bias_sub, nullptr, 0, 0, 0, c.lam, c.lam_last, c.scale_lam, c.scale_lam_sideinfo,
                c.scale_bias_const, CHOL_EXPLICIT};
    cc.cg_wide = &P;
    cc.row_limit = n_wide;
    int rc = launch_chol(dev, cc, &X);                    // positions [0, n_wide)
    if (rc) return rc;
    CgCall rest = c;
    rest.skip_first = n_wide;
    return launch_cg(dev, rest, X, nullptr);              // the shorter rows, and a block system's rows without entries
}
static int launch_cg_any(const DeviceInfo &dev, const CgCall &c, const SparseShard &X, BinTimers *tm = nullptr)
{
    const int rc = launch_cg_wide(dev, c, X);
    return rc >= 0 ? rc : launch_cg(dev, c, X, tm);
}

// Eigenvectors / values of one side's shared matrix w C^T C (eig_kernels.hpp: tridiagonalisation + implicit QL, two launches
// of 1 and ceil(k_c / 64) workgroups -- a few milliseconds of mostly sequential work on a handful of CUs), so
//  * inside a half-step it is enqueued BEHIND the launches of the rows that do not need it (the producer / consumer batches of
//    the long rows), on the eigen stream, waiting only for an event recorded in front of them;
//  * a half-step also enqueues the decomposition the NEXT half-step of the other side will ask for (`wanted`: that side took
//    the low-rank path before), since C and D are final once their own updates have run (cmfrec's order C, D, B, A):
//    `fresh` says the cached vectors belong to the side matrix as it stands; every update / upload of C or D clears it.
// (Rounds 3-5 bound rocSOLVER's dsyevd here -- ~4000 launches per decomposition, 6 ms on an idle device and 33 ms beside the
//  persistent batches, profiles/r05/r05_zg_*; round 6 replaced it and the link against rocBLAS went with it.)
struct EigCache {
    DevBuf<double> W, V, D, E, Tau;
    DevBuf<int> info;
    DevBuf<real_t> Q, Qt, Lam, M;     // M: the matrix a prefetch decomposes (the half-step's own one lives in the session's ctc)
    hipEvent_t ev = nullptr;          // recorded behind the chain on the eigen stream
    bool fresh = false, wanted = false;
    int kind = 0;                     // 3 tridiagonalisation + QL (eig_kernels.hpp), 2 the one-workgroup Jacobi kernel
    int checked_kc = 0;               // the QL kernel's status word has been read back for this matrix size (once per size and cache)
    ~EigCache() { if (ev) (void)hipEventDestroy(ev); }
};

struct LowRankScratch {
    DevBuf<real_t> Ct, Bt, R, T;
    EigCache own;                 // callers without a cache per side (the stand-alone operator)
    int last_rows = 0;            // rows the low-rank kernels solved in the most recent launch (0: path not taken)
    int last_eig = 0;             // its eigen-decomposition: 3 tridiagonalisation + QL, 2 the one-workgroup Jacobi kernel
};

// Enqueues the decomposition of the kc x kc matrix Minit on the eigen stream, behind `after` (an event of the main stream:
// what produced Minit), and records E.ev behind it.
static void issue_eig(DeviceInfo &d, EigCache &E, const real_t *Minit, int kc, hipEvent_t after)
{
    E.W.alloc_at_least((size_t)kc * kc); E.V.alloc_at_least((size_t)kc * kc);
    E.Q.alloc_at_least((size_t)kc * kc); E.Qt.alloc_at_least((size_t)kc * kc); E.Lam.alloc_at_least((size_t)kc);
    if (!E.ev) HIP_CHECK(hipEventCreateWithFlags(&E.ev, hipEventDisableTiming));
    HIP_CHECK(hipStreamWaitEvent(d.eig_stream(), after, 0));
    // default: Householder tridiagonalisation + implicit QL with the rotations applied row-parallel (eig_kernels.hpp);
    // CMFREC_HIP_EIG=jacobi takes the one-workgroup Jacobi kernel (on-device cross-check; also the fallback should a QL iteration
    // ever fail to converge: the status word is read back the first time a matrix of this size goes through a cache)
    static bool ql_bad = false;
    bool done = false;
    if (!ql_bad && !switches().eig_jacobi && kc <= EIG_MAX_N) {
        E.D.alloc_at_least((size_t)kc); E.E.alloc_at_least((size_t)kc); E.Tau.alloc_at_least((size_t)kc);
        if (E.info.n < 1) { E.info.alloc(1); HIP_CHECK(hipMemsetAsync(E.info.ptr, 0, sizeof(int), d.eig_stream())); }
        hipLaunchKernelGGL(eig_tridiag_kernel<real_t>, dim3(1), dim3(EIG_TRIDIAG_THREADS), 0, d.eig_stream(), Minit, kc, E.W.ptr, E.D.ptr, E.E.ptr, E.Tau.ptr);
        const bool wide = kc > 288;                          // rows of the eigenvector matrix per workgroup: 64, or 32 (LDS: kc rows doubles)
        const int rows = wide ? 32 : 64;
        const size_t smem = ((size_t)kc * rows + 4 * (size_t)kc) * sizeof(double);
        auto kern = wide ? eig_ql_rows_kernel<real_t, 32> : eig_ql_rows_kernel<real_t, 64>;
        static thread_local bool attr_set[MAX_DEVICES][2] = {{false}};
        bool &as = attr_set[std::min(std::max(d.device, 0), MAX_DEVICES - 1)][wide ? 1 : 0];
        if (!as) { HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); as = true; }
        hipLaunchKernelGGL(kern, dim3((kc + rows - 1) / rows), dim3(64), smem, d.eig_stream(), kc, E.W.ptr, E.D.ptr, E.E.ptr, E.Tau.ptr, E.Q.ptr,
                           E.Qt.ptr, (size_t)kc, E.Lam.ptr, E.info.ptr);
        HIP_CHECK(hipGetLastError());
        done = true;
        if (E.checked_kc != kc) {
            int h_info = 0;
            HIP_CHECK(hipMemcpyAsync(&h_info, E.info.ptr, sizeof(int), hipMemcpyDeviceToHost, d.eig_stream()));
            HIP_CHECK(hipStreamSynchronize(d.eig_stream()));
            E.checked_kc = kc;
            if (h_info != 0) { ql_bad = true; done = false; }
        }
    }
    if (!done)
        hipLaunchKernelGGL(jacobi_eig_kernel<real_t>, dim3(1), dim3(1024), 0, d.eig_stream(), Minit, kc, E.W.ptr, E.V.ptr, E.Q.ptr, E.Qt.ptr,
                           (size_t)kc, E.Lam.ptr, 30, sizeof(real_t) == 4 ? 1e-9 : 1e-13);
    E.kind = done ? 3 : 2;
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(E.ev, d.eig_stream()));
}

// Collective Cholesky half-step (mode CHOL_COLLECTIVE, dense side information on every row of the block) with the rows of
// few entries solved by the low-rank update of a diagonalised shared matrix instead of a k_t^3 / 3 factorisation per row
// (config 5's users: 20 entries against k_t = 257 unknowns).  c: the call as launch_chol would take it (right-hand sides
// w U C already in the rows, c.Minit = w C^T C); Um: the block's rows of U; k: the factors shared with X; rows_b: rows of
// the opposing matrix.  Returns -1 when the path does not apply (the caller launches as usual), else the return code.
// CMFREC_HIP_LOWRANK=0 switches it off, =1 forces it whenever the shapes allow (tests).
// mine: this side's cache (NULL: the scratch's own, never fresh); next / kc_next: the other side's cache with its matrix in
// next->M, decomposed behind this one for the half-step that follows (NULL: nothing to prefetch).
static int launch_collective_lowrank(const DeviceInfo &dev, LowRankScratch &S, CholCall c, const SparseShard &X, const real_t *Cm,
                                     const real_t *Um, int p_self, real_t w, int k, int rows_b, EigCache *mine = nullptr,
                                     EigCache *next = nullptr, int kc_next = 0)
{
    const int lr_sw = switches().lowrank;       // CMFREC_HIP_LOWRANK: 0 / 1 force the path off / on
    const int kt = c.kt, kc = c.kc, k_side_self = c.koff;
    // rows of up to lr_max entries: the s x s system is at most half the k_t x k_t one (and 8 blocks in single, 4 in double)
    const int lr_type_max = (sizeof(real_t) == 4) ? 128 : 64;
    const int lr_max = (kt >= 256 && lr_type_max >= 128) ? 128 : (kt >= 128 ? 64 : 32);
    const int n_full = X.rows_longer_than(lr_max, X.nrows);          // positions [0, n_full): full factorisation
    const int n_light = X.nrows - n_full;
    const bool lr_ok = lr_sw != 0 && c.mode == CHOL_COLLECTIVE && c.Mfull == nullptr && c.X2 == nullptr &&
                       !X.weighted() &&
                       !c.scale_bias_const && !dev.nonneg_now && !c.nonneg && dev.l1_now == (real_t)0 && dev.l1_last_now == (real_t)0 &&
                       c.rows_with_u >= X.nrows && !c.rhs_prefilled_all && kt <= 320 && kt >= 64 && kc > 0 && p_self > 0 &&
                       // the penalty must be a multiple of the identity on the rotated block: the last unknown's own lambda
                       // (a bias) has to sit outside of it
                       (kt - 1 >= kc || c.lam_last == c.lam) &&
                       (lr_sw == 1 || (n_light >= 32768 && kt >= 96));
    S.last_rows = 0; S.last_eig = 0;
    if (!lr_ok || n_light <= 0) return -1;
    hipStream_t st = dev.stream;
    const int ngr = (kt + 15) / 16;
    const size_t ldbt = (size_t)16 * ngr;
    S.Ct.alloc_at_least((size_t)p_self * kc);
    S.Bt.alloc_at_least((size_t)rows_b * ldbt + 64);
    S.R.alloc_at_least((size_t)X.nrows * kc);
    S.T.alloc_at_least((size_t)n_light * kc);
    EigCache &E = mine ? *mine : S.own;
    if (mine) mine->wanted = true;
    // w C^T C = Q L Q^T on the eigen stream, beside the full factorisations of the longer rows (which do not need it and are
    // enqueued first: see EigCache)
    {
        DeviceInfo &d = const_cast<DeviceInfo &>(dev);
        d.ensure_aux();
        const bool have = mine != nullptr && mine->fresh && mine->ev != nullptr;
        const bool ahead = next != nullptr && kc_next > 0;
        if (!have || ahead) {
            if (!d.eig_fork) HIP_CHECK(hipEventCreateWithFlags(&d.eig_fork, hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(d.eig_fork, st));
        }
        c.row_limit = n_full;
        const int rc = (n_full > 0) ? launch_chol(dev, c, &X) : 0;
        if (rc) return rc;
        if (!have) {
            issue_eig(d, E, c.Minit, kc, d.eig_fork);
            if (mine) mine->fresh = true;          // (c.Minit is the session's w C^T C of this side as it stands)
        }
        if (ahead) {
            issue_eig(d, *next, next->M.ptr, kc_next, d.eig_fork);
            next->fresh = true;
        }
        S.last_eig = E.kind;
        S.last_rows = n_light;
        HIP_CHECK(hipStreamWaitEvent(st, E.ev, 0));
    }
    // C~ = C Q ;  B~ = [B(:, :k) Q(k_side:, :) | B(:, k:)] ;  R = w U C~ (natural row order)
    launch_gemm<false>(dev, p_self, kc, kc, (real_t)1, Cm, (size_t)kc, E.Q.ptr, (size_t)kc, S.Ct.ptr, (size_t)kc);
    launch_gemm<false>(dev, rows_b, kc, k, (real_t)1, c.B, c.ldb, E.Q.ptr + (size_t)k_side_self * kc, (size_t)kc, S.Bt.ptr, ldbt);
    if (kt > kc)
        hipLaunchKernelGGL(copy_cols_kernel<real_t>, grid1d((size_t)rows_b * (kt - kc)), dim3(256), 0, st, S.Bt.ptr, ldbt, kc, c.B, c.ldb, k,
                           kt - kc, (size_t)rows_b);
    launch_gemm<false>(dev, X.nrows, kc, p_self, w, Um, (size_t)p_self, S.Ct.ptr, (size_t)kc, S.R.ptr, (size_t)kc);
    LrParams<real_t> L;
    L.A = c.A; L.lda = c.lda; L.pre = S.R.ptr; L.ldpre = (size_t)kc;
    L.Tc = S.T.ptr; L.ldt = (size_t)kc; L.pos0 = n_full;
    L.Bt = S.Bt.ptr; L.ldbt = ldbt; L.lam_eig = E.Lam.ptr;
    L.kt = kt; L.kc = kc; L.koff = k_side_self; L.rotated = 1;
    L.indptr = X.p.ptr; L.indices = X.i.ptr; L.values = X.v.ptr; L.bias_sub = c.bias_sub;
    L.lam = c.lam; L.lam_last = c.lam_last;
    L.scale_lam = c.scale_lam ? 1 : 0; L.scale_lam_sideinfo = c.scale_lam_sideinfo ? 1 : 0;
    L.scale_bias_const = 0; L.p_side = p_self; L.collective = 1;
    if (dev.row_counter.n < ROW_COUNTER_INTS) const_cast<DeviceInfo &>(dev).row_counter.alloc(ROW_COUNTER_INTS);
    HIP_CHECK(hipMemsetAsync(dev.row_counter.ptr + 8, 0, 4 * sizeof(int), st));
    // rows sorted by length: (64, 128] entries -> 8 blocks (single precision only), (32, 64] -> 4, the rest -> 2
    const int n_gt64 = X.rows_longer_than(64, X.nrows), n_gt32 = X.rows_longer_than(32, X.nrows);
    auto lr_launch = [&](auto kern, int nb_, int wps, int first, int last, int counter) {
        if (last <= first) return;
        LrParams<real_t> Lc = L;
        Lc.row_first = first; Lc.nrows = last; Lc.counter = dev.row_counter.ptr + 8 + counter;
        const size_t smem = 4 * lowrank_lds_elems<real_t>(nb_) * sizeof(real_t);
        if (smem > 48 * 1024)
            HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const int grid = std::min((last - first + 3) / 4, dev.num_cus * wps);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, Lc, X.desc.ptr);
    };
    // (two wavefronts per SIMD at least: a single wave cannot keep the matrix pipe issuing, see gramk_kernels.hpp)
#ifdef CMFREC_HIP_FLOAT
    lr_launch(lowrank_rows_kernel<real_t, 8, 2>, 8, 2, n_full, std::max(n_full, n_gt64), 0);
    lr_launch(lowrank_rows_kernel<real_t, 4, 3>, 4, 3, std::max(n_full, n_gt64), std::max(n_full, n_gt32), 1);
#else
    lr_launch(lowrank_rows_kernel<real_t, 4, 2>, 4, 2, std::max(n_full, n_gt64), std::max(n_full, n_gt32), 1);
#endif
#ifdef CMFREC_HIP_FLOAT
    lr_launch(lowrank_rows_kernel<real_t, 2, 4>, 2, 4, std::max(n_full, n_gt32), X.nrows, 2);
#else
    lr_launch(lowrank_rows_kernel<real_t, 2, 3>, 2, 3, std::max(n_full, n_gt32), X.nrows, 2);
#endif
    HIP_CHECK(hipGetLastError());
    // x[:kc] = Q x~[:kc]: one GEMM over the light rows (in processing order), then back to their rows
    launch_gemm<false>(dev, n_light, kc, kc, (real_t)1, S.T.ptr, (size_t)kc, E.Qt.ptr, (size_t)kc, S.R.ptr, (size_t)kc);
    hipLaunchKernelGGL(scatter_rows_kernel<real_t>, grid1d((size_t)n_light * kc), dim3(256), 0, st, c.A, c.lda, X.desc.ptr, n_full, S.R.ptr,
                       (size_t)kc, kc, (size_t)n_light);
    HIP_CHECK(hipGetLastError());
    return 0;
}

// Plain closed-form rows (factors_closed_form, common.c:978-1070: no side information) with few entries against many unknowns: the
// row's matrix is lam_i I + sum_j B_j B_j^T -- already diagonal plus a rank-s update, so the low-rank kernel applies without any
// rotation (lowrank_kernels.hpp, !rotated / !collective): an s x s system instead of a k_t^3 / 3 factorisation.  Config 3's users
// (k_t = 129, double): 38 % of the rows hold at most 64 entries.  -1 when the path does not apply.
#ifndef CMF_LR_MAX_F64
#define CMF_LR_MAX_F64 96      // 64: the four-block build only (the A/B build of round 5)
#endif
static int launch_plain_lowrank(const DeviceInfo &dev, const CholCall &c, const SparseShard &X)
{
    const int lr_sw = switches().lowrank;       // CMFREC_HIP_LOWRANK: 0 / 1 force the path off / on
    const int kt = c.kt;
    // (double precision, k_t >= 128: rows of 65..96 entries on a six-block build, 21 tiles at one wavefront per SIMD -- the full
    //  path costs such a row the 36-tile rank-k update, a 133 KB round trip through HBM and the eight-block factorisation)
    const int lr_type_max = (sizeof(real_t) == 4) ? 128 : 64;
    const int lr_max = (kt >= 256 && lr_type_max >= 128) ? 128 : (kt >= 128 ? (sizeof(real_t) == 8 ? CMF_LR_MAX_F64 : 64) : 32);
    const int n_rows = X.n_nonempty;                                  // rows without entries stay as they are (common.c:3270)
    const int n_full = X.rows_longer_than(lr_max, n_rows);
    const int n_light = n_rows - n_full;
    const bool ok = lr_sw != 0 && !X.weighted() && c.koff == 0 && c.X2 == nullptr && !c.rhs_only &&
                    c.values_override == nullptr && !c.nonneg && !dev.nonneg_now && dev.l1_now == (real_t)0 && dev.l1_last_now == (real_t)0 &&
                    kt <= 320 && (lr_sw == 1 ? kt >= 40 : (kt >= 96 && n_light >= 2048));
    if (!ok || n_light <= 0) return -1;
    CholCall cf = c;
    cf.row_limit = n_full;
    const int rc = (n_full > 0) ? launch_chol(dev, cf, &X) : 0;
    if (rc) return rc;
    hipStream_t st = dev.stream;
    LrParams<real_t> L;
    L.A = c.A; L.lda = c.lda; L.pre = nullptr; L.ldpre = 0; L.Tc = nullptr; L.ldt = 0; L.pos0 = n_full;
    L.Bt = c.B; L.ldbt = c.ldb; L.lam_eig = nullptr;
    L.kt = kt; L.kc = 0; L.koff = 0; L.rotated = 0;
    L.indptr = X.p.ptr; L.indices = X.i.ptr; L.values = X.v.ptr; L.bias_sub = c.bias_sub;
    L.lam = c.lam; L.lam_last = c.lam_last;
    L.scale_lam = c.scale_lam ? 1 : 0; L.scale_lam_sideinfo = 0; L.scale_bias_const = c.scale_bias_const ? 1 : 0; L.p_side = 0; L.collective = 0;
    if (dev.row_counter.n < ROW_COUNTER_INTS) const_cast<DeviceInfo &>(dev).row_counter.alloc(ROW_COUNTER_INTS);
    HIP_CHECK(hipMemsetAsync(dev.row_counter.ptr + 8, 0, 4 * sizeof(int), st));
    const int n_gt64 = X.rows_longer_than(64, n_rows), n_gt32 = X.rows_longer_than(32, n_rows);
    auto lr_launch = [&](auto kern, int nb_, int wps, int first, int last, int counter) {
        if (last <= first) return;
        LrParams<real_t> Lc = L;
        Lc.row_first = first; Lc.nrows = last; Lc.counter = dev.row_counter.ptr + 8 + counter;
        const size_t smem = 4 * lowrank_lds_elems<real_t>(nb_) * sizeof(real_t);
        if (smem > 48 * 1024)
            HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const int grid = std::min((last - first + 3) / 4, dev.num_cus * wps);
        poison_lds(st, dev.num_cus);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, Lc, X.desc.ptr);
    };
#ifdef CMFREC_HIP_FLOAT
    lr_launch(lowrank_rows_kernel<real_t, 8, 2>, 8, 2, n_full, std::max(n_full, n_gt64), 0);
    lr_launch(lowrank_rows_kernel<real_t, 4, 3>, 4, 3, std::max(n_full, n_gt64), std::max(n_full, n_gt32), 1);
    lr_launch(lowrank_rows_kernel<real_t, 2, 4>, 2, 4, std::max(n_full, n_gt32), n_rows, 2);
#else
    if (lr_max > 64) lr_launch(lowrank_rows_kernel<real_t, 6, 1>, 6, 1, n_full, std::max(n_full, n_gt64), 0);
    lr_launch(lowrank_rows_kernel<real_t, 4, 2>, 4, 2, std::max(n_full, n_gt64), std::max(n_full, n_gt32), 1);
    lr_launch(lowrank_rows_kernel<real_t, 2, 3>, 2, 3, std::max(n_full, n_gt32), n_rows, 2);
#endif
    HIP_CHECK(hipGetLastError());
    return 0;
}

static void init_device(DeviceInfo &dev, int device)
{
    switches_mut().reload();           // the environment switches, once per session / operator call (device.hpp, Switches)
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        g_last_error = "cmfrec_hip: no HIP device available (this library has no CPU fallback)";
        fprintf(stderr, "%s\n", g_last_error.c_str());
        throw HipError{4};
    }
    if (device >= 0) HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipGetDevice(&dev.device));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, dev.device));
    dev.num_cus = prop.multiProcessorCount;
    HIP_CHECK(hipStreamCreateWithFlags(&dev.stream, hipStreamNonBlocking));
}

}  // namespace cmfhip

using namespace cmfhip;

struct cmfrec_hip_session {
    cmfrec_hip_model mdl;
    DeviceInfo dev;
    int k_totA = 0, k_totB = 0, has_bias = 0;
    size_t ldA = 0, ldB = 0;
    DevBuf<real_t> A, B, biasA, biasB, C, D, U, II;
    SparseShard Xr, Xc;
    // sparse side information (missing = absent): CSR by user / item for the factor updates, CSC by attribute for C / D
    SparseShard Usr, Usc, Isr, Isc;
    bool sparseU = false, sparseI = false;
    // non-negativity constraints (solve_nonneg instead of the Cholesky solve; they switch the CG off for that matrix)
    bool nonneg = false, nonneg_C = false, nonneg_D = false;
    int max_cd_steps = 100;
    // implicit features of the explicit model (add_implicit_features): Ai [m, k+k_main], Bi [n, k+k_main] factorise the
    // binary "was observed" matrix alongside X (collective.c:8448-8534); Cholesky-type solves only
    bool implicit_feats = false;
    real_t w_implicit = 1;
    DevBuf<real_t> Ai, Bi, bitbi, bitbi_full, ones;
    // segment tables of the two-stage right-hand-side gather of the Ai / Bi updates ([0]: rows of X, [1]: columns)
    struct GatherSegs { DevBuf<int> seg_row, seg_off, row_first; int nseg = 0; } gsegs[2];
    DevBuf<real_t> gpartial, grhs;
    // per-matrix penalties, the reference's lam_unique / l1_lam_unique order (collective.c:430): user bias, item bias, A, B,
    // C, D -- after the w_main rescaling.  Scalar lam / l1_lam fill all six.
    real_t lam6[6] = {0, 0, 0, 0, 0, 0}, l16[6] = {0, 0, 0, 0, 0, 0};
    bool scale_bias_const = false;   // explicit model: the bias' lambda is not scaled row by row (lam6[0], [1] carry a constant factor)
    // NA_as_zero for the main matrix (explicit model without side information): absent entries of X are zeros.  Every
    // half-step is optimizeA Case 3 (common.c:3118-3205): one shared matrix, the stored values uncentred, and the constant
    // -sum over all opposing rows of (their bias + naz_mean) x row on every right-hand side (collective.c:8573-8600, :8756-8787)
    bool naz_X = false, naz_center = false;
    real_t x_subtract = 0;        // what the most recent set_X_coo* subtracted from the values (cmfrec_hip_session_set_NA_as_zero_X checks it)
    real_t naz_mean = 0;
    DevBuf<real_t> naz_part, naz_vec, naz_M, naz_rhs;
    // ... with observation weights (update_factor_naz_weighted): per entry w - 1 and the right-hand-side bracket, a zero array in
    // the place of the values for the CG kernels
    DevBuf<real_t> naz_g, naz_xt, naz_zero;
    DevBuf<real_t> naz_mult;            // NA_as_zero_X with sparse side information: the rows' lambda multiplier (every cell counts)
    DevBuf<int> zrowsA, zrowsB;   // rows every update of A / B leaves at zero (cmfrec_hip_session_set_zero_rows)
    int n_zrowsA = 0, n_zrowsB = 0;
    DevBuf<unsigned char> cfmaskA, cfmaskB;   // rows that take the closed form inside a CG update (cmfrec_hip_session_set_closed_form_rows)
    // ... and the attributes of dense side information with NaN (C / D updates): 1 = closed form, 2 = CG from zero with k_side + k steps
    DevBuf<unsigned char> cfmaskC, cfmaskD;
    bool cfC_any[3] = {false, false, false}, cfD_any[3] = {false, false, false};   // which mask values occur
    bool has_cfA = false, has_cfB = false;
    DevBuf<real_t> cf_keep;
    real_t l1_lam = 0;              // L1 penalty (after the w_main rescaling); C / D use l1_lam / w_user, / w_item
    // optional split of the local rows of A into contiguous parts, each with its own processing order: an A-step then
    // finishes part by part (one event each), so the all-gather of a finished part overlaps the rest of the step
    std::vector<std::unique_ptr<SparseShard>> XrParts;
    std::vector<int> partBegin;          // nparts + 1 local row offsets
    std::vector<hipEvent_t> partEv;
    DevBuf<real_t> gram, ctc, betbe, ucA, ucB;
    // low-rank row updates (lowrank_kernels.hpp): eigenvectors / values of w C^T C, rotated C and opposing factors,
    // rotated right-hand sides and solutions
    LowRankScratch lr;
    EigCache eigA, eigB;          // eigenvectors of w_user C^T C (A-steps) / w_item D^T D (B-steps), see EigCache
    // row-block shards of the explicit / collective model (distributed.py): U / I hold only the rows of this
    // rank's block, and the C / D update is split into partial sums over the local rows (side_part: [kc x kc | p x kc]),
    // an all-reduce by the caller, and the identical small solve on every rank
    bool side_local = false;
    DevBuf<real_t> side_part;
    GramWorkspace gws;
    std::vector<EventPair> evA, evB;     // whole half-steps
    BinTimers binA, binB;                // row-update kernel launches per nnz bin

    hipEvent_t new_event()
    {
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        return e;
    }
    ~cmfrec_hip_session()
    {
        for (auto &p : evA) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        for (auto &p : evB) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        binA.clear(); binB.clear();
        for (auto e : partEv) (void)hipEventDestroy(e);
    }
};

static void trim_events(std::vector<EventPair> &v)
{
    constexpr size_t CAP = 2048;
    if (v.size() < CAP) return;
    for (size_t e = 0; e < CAP / 2; e++) { (void)hipEventDestroy(v[e].a); (void)hipEventDestroy(v[e].b); }
    v.erase(v.begin(), v.begin() + CAP / 2);
}

static int guarded(const std::function<int()> &f)
{
    try {
        return f();
    } catch (const HipError &e) {
        return e.code;
    } catch (const std::bad_alloc &) {
        g_last_error = "cmfrec_hip: host out of memory";
        return 1;
    }
}

extern "C" {

int cmfrec_hip_sizeof_real(void) { return (int)sizeof(real_t); }
int cmfrec_hip_sizeof_model(void) { return (int)sizeof(cmfrec_hip_model); }
const char *cmfrec_hip_build_info(void)
{
#ifdef CMFREC_HIP_FLOAT
    return "cmfrec_hip float gfx950";
#else
    return "cmfrec_hip double gfx950";
#endif
}
const char *cmfrec_hip_last_error(void) { return g_last_error.c_str(); }
void cmfrec_hip_reload_switches(void) { switches_mut().reload(); }

cmfrec_hip_session *cmfrec_hip_session_create(const cmfrec_hip_model *model, int device)
{
    cmfrec_hip_session *s = nullptr;
    int rc = guarded([&]() {
        const cmfrec_hip_model &m = *model;
        if (m.m <= 0 || m.n <= 0 || m.k < 0 || m.row_begin < 0 || m.row_end > m.m || m.col_begin < 0 ||
            m.col_end > m.n || m.row_begin > m.row_end || m.col_begin > m.col_end) {
            g_last_error = "cmfrec_hip: invalid model sizes";
            return 2;
        }
        if (m.implicit && (m.user_bias || m.item_bias)) {
            g_last_error = "cmfrec_hip: the implicit model has no biases";
            return 2;
        }
        if ((m.k_user && !m.p) || (m.k_item && !m.q)) {
            g_last_error = "cmfrec_hip: k_user / k_item need side information";
            return 2;
        }
        if ((m.p > 0 && m.m_u > m.m) || (m.q > 0 && m.n_i > m.n) || m.m_x > m.m || m.n_x > m.n) {
            g_last_error = "cmfrec_hip: m / n must be the rows of A / B: max(m_x, m_u) and max(n_x, n_i)";
            return 2;
        }
        s = new cmfrec_hip_session();
        s->mdl = m;
        init_device(s->dev, device);
        s->k_totA = m.k_user + m.k + m.k_main;
        s->k_totB = m.k_item + m.k + m.k_main;
        s->has_bias = (m.user_bias || m.item_bias) ? 1 : 0;
        s->ldA = (size_t)(s->k_totA + s->has_bias);
        s->ldB = (size_t)(s->k_totB + s->has_bias);
        s->A.alloc((size_t)m.m * s->ldA);
        s->B.alloc((size_t)m.n * s->ldB);
        HIP_CHECK(hipMemsetAsync(s->A.ptr, 0, s->A.n * sizeof(real_t), s->dev.stream));
        HIP_CHECK(hipMemsetAsync(s->B.ptr, 0, s->B.n * sizeof(real_t), s->dev.stream));
        if (s->has_bias) {
            s->biasA.alloc(m.m);
            s->biasB.alloc(m.n);
            HIP_CHECK(hipMemsetAsync(s->biasA.ptr, 0, (size_t)m.m * sizeof(real_t), s->dev.stream));
            HIP_CHECK(hipMemsetAsync(s->biasB.ptr, 0, (size_t)m.n * sizeof(real_t), s->dev.stream));
        }
        int kmax = std::max(s->k_totA, s->k_totB) + 1;
        for (int e = 0; e < 6; e++) s->lam6[e] = s->mdl.lam;
        s->gram.alloc((size_t)kmax * kmax);
        s->ctc.alloc((size_t)kmax * kmax);
        s->betbe.alloc((size_t)kmax * kmax);
        if (m.p > 0) s->C.alloc((size_t)m.p * (m.k_user + m.k));
        if (m.q > 0) s->D.alloc((size_t)m.q * (m.k_item + m.k));
        if (m.p > 0 && m.use_cg) s->ucA.alloc((size_t)std::max(1, m.m_u) * (m.k_user + m.k));     // U C of the block CG
        if (m.q > 0 && m.use_cg) s->ucB.alloc((size_t)std::max(1, m.n_i) * (m.k_item + m.k));
        return 0;
    });
    g_last_rc = rc;
    if (rc != 0) {
        delete s;
        return nullptr;
    }
    return s;
}
int cmfrec_hip_last_error_code(void) { return g_last_rc; }

void cmfrec_hip_session_destroy(cmfrec_hip_session *s)
{
    if (!s) return;
    (void)hipStreamSynchronize(s->dev.stream);
    delete s;
}

static int weights_allowed(const cmfrec_hip_session *s, const char *fn)
{
    if (s->mdl.implicit) { g_last_error = std::string(fn) + ": observation weights belong to the explicit model"; return 2; }
    return 0;
}

// NA_as_zero_X with observation weights: an absent entry is a zero of weight one, so the rows' lambda multipliers under scale_lam
// count them (wsumA / wsumB, collective.c:7991-8022).  Kept in SparseShard::wsum_naz beside the plain sums of the weights and
// rebuilt whenever the flag is switched on or X is uploaded again (the plain sums stay what the ordinary weighted updates and
// cmfrec_hip_session_set_lambda_multipliers('A' / 'B') use).
static void refresh_naz_multipliers(cmfrec_hip_session *s)
{
    if (!s->naz_X || !s->Xr.weighted()) { s->Xr.wsum_naz.release(); s->Xc.wsum_naz.release(); return; }
    HIP_CHECK(hipSetDevice(s->dev.device));
    hipStream_t st = s->dev.stream;
    s->Xr.wsum_naz.alloc_at_least((size_t)s->Xr.nrows); s->Xc.wsum_naz.alloc_at_least((size_t)s->Xc.nrows);
    hipLaunchKernelGGL(naz_wsum_kernel<real_t>, grid1d((size_t)s->Xr.nrows), dim3(256), 0, st, s->Xr.p.ptr, s->Xr.w.ptr, s->Xr.nrows,
                       s->mdl.n, s->Xr.wsum_naz.ptr);
    hipLaunchKernelGGL(naz_wsum_kernel<real_t>, grid1d((size_t)s->Xc.nrows), dim3(256), 0, st, s->Xc.p.ptr, s->Xc.w.ptr, s->Xc.nrows,
                       s->mdl.m, s->Xc.wsum_naz.ptr);
    HIP_CHECK(hipGetLastError());
}

int cmfrec_hip_session_set_X_weighted(cmfrec_hip_session *s, const size_t *csr_p, const int_t *csr_i, const real_t *csr_v,
                                      const real_t *csr_w, const size_t *csc_p, const int_t *csc_i, const real_t *csc_v,
                                      const real_t *csc_w)
{
    return guarded([&]() {
        if ((csr_w != nullptr) != (csc_w != nullptr)) { g_last_error = "cmfrec_hip_session_set_X_weighted: weights for both orientations or none"; return 2; }
        if (csr_w != nullptr) { const int rc = weights_allowed(s, "cmfrec_hip_session_set_X_weighted"); if (rc) return rc; }
        HIP_CHECK(hipSetDevice(s->dev.device));
        s->Xr.opp_row_bytes_hint = s->Xc.opp_row_bytes_hint = (size_t)(s->mdl.k + s->mdl.k_main) * sizeof(real_t);
        shard_from_csr(s->Xr, s->mdl.row_end - s->mdl.row_begin, csr_p, csr_i, csr_v, s->mdl.n, s->dev.stream, csr_w);
        shard_from_csr(s->Xc, s->mdl.col_end - s->mdl.col_begin, csc_p, csc_i, csc_v, s->mdl.m, s->dev.stream, csc_w);
        refresh_naz_multipliers(s);
        return 0;
    });
}

int cmfrec_hip_session_set_X(cmfrec_hip_session *s, const size_t *csr_p, const int_t *csr_i, const real_t *csr_v,
                             const size_t *csc_p, const int_t *csc_i, const real_t *csc_v)
{
    return cmfrec_hip_session_set_X_weighted(s, csr_p, csr_i, csr_v, nullptr, csc_p, csc_i, csc_v, nullptr);
}

int cmfrec_hip_session_set_A_parts(cmfrec_hip_session *s, const size_t *csr_p, const int_t *csr_i, const real_t *csr_v,
                                   int nparts)
{
    return guarded([&]() {
        HIP_CHECK(hipSetDevice(s->dev.device));
        const int nrows = s->mdl.row_end - s->mdl.row_begin;
        s->XrParts.clear(); s->partBegin.clear();
        for (auto e : s->partEv) (void)hipEventDestroy(e);
        s->partEv.clear();
        if (nparts <= 1) return 0;
        if (s->mdl.p > 0) { g_last_error = "cmfrec_hip_session_set_A_parts: not with user side information"; return 2; }
        if (s->Xr.weighted()) { g_last_error = "cmfrec_hip_session_set_A_parts: not with observation weights"; return 2; }
        // no host CSR given: cut the resident one (shards built on the device)
        std::vector<size_t> hp; std::vector<int_t> hi; std::vector<real_t> hv;
        if (csr_p == nullptr) {
            if (s->Xr.nrows != nrows) { g_last_error = "cmfrec_hip_session_set_A_parts: set X first"; return 2; }
            hp.resize((size_t)nrows + 1); hi.resize(std::max<size_t>(s->Xr.nnz, 1)); hv.resize(std::max<size_t>(s->Xr.nnz, 1));
            s->Xr.p.download(hp.data(), (size_t)nrows + 1, s->dev.stream);
            s->Xr.i.download(hi.data(), s->Xr.nnz, s->dev.stream);
            s->Xr.v.download(hv.data(), s->Xr.nnz, s->dev.stream);
            HIP_CHECK(hipStreamSynchronize(s->dev.stream));
            csr_p = hp.data(); csr_i = hi.data(); csr_v = hv.data();
        }
        nparts = std::min(nparts, std::max(1, nrows));
        const int step = (nrows + nparts - 1) / nparts;
        for (int c = 0; c <= nparts; c++) s->partBegin.push_back(std::min(c * step, nrows));
        std::vector<size_t> pp;
        for (int c = 0; c < nparts; c++) {
            const int r0 = s->partBegin[c], r1 = s->partBegin[c + 1];
            pp.resize((size_t)(r1 - r0) + 1);
            for (int r = r0; r <= r1; r++) pp[r - r0] = csr_p[r] - csr_p[r0];
            s->XrParts.emplace_back(new SparseShard());
            s->XrParts.back()->is_part = true;
            s->XrParts.back()->opp_row_bytes_hint = (size_t)(s->mdl.k + s->mdl.k_main) * sizeof(real_t);
            shard_from_csr(*s->XrParts.back(), r1 - r0, pp.data(), csr_i + csr_p[r0], csr_v + csr_p[r0], s->mdl.n, s->dev.stream);
            HIP_CHECK(hipStreamSynchronize(s->dev.stream));              // pp is reused
            hipEvent_t e;
            HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            s->partEv.push_back(e);
        }
        return 0;
    });
}

int cmfrec_hip_session_part_range(cmfrec_hip_session *s, int part, int *begin, int *end)
{
    if (part < 0 || part >= (int)s->XrParts.size()) return 2;
    if (begin) *begin = s->partBegin[part];
    if (end) *end = s->partBegin[part + 1];
    return 0;
}

int cmfrec_hip_session_nparts(cmfrec_hip_session *s) { return (int)s->XrParts.size(); }

int cmfrec_hip_session_stream_wait_part(cmfrec_hip_session *s, int part, void *stream)
{
    return guarded([&]() {
        if (part < 0 || part >= (int)s->partEv.size()) { g_last_error = "cmfrec_hip_session_stream_wait_part: no such part"; return 2; }
        HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, s->partEv[part], 0));
        return 0;
    });
}

// weight: one observation weight per entry (explicit model), or null
int cmfrec_hip_session_set_X_coo_weighted(cmfrec_hip_session *s, const int_t *row, const int_t *col, const real_t *val,
                                          const real_t *weight, size_t nnz, real_t subtract, real_t alpha)
{
    return guarded([&]() {
        const cmfrec_hip_model &m = s->mdl;
        if (m.row_begin != 0 || m.row_end != m.m || m.col_begin != 0 || m.col_end != m.n) {
            g_last_error = "cmfrec_hip_session_set_X_coo: only for sessions that own all rows and columns";
            return 2;
        }
        if (weight != nullptr) { const int rc = weights_allowed(s, "cmfrec_hip_session_set_X_coo_weighted"); if (rc) return rc; }
        HIP_CHECK(hipSetDevice(s->dev.device));
        DevBuf<int> dr, dc; DevBuf<real_t> dv, dw;
        dr.upload(row, nnz, s->dev.stream); dc.upload(col, nnz, s->dev.stream); dv.upload(val, nnz, s->dev.stream);
        if (weight != nullptr) dw.upload(weight, nnz, s->dev.stream);
        s->Xr.opp_row_bytes_hint = s->Xc.opp_row_bytes_hint = (size_t)(m.k + m.k_main) * sizeof(real_t);
        s->x_subtract = subtract;
        shard_from_coo(s->Xr, m.m, m.n, dr.ptr, dc.ptr, dv.ptr, nnz, subtract, alpha, s->dev.stream, dw.ptr);
        shard_from_coo(s->Xc, m.n, m.m, dc.ptr, dr.ptr, dv.ptr, nnz, subtract, alpha, s->dev.stream, dw.ptr);
        refresh_naz_multipliers(s);
        HIP_CHECK(hipStreamSynchronize(s->dev.stream));
        return 0;
    });
}

int cmfrec_hip_session_set_X_coo(cmfrec_hip_session *s, const int_t *row, const int_t *col, const real_t *val,
                                 size_t nnz, real_t subtract, real_t alpha)
{
    return cmfrec_hip_session_set_X_coo_weighted(s, row, col, val, nullptr, nnz, subtract, alpha);
}

// One shard from a COO triplet that already lives in HBM (multi-GPU set-up: the triplets come out of the
// all-to-all exchange of the ranks' blocks, distributed.py): which = 'r' -> CSR of the local rows (d_key = row - row_begin,
// d_other = global column), 'c' -> CSC of the local columns (d_key = column - col_begin, d_other = global row).
// x := (x - subtract) * alpha as in cmfrec_hip_session_set_X_coo.  The pointers are device pointers.
int cmfrec_hip_session_set_X_coo_device(cmfrec_hip_session *s, int which, const int_t *d_key, const int_t *d_other,
                                        const real_t *d_val, size_t nnz, real_t subtract, real_t alpha)
{
    return guarded([&]() {
        const cmfrec_hip_model &m = s->mdl;
        if (which != 'r' && which != 'c') { g_last_error = "cmfrec_hip_session_set_X_coo_device: which must be 'r' or 'c'"; return 2; }
        HIP_CHECK(hipSetDevice(s->dev.device));
        s->Xr.opp_row_bytes_hint = s->Xc.opp_row_bytes_hint = (size_t)(m.k + m.k_main) * sizeof(real_t);
        if (which == 'r') shard_from_coo(s->Xr, m.row_end - m.row_begin, m.n, d_key, d_other, d_val, nnz, subtract, alpha, s->dev.stream);
        else shard_from_coo(s->Xc, m.col_end - m.col_begin, m.m, d_key, d_other, d_val, nnz, subtract, alpha, s->dev.stream);
        refresh_naz_multipliers(s);       // (this upload carries no weights: the multipliers of a weighted predecessor go)
        HIP_CHECK(hipStreamSynchronize(s->dev.stream));
        return 0;
    });
}

int cmfrec_hip_session_init_biases(cmfrec_hip_session *s, real_t lam_user, real_t lam_item)
{
    return guarded([&]() {
        const cmfrec_hip_model &m = s->mdl;
        if (m.implicit || !s->has_bias) { g_last_error = "cmfrec_hip_session_init_biases: the model has no biases"; return 2; }
        if (m.row_begin != 0 || m.row_end != m.m || m.col_begin != 0 || m.col_end != m.n) {
            g_last_error = "cmfrec_hip_session_init_biases: only for sessions that own all rows and columns";
            return 2;
        }
        HIP_CHECK(hipSetDevice(s->dev.device));
        hipStream_t st = s->dev.stream;
        // under scale_lam_sideinfo the rows that have side information count its attributes too (wsumA / wsumB,
        // collective.c:8071-8104).  For sparse side information the reference adds `U_csr_p[row+1] - U_csr[row]` there -- an
        // index minus a value (:8086): nothing to pin, so that combination is refused
        if ((s->Xr.weighted() || s->Xc.weighted()) && (m.scale_lam_sideinfo || s->scale_bias_const)) {
            g_last_error = "cmfrec_hip: bias start values with observation weights and scale_lam_sideinfo / scale_bias_const are not implemented";
            return 2;
        }
        if (m.scale_lam_sideinfo && ((s->sparseU && m.user_bias) || (s->sparseI && m.item_bias))) {
            g_last_error = "cmfrec_hip: bias start values with scale_lam_sideinfo and sparse side information are not defined by the reference";
            return 2;
        }
        auto sweep = [&](const SparseShard &X, const real_t *other, real_t lam_b, int user_rule, real_t *bias) {
            const bool users = (&X == &s->Xr);
            if (X.weighted()) {                                           // common.c:4180-4205, :4672-4692, :4826-4847
                const int onesided = (m.user_bias != m.item_bias) ? 1 : 0;
                const real_t *wd = m.scale_lam ? X.wsum.ptr : nullptr;
                if (X.n_long > 0)
                    hipLaunchKernelGGL(bias_sweep_weighted_long_kernel, dim3(X.n_long), dim3(64), 0, st, X.p.ptr, X.i.ptr, X.v.ptr, X.w.ptr,
                                       other, X.order.ptr, X.n_long, lam_b, wd, onesided, bias);
                if (X.nrows > X.n_long)
                    hipLaunchKernelGGL(bias_sweep_weighted_kernel, dim3((X.nrows - X.n_long + 63) / 64), dim3(64), 0, st, X.p.ptr, X.i.ptr,
                                       X.v.ptr, X.w.ptr, other, X.order.ptr, X.n_long, X.nrows, lam_b, wd, onesided, bias);
                return;
            }
            const int extra = m.scale_lam_sideinfo ? (users ? m.p : m.q) : 0, extra_rows = users ? m.m_u : m.n_i;
            if (X.n_long > 0)
                hipLaunchKernelGGL(bias_sweep_long_kernel, dim3(X.n_long), dim3(64), 0, st, X.p.ptr, X.i.ptr, X.v.ptr, other,
                                   X.order.ptr, X.n_long, lam_b, (int)m.scale_lam, user_rule, bias, extra, extra_rows);
            if (X.nrows > X.n_long)
                hipLaunchKernelGGL(bias_sweep_kernel, dim3((X.nrows - X.n_long + 63) / 64), dim3(64), 0, st, X.p.ptr, X.i.ptr,
                                   X.v.ptr, other, X.order.ptr, X.n_long, X.nrows, lam_b, (int)m.scale_lam, user_rule, bias, extra,
                                   extra_rows);
        };
        if (m.user_bias && !m.item_bias) {                                // collective.c:8166-8185
            sweep(s->Xr, nullptr, lam_user, 0, s->biasA.ptr);
        } else if (m.item_bias && !m.user_bias) {                         // :8187-8204 (only when the B-step uses CG)
            if (m.use_cg) sweep(s->Xc, nullptr, lam_item, 0, s->biasB.ptr);
        } else {                                                          // common.c:4410-4909, five sweeps, items first
            HIP_CHECK(hipMemsetAsync(s->biasA.ptr, 0, (size_t)m.m * sizeof(real_t), st));
            HIP_CHECK(hipMemsetAsync(s->biasB.ptr, 0, (size_t)m.n * sizeof(real_t), st));
            for (int it = 0; it < 5; it++) {
                sweep(s->Xc, s->biasA.ptr, lam_item, 0, s->biasB.ptr);
                sweep(s->Xr, s->biasB.ptr, lam_user, 1, s->biasA.ptr);
            }
        }
        if (m.user_bias)
            hipLaunchKernelGGL(col_insert_kernel<real_t>, grid1d(m.m), dim3(256), 0, st, s->A.ptr, s->ldA, m.m, s->k_totA, s->biasA.ptr);
        if (m.item_bias)
            hipLaunchKernelGGL(col_insert_kernel<real_t>, grid1d(m.n), dim3(256), 0, st, s->B.ptr, s->ldB, m.n, s->k_totB, s->biasB.ptr);
        HIP_CHECK(hipGetLastError());
        return 0;
    });
}

int cmfrec_hip_session_get_X(cmfrec_hip_session *s, int which, size_t *indptr, int_t *indices, real_t *values,
                             int_t *order)
{
    return guarded([&]() {
        HIP_CHECK(hipSetDevice(s->dev.device));
        const SparseShard &X = (which == 'r' || which == 'R') ? s->Xr : s->Xc;
        if (indptr) X.p.download(indptr, (size_t)X.nrows + 1, s->dev.stream);
        if (indices) X.i.download(indices, X.nnz, s->dev.stream);
        if (values) X.v.download(values, X.nnz, s->dev.stream);
        if (order) X.order.download(order, (size_t)X.nrows, s->dev.stream);
        HIP_CHECK(hipStreamSynchronize(s->dev.stream));
        return 0;
    });
}

static void upload_padded(cmfrec_hip_session *s, const real_t *host, size_t rows, int cols, real_t *dst, size_t ldd)
{
    if (!host || rows == 0) return;
    if ((size_t)cols == ldd) {
        HIP_CHECK(hipMemcpyAsync(dst, host, rows * cols * sizeof(real_t), hipMemcpyHostToDevice, s->dev.stream));
    } else {
        HIP_CHECK(hipMemcpy2DAsync(dst, ldd * sizeof(real_t), host, (size_t)cols * sizeof(real_t),
                                   (size_t)cols * sizeof(real_t), rows, hipMemcpyHostToDevice, s->dev.stream));
    }
}
static void download_padded(cmfrec_hip_session *s, real_t *host, size_t rows, int cols, const real_t *src, size_t lds)
{
    if (!host || rows == 0) return;
    if ((size_t)cols == lds) {
        HIP_CHECK(hipMemcpyAsync(host, src, rows * cols * sizeof(real_t), hipMemcpyDeviceToHost, s->dev.stream));
    } else {
        HIP_CHECK(hipMemcpy2DAsync(host, (size_t)cols * sizeof(real_t), src, lds * sizeof(real_t),
                                   (size_t)cols * sizeof(real_t), rows, hipMemcpyDeviceToHost, s->dev.stream));
    }
}

int cmfrec_hip_session_set_factors(cmfrec_hip_session *s, const real_t *A, const real_t *B, const real_t *biasA,
                                   const real_t *biasB, const real_t *C, const real_t *D)
{
    return guarded([&]() {
        HIP_CHECK(hipSetDevice(s->dev.device));
        const cmfrec_hip_model &m = s->mdl;
        hipStream_t st = s->dev.stream;
        s->eigA.fresh = false; s->eigB.fresh = false;
        upload_padded(s, A, m.m, s->k_totA, s->A.ptr, s->ldA);
        upload_padded(s, B, m.n, s->k_totB, s->B.ptr, s->ldB);
        if (s->has_bias) {
            // bias column of the [rows, k_tot+1] layout: the bias itself or ones (collective.c:8283-8317)
            if (biasA && m.user_bias) s->biasA.upload(biasA, m.m, st);
            if (biasB && m.item_bias) s->biasB.upload(biasB, m.n, st);
            if (m.user_bias)
                hipLaunchKernelGGL(col_insert_kernel<real_t>, grid1d(m.m), dim3(256), 0, st, s->A.ptr, s->ldA, m.m, s->k_totA, s->biasA.ptr);
            else
                hipLaunchKernelGGL(col_fill_kernel<real_t>, grid1d(m.m), dim3(256), 0, st, s->A.ptr, s->ldA, m.m, s->k_totA, (real_t)1);
            if (m.item_bias)
                hipLaunchKernelGGL(col_insert_kernel<real_t>, grid1d(m.n), dim3(256), 0, st, s->B.ptr, s->ldB, m.n, s->k_totB, s->biasB.ptr);
            else
                hipLaunchKernelGGL(col_fill_kernel<real_t>, grid1d(m.n), dim3(256), 0, st, s->B.ptr, s->ldB, m.n, s->k_totB, (real_t)1);
            HIP_CHECK(hipGetLastError());
        }
        if (C && m.p > 0) s->C.upload(C, (size_t)m.p * (m.k_user + m.k), st);
        if (D && m.q > 0) s->D.upload(D, (size_t)m.q * (m.k_item + m.k), st);
        HIP_CHECK(hipStreamSynchronize(st));
        return 0;
    });
}

int cmfrec_hip_session_get_factors(cmfrec_hip_session *s, real_t *A, real_t *B, real_t *biasA, real_t *biasB,
                                   real_t *C, real_t *D)
{
    return guarded([&]() {
        HIP_CHECK(hipSetDevice(s->dev.device));
        const cmfrec_hip_model &m = s->mdl;
        hipStream_t st = s->dev.stream;
        download_padded(s, A, m.m, s->k_totA, s->A.ptr, s->ldA);
        download_padded(s, B, m.n, s->k_totB, s->B.ptr, s->ldB);
        if (biasA && m.user_bias) s->biasA.download(biasA, m.m, st);
        if (biasB && m.item_bias) s->biasB.download(biasB, m.n, st);
        if (C && m.p > 0) s->C.download(C, (size_t)m.p * (m.k_user + m.k), st);
        if (D && m.q > 0) s->D.download(D, (size_t)m.q * (m.k_item + m.k), st);
        HIP_CHECK(hipStreamSynchronize(st));
        return 0;
    });
}

int cmfrec_hip_session_set_sideinfo(cmfrec_hip_session *s, const real_t *U, const real_t *II)
{
    return guarded([&]() {
        HIP_CHECK(hipSetDevice(s->dev.device));
        const cmfrec_hip_model &m = s->mdl;
        if (U && m.p > 0) s->U.upload(U, (size_t)m.m_u * m.p, s->dev.stream);
        if (II && m.q > 0) s->II.upload(II, (size_t)m.n_i * m.q, s->dev.stream);
        HIP_CHECK(hipStreamSynchronize(s->dev.stream));
        return 0;
    });
}

// Row-block shards: U holds the rows [row_begin, min(row_end, m_u)) only, I the rows [col_begin, min(col_end, n_i)).
int cmfrec_hip_session_set_sideinfo_local(cmfrec_hip_session *s, const real_t *U_local, const real_t *I_local)
{
    return guarded([&]() {
        HIP_CHECK(hipSetDevice(s->dev.device));
        const cmfrec_hip_model &m = s->mdl;
        const int ru = std::max(0, std::min(m.row_end, m.m_u) - m.row_begin), ri = std::max(0, std::min(m.col_end, m.n_i) - m.col_begin);
        if (U_local && m.p > 0 && ru > 0) s->U.upload(U_local, (size_t)ru * m.p, s->dev.stream);
        if (I_local && m.q > 0 && ri > 0) s->II.upload(I_local, (size_t)ri * m.q, s->dev.stream);
        s->side_local = true;
        HIP_CHECK(hipStreamSynchronize(s->dev.stream));
        return 0;
    });
}

// C / D update of a row-block shard, first half (optimizeA Case 1, common.c:2793-2991, split over the ranks): the partial
// sums over the LOCAL rows of the factor matrix F (A for 'C', B for 'D') that carry side information,
//   side_part = [ F_loc^T F_loc  (kc x kc) | U_loc^T F_loc  (p x kc) ],
// for the caller to all-reduce (cmfrec_hip_session_device_ptr(s, 'P')).  Dense side information only.
int cmfrec_hip_session_sideinfo_partial(cmfrec_hip_session *s, int which)
{
    return guarded([&]() {
        HIP_CHECK(hipSetDevice(s->dev.device));
        const cmfrec_hip_model &m = s->mdl;
        const DeviceInfo &dev = s->dev;
        const bool isC = (which == 'C');
        const int p = isC ? m.p : m.q;
        if ((which != 'C' && which != 'D') || p <= 0 || (isC ? s->sparseU : s->sparseI)) {
            g_last_error = "cmfrec_hip_session_sideinfo_partial: dense side information on that side is required";
            return 2;
        }
        const int rows_u = isC ? m.m_u : m.n_i, begin = isC ? m.row_begin : m.col_begin, end = isC ? m.row_end : m.col_end;
        const int local = std::max(0, std::min(end, rows_u) - begin);
        const int kc = (isC ? m.k_user : m.k_item) + m.k;
        const real_t *F = (isC ? s->A.ptr : s->B.ptr) + (size_t)begin * (isC ? s->ldA : s->ldB);
        const size_t ldF = isC ? s->ldA : s->ldB;
        const real_t *Um = (isC ? s->U.ptr : s->II.ptr) + (s->side_local ? (size_t)0 : (size_t)begin * p);
        s->side_part.alloc_at_least((size_t)kc * kc + (size_t)p * kc);
        HIP_CHECK(hipMemsetAsync(s->side_part.ptr, 0, ((size_t)kc * kc + (size_t)p * kc) * sizeof(real_t), dev.stream));
        if (local > 0) {
            launch_gram(dev, s->gws, F, ldF, local, kc, s->side_part.ptr, (real_t)1, (real_t)0);
            launch_gemm<true>(dev, p, kc, local, (real_t)1, Um, (size_t)p, F, ldF, s->side_part.ptr + (size_t)kc * kc, (size_t)kc);
        }
        return 0;
    });
}

// ... second half, on the all-reduced sums: C := (U^T F) (F^T F + lam I)^-1, the same on every rank
int cmfrec_hip_session_sideinfo_finish(cmfrec_hip_session *s, int which)
{
    return guarded([&]() {
        HIP_CHECK(hipSetDevice(s->dev.device));
        const cmfrec_hip_model &m = s->mdl;
        const DeviceInfo &dev = s->dev;
        const bool isC = (which == 'C');
        const int p = isC ? m.p : m.q;
        if ((which != 'C' && which != 'D') || p <= 0 || (isC ? s->sparseU : s->sparseI)) {
            g_last_error = "cmfrec_hip_session_sideinfo_finish: dense side information on that side is required";
            return 2;
        }
        const int rows_u = isC ? m.m_u : m.n_i;
        const int kc = (isC ? m.k_user : m.k_item) + m.k;
        real_t *Cm = isC ? s->C.ptr : s->D.ptr;
        (isC ? s->eigA : s->eigB).fresh = false;
        const real_t w = isC ? m.w_user : m.w_item;
        const bool scale_lam = m.scale_lam || m.scale_lam_sideinfo;
        const real_t lam = s->lam6[isC ? 4 : 5] / w;                                               // collective.c:8367, :8418
        const real_t diag = scale_lam ? lam * (real_t)rows_u : lam;                                // common.c:2832
        HIP_CHECK(hipMemcpyAsync(s->gram.ptr, s->side_part.ptr, (size_t)kc * kc * sizeof(real_t), hipMemcpyDeviceToDevice, dev.stream));
        hipLaunchKernelGGL(add_diag_kernel<real_t>, dim3((kc + 255) / 256), dim3(256), 0, dev.stream, s->gram.ptr, kc, 0, kc, diag);
        HIP_CHECK(hipMemcpyAsync(Cm, s->side_part.ptr + (size_t)kc * kc, (size_t)p * kc * sizeof(real_t), hipMemcpyDeviceToDevice, dev.stream));
        struct Scope {                                            // nonneg_C / nonneg_D, L1 on C / D as in cmfrec_hip_session_update
            const DeviceInfo &d;
            Scope(const DeviceInfo &d_, bool nn, int steps, real_t l1, real_t sc) : d(d_) { d.nonneg_now = nn; d.max_cd_steps = steps; d.l1_now = l1; d.l1_last_now = l1; d.l1_scale = sc; }
            ~Scope() { d.nonneg_now = false; d.l1_now = 0; d.l1_last_now = 0; d.l1_scale = 1; }
        } scope(dev, isC ? s->nonneg_C : s->nonneg_D, s->max_cd_steps, s->l16[isC ? 4 : 5] / w, scale_lam ? (real_t)rows_u : (real_t)1);
        CholCall c{Cm, (size_t)kc, nullptr, 0, kc, 0, nullptr, s->gram.ptr, 0, 0, 0, 0, 0, false, false, false, CHOL_PREFILLED};
        return launch_chol(dev, c, nullptr, p);                                                    // common.c:2872-2875
    });
}

int cmfrec_hip_session_set_nonneg(cmfrec_hip_session *s, int nonneg, int nonneg_C, int nonneg_D, int max_cd_steps)
{
    s->nonneg = nonneg != 0; s->nonneg_C = nonneg_C != 0; s->nonneg_D = nonneg_D != 0;
    s->max_cd_steps = max_cd_steps;
    return 0;
}

int cmfrec_hip_session_set_implicit_features(cmfrec_hip_session *s, real_t w_implicit, const real_t *Ai, const real_t *Bi)
{
    return guarded([&]() {
        const cmfrec_hip_model &m = s->mdl;
        HIP_CHECK(hipSetDevice(s->dev.device));
        if (m.implicit) { g_last_error = "cmfrec_hip: add_implicit_features belongs to the explicit-feedback model"; return 2; }
        if (m.row_begin != 0 || m.row_end != m.m || m.col_begin != 0 || m.col_end != m.n || !s->XrParts.empty()) {
            g_last_error = "cmfrec_hip: add_implicit_features: sharded sessions are not built";
            return 2;
        }
        if ((m.m_x > 0 && m.m_x < m.m) || (m.n_x > 0 && m.n_x < m.n)) {
            g_last_error = "cmfrec_hip: add_implicit_features: side information must not cover rows / columns beyond X";
            return 2;
        }
        if (s->Xr.p.ptr == nullptr || s->Xc.p.ptr == nullptr) {
            g_last_error = "cmfrec_hip: add_implicit_features: set X first";
            return 2;
        }
        if (!(w_implicit > 0)) { g_last_error = "cmfrec_hip: w_implicit must be positive"; return 2; }
        const int kk = m.k + m.k_main;
        const int ktmax = std::max(s->k_totA, s->k_totB) + 1;
        s->Ai.alloc((size_t)m.m * kk); s->Bi.alloc((size_t)m.n * kk);
        s->bitbi.alloc((size_t)kk * kk); s->bitbi_full.alloc((size_t)ktmax * ktmax);
        const size_t nnz = s->Xr.nnz;
        s->ones.alloc(std::max<size_t>(nnz, 1));
        std::vector<real_t> one(std::max<size_t>(nnz, 1), (real_t)1);
        HIP_CHECK(hipMemcpyAsync(s->ones.ptr, one.data(), one.size() * sizeof(real_t), hipMemcpyHostToDevice, s->dev.stream));
        if (Ai) HIP_CHECK(hipMemcpyAsync(s->Ai.ptr, Ai, (size_t)m.m * kk * sizeof(real_t), hipMemcpyHostToDevice, s->dev.stream));
        else HIP_CHECK(hipMemsetAsync(s->Ai.ptr, 0, (size_t)m.m * kk * sizeof(real_t), s->dev.stream));
        if (Bi) HIP_CHECK(hipMemcpyAsync(s->Bi.ptr, Bi, (size_t)m.n * kk * sizeof(real_t), hipMemcpyHostToDevice, s->dev.stream));
        else HIP_CHECK(hipMemsetAsync(s->Bi.ptr, 0, (size_t)m.n * kk * sizeof(real_t), s->dev.stream));
        HIP_CHECK(hipStreamSynchronize(s->dev.stream));
        // segments of up to GSUM_SEG entries per row, in row order
        size_t max_seg = 0;
        for (int o = 0; o < 2; o++) {
            const SparseShard &X = o == 0 ? s->Xr : s->Xc;
            std::vector<size_t> hp((size_t)X.nrows + 1);
            X.p.download(hp.data(), hp.size(), s->dev.stream);
            HIP_CHECK(hipStreamSynchronize(s->dev.stream));
            std::vector<int> srow, soff, rfirst((size_t)X.nrows + 1, 0);
            for (int r = 0; r < X.nrows; r++) {
                rfirst[r] = (int)srow.size();
                for (size_t off = 0; off < hp[r + 1] - hp[r]; off += GSUM_SEG) { srow.push_back(r); soff.push_back((int)off); }
            }
            rfirst[X.nrows] = (int)srow.size();
            auto &G = s->gsegs[o];
            G.nseg = (int)srow.size();
            G.seg_row.upload(srow.data(), std::max<size_t>(srow.size(), 1), s->dev.stream);
            G.seg_off.upload(soff.data(), std::max<size_t>(soff.size(), 1), s->dev.stream);
            G.row_first.upload(rfirst.data(), rfirst.size(), s->dev.stream);
            HIP_CHECK(hipStreamSynchronize(s->dev.stream));
            max_seg = std::max(max_seg, srow.size());
        }
        s->gpartial.alloc(std::max<size_t>(max_seg, 1) * kk);
        s->grhs.alloc((size_t)std::max(m.m, m.n) * kk);
        s->w_implicit = w_implicit;
        s->implicit_feats = true;
        return 0;
    });
}

int cmfrec_hip_session_get_implicit_features(cmfrec_hip_session *s, real_t *Ai, real_t *Bi)
{
    return guarded([&]() {
        const cmfrec_hip_model &m = s->mdl;
        if (!s->implicit_feats) { g_last_error = "cmfrec_hip: the session has no implicit features"; return 2; }
        HIP_CHECK(hipSetDevice(s->dev.device));
        const int kk = m.k + m.k_main;
        if (Ai) HIP_CHECK(hipMemcpyAsync(Ai, s->Ai.ptr, (size_t)m.m * kk * sizeof(real_t), hipMemcpyDeviceToHost, s->dev.stream));
        if (Bi) HIP_CHECK(hipMemcpyAsync(Bi, s->Bi.ptr, (size_t)m.n * kk * sizeof(real_t), hipMemcpyDeviceToHost, s->dev.stream));
        HIP_CHECK(hipStreamSynchronize(s->dev.stream));
        return 0;
    });
}

int cmfrec_hip_session_set_NA_as_zero_X(cmfrec_hip_session *s, int on, int center, real_t glob_mean)
{
    return guarded([&]() {
        if (on && s->mdl.implicit) { g_last_error = "cmfrec_hip_session_set_NA_as_zero_X: explicit model only"; return 2; }
        // the mean enters through the right-hand-side constant: an X that was uploaded centred would count it twice; the weighted
        // and the sharded forms are not built (the updates would refuse them one by one)
        if (on && (s->x_subtract != (real_t)0 || s->mdl.row_begin != 0 || s->mdl.row_end != s->mdl.m ||
                   s->mdl.col_begin != 0 || s->mdl.col_end != s->mdl.n)) {
            g_last_error = "cmfrec_hip_session_set_NA_as_zero_X: X must have been set uncentred (subtract = 0) on the whole row and "
                           "column range";
            return 2;
        }
        s->naz_X = on != 0; s->naz_center = center != 0; s->naz_mean = glob_mean;
        refresh_naz_multipliers(s);
        return 0;
    });
}

int cmfrec_hip_session_set_zero_rows(cmfrec_hip_session *s, int which, const int_t *rows, int count)
{
    return guarded([&]() {
        HIP_CHECK(hipSetDevice(s->dev.device));
        if (which != 'A' && which != 'B') { g_last_error = "cmfrec_hip_session_set_zero_rows: which must be 'A' or 'B'"; return 2; }
        const int limit = which == 'A' ? s->mdl.m : s->mdl.n;
        for (int e = 0; e < count; e++)
            if (rows == nullptr || rows[e] < 0 || rows[e] >= limit) { g_last_error = "cmfrec_hip_session_set_zero_rows: row out of range"; return 2; }
        DevBuf<int> &buf = which == 'A' ? s->zrowsA : s->zrowsB;
        (which == 'A' ? s->n_zrowsA : s->n_zrowsB) = std::max(count, 0);
        if (count > 0) {
            std::vector<int> h(rows, rows + count);
            buf.alloc_at_least((size_t)count);
            HIP_CHECK(hipMemcpyAsync(buf.ptr, h.data(), (size_t)count * sizeof(int), hipMemcpyHostToDevice, s->dev.stream));
            HIP_CHECK(hipStreamSynchronize(s->dev.stream));
        }
        return 0;
    });
}

int cmfrec_hip_session_set_closed_form_rows(cmfrec_hip_session *s, int which, const unsigned char *mask)
{
    return guarded([&]() {
        HIP_CHECK(hipSetDevice(s->dev.device));
        if (which == 'C' || which == 'D') {
            // attributes of DENSE side information with NaN, which the session holds as the sparse matrix of its present values: what
            // the reference's dense C / D update does per attribute inside a CG update (0: CG as asked for, 1: closed form, 2: CG from
            // zero with k_side + k steps; fit.hip, DenseNanSide::rules)
            const bool isC = which == 'C';
            const size_t rows = (size_t)(isC ? s->mdl.p : s->mdl.q);
            bool *any = isC ? s->cfC_any : s->cfD_any;
            any[0] = any[1] = any[2] = false;
            if (mask == nullptr) return 0;
            if (!(isC ? s->sparseU : s->sparseI)) { g_last_error = "cmfrec_hip_session_set_closed_form_rows: 'C' / 'D' need sparse side information on that side"; return 2; }
            for (size_t r = 0; r < rows; r++) {
                if (mask[r] > 2) { g_last_error = "cmfrec_hip_session_set_closed_form_rows: mask values 0, 1, 2"; return 2; }
                any[mask[r]] = true;
            }
            DevBuf<unsigned char> &buf = isC ? s->cfmaskC : s->cfmaskD;
            buf.alloc_at_least(rows);
            HIP_CHECK(hipMemcpyAsync(buf.ptr, mask, rows, hipMemcpyHostToDevice, s->dev.stream));
            HIP_CHECK(hipStreamSynchronize(s->dev.stream));
            return 0;
        }
        if (which != 'A' && which != 'B') { g_last_error = "cmfrec_hip_session_set_closed_form_rows: which must be 'A', 'B', 'C' or 'D'"; return 2; }
        const size_t rows = (size_t)(which == 'A' ? s->mdl.m : s->mdl.n);
        (which == 'A' ? s->has_cfA : s->has_cfB) = (mask != nullptr);
        if (mask != nullptr) {
            DevBuf<unsigned char> &buf = which == 'A' ? s->cfmaskA : s->cfmaskB;
            buf.alloc_at_least(rows);
            HIP_CHECK(hipMemcpyAsync(buf.ptr, mask, rows, hipMemcpyHostToDevice, s->dev.stream));
            HIP_CHECK(hipStreamSynchronize(s->dev.stream));
        }
        return 0;
    });
}

int cmfrec_hip_session_set_lambda_multipliers(cmfrec_hip_session *s, int which, const real_t *mult)
{
    return guarded([&]() {
        HIP_CHECK(hipSetDevice(s->dev.device));
        if (mult != nullptr && (which == 'C' || which == 'D')) {
            // attributes of dense side information with NaN under scale_lam (DenseNanSide::rules): unit weights on the attribute-major
            // shard of the present values (a multiplication by one: the unweighted numbers bit for bit) carry the multipliers
            SparseShard &Us = which == 'C' ? s->Usc : s->Isc;
            if (!(which == 'C' ? s->sparseU : s->sparseI) || Us.nnz == 0) {
                g_last_error = "cmfrec_hip_session_set_lambda_multipliers: 'C' / 'D' need sparse side information on that side";
                return 2;
            }
            Us.w.alloc_at_least(Us.nnz);
            hipLaunchKernelGGL(fill_kernel<real_t>, grid1d(Us.nnz), dim3(256), 0, s->dev.stream, Us.w.ptr, Us.nnz, (real_t)1);
            HIP_CHECK(hipGetLastError());
            Us.wsum.upload(mult, (size_t)Us.nrows, s->dev.stream);
            HIP_CHECK(hipStreamSynchronize(s->dev.stream));
            return 0;
        }
        if ((which != 'A' && which != 'B') || mult == nullptr) { g_last_error = "cmfrec_hip_session_set_lambda_multipliers: which must be 'A', 'B', 'C' or 'D', mult non-null"; return 2; }
        SparseShard &X = which == 'A' ? s->Xr : s->Xc;
        if (!X.weighted() || s->mdl.implicit) {
            g_last_error = "cmfrec_hip_session_set_lambda_multipliers: the explicit model with observation weights on X (unit weights will do)";
            return 2;
        }
        X.wsum.upload(mult, (size_t)X.nrows, s->dev.stream);
        HIP_CHECK(hipStreamSynchronize(s->dev.stream));
        return 0;
    });
}

int cmfrec_hip_session_set_scale_bias_const(cmfrec_hip_session *s, int on)
{
    s->scale_bias_const = on != 0;
    return 0;
}

int cmfrec_hip_session_set_lam_unique(cmfrec_hip_session *s, const real_t *lam_unique, const real_t *l1_lam_unique, int max_cd_steps)
{
    if (lam_unique) for (int e = 0; e < 6; e++) s->lam6[e] = lam_unique[e];
    if (l1_lam_unique) {
        for (int e = 0; e < 6; e++) s->l16[e] = l1_lam_unique[e];
        s->max_cd_steps = max_cd_steps;
    }
    return 0;
}

int cmfrec_hip_session_set_l1(cmfrec_hip_session *s, real_t l1_lam, int max_cd_steps)
{
    s->l1_lam = l1_lam;
    for (int e = 0; e < 6; e++) s->l16[e] = l1_lam;
    s->max_cd_steps = max_cd_steps;
    return 0;
}

int cmfrec_hip_session_set_sideinfo_sparse(cmfrec_hip_session *s, int which, const int_t *row, const int_t *col,
                                           const real_t *val, size_t nnz)
{
    return guarded([&]() {
        HIP_CHECK(hipSetDevice(s->dev.device));
        const cmfrec_hip_model &m = s->mdl;
        const bool isU = (which == 'U');
        const int rows = isU ? m.m_u : m.n_i, cols = isU ? m.p : m.q;
        if ((which != 'U' && which != 'I') || rows <= 0 || cols <= 0 || nnz == 0) {
            g_last_error = "cmfrec_hip_session_set_sideinfo_sparse: needs 'U' / 'I', the model's m_u, p / n_i, q and entries";
            return 2;
        }
        if (rows > (isU ? (m.m_x > 0 ? m.m_x : m.m) : (m.n_x > 0 ? m.n_x : m.n)) || m.row_begin != 0 || m.row_end != m.m ||
            m.col_begin != 0 || m.col_end != m.n) {
            g_last_error = "cmfrec_hip: sparse side information: rows beyond X and sharded sessions are not supported";
            return 2;
        }
        DevBuf<int> dr, dc; DevBuf<real_t> dv;
        dr.upload(row, nnz, s->dev.stream); dc.upload(col, nnz, s->dev.stream); dv.upload(val, nnz, s->dev.stream);
        // CSR over ALL rows of the factor matrix (rows beyond m_u / n_i are empty), CSC over the attributes
        shard_from_coo(isU ? s->Usr : s->Isr, isU ? m.m : m.n, cols, dr.ptr, dc.ptr, dv.ptr, nnz, (real_t)0, (real_t)1, s->dev.stream);
        shard_from_coo(isU ? s->Usc : s->Isc, cols, rows, dc.ptr, dr.ptr, dv.ptr, nnz, (real_t)0, (real_t)1, s->dev.stream);
        HIP_CHECK(hipStreamSynchronize(s->dev.stream));
        (isU ? s->sparseU : s->sparseI) = true;
        return 0;
    });
}

// ---- one half-step of the ALS loop on the local block -------------------------------------
// Explicit model: rows that exist only in the side information (beyond the shape of X) are fitted to it alone,
//   a[:kc] = (C^T C + (lam / w) (p if scale_lam) I)^-1 C^T u     (optimizeA Case 1 on U[m_x:], collective.c:4967-5101)
// their remaining unknowns are zero under Cholesky (A was zeroed) and left alone under CG.
static int solve_sideinfo_only_rows(cmfrec_hip_session *s, bool isA, bool chol, int first, int count)
{
    if (count <= 0) return 0;
    const cmfrec_hip_model &m = s->mdl;
    const DeviceInfo &dev = s->dev;
    real_t *self = isA ? s->A.ptr : s->B.ptr;
    const size_t ld_self = isA ? s->ldA : s->ldB;
    const int begin = isA ? m.row_begin : m.col_begin;
    const int p_self = isA ? m.p : m.q, kc = (isA ? m.k_user : m.k_item) + m.k;
    const real_t *Cm = isA ? s->C.ptr : s->D.ptr;
    const real_t *Um = isA ? s->U.ptr : s->II.ptr;
    const real_t w = isA ? m.w_user : m.w_item;
    const bool scale_lam = m.scale_lam || m.scale_lam_sideinfo;                                    // :7465
    real_t *rows = self + (size_t)(begin + first) * ld_self;
    if (chol) HIP_CHECK(hipMemsetAsync(rows, 0, (size_t)count * ld_self * sizeof(real_t), dev.stream));
    launch_gram(dev, s->gws, Cm, (size_t)kc, p_self, kc, s->betbe.ptr, (real_t)1,
                (s->lam6[isA ? 2 : 3] / w) * (real_t)(scale_lam ? p_self : 1));                                   // common.c:2824-2832
    launch_gemm<false>(dev, count, kc, p_self, (real_t)1, Um + (size_t)((s->side_local ? 0 : begin) + first) * p_self, (size_t)p_self, Cm,
                       (size_t)kc, rows, ld_self);                                                 // common.c:2847-2855
    CholCall c{rows, ld_self, nullptr, 0, kc, 0, nullptr, s->betbe.ptr, 0, 0, 0, 0, 0, false, false, false, CHOL_PREFILLED};
    return launch_chol(dev, c, nullptr, count);                                                    // common.c:2872-2875
}

// sum of the rows of F (Bi / Ai) at every row's observed positions, by the two-stage segmented gather-sum; result in s->grhs
// [X.nrows, kk].  Returns false when the tables / widths do not allow it (the caller then gathers inside the row kernel).
static bool launch_gsum(cmfrec_hip_session *s, bool isA, const SparseShard &X, const real_t *Fi, int kk)
{
    auto &G = s->gsegs[isA ? 0 : 1];
    if (kk > 64 * GSUM_MAXC || s->grhs.ptr == nullptr) return false;
    hipStream_t st = s->dev.stream;
    if (G.nseg > 0)
        hipLaunchKernelGGL(gather_sum_segments_kernel<real_t>, dim3((G.nseg + 3) / 4), dim3(256), 0, st, X.p.ptr, X.i.ptr, Fi, (size_t)kk, kk,
                           G.seg_row.ptr, G.seg_off.ptr, G.nseg, s->gpartial.ptr);
    hipLaunchKernelGGL(gather_sum_rows_kernel<real_t>, dim3((X.nrows + 3) / 4), dim3(256), 0, st, s->gpartial.ptr, G.row_first.ptr, X.nrows, kk,
                       s->grhs.ptr, (size_t)kk);
    HIP_CHECK(hipGetLastError());
    return true;
}

// NA_as_zero_X on a side with SPARSE side information (missing = absent; round 5): no matrix is shared by the rows any more -- each
// adds the rank-1 terms of its own attributes -- so the reference solves row by row (collective_closed_form_block's general
// branch with prefer_BtB, collective.c:1534-1846):
//     M_i   = blockdiag(0, B^T B) + w sum_u c_u c_u^T + lam mult_i I,   mult_i = n (+ the row's attributes under scale_lam_sideinfo)
//     rhs_i = [w sum_u u_iu c_u ; sum_j x_j b_j + cst]
// on the two-source build of the row Cholesky kernel: B^T B embedded as the matrix every row starts from (Mfull), the entries of X
// right-hand side only, the attributes as the second gather source with their rank-1 terms, the constant prefilled.  Closed
// form only (the block CG with NA_as_zero_X, collective.c:2134-2903, is not restated).
static int update_factor_naz_sparse_side(cmfrec_hip_session *s, bool isA, bool chol)
{
    const cmfrec_hip_model &m = s->mdl;
    const DeviceInfo &dev = s->dev;
    hipStream_t st = dev.stream;
    const int p_self = isA ? m.p : m.q, rows_u = isA ? m.m_u : m.n_i;
    const int rows_self = isA ? m.m : m.n, rows_opp = isA ? m.n : m.m;
    if (rows_u != rows_self) {
        g_last_error = "cmfrec_hip: NA_as_zero_X with side information: side information on exactly the rows / columns of X";
        return 2;
    }
    real_t *self = isA ? s->A.ptr : s->B.ptr;
    real_t *opp = isA ? s->B.ptr : s->A.ptr;
    const size_t ld_self = isA ? s->ldA : s->ldB, ld_opp = isA ? s->ldB : s->ldA;
    const int k_side_self = isA ? m.k_user : m.k_item, k_side_opp = isA ? m.k_item : m.k_user;
    const SparseShard &X = isA ? s->Xr : s->Xc;
    const SparseShard &Us = isA ? s->Usr : s->Isr;
    const real_t *Cm = isA ? s->C.ptr : s->D.ptr;
    const real_t w = isA ? m.w_user : m.w_item;
    const bool self_bias = isA ? m.user_bias : m.item_bias, opp_bias = isA ? m.item_bias : m.user_bias;
    const real_t lam_self = s->lam6[isA ? 2 : 3];
    const real_t lam_last_self = self_bias ? s->lam6[isA ? 0 : 1] : lam_self;
    const int kk = m.k + m.k_main, ks = kk + (self_bias ? 1 : 0), kc = k_side_self + m.k, kt = k_side_self + ks;
    const real_t *oppx = opp + k_side_opp;
    if (self_bias)
        hipLaunchKernelGGL(col_fill_kernel<real_t>, grid1d(rows_opp), dim3(256), 0, st, opp, ld_opp, rows_opp, isA ? s->k_totB : s->k_totA, (real_t)1);
    launch_gram(dev, s->gws, oppx, ld_opp, rows_opp, ks, s->gram.ptr, (real_t)1, (real_t)0);
    // implicit features (round 6, fixture g37): w_i Bi^T Bi on the first k + k_main unknowns of the X block (collective.c:1704-1707;
    // the block CG keeps it apart, unweighted: :2301-2304, :2626-2629), w_i times the sum of the opposing implicit factors at the
    // row's observed positions in the right-hand side (:1757-1771)
    const real_t *Fi = s->implicit_feats ? (isA ? s->Bi.ptr : s->Ai.ptr) : nullptr;
    if (Fi != nullptr) {
        launch_gram(dev, s->gws, Fi, (size_t)kk, rows_opp, kk, s->bitbi.ptr, chol ? s->w_implicit : (real_t)1, (real_t)0);
        if (chol) {
            hipLaunchKernelGGL(add_block_kernel<real_t>, grid1d((size_t)kk * kk), dim3(256), 0, st, s->bitbi.ptr, kk, (real_t)1, s->gram.ptr, ks, 0);
            if (!launch_gsum(s, isA, X, Fi, kk)) {
                g_last_error = "cmfrec_hip: NA_as_zero_X with implicit features: k + k_main too wide for the gather-sum";
                return 2;
            }
        }
    }
    if (!chol) {
        // Block CG (round 6; collective_block_cg with NA_as_zero_X and u_vec_sp, collective.c:2134-2903): the X block of every row's
        // system is the shared B^T B (:2430-2445 first residual, :2700-2760 the products), its right-hand side sum_j x_j b_j + cst;
        // the row's attributes gather rows of C.  On the lane <-> unknown kernel: B^T B as its shared matrix (CgParams::gx), the
        // right-hand side of the X block as the per-row constant of the first residual, the entries of X with rank-1 weight zero.
        s->naz_rhs.alloc_at_least((size_t)rows_self * ks);
        HIP_CHECK(hipMemsetAsync(s->naz_rhs.ptr, 0, (size_t)rows_self * ks * sizeof(real_t), st));
        {
            CholCall cr{s->naz_rhs.ptr, (size_t)ks, oppx, ld_opp, ks, 0, nullptr, s->gram.ptr, 0, 0, 0, lam_self, lam_last_self, false, false, false, CHOL_NAZ};
            cr.rhs_only = true;
            int rc = launch_chol(dev, cr, &X);
            if (rc) return rc;
        }
        if (opp_bias || s->naz_center) {
            const int nb = (rows_opp + COLSUM_ROWS - 1) / COLSUM_ROWS;
            s->naz_part.alloc_at_least((size_t)nb * ks); s->naz_vec.alloc_at_least((size_t)ks);
            const real_t *bias = opp_bias ? (isA ? s->biasB.ptr : s->biasA.ptr) : nullptr;
            hipLaunchKernelGGL(weighted_colsum_partial_kernel<real_t>, dim3(nb), dim3(256), 0, st, oppx, ld_opp, rows_opp, ks, bias,
                               s->naz_center ? s->naz_mean : (real_t)0, s->naz_part.ptr);
            hipLaunchKernelGGL(colsum_finish_kernel<real_t>, grid1d(ks), dim3(256), 0, st, s->naz_part.ptr, nb, ks, (real_t)-1, s->naz_vec.ptr);
            hipLaunchKernelGGL(add_rowvec_kernel<real_t>, grid1d((size_t)rows_self * ks), dim3(256), 0, st, s->naz_rhs.ptr, (size_t)ks, (size_t)rows_self, ks,
                               s->naz_vec.ptr);
        }
        const size_t nnz = std::max<size_t>(X.nnz, 1);
        if (s->naz_zero.n < nnz) {
            s->naz_zero.alloc(nnz);
            HIP_CHECK(hipMemsetAsync(s->naz_zero.ptr, 0, nnz * sizeof(real_t), st));
        }
        const bool scaled_cg = m.scale_lam || m.scale_lam_sideinfo;
        if (scaled_cg) {
            s->naz_mult.alloc_at_least((size_t)rows_self);
            hipLaunchKernelGGL(fill_kernel<real_t>, grid1d((size_t)rows_self), dim3(256), 0, st, s->naz_mult.ptr, (size_t)rows_self, (real_t)rows_opp);
        }
        HIP_CHECK(hipGetLastError());
        CgCall c{self, ld_self, oppx, ld_opp, ks, nullptr, nullptr, lam_self, lam_last_self, scaled_cg, false, m.max_cg_steps, false,
                 (bool)m.precondition_cg};
        c.koff = k_side_self; c.kc = kc; c.w_side = w; c.rows_with_u = rows_u; c.p_side = p_self;
        c.scale_lam_sideinfo = (bool)m.scale_lam_sideinfo; c.X2 = &Us; c.C2 = Cm;
        c.Gx = s->gram.ptr; c.rconst_x = s->naz_rhs.ptr; c.ldr_x = (size_t)ks;
        c.values_override = s->naz_zero.ptr; c.weights_override = s->naz_zero.ptr;
        if (scaled_cg) c.wsum_override = s->naz_mult.ptr;
        c.gx_all_rows = opp_bias || s->naz_center;         // rows with neither entries nor attributes: solved from the constant, zero without it
        if (Fi != nullptr) { c.Bi = Fi; c.BiTBi = s->bitbi.ptr; c.ki = kk; c.w_imp = s->w_implicit; }     // (the kernel gathers Bi_j at the row's entries itself)
        return launch_cg(dev, c, X);
    }
    s->naz_M.alloc_at_least((size_t)kt * kt);
    hipLaunchKernelGGL(betbe_base_kernel<real_t>, grid1d((size_t)kt * kt), dim3(256), 0, st, s->gram.ptr, ks, k_side_self, (real_t)0, s->naz_M.ptr);
    // right-hand sides start from [0 ; cst]
    HIP_CHECK(hipMemset2DAsync(self, ld_self * sizeof(real_t), 0, (size_t)kt * sizeof(real_t), (size_t)rows_self, st));
    if (opp_bias || s->naz_center) {
        const int nb = (rows_opp + COLSUM_ROWS - 1) / COLSUM_ROWS;
        s->naz_part.alloc_at_least((size_t)nb * ks); s->naz_vec.alloc_at_least((size_t)ks);
        const real_t *bias = opp_bias ? (isA ? s->biasB.ptr : s->biasA.ptr) : nullptr;
        hipLaunchKernelGGL(weighted_colsum_partial_kernel<real_t>, dim3(nb), dim3(256), 0, st, oppx, ld_opp, rows_opp, ks, bias,
                           s->naz_center ? s->naz_mean : (real_t)0, s->naz_part.ptr);
        hipLaunchKernelGGL(colsum_finish_kernel<real_t>, grid1d(ks), dim3(256), 0, st, s->naz_part.ptr, nb, ks, (real_t)-1, s->naz_vec.ptr);
        hipLaunchKernelGGL(add_rowvec_kernel<real_t>, grid1d((size_t)rows_self * ks), dim3(256), 0, st, self + k_side_self, ld_self, (size_t)rows_self, ks,
                           s->naz_vec.ptr);
    }
    if (Fi != nullptr)
        hipLaunchKernelGGL(add_cols_scaled_kernel<real_t>, grid1d((size_t)X.nrows * kk), dim3(256), 0, st, self, ld_self, k_side_self, s->grhs.ptr, kk,
                           s->w_implicit, (size_t)X.nrows);
    const bool scaled = m.scale_lam || m.scale_lam_sideinfo;
    if (scaled) {
        s->naz_mult.alloc_at_least((size_t)rows_self);
        hipLaunchKernelGGL(fill_kernel<real_t>, grid1d((size_t)rows_self), dim3(256), 0, st, s->naz_mult.ptr, (size_t)rows_self, (real_t)rows_opp);
    }
    HIP_CHECK(hipGetLastError());
    CholCall c{self, ld_self, oppx, ld_opp, kt, k_side_self, nullptr, nullptr, kc, rows_u, p_self, lam_self, lam_last_self, scaled,
               (bool)m.scale_lam_sideinfo, false, CHOL_COLLECTIVE, s->naz_M.ptr};
    c.X2 = &Us; c.B2 = Cm; c.ldb2 = (size_t)kc; c.kc2 = kc; c.w2 = w;
    c.rhs_prefilled_all = true; c.x_rhs_only = true;
    if (scaled) c.mult_override = s->naz_mult.ptr;
    return launch_chol(dev, c, &X);
}

// One half-step of the explicit model with the main matrix missing-as-zero (see cmfrec_hip_session::naz_X).
// Without side information on this side: optimizeA Case 3 (common.c:3116-3205).  With dense, complete side information that
// covers exactly the rows of X: optimizeA_collective with bufferBeTBeChol (collective.c:5566-5968, :5607-5617) -- every row
// shares  blockdiag(0, B^T B) + w C^T C + lam mult I  (mult = n + p | n | 1, :4787-4799, :5700-5716), right-hand sides
// X B + w U C + the bias / mean constant on the columns behind k_user (:5753-5770, :5815-5821), one factorisation (:5715) and
// the substitutions of collective_closed_form_block's first branch (:1364-1460).
static int update_factor_naz(cmfrec_hip_session *s, bool isA, bool chol)
{
    const cmfrec_hip_model &m = s->mdl;
    const DeviceInfo &dev = s->dev;
    hipStream_t st = dev.stream;
    const int p_self = isA ? m.p : m.q;
    const int rows_u = isA ? m.m_u : m.n_i;
    const int rows_self = isA ? m.m : m.n, rows_opp = isA ? m.n : m.m;
    if (s->scale_bias_const || dev.nonneg_now || dev.l1_now != (real_t)0 ||
        dev.l1_last_now != (real_t)0 || m.row_begin != 0 || m.row_end != m.m || m.col_begin != 0 || m.col_end != m.n ||
        s->Xr.weighted() || s->side_local) {
        g_last_error = "cmfrec_hip: NA_as_zero_X: the explicit model on one device without weights, nonneg / L1, "
                       "scale_bias_const, incomplete side information";
        return 2;
    }
    // implicit features (round 5): the model without side information -- the reference runs those half-steps through
    // optimizeA_collective's general branch (collective.c:8612 / :8783 -> :1534-1846 with prefer_BtB), which without side
    // information is ONE matrix for all rows, B^T B + w_i Bi^T Bi + lam mult I, and right-hand sides X B + w_i sum_{observed} Bi_j
    // + the constant: the shared-matrix half-step below with two more terms.  Closed form (the block CG is not restated).
    // (use_cg: the reference takes its closed-form Case 1 here whatever the solver asked for -- collective.c:5121-5130 -- and returns
    //  the same numbers bit for bit)
    // (round 6, fixture g37: with side information too -- dense: the shared block matrix takes w_i Bi^T Bi on its X block and the
    //  right-hand sides the gather-sum like below; sparse: update_factor_naz_sparse_side)
    if (p_self > 0 && (isA ? s->sparseU : s->sparseI)) return update_factor_naz_sparse_side(s, isA, chol);
    if (p_self > 0 && rows_u != rows_self) {
        // (m > m_u takes optimizeA Case 3 for the rows beyond in the reference -- its build corrupts the heap there, so nothing
        //  pins that branch)
        g_last_error = "cmfrec_hip: NA_as_zero_X with side information: side information on exactly the rows / columns of X";
        return 2;
    }
    // (use_cg changes nothing here, like in Case 3: collective_closed_form_block takes the factorised block matrix before it looks
    //  at the solver, collective.c:1364-1460 -- the reference's fit with use_cg = true returns the closed-form numbers, g20)
    (void)chol;
    if (p_self == 0 && (isA ? m.k_user : m.k_item) != 0) {
        g_last_error = "cmfrec_hip: NA_as_zero_X: k_user / k_item without side information on that side";
        return 2;
    }
    real_t *self = isA ? s->A.ptr : s->B.ptr;
    real_t *opp = isA ? s->B.ptr : s->A.ptr;
    const size_t ld_self = isA ? s->ldA : s->ldB, ld_opp = isA ? s->ldB : s->ldA;
    const int k_side_self = isA ? m.k_user : m.k_item, k_side_opp = isA ? m.k_item : m.k_user;
    const SparseShard &X = isA ? s->Xr : s->Xc;
    const bool self_bias = isA ? m.user_bias : m.item_bias, opp_bias = isA ? m.item_bias : m.user_bias;
    const real_t lam_self = s->lam6[isA ? 2 : 3];
    const real_t lam_last_self = self_bias ? s->lam6[isA ? 0 : 1] : lam_self;
    const int kk = m.k + m.k_main, ks = kk + (self_bias ? 1 : 0), kc = k_side_self + m.k, kt = k_side_self + ks;
    const real_t *oppx = opp + k_side_opp;                        // the columns X refers to
    if (self_bias) {                          // the opposing bias column is fixed to 1 (collective.c:8538-8543, :8728-8732)
        hipLaunchKernelGGL(col_fill_kernel<real_t>, grid1d(rows_opp), dim3(256), 0, st, opp, ld_opp, rows_opp, isA ? s->k_totB : s->k_totA,
                           (real_t)1);
    }
    // shared matrix: opp^T opp + diag(lam .. lam, lam_last), x rows_opp (+ p with side information on this side under
    // scale_lam_sideinfo) under scale_lam (common.c:3128-3138; collective.c:4787-4799)
    const real_t mult = (p_self > 0 && m.scale_lam_sideinfo) ? (real_t)(rows_opp + p_self)
                        : ((m.scale_lam || m.scale_lam_sideinfo) ? (real_t)rows_opp : (real_t)1);
    launch_gram(dev, s->gws, oppx, ld_opp, rows_opp, ks, s->gram.ptr, (real_t)1, lam_self * mult);
    if (lam_last_self != lam_self)
        hipLaunchKernelGGL(add_diag_kernel<real_t>, dim3(1), dim3(64), 0, st, s->gram.ptr, ks, ks - 1, ks, (lam_last_self - lam_self) * mult);
    const real_t *Fi = s->implicit_feats ? (isA ? s->Bi.ptr : s->Ai.ptr) : nullptr;      // the opposing side's implicit factors [rows_opp, kk]
    if (Fi != nullptr) {                                                                   // + w_i Bi^T Bi on the first k + k_main unknowns (:1704-1707)
        launch_gram(dev, s->gws, Fi, (size_t)kk, rows_opp, kk, s->bitbi.ptr, s->w_implicit, (real_t)0);
        hipLaunchKernelGGL(add_block_kernel<real_t>, grid1d((size_t)kk * kk), dim3(256), 0, st, s->bitbi.ptr, kk, (real_t)1, s->gram.ptr, ks, 0);
    }
    real_t *Msh = s->gram.ptr;                // the matrix the rows share, [kt, kt]
    real_t *rhs_x = self;                     // where the row kernel leaves sum_j x_j opp_j: [rows, ks], ld
    size_t ld_rhs = ld_self;
    if (p_self > 0) {
        const real_t *Cm = isA ? s->C.ptr : s->D.ptr;
        const real_t w = isA ? m.w_user : m.w_item;
        launch_gram(dev, s->gws, Cm, (size_t)kc, p_self, kc, s->ctc.ptr, w, (real_t)0);            // :5658-5668
        s->naz_M.alloc_at_least((size_t)kt * kt);
        hipLaunchKernelGGL(naz_block_matrix_kernel<real_t>, grid1d((size_t)kt * kt), dim3(256), 0, st, s->gram.ptr, ks, s->ctc.ptr, kc, k_side_self,
                           lam_self * mult, s->naz_M.ptr);
        Msh = s->naz_M.ptr;
        s->naz_rhs.alloc_at_least((size_t)rows_self * ks);
        rhs_x = s->naz_rhs.ptr; ld_rhs = (size_t)ks;
    }
    // right-hand sides: sum_j x_j opp_j over the row's entries (tgemm_sp_dense, :3145-3151 / :5753-5762) ...
    real_t *rhs = self;
    const size_t ld_r = ld_self;
    HIP_CHECK(hipMemset2DAsync(rhs, ld_r * sizeof(real_t), 0, (size_t)kt * sizeof(real_t), (size_t)rows_self, st));
    if (p_self > 0) HIP_CHECK(hipMemsetAsync(rhs_x, 0, (size_t)rows_self * ks * sizeof(real_t), st));
    CholCall c{rhs_x, ld_rhs, oppx, ld_opp, ks, 0, nullptr, s->gram.ptr, 0, 0, 0, lam_self, lam_last_self, false, false, false, CHOL_NAZ};
    c.rhs_only = true;
    int rc = launch_chol(dev, c, &X);
    if (rc) return rc;
    if (p_self > 0) {
        // ... w U C on the first k_side + k columns, the gathered part behind k_side
        const real_t *Cm = isA ? s->C.ptr : s->D.ptr;
        const real_t *Um = isA ? s->U.ptr : s->II.ptr;
        launch_gemm<false>(dev, rows_self, kc, p_self, isA ? m.w_user : m.w_item, Um, (size_t)p_self, Cm, (size_t)kc, rhs, ld_r);
        hipLaunchKernelGGL(add_cols_scaled_kernel<real_t>, grid1d((size_t)rows_self * ks), dim3(256), 0, st, rhs, ld_r, k_side_self, rhs_x, ks,
                           (real_t)1, (size_t)rows_self);
    }
    if (Fi != nullptr) {
        // ... w_i times the sum of the opposing implicit factors at the row's observed positions (:1757-1771)
        if (!launch_gsum(s, isA, X, Fi, kk)) {
            g_last_error = "cmfrec_hip: NA_as_zero_X with implicit features: k + k_main too wide for the gather-sum";
            return 2;
        }
        hipLaunchKernelGGL(add_cols_scaled_kernel<real_t>, grid1d((size_t)X.nrows * kk), dim3(256), 0, st, rhs, ld_r, k_side_self, s->grhs.ptr, kk,
                           s->w_implicit, (size_t)X.nrows);
    }
    // ... plus the constant of the opposing biases and the mean (:3152-3157 / :5815-5821)
    if (opp_bias || s->naz_center) {
        const int nb = (rows_opp + COLSUM_ROWS - 1) / COLSUM_ROWS;
        s->naz_part.alloc_at_least((size_t)nb * ks); s->naz_vec.alloc_at_least((size_t)ks);
        const real_t *bias = opp_bias ? (isA ? s->biasB.ptr : s->biasA.ptr) : nullptr;
        hipLaunchKernelGGL(weighted_colsum_partial_kernel<real_t>, dim3(nb), dim3(256), 0, st, oppx, ld_opp, rows_opp, ks, bias,
                           s->naz_center ? s->naz_mean : (real_t)0, s->naz_part.ptr);
        hipLaunchKernelGGL(colsum_finish_kernel<real_t>, grid1d(ks), dim3(256), 0, st, s->naz_part.ptr, nb, ks, (real_t)-1, s->naz_vec.ptr);
        hipLaunchKernelGGL(add_rowvec_kernel<real_t>, grid1d((size_t)rows_self * ks), dim3(256), 0, st, rhs + k_side_self, ld_r, (size_t)rows_self, ks,
                           s->naz_vec.ptr);
    }
    // one factorisation, all rows (with and without entries) through the triangular solves (:3171-3175 / :5715, :1455-1458)
    hipLaunchKernelGGL(potrf_upper_kernel<real_t>, dim3(1), dim3(256), 0, st, Msh, kt);
    HIP_CHECK(hipGetLastError());
    launch_potrs_rows(dev, rows_self, kt, Msh, self, ld_self);
    return 0;
}

// ... WITH observation weights on a side that carries DENSE side information (round 5): a row with entries leaves the shared
// factorisation (collective.c:1367-1372 wants weight == NULL || nnz == 0) for collective_closed_form_block's general branch,
//     M_i   = w C^T C (+) B^T B + sum_j (w_j - 1) b_j b_j^T + lam mult_i I,   mult_i = sum of the row's weights + absent entries (+ p)
//     rhs_i = [w U C ; sum_j (w_j x_j - (w_j - 1)(mean + bias_j)) b_j + cst]
// on the row Cholesky kernel's collective mode: w C^T C for the rows with side information (all of them), blockdiag(0, B^T B) as the
// matrix every row starts from, the entries' weight pairs as they are (entry_pairs), the right-hand sides prefilled.  Closed form.
static int update_factor_naz_weighted_side(cmfrec_hip_session *s, bool isA, bool chol)
{
    const cmfrec_hip_model &m = s->mdl;
    const DeviceInfo &dev = s->dev;
    hipStream_t st = dev.stream;
    // has_side == false (round 6, fixture g39): no side information on this side, but implicit features -- the reference still takes its
    // collective route (collective.c:8612, :8783), whose rows without entries are zeroed unless the bias / mean constant exists
    // (:1258-1268); every row then counts as "with side information" of zero attributes
    const int p_self = isA ? m.p : m.q;
    const bool has_side = p_self > 0;
    const int rows_self = isA ? m.m : m.n, rows_opp = isA ? m.n : m.m;
    const int rows_u = has_side ? (isA ? m.m_u : m.n_i) : rows_self;
    if (rows_u != rows_self) {
        g_last_error = "cmfrec_hip: NA_as_zero_X with side information: side information on exactly the rows / columns of X";
        return 2;
    }
    real_t *self = isA ? s->A.ptr : s->B.ptr;
    real_t *opp = isA ? s->B.ptr : s->A.ptr;
    const size_t ld_self = isA ? s->ldA : s->ldB, ld_opp = isA ? s->ldB : s->ldA;
    const int k_side_self = isA ? m.k_user : m.k_item, k_side_opp = isA ? m.k_item : m.k_user;
    const SparseShard &X = isA ? s->Xr : s->Xc;
    const real_t *Cm = isA ? s->C.ptr : s->D.ptr;
    const real_t *Um = isA ? s->U.ptr : s->II.ptr;
    // sparse side information (missing = absent; round 6, fixture g36): the row's attributes as the second gather source instead of the
    // shared w C^T C / w U C (collective.c:1636-1653 / :2292-2298 beside the weight branches)
    const bool sparse_side = has_side && (isA ? s->sparseU : s->sparseI);
    const SparseShard &Us = isA ? s->Usr : s->Isr;
    const real_t w = isA ? m.w_user : m.w_item;
    const bool self_bias = isA ? m.user_bias : m.item_bias, opp_bias = isA ? m.item_bias : m.user_bias;
    const real_t lam_self = s->lam6[isA ? 2 : 3];
    const real_t lam_last_self = self_bias ? s->lam6[isA ? 0 : 1] : lam_self;
    const int kk = m.k + m.k_main, ks = kk + (self_bias ? 1 : 0), kc = has_side ? k_side_self + m.k : 0, kt = k_side_self + ks;
    const real_t *oppx = opp + k_side_opp;
    if (self_bias)
        hipLaunchKernelGGL(col_fill_kernel<real_t>, grid1d(rows_opp), dim3(256), 0, st, opp, ld_opp, rows_opp, isA ? s->k_totB : s->k_totA, (real_t)1);
    launch_gram(dev, s->gws, oppx, ld_opp, rows_opp, ks, s->gram.ptr, (real_t)1, (real_t)0);
    // implicit features: w_i Bi^T Bi on the first k + k_main unknowns of the X block (closed form: inside the matrix every row starts
    // from; block CG: the kernel's own product with the unweighted matrix), w_i times the gather-sum of the opposing implicit factors in
    // the right-hand sides (closed form; the CG kernel gathers them itself)
    const real_t *Fi = s->implicit_feats ? (isA ? s->Bi.ptr : s->Ai.ptr) : nullptr;
    if (Fi != nullptr) {
        launch_gram(dev, s->gws, Fi, (size_t)kk, rows_opp, kk, s->bitbi.ptr, chol ? s->w_implicit : (real_t)1, (real_t)0);
        if (chol) {
            hipLaunchKernelGGL(add_block_kernel<real_t>, grid1d((size_t)kk * kk), dim3(256), 0, st, s->bitbi.ptr, kk, (real_t)1, s->gram.ptr, ks, 0);
            if (!launch_gsum(s, isA, X, Fi, kk)) {
                g_last_error = "cmfrec_hip: NA_as_zero_X with implicit features: k + k_main too wide for the gather-sum";
                return 2;
            }
        }
    }
    s->naz_M.alloc_at_least((size_t)kt * kt);
    hipLaunchKernelGGL(betbe_base_kernel<real_t>, grid1d((size_t)kt * kt), dim3(256), 0, st, s->gram.ptr, ks, k_side_self, (real_t)0, s->naz_M.ptr);
    const bool has_cst = opp_bias || s->naz_center;
    const real_t *bias = opp_bias ? (isA ? s->biasB.ptr : s->biasA.ptr) : nullptr;
    if (!chol) {
        // Block CG (round 6; fixture g35).  The rows WITH entries: collective_block_cg with NA_as_zero_X and weights (collective.c:2446-2493
        // first residual, :2700-2760 products) -- the shared B^T B, the corrections (w_j - 1) b_j b_j^T of the row's entries, the dense
        // side-information block through C^T C and U C -- on the lane <-> unknown kernel (CgParams::gx).  The rows WITHOUT entries run
        // the same CG from their side information and the constant: the reference factorises the shared block matrix for them only
        // when the caller hands it the buffers of precompute_for_predictions (filled_BtB, :5702-5716), which this model does not offer.
        real_t *uc = isA ? s->ucA.ptr : s->ucB.ptr;
        if (has_side && !sparse_side) {
            launch_gram(dev, s->gws, Cm, (size_t)kc, p_self, kc, s->ctc.ptr, (real_t)1, (real_t)0);                 // C^T C, unweighted
            launch_gemm<false>(dev, rows_self, kc, p_self, (real_t)1, Um, (size_t)p_self, Cm, (size_t)kc, uc, (size_t)kc);   // U C
        }
        const size_t nnzc = X.nnz;
        s->naz_g.alloc_at_least(std::max<size_t>(nnzc, 1)); s->naz_xt.alloc_at_least(std::max<size_t>(nnzc, 1));
        if (nnzc > 0)
            hipLaunchKernelGGL(naz_entry_transform_kernel<real_t>, grid1d(nnzc), dim3(256), 0, st, X.v.ptr, X.w.ptr, X.i.ptr, nnzc,
                               has_cst ? bias : nullptr, s->naz_center ? s->naz_mean : (real_t)0, s->naz_g.ptr, s->naz_xt.ptr);
        s->naz_rhs.alloc_at_least((size_t)rows_self * ks);
        HIP_CHECK(hipMemsetAsync(s->naz_rhs.ptr, 0, (size_t)rows_self * ks * sizeof(real_t), st));
        {
            CholCall cr{s->naz_rhs.ptr, (size_t)ks, oppx, ld_opp, ks, 0, nullptr, s->gram.ptr, 0, 0, 0, lam_self, lam_last_self, false, false, false, CHOL_NAZ};
            cr.rhs_only = true; cr.values_override = s->naz_xt.ptr;
            int rc = launch_chol(dev, cr, &X);
            if (rc) return rc;
        }
        if (has_cst) {
            const int nb = (rows_opp + COLSUM_ROWS - 1) / COLSUM_ROWS;
            s->naz_part.alloc_at_least((size_t)nb * ks); s->naz_vec.alloc_at_least((size_t)ks);
            hipLaunchKernelGGL(weighted_colsum_partial_kernel<real_t>, dim3(nb), dim3(256), 0, st, oppx, ld_opp, rows_opp, ks, bias,
                               s->naz_center ? s->naz_mean : (real_t)0, s->naz_part.ptr);
            hipLaunchKernelGGL(colsum_finish_kernel<real_t>, grid1d(ks), dim3(256), 0, st, s->naz_part.ptr, nb, ks, (real_t)-1, s->naz_vec.ptr);
            hipLaunchKernelGGL(add_rowvec_kernel<real_t>, grid1d((size_t)rows_self * ks), dim3(256), 0, st, s->naz_rhs.ptr, (size_t)ks, (size_t)rows_self, ks,
                               s->naz_vec.ptr);
        }
        const size_t nz = std::max<size_t>(nnzc, 1);
        if (s->naz_zero.n < nz) {
            s->naz_zero.alloc(nz);
            HIP_CHECK(hipMemsetAsync(s->naz_zero.ptr, 0, nz * sizeof(real_t), st));
        }
        HIP_CHECK(hipGetLastError());
        const bool scaled_cg = m.scale_lam || m.scale_lam_sideinfo;
        CgCall c{self, ld_self, oppx, ld_opp, ks, nullptr, nullptr, lam_self, lam_last_self, scaled_cg, false, m.max_cg_steps, false,
                 (bool)m.precondition_cg};
        c.koff = k_side_self; c.kc = kc; c.w_side = w; c.rows_with_u = rows_u; c.p_side = p_self;
        if (sparse_side) { c.X2 = &Us; c.C2 = Cm; } else if (has_side) { c.CtC = s->ctc.ptr; c.UC = uc; }
        if (Fi != nullptr) { c.Bi = Fi; c.BiTBi = s->bitbi.ptr; c.ki = kk; c.w_imp = s->w_implicit; }
        c.scale_lam_sideinfo = (bool)m.scale_lam_sideinfo;
        c.Gx = s->gram.ptr; c.rconst_x = s->naz_rhs.ptr; c.ldr_x = (size_t)ks;
        c.values_override = s->naz_zero.ptr; c.weights_override = s->naz_g.ptr;      // (the multipliers: X.wsum_naz, launch_cg)
        c.gx_all_rows = has_cst;
        return launch_cg(dev, c, X);
    }
    if (has_side && !sparse_side) launch_gram(dev, s->gws, Cm, (size_t)kc, p_self, kc, s->ctc.ptr, w, (real_t)0);
    // right-hand sides start from [w U C ; cst]  (sparse side information: [0 ; cst], the attributes are gathered by the row kernel)
    HIP_CHECK(hipMemset2DAsync(self, ld_self * sizeof(real_t), 0, (size_t)kt * sizeof(real_t), (size_t)rows_self, st));
    if (has_side && !sparse_side) launch_gemm<false>(dev, rows_self, kc, p_self, w, Um, (size_t)p_self, Cm, (size_t)kc, self, ld_self);
    if (Fi != nullptr)
        hipLaunchKernelGGL(add_cols_scaled_kernel<real_t>, grid1d((size_t)X.nrows * kk), dim3(256), 0, st, self, ld_self, k_side_self, s->grhs.ptr, kk,
                           s->w_implicit, (size_t)X.nrows);
    if (has_cst) {
        const int nb = (rows_opp + COLSUM_ROWS - 1) / COLSUM_ROWS;
        s->naz_part.alloc_at_least((size_t)nb * ks); s->naz_vec.alloc_at_least((size_t)ks);
        hipLaunchKernelGGL(weighted_colsum_partial_kernel<real_t>, dim3(nb), dim3(256), 0, st, oppx, ld_opp, rows_opp, ks, bias,
                           s->naz_center ? s->naz_mean : (real_t)0, s->naz_part.ptr);
        hipLaunchKernelGGL(colsum_finish_kernel<real_t>, grid1d(ks), dim3(256), 0, st, s->naz_part.ptr, nb, ks, (real_t)-1, s->naz_vec.ptr);
        hipLaunchKernelGGL(add_rowvec_kernel<real_t>, grid1d((size_t)rows_self * ks), dim3(256), 0, st, self + k_side_self, ld_self, (size_t)rows_self, ks,
                           s->naz_vec.ptr);
    }
    const size_t nnz = X.nnz;
    s->naz_g.alloc_at_least(std::max<size_t>(nnz, 1)); s->naz_xt.alloc_at_least(std::max<size_t>(nnz, 1));
    if (nnz > 0)
        hipLaunchKernelGGL(naz_entry_transform_kernel<real_t>, grid1d(nnz), dim3(256), 0, st, X.v.ptr, X.w.ptr, X.i.ptr, nnz,
                           has_cst ? bias : nullptr, s->naz_center ? s->naz_mean : (real_t)0, s->naz_g.ptr, s->naz_xt.ptr);
    HIP_CHECK(hipGetLastError());
    const bool scaled = m.scale_lam || m.scale_lam_sideinfo;
    CholCall c{self, ld_self, oppx, ld_opp, kt, k_side_self, nullptr, (sparse_side || !has_side) ? nullptr : s->ctc.ptr, kc, rows_u, p_self, lam_self, lam_last_self,
               scaled, (bool)m.scale_lam_sideinfo, false, CHOL_COLLECTIVE, s->naz_M.ptr};
    c.rhs_prefilled_all = true; c.entry_pairs = true; c.values_override = s->naz_xt.ptr; c.weights_override = s->naz_g.ptr;
    if (sparse_side) { c.X2 = &Us; c.B2 = Cm; c.ldb2 = (size_t)kc; c.kc2 = kc; c.w2 = w; }
    if (scaled) c.mult_override = X.wsum_naz.ptr;        // sum of the row's weights + its absent entries (cmfrec_hip_session_set_NA_as_zero_X)
    return launch_chol(dev, c, &X);
}

// The same half-step WITH observation weights and without side information (optimizeA Case 4 with NA_as_zero && weight,
// common.c:3209-3302; driver collective.c:8573-8600 + :8680-8717, :8756-8787 + :8847-8876): an absent entry is a zero of weight
// one, so every row's system is the shared opp^T opp plus the correction of its present entries,
//     M_i   = opp^T opp + sum_j (w_j - 1) opp_j opp_j^T + diag(lam_i .. lam_i, lam_last_i)
//     rhs_i = sum_j [w_j x_j - (w_j - 1)(mean + bias_j)] opp_j + cst,    cst = - sum over ALL opposing rows of (bias + mean) x row
// (factors_closed_form :846-907; factors_explicit_cg_NA_as_zero_weighted :1293-1441), lam_i = lam x (sum of the row's weights +
// number of its absent entries) under scale_lam.  Closed form: the row Cholesky kernel in its CHOL_NAZ_W mode (B^T B as the
// initial matrix, the corrections on the matrix cores).  CG: the right-hand sides by the gather-only launch, then the tiled CG
// kernels of the explicit model with the shared matrix in LDS (GRAMX builds), weights w - 1 and zero values --
//     r = rhs_i - M_i a,   Ap = M_i p
// are the reference's :1321-1371 / :1391-1406 regrouped.  Rows without entries are solved when cst exists (:3270-3271).
static int update_factor_naz_weighted(cmfrec_hip_session *s, bool isA, bool chol)
{
    const cmfrec_hip_model &m = s->mdl;
    const DeviceInfo &dev = s->dev;
    hipStream_t st = dev.stream;
    const int p_self = isA ? m.p : m.q;
    if (dev.nonneg_now || dev.l1_now != (real_t)0 || dev.l1_last_now != (real_t)0 || s->side_local) {
        g_last_error = "cmfrec_hip: NA_as_zero_X with observation weights: the model without nonneg / L1";
        return 2;
    }
    if (p_self > 0 || s->implicit_feats) return update_factor_naz_weighted_side(s, isA, chol);
    const int rows_self = isA ? m.m : m.n, rows_opp = isA ? m.n : m.m;
    real_t *self = isA ? s->A.ptr : s->B.ptr;
    real_t *opp = (isA ? s->B.ptr : s->A.ptr) + (isA ? m.k_item : m.k_user);      // the columns X refers to (the other side may carry k_item / k_user)
    const size_t ld_self = isA ? s->ldA : s->ldB, ld_opp = isA ? s->ldB : s->ldA;
    const SparseShard &X = isA ? s->Xr : s->Xc;
    const bool self_bias = isA ? m.user_bias : m.item_bias, opp_bias = isA ? m.item_bias : m.user_bias;
    const real_t lam_self = s->lam6[isA ? 2 : 3];
    const real_t lam_last_self = self_bias ? s->lam6[isA ? 0 : 1] : lam_self;
    const int ks = m.k + m.k_main + (self_bias ? 1 : 0);
    if (!chol && ks > 64 && !m.precondition_cg) {
        g_last_error = "cmfrec_hip: NA_as_zero_X with observation weights under CG: at most 64 unknowns per row (the preconditioned solver takes more)";
        return 2;
    }
    if (self_bias)                            // the opposing bias column is fixed to 1 (collective.c:8538-8543, :8728-8732)
        hipLaunchKernelGGL(col_fill_kernel<real_t>, grid1d(rows_opp), dim3(256), 0, st, isA ? s->B.ptr : s->A.ptr, ld_opp, rows_opp,
                           isA ? s->k_totB : s->k_totA, (real_t)1);
    launch_gram(dev, s->gws, opp, ld_opp, rows_opp, ks, s->gram.ptr, (real_t)1, (real_t)0);       // :3233-3236, no diagonal
    // cst (bias_BtX) and the per-entry pairs
    const bool has_cst = opp_bias || s->naz_center;
    const real_t *bias = opp_bias ? (isA ? s->biasB.ptr : s->biasA.ptr) : nullptr;
    if (has_cst) {
        const int nb = (rows_opp + COLSUM_ROWS - 1) / COLSUM_ROWS;
        s->naz_part.alloc_at_least((size_t)nb * ks); s->naz_vec.alloc_at_least((size_t)ks);
        hipLaunchKernelGGL(weighted_colsum_partial_kernel<real_t>, dim3(nb), dim3(256), 0, st, opp, ld_opp, rows_opp, ks, bias,
                           s->naz_center ? s->naz_mean : (real_t)0, s->naz_part.ptr);
        hipLaunchKernelGGL(colsum_finish_kernel<real_t>, grid1d(ks), dim3(256), 0, st, s->naz_part.ptr, nb, ks, (real_t)-1, s->naz_vec.ptr);
    }
    const real_t *cst = has_cst ? s->naz_vec.ptr : nullptr;
    const size_t nnz = X.nnz;
    s->naz_g.alloc_at_least(std::max<size_t>(nnz, 1)); s->naz_xt.alloc_at_least(std::max<size_t>(nnz, 1));
    if (nnz > 0)
        hipLaunchKernelGGL(naz_entry_transform_kernel<real_t>, grid1d(nnz), dim3(256), 0, st, X.v.ptr, X.w.ptr, X.i.ptr, nnz,
                           has_cst ? bias : nullptr, s->naz_center ? s->naz_mean : (real_t)0, s->naz_g.ptr, s->naz_xt.ptr);
    HIP_CHECK(hipGetLastError());
    if (chol) {
        // right-hand sides start from cst (or zero), every row that is solved is overwritten
        const int nsolve = has_cst ? X.nrows : X.n_nonempty;
        s->naz_rhs.alloc_at_least((size_t)rows_self * ks);
        hipLaunchKernelGGL(set_rowvec_kernel<real_t>, grid1d((size_t)rows_self * ks), dim3(256), 0, st, s->naz_rhs.ptr, (size_t)ks, (size_t)rows_self, ks, cst);
        // (rows without entries and without cst keep their factors: the kernel solves into naz_rhs, copied back below for the solved rows)
        CholCall c{s->naz_rhs.ptr, (size_t)ks, opp, ld_opp, ks, 0, nullptr, s->gram.ptr, 0, 0, 0, lam_self, lam_last_self, m.scale_lam != 0, false,
                   s->scale_bias_const, CHOL_NAZ_W};
        c.values_override = s->naz_xt.ptr; c.weights_override = s->naz_g.ptr; c.rhs_prefilled_all = true; c.all_rows = has_cst;
        int rc = launch_chol(dev, c, &X);
        if (rc) return rc;
        hipLaunchKernelGGL(copy_rows_by_order_kernel<real_t>, grid1d((size_t)nsolve * ks), dim3(256), 0, st, s->naz_rhs.ptr, (size_t)ks, self, ld_self,
                           X.order.ptr, nsolve, ks);
        HIP_CHECK(hipGetLastError());
        return 0;
    }
    // CG: rhs_i by the gather-only launch (sum_j xt_j opp_j) + cst ...
    s->naz_rhs.alloc_at_least((size_t)rows_self * ks);
    HIP_CHECK(hipMemsetAsync(s->naz_rhs.ptr, 0, (size_t)rows_self * ks * sizeof(real_t), st));
    {
        CholCall c{s->naz_rhs.ptr, (size_t)ks, opp, ld_opp, ks, 0, nullptr, s->gram.ptr, 0, 0, 0, lam_self, lam_last_self, false, false, false, CHOL_NAZ};
        c.rhs_only = true; c.values_override = s->naz_xt.ptr;
        int rc = launch_chol(dev, c, &X);
        if (rc) return rc;
    }
    if (has_cst)
        hipLaunchKernelGGL(add_rowvec_kernel<real_t>, grid1d((size_t)rows_self * ks), dim3(256), 0, st, s->naz_rhs.ptr, (size_t)ks, (size_t)rows_self, ks, cst);
    // ... then the tiled kernels: shared matrix in LDS, rank-1 weights w - 1, values zero
    if (s->naz_zero.n < std::max<size_t>(nnz, 1)) {
        s->naz_zero.alloc(std::max<size_t>(nnz, 1));
        HIP_CHECK(hipMemsetAsync(s->naz_zero.ptr, 0, std::max<size_t>(nnz, 1) * sizeof(real_t), st));
    }
    CgCall c{self, ld_self, opp, ld_opp, ks, nullptr, nullptr, lam_self, lam_last_self, m.scale_lam != 0, s->scale_bias_const, m.max_cg_steps, false};
    c.Gx = s->gram.ptr; c.rconst_x = s->naz_rhs.ptr; c.ldr_x = (size_t)ks;
    c.values_override = s->naz_zero.ptr; c.weights_override = s->naz_g.ptr;
    // precondition_cg (round 6; factors_explicit_pcg_NA_as_zero_weighted, common.c:1443-1613): the lane <-> unknown kernel with the
    // shared matrix, its diagonal in the Jacobi preconditioner, the rows without entries in the same launch when the constant exists
    c.precond = m.precondition_cg != 0; c.gx_all_rows = c.precond && has_cst;
    int rc = launch_cg(dev, c, X);
    if (rc) return rc;
    if (!c.precond && has_cst && X.nrows > X.n_nonempty) {
        const int cnt = X.nrows - X.n_nonempty;
        hipLaunchKernelGGL(cg_shared_matrix_rows_kernel<real_t>, dim3((cnt + 3) / 4), dim3(256), 0, st, self, ld_self, X.order.ptr + X.n_nonempty, cnt, ks,
                           s->gram.ptr, cst, lam_self, lam_last_self, m.scale_lam ? X.wsum_naz.ptr : nullptr, s->scale_bias_const ? 1 : 0, m.max_cg_steps);
        HIP_CHECK(hipGetLastError());
    }
    return 0;
}

static int update_factor(cmfrec_hip_session *s, bool isA, bool chol, int part = -1)
{
    if (s->naz_X && !s->mdl.implicit && s->Xr.weighted()) return update_factor_naz_weighted(s, isA, chol);
    if (s->naz_X && !s->mdl.implicit) return update_factor_naz(s, isA, chol);
    const cmfrec_hip_model &m = s->mdl;
    const DeviceInfo &dev = s->dev;
    hipStream_t st = dev.stream;
    // "self" = matrix being updated, "opp" = the fixed one
    real_t *self = isA ? s->A.ptr : s->B.ptr;
    real_t *opp = isA ? s->B.ptr : s->A.ptr;
    const size_t ld_self = isA ? s->ldA : s->ldB, ld_opp = isA ? s->ldB : s->ldA;
    const int rows_opp = isA ? (m.n_x > 0 ? m.n_x : m.n) : (m.m_x > 0 ? m.m_x : m.m);   // rows of the opposing matrix that X refers to (the Gramian's rows)
    const int k_side_self = isA ? m.k_user : m.k_item, k_side_opp = isA ? m.k_item : m.k_user;
    const int begin = isA ? m.row_begin : m.col_begin;
    const SparseShard &X = (part >= 0) ? *s->XrParts[part] : (isA ? s->Xr : s->Xc);
    const bool self_bias = isA ? m.user_bias : m.item_bias;
    const bool opp_bias = isA ? m.item_bias : m.user_bias;
    const int p_self = isA ? m.p : m.q;
    // lam_unique[2] / [3] for A / B; the bias, when fitted, takes lam_unique[0] / [1] (collective.c:8649-8654, :8820-8825)
    const bool sbc = s->scale_bias_const && !m.implicit;      // rows without side information: common.c:679-723
    const real_t lam_self = s->lam6[isA ? 2 : 3];
    const real_t lam_last_self = (!m.implicit && self_bias) ? s->lam6[isA ? 0 : 1] : lam_self;
    real_t *self_blk = self + (size_t)(begin + (part >= 0 ? s->partBegin[part] : 0)) * ld_self;
    if (part >= 0 && (!isA || p_self > 0)) {
        g_last_error = "cmfrec_hip: row parts are only built for A-steps without side information";
        return 2;
    }
    const int kk = m.k + m.k_main;
    const int rows_x_self = isA ? (m.m_x > 0 ? m.m_x : m.m) : (m.n_x > 0 ? m.n_x : m.n);          // rows of this matrix X has

    const bool sparse_side = isA ? s->sparseU : s->sparseI;
    if (p_self > 0 && sparse_side) {
        // sparse side information (missing = absent): the row's attributes are a second gather source of the same
        // Cholesky launch (collective.c:1636-1653, :1719-1731 / :2003-2021) or of the lane <-> unknown CG kernel
        const real_t *Cm = isA ? s->C.ptr : s->D.ptr;
        const SparseShard &Us = isA ? s->Usr : s->Isr;
        const int rows_u = isA ? m.m_u : m.n_i;
        const int kc = k_side_self + m.k;
        const real_t w = isA ? m.w_user : m.w_item;
        if (!chol) {
            // block CG / PCG with the attributes as a second gathered term (collective_block_cg u_vec_sp branches,
            // collective.c:2292-2298, :2609-2621, :2847-2860; implicit: :2993-2999 ff.), generic kernel
            const real_t *bias_sub_cg = nullptr;
            int kx = kk;
            if (m.implicit) {
                launch_gram(dev, s->gws, opp + k_side_opp, ld_opp, rows_opp, kk, s->gram.ptr, (real_t)1, (real_t)0);
            } else {
                if (self_bias) {
                    const int rows_fill = isA ? m.n : m.m;
                    hipLaunchKernelGGL(col_fill_kernel<real_t>, grid1d(rows_fill), dim3(256), 0, st, opp, ld_opp, rows_fill,
                                       isA ? s->k_totB : s->k_totA, (real_t)1);
                    kx += 1;
                }
                if (opp_bias) bias_sub_cg = isA ? s->biasB.ptr : s->biasA.ptr;
            }
            CgCall c{self_blk, ld_self, opp + k_side_opp, ld_opp, kx, bias_sub_cg, m.implicit ? s->gram.ptr : nullptr,
                     lam_self, lam_last_self, (bool)(m.scale_lam || m.scale_lam_sideinfo), false, m.max_cg_steps, (bool)m.implicit,
                     (bool)m.precondition_cg};
            c.koff = k_side_self; c.kc = kc; c.w_side = w; c.rows_with_u = rows_u; c.p_side = p_self;
            c.scale_lam_sideinfo = (bool)m.scale_lam_sideinfo; c.X2 = &Us; c.C2 = Cm;
            if (s->implicit_feats && !m.implicit) {             // collective.c:2301-2304, :2624-2643, :2862-2868 (round 6: with u_vec_sp too)
                const real_t *Fc = isA ? s->Bi.ptr : s->Ai.ptr;
                launch_gram(dev, s->gws, Fc, (size_t)kk, rows_opp, kk, s->bitbi.ptr, (real_t)1, (real_t)0);
                c.Bi = Fc; c.BiTBi = s->bitbi.ptr; c.ki = kk; c.w_imp = s->w_implicit;
                if (part < 0 && launch_gsum(s, isA, X, Fc, kk)) c.gsum = s->grhs.ptr;
            }
            return launch_cg_any(dev, c, X, nullptr);
        }
        if (m.implicit) {
            if (X.nrows) HIP_CHECK(hipMemsetAsync(self_blk, 0, (size_t)X.nrows * ld_self * sizeof(real_t), st));   // :6018-6019
            const int kt = k_side_self + kk;
            launch_gram(dev, s->gws, opp + k_side_opp, ld_opp, rows_opp, kk, s->gram.ptr, (real_t)1, lam_self);
            hipLaunchKernelGGL(betbe_base_kernel<real_t>, grid1d(kt * kt), dim3(256), 0, st, s->gram.ptr, kk, k_side_self, lam_self,
                               s->betbe.ptr);
            CholCall c{self_blk, ld_self, opp + k_side_opp, ld_opp, kt, k_side_self, nullptr, nullptr, kc, rows_u, p_self, lam_self,
                       lam_last_self, false, false, false, CHOL_COLLECTIVE_IMPLICIT, s->betbe.ptr};
            c.X2 = &Us; c.B2 = Cm; c.ldb2 = (size_t)kc; c.kc2 = kc; c.w2 = w;
            return launch_chol(dev, c, &X);
        }
        // (the explicit model's closed form: below, after the implicit-features term it may carry)
    }
    if (p_self > 0 && !chol) {
        // block CG on the collective system, dense full side information: collective_block_cg (explicit,
        // collective.c:2134-2903) / collective_block_cg_implicit (:2905-3303), prefer_CtC branch
        const real_t *Cm = isA ? s->C.ptr : s->D.ptr;
        const real_t *Um = isA ? s->U.ptr : s->II.ptr;
        const int rows_u = isA ? m.m_u : m.n_i;
        const int kc = k_side_self + m.k;
        const real_t w = isA ? m.w_user : m.w_item;
        const int local_u = std::max(0, std::min(rows_u - begin, X.nrows));
        real_t *uc = isA ? s->ucA.ptr : s->ucB.ptr;
        launch_gram(dev, s->gws, Cm, (size_t)kc, p_self, kc, s->ctc.ptr, (real_t)1, (real_t)0);   // C^T C, unweighted
        launch_gemm<false>(dev, local_u, kc, p_self, (real_t)1, Um + (size_t)(s->side_local ? 0 : begin) * p_self, (size_t)p_self, Cm, (size_t)kc,
                           uc, (size_t)kc);                                                        // U C
        const real_t *bias_sub_cg = nullptr;
        int kx = kk;
        if (m.implicit) {
            launch_gram(dev, s->gws, opp + k_side_opp, ld_opp, rows_opp, kk, s->gram.ptr, (real_t)1, (real_t)0);
        } else {
            if (self_bias) {
                const int rows_fill = isA ? m.n : m.m;
                hipLaunchKernelGGL(col_fill_kernel<real_t>, grid1d(rows_fill), dim3(256), 0, st, opp, ld_opp, rows_fill,
                                   isA ? s->k_totB : s->k_totA, (real_t)1);
                HIP_CHECK(hipGetLastError());
                kx += 1;
            }
            if (opp_bias) bias_sub_cg = isA ? s->biasB.ptr : s->biasA.ptr;
        }
        CgCall c{self_blk, ld_self, opp + k_side_opp, ld_opp, kx, bias_sub_cg, m.implicit ? s->gram.ptr : nullptr,
                 lam_self, lam_last_self, (bool)(m.scale_lam || m.scale_lam_sideinfo), sbc, m.max_cg_steps, (bool)m.implicit,
                 (bool)m.precondition_cg};
        // explicit model: rows beyond X are not part of the block system (solve_sideinfo_only_rows)
        const int local_x = std::max(0, std::min(rows_x_self - begin, X.nrows));
        const int local_u_main = m.implicit ? local_u : std::min(local_u, local_x);
        c.koff = k_side_self; c.kc = kc; c.CtC = s->ctc.ptr; c.UC = uc; c.w_side = w; c.rows_with_u = local_u_main;
        c.p_side = p_self; c.scale_lam_sideinfo = (bool)m.scale_lam_sideinfo;
        if (s->implicit_feats && !m.implicit) {                 // collective.c:2301-2304, :2624-2643, :2862-2868
            const real_t *Fi = isA ? s->Bi.ptr : s->Ai.ptr;
            launch_gram(dev, s->gws, Fi, (size_t)kk, rows_opp, kk, s->bitbi.ptr, (real_t)1, (real_t)0);
            c.Bi = Fi; c.BiTBi = s->bitbi.ptr; c.ki = kk; c.w_imp = s->w_implicit;
            if (part < 0 && launch_gsum(s, isA, X, Fi, kk)) c.gsum = s->grhs.ptr;
        }
        int rc = launch_cg_any(dev, c, X, nullptr);
        if (rc) return rc;
        return solve_sideinfo_only_rows(s, isA, false, local_u_main, local_u - local_u_main);
    }
    if (m.implicit && p_self > 0) {
        // optimizeA_collective_implicit, Cholesky, dense full side information (collective.c:5971-6244)
        const real_t *Cm = isA ? s->C.ptr : s->D.ptr;
        const real_t *Um = isA ? s->U.ptr : s->II.ptr;
        const int rows_u = isA ? m.m_u : m.n_i;
        const int kc = k_side_self + m.k, kt = k_side_self + kk;
        const real_t w = isA ? m.w_user : m.w_item;
        launch_gram(dev, s->gws, opp + k_side_opp, ld_opp, rows_opp, kk, s->gram.ptr, (real_t)1, lam_self);     // :6056-6061
        hipLaunchKernelGGL(betbe_base_kernel<real_t>, grid1d(kt * kt), dim3(256), 0, st, s->gram.ptr, kk, k_side_self, lam_self,
                           s->betbe.ptr);                                                           // :6121-6135
        launch_gram(dev, s->gws, Cm, (size_t)kc, p_self, kc, s->ctc.ptr, w, (real_t)0);            // :6138-6160
        if (X.nrows) HIP_CHECK(hipMemsetAsync(self_blk, 0, (size_t)X.nrows * ld_self * sizeof(real_t), st));   // :6018-6019
        const int local_u = std::max(0, std::min(rows_u - begin, X.nrows));
        launch_gemm<false>(dev, local_u, kc, p_self, w, Um + (size_t)(s->side_local ? 0 : begin) * p_self, (size_t)p_self, Cm, (size_t)kc,
                           self_blk, ld_self);                                                      // :6163-6168
        CholCall c{self_blk, ld_self, opp + k_side_opp, ld_opp, kt, k_side_self, nullptr, s->ctc.ptr, kc, local_u,
                   p_self, lam_self, lam_last_self, false, false, false, CHOL_COLLECTIVE_IMPLICIT, s->betbe.ptr};
        return launch_chol(dev, c, &X);
    }
    if (m.implicit) {
        // optimizeA_implicit, common.c:3305-3421 (the Gramian once per half-step: parts > 0 reuse it)
        if (part <= 0)
            launch_gram(dev, s->gws, opp + k_side_opp, ld_opp, rows_opp, kk, s->gram.ptr, (real_t)1, chol ? lam_self : (real_t)0);
        if (chol) {
            if (X.nrows) HIP_CHECK(hipMemsetAsync(self_blk, 0, (size_t)X.nrows * ld_self * sizeof(real_t), st)); // :3334
            CholCall c{self_blk + k_side_self, ld_self, opp + k_side_opp, ld_opp, kk, 0, nullptr,
                       s->gram.ptr, 0, 0, 0, lam_self, lam_last_self, false, false, false, CHOL_IMPLICIT};
            return launch_chol(dev, c, &X);
        }
        CgCall c{self_blk + k_side_self, ld_self, opp + k_side_opp, ld_opp, kk, nullptr, s->gram.ptr,
                 lam_self, lam_last_self, false, false, m.max_cg_steps, true, (bool)m.precondition_cg};
        return launch_cg_any(dev, c, X, isA ? &s->binA : &s->binB);
    }

    // ---- explicit ----
    // the opposing matrix' bias column is fixed to 1 while this one's bias is fitted (collective.c:8538-8543, :8728-8732)
    if (self_bias) {
        const int rows_fill = isA ? m.n : m.m;
        hipLaunchKernelGGL(col_fill_kernel<real_t>, grid1d(rows_fill), dim3(256), 0, st, opp, ld_opp, rows_fill,
                           isA ? s->k_totB : s->k_totA, (real_t)1);
        HIP_CHECK(hipGetLastError());
    }
    const real_t *bias_sub = nullptr;
    if (opp_bias) bias_sub = isA ? s->biasB.ptr : s->biasA.ptr;             // fused "X - bias" (:8566-8570, :8750-8754)
    const int ksolve = kk + (self_bias ? 1 : 0);
    // implicit features: w_i Bi^T Bi joins the X block of every row's matrix (collective.c:1704-1707) and
    // w_i sum_{j observed} Bi_j its right-hand side (:1757-1771).  The second gather source of the Cholesky launch reads
    // the same sparsity pattern with unit values from Bi, right-hand side only.
    const real_t *Fi = nullptr;
    if (s->implicit_feats && chol) {
        Fi = isA ? s->Bi.ptr : s->Ai.ptr;
        const int kt_i = k_side_self + ksolve;
        launch_gram(dev, s->gws, Fi, (size_t)kk, rows_opp, kk, s->bitbi.ptr, s->w_implicit, (real_t)0);
        hipLaunchKernelGGL(embed_block_kernel<real_t>, grid1d(kt_i * kt_i), dim3(256), 0, st, s->bitbi.ptr, kk, k_side_self,
                           kt_i, s->bitbi_full.ptr);
        HIP_CHECK(hipGetLastError());
    }
    auto add_implicit_term = [&](CholCall &c) {
        if (Fi == nullptr) return;
        c.Mfull = s->bitbi_full.ptr;
        auto &G = s->gsegs[isA ? 0 : 1];
        if (kk <= 64 * GSUM_MAXC && part < 0) {
            // the right-hand-side term by the segmented gather-sum, added to the (zeroed / w U C) rows the launch then starts
            // from: the row kernel gathers X only (first version: Bi as a second gather source of the same launch, c3 +
            // implicit features A-step 34.4 ms, B-step 20.5 ms)
            if (G.nseg > 0)
                hipLaunchKernelGGL(gather_sum_segments_kernel<real_t>, dim3((G.nseg + 3) / 4), dim3(256), 0, st, X.p.ptr, X.i.ptr, Fi,
                                   (size_t)kk, kk, G.seg_row.ptr, G.seg_off.ptr, G.nseg, s->gpartial.ptr);
            hipLaunchKernelGGL(gather_sum_rows_kernel<real_t>, dim3((X.nrows + 3) / 4), dim3(256), 0, st, s->gpartial.ptr,
                               G.row_first.ptr, X.nrows, kk, s->grhs.ptr, (size_t)kk);
            hipLaunchKernelGGL(add_cols_scaled_kernel<real_t>, grid1d((size_t)X.nrows * kk), dim3(256), 0, st, self_blk, ld_self,
                               k_side_self, s->grhs.ptr, kk, s->w_implicit, (size_t)X.nrows);
            c.rhs_prefilled_all = true;
            return;
        }
        c.X2 = &X; c.values2 = s->ones.ptr; c.B2 = Fi; c.ldb2 = (size_t)kk; c.kc2 = kk; c.koff2 = k_side_self;
        c.w2 = s->w_implicit; c.w2_syr_zero = true; c.rows2 = X.nrows;
    };
    if (p_self > 0 && sparse_side) {
        // sparse side information, closed form: the row's attributes as the second gather source (collective.c:1636-1653, :1719-1731);
        // round 6: together with the implicit-features term (its right-hand side by the segmented gather-sum)
        const real_t *Cm = isA ? s->C.ptr : s->D.ptr;
        const SparseShard &Us = isA ? s->Usr : s->Isr;
        const int rows_u = isA ? m.m_u : m.n_i;
        const int kc = k_side_self + m.k, kt = k_side_self + ksolve;
        if (X.nrows) HIP_CHECK(hipMemsetAsync(self_blk, 0, (size_t)X.nrows * ld_self * sizeof(real_t), st));   // :4817-4822
        CholCall c{self_blk, ld_self, opp + k_side_opp, ld_opp, kt, k_side_self, bias_sub, nullptr, kc, rows_u, p_self, lam_self,
                   lam_last_self, (bool)(m.scale_lam || m.scale_lam_sideinfo), (bool)m.scale_lam_sideinfo, false, CHOL_COLLECTIVE};
        add_implicit_term(c);
        if (c.X2 != nullptr) {
            g_last_error = "cmfrec_hip: implicit features with sparse side information: k + k_main beyond the gather-sum's width";
            return 2;
        }
        c.X2 = &Us; c.B2 = Cm; c.ldb2 = (size_t)kc; c.kc2 = kc; c.w2 = isA ? m.w_user : m.w_item;
        return launch_chol(dev, c, &X);
    }
    if (p_self > 0) {
        // optimizeA_collective general branch, Cholesky (collective.c:5566-5968)
        const real_t *Cm = isA ? s->C.ptr : s->D.ptr;
        const real_t *Um = isA ? s->U.ptr : s->II.ptr;
        const int rows_u = isA ? m.m_u : m.n_i;
        const int kc = k_side_self + m.k, kt = k_side_self + ksolve;
        const real_t w = isA ? m.w_user : m.w_item;
        launch_gram(dev, s->gws, Cm, (size_t)kc, p_self, kc, s->ctc.ptr, w, (real_t)0);            // :5658-5668
        if (X.nrows) HIP_CHECK(hipMemsetAsync(self_blk, 0, (size_t)X.nrows * ld_self * sizeof(real_t), st));   // :4817-4822
        const int local_u = std::max(0, std::min(rows_u - begin, X.nrows));
        launch_gemm<false>(dev, local_u, kc, p_self, w, Um + (size_t)(s->side_local ? 0 : begin) * p_self, (size_t)p_self, Cm, (size_t)kc,
                           self_blk, ld_self);                                                      // :5768-5773
        const int local_x = std::max(0, std::min(rows_x_self - begin, X.nrows));
        const int local_u_main = std::min(local_u, local_x);              // rows beyond X: solve_sideinfo_only_rows
        CholCall c{self_blk, ld_self, opp + k_side_opp, ld_opp, kt, k_side_self, bias_sub, s->ctc.ptr, kc, local_u_main,
                   p_self, lam_self, lam_last_self, (bool)(m.scale_lam || m.scale_lam_sideinfo), (bool)m.scale_lam_sideinfo, sbc,
                   CHOL_COLLECTIVE};
        add_implicit_term(c);
        // rows with few entries against many unknowns: low-rank path (lowrank_kernels.hpp)
        if (Fi == nullptr && !sbc && local_u_main == X.nrows && part < 0) {
            // the other side's w D^T D, if its next half-step will want the eigenvectors and they are not there yet: decomposed
            // behind this side's (EigCache)
            EigCache *mine = isA ? &s->eigA : &s->eigB, *next = isA ? &s->eigB : &s->eigA;
            const int p_next = isA ? m.q : m.p, kc_next = (isA ? m.k_item : m.k_user) + m.k;
            if (next->wanted && !next->fresh && p_next > 0 && !(isA ? s->sparseI : s->sparseU)) {
                next->M.alloc_at_least((size_t)kc_next * kc_next);
                launch_gram(dev, s->gws, isA ? s->D.ptr : s->C.ptr, (size_t)kc_next, p_next, kc_next, next->M.ptr, isA ? m.w_item : m.w_user,
                            (real_t)0);
            } else {
                next = nullptr;
            }
            const int rc_lr = launch_collective_lowrank(dev, s->lr, c, X, Cm, Um + (size_t)(s->side_local ? 0 : begin) * p_self, p_self, w, m.k,
                                                        isA ? m.n : m.m, mine, next, kc_next);
            if (rc_lr >= 0) return rc_lr;
        }
        int rc = launch_chol(dev, c, &X);
        if (rc) return rc;
        return solve_sideinfo_only_rows(s, isA, true, local_u_main, local_u - local_u_main);
    }
    const bool scale_lam = m.scale_lam || m.scale_lam_sideinfo;                                     // :7465
    if (Fi != nullptr) {
        // without side information on this side the reference still takes optimizeA_collective (collective.c:8612, :8783)
        if (X.nrows) HIP_CHECK(hipMemsetAsync(self_blk, 0, (size_t)X.nrows * ld_self * sizeof(real_t), st));   // :4817-4822
        CholCall c{self_blk, ld_self, opp + k_side_opp, ld_opp, ksolve, 0, bias_sub, nullptr, 0, 0, 0,
                   lam_self, lam_last_self, scale_lam, false, false, CHOL_COLLECTIVE};
        add_implicit_term(c);
        return launch_chol(dev, c, &X);
    }
    if (chol) {
        CholCall c{self_blk + k_side_self, ld_self, opp + k_side_opp, ld_opp, ksolve, 0, bias_sub, nullptr, 0, 0, 0,
                   lam_self, lam_last_self, scale_lam, false, sbc, CHOL_EXPLICIT};
        return launch_chol(dev, c, &X);
    }
    CgCall c{self_blk + k_side_self, ld_self, opp + k_side_opp, ld_opp, ksolve, bias_sub, nullptr,
             lam_self, lam_last_self, scale_lam, sbc, m.max_cg_steps, false, (bool)m.precondition_cg};
    if (s->implicit_feats) {
        // block CG with the implicit-features term (collective_block_cg without side information on this side,
        // collective.c:2624-2643, :2862-2868): generic kernel; rows without entries are zeroed (:1258-1268)
        const real_t *Fc = isA ? s->Bi.ptr : s->Ai.ptr;
        launch_gram(dev, s->gws, Fc, (size_t)kk, rows_opp, kk, s->bitbi.ptr, (real_t)1, (real_t)0);
        c.Bi = Fc; c.BiTBi = s->bitbi.ptr; c.ki = kk; c.w_imp = s->w_implicit;
        if (part < 0 && launch_gsum(s, isA, X, Fc, kk)) c.gsum = s->grhs.ptr;
    }
    return launch_cg_any(dev, c, X, isA ? &s->binA : &s->binB);
}

// Ai / Bi update: optimizeA Case 3 on the binary indicator of X (collective.c:8448-8534; common.c:3116-3205): one
// shared matrix F^T F + lam/w_i (x rows of F under scale_lam), right-hand sides sum_{j observed} F_j
static int update_implicit_feats(cmfrec_hip_session *s, bool isAi)
{
    const cmfrec_hip_model &m = s->mdl;
    const DeviceInfo &dev = s->dev;
    const int kk = m.k + m.k_main;
    real_t *self = isAi ? s->Ai.ptr : s->Bi.ptr;
    const real_t *F = isAi ? s->B.ptr + m.k_item : s->A.ptr + m.k_user;      // B_bias + k_item / A_bias + k_user
    const size_t ldf = isAi ? s->ldB : s->ldA;
    const int rows_f = isAi ? m.n : m.m, rows_self = isAi ? m.m : m.n;
    const SparseShard &X = isAi ? s->Xr : s->Xc;
    const bool scale_lam = m.scale_lam || m.scale_lam_sideinfo;
    const real_t lam = (s->lam6[isAi ? 2 : 3] / s->w_implicit) * (scale_lam ? (real_t)rows_f : (real_t)1);   // :8469, :8510
    launch_gram(dev, s->gws, F, ldf, rows_f, kk, s->gram.ptr, (real_t)1, lam);
    HIP_CHECK(hipMemsetAsync(self, 0, (size_t)rows_self * kk * sizeof(real_t), dev.stream));
    CholCall c{self, (size_t)kk, F, ldf, kk, 0, nullptr, s->gram.ptr, 0, 0, 0, lam, lam, false, false, false, CHOL_NAZ};
    c.values_override = s->ones.ptr;
    if (dev.nonneg_now || dev.l1_now != (real_t)0 || dev.l1_last_now != (real_t)0)
        return launch_chol(dev, c, &X);             // coordinate descent: per row on the assembled system
    // one factorisation of the shared matrix, the row kernel only gathers the right-hand sides (first version: the
    // matrix factorised once per row -- c1 + implicit features 8 ms for Bi + Ai)
    // ... first by the row kernel itself (rhs_only, still the path of wider systems), whose staging loop made the longest
    // row the critical path (Bi at the C1 shape 4.5 ms); now a two-stage segmented gather-sum
    auto &G = s->gsegs[isAi ? 0 : 1];
    if (kk <= 64 * GSUM_MAXC) {
        if (G.nseg > 0)
            hipLaunchKernelGGL(gather_sum_segments_kernel<real_t>, dim3((G.nseg + 3) / 4), dim3(256), 0, dev.stream, X.p.ptr, X.i.ptr, F, ldf,
                               kk, G.seg_row.ptr, G.seg_off.ptr, G.nseg, s->gpartial.ptr);
        hipLaunchKernelGGL(gather_sum_rows_kernel<real_t>, dim3((rows_self + 3) / 4), dim3(256), 0, dev.stream, s->gpartial.ptr,
                           G.row_first.ptr, rows_self, kk, self, (size_t)kk);
        HIP_CHECK(hipGetLastError());
    } else {
        c.rhs_only = true;
        int rc = launch_chol(dev, c, &X);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(potrf_upper_kernel<real_t>, dim3(1), dim3(256), 0, dev.stream, s->gram.ptr, kk);
    HIP_CHECK(hipGetLastError());
    launch_potrs_rows(dev, rows_self, kk, s->gram.ptr, self, (size_t)kk);
    return 0;
}

// C / D update: optimizeA Case 1 with do_B (common.c:2793-2991; Q6: always the transposed gemm)
static int update_sideinfo(cmfrec_hip_session *s, bool isC, bool chol, int cg_steps = -1)
{
    const cmfrec_hip_model &m = s->mdl;
    const DeviceInfo &dev = s->dev;
    const int p = isC ? m.p : m.q;
    if (p <= 0) return 0;
    (isC ? s->eigA : s->eigB).fresh = false;
    const int rows_u = isC ? m.m_u : m.n_i;
    const int kc = (isC ? m.k_user : m.k_item) + m.k;
    real_t *Cm = isC ? s->C.ptr : s->D.ptr;
    const real_t *Um = isC ? s->U.ptr : s->II.ptr;
    const real_t *F = isC ? s->A.ptr : s->B.ptr;
    const size_t ldF = isC ? s->ldA : s->ldB;
    const real_t w = isC ? m.w_user : m.w_item;
    const bool scale_lam = m.scale_lam || m.scale_lam_sideinfo;
    real_t lam = s->lam6[isC ? 4 : 5] / w;                                                                    // collective.c:8367, :8418
    if (isC ? s->sparseU : s->sparseI) {
        // sparse side information: optimizeA Case 4 on its CSC -- one row of C per attribute, gathered from the
        // first k_side+k columns of the factor matrix (collective.c:8354-8386; attributes nobody has stay untouched)
        const SparseShard &Uc = isC ? s->Usc : s->Isc;
        if (!chol) {
            CgCall cg{Cm, (size_t)kc, F, ldF, kc, nullptr, nullptr, lam, lam, scale_lam, false, cg_steps > 0 ? cg_steps : m.max_cg_steps, false,
                      (bool)m.precondition_cg};
            return launch_cg_any(dev, cg, Uc);
        }
        CholCall c{Cm, (size_t)kc, F, ldF, kc, 0, nullptr, nullptr, 0, 0, 0, lam, lam, scale_lam, false, false, CHOL_EXPLICIT};
        return launch_chol(dev, c, &Uc);
    }
    real_t diag = scale_lam ? lam * (real_t)rows_u : lam;                                          // common.c:2832
    launch_gram(dev, s->gws, F, ldF, rows_u, kc, s->gram.ptr, (real_t)1, diag);                    // common.c:2824
    launch_gemm<true>(dev, p, kc, rows_u, (real_t)1, Um, (size_t)p, F, ldF, Cm, (size_t)kc);       // common.c:2852-2855
    CholCall c{Cm, (size_t)kc, nullptr, 0, kc, 0, nullptr, s->gram.ptr, 0, 0, 0, 0, 0, false, false, false, CHOL_PREFILLED};
    return launch_chol(dev, c, nullptr, p);                                                        // common.c:2872-2875
}

// The matrices the reference keeps for predictions on new data (the step after the path),
// from the factors resident in the session:
//   implicit (epilogue of fit_collective_implicit_als, src/collective.c:10056-10115):
//     BtB = B^T B + lam I;  with user side information  BeTBe = blockdiag(lam I, BtB) + w C^T C  and its Cholesky factor
//   explicit (epilogue of fit_collective_explicit_als, :8936-9249), Bp = [B(:, k_item:) | 1 if user_bias]:
//     BtB = Bp^T Bp;  TransBtBinvBt = Bp (BtB + lam (n if scale_lam) I)^-1;  CtCw = w C^T C;
//     TransCtCinvCt = C (C^T C + lam (p if scale_lam) / w I)^-1;
//     BeTBeChol = chol(blockdiag(0, BtB) + CtCw + lam mult I),  mult = p + n | n | 1
// Host output buffers, NULL = skip; square matrices are complete (both triangles) except the Cholesky factors
// (upper triangle R, M = R^T R; the strict lower triangle holds the lower triangle of M).
int cmfrec_hip_session_precompute(cmfrec_hip_session *s, int last_step_cholesky, int include_all_X, real_t *BtB,
                                  real_t *TransBtBinvBt, real_t *BeTBe, real_t *BeTBeChol, real_t *CtCw,
                                  real_t *TransCtCinvCt)
{
    return guarded([&]() {
        HIP_CHECK(hipSetDevice(s->dev.device));
        cmfrec_hip_model m = s->mdl;
        // rows of B that enter: n_max with include_all_X (explicit model only), else the columns of X (collective.c:9013-9016)
        if (!(include_all_X && !m.implicit) && m.n_x > 0) m.n = m.n_x;
        const DeviceInfo &dev = s->dev;
        hipStream_t st = dev.stream;
        const int kk = m.k + m.k_main, ub = (!m.implicit && m.user_bias) ? 1 : 0;
        const int kp = kk + ub, kc = m.k_user + m.k, kq = m.k_user + kp;
        const bool scale_lam = m.scale_lam || m.scale_lam_sideinfo;
        DevBuf<real_t> Bp, G, M, T1;
        Bp.alloc((size_t)m.n * kp); G.alloc((size_t)kp * kp);
        hipLaunchKernelGGL(copy_mat_kernel<real_t>, grid1d((size_t)m.n * kk), dim3(256), 0, st, s->B.ptr + m.k_item, s->ldB, Bp.ptr,
                           (size_t)kp, (size_t)m.n, kk);
        if (ub) hipLaunchKernelGGL(col_fill_kernel<real_t>, grid1d(m.n), dim3(256), 0, st, Bp.ptr, (size_t)kp, m.n, kk, (real_t)1);
        // lam_unique[2] for the users' systems, the bias' own lam_unique[0] as a correction of the last diagonal entry
        // (collective.c:9066-9074, :9225-9238).  Implicit model: the Gramian keeps the lambda of the last A-step when that ran
        // Cholesky (lam_unique[2]), else the epilogue adds the scalar lam (:10062-10075).
        const real_t lamA = m.implicit ? (last_step_cholesky ? s->lam6[2] : m.lam) : s->lam6[2];
        const real_t lam_bias_diff = ub ? s->lam6[0] - s->lam6[2] : (real_t)0;
        launch_gram(dev, s->gws, Bp.ptr, (size_t)kp, m.n, kp, G.ptr, (real_t)1, m.implicit ? lamA : (real_t)0);
        if (BtB) G.download(BtB, (size_t)kp * kp, st);
        if (TransBtBinvBt && !m.implicit) {                               // collective.c:9034-9082
            M.alloc((size_t)kp * kp);
            HIP_CHECK(hipMemcpyAsync(M.ptr, G.ptr, (size_t)kp * kp * sizeof(real_t), hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(add_diag_kernel<real_t>, grid1d(kp), dim3(256), 0, st, M.ptr, kp, 0, kp,
                               lamA * (real_t)(m.scale_lam ? m.n : 1));
            if (lam_bias_diff != 0)
                hipLaunchKernelGGL(add_diag_kernel<real_t>, grid1d(1), dim3(256), 0, st, M.ptr, kp, kp - 1, kp,
                                   lam_bias_diff * (real_t)(m.scale_lam ? m.n : 1));
            CholCall c{Bp.ptr, (size_t)kp, nullptr, 0, kp, 0, nullptr, M.ptr, 0, 0, 0, 0, 0, false, false, false, CHOL_PREFILLED};
            int rc = launch_chol(dev, c, nullptr, m.n);
            if (rc) return rc;
            Bp.download(TransBtBinvBt, (size_t)m.n * kp, st);
        }
        if (m.p > 0) {
            const real_t w = m.w_user;
            DevBuf<real_t> CtC, Cc;
            CtC.alloc((size_t)kc * kc);
            launch_gram(dev, s->gws, s->C.ptr, (size_t)kc, m.p, kc, CtC.ptr, (real_t)1, (real_t)0);
            if (TransCtCinvCt && !m.implicit) {                           // :9083-9142
                M.alloc((size_t)kc * kc); Cc.alloc((size_t)m.p * kc);
                HIP_CHECK(hipMemcpyAsync(M.ptr, CtC.ptr, (size_t)kc * kc * sizeof(real_t), hipMemcpyDeviceToDevice, st));
                HIP_CHECK(hipMemcpyAsync(Cc.ptr, s->C.ptr, (size_t)m.p * kc * sizeof(real_t), hipMemcpyDeviceToDevice, st));
                hipLaunchKernelGGL(add_diag_kernel<real_t>, grid1d(kc), dim3(256), 0, st, M.ptr, kc, 0, kc,
                                   lamA * (real_t)(m.scale_lam ? m.p : 1) / w);
                CholCall c{Cc.ptr, (size_t)kc, nullptr, 0, kc, 0, nullptr, M.ptr, 0, 0, 0, 0, 0, false, false, false, CHOL_PREFILLED};
                int rc = launch_chol(dev, c, nullptr, m.p);
                if (rc) return rc;
                Cc.download(TransCtCinvCt, (size_t)m.p * kc, st);
            }
            if (CtCw) {                                                   // w C^T C
                T1.alloc((size_t)kc * kc);
                HIP_CHECK(hipMemsetAsync(T1.ptr, 0, (size_t)kc * kc * sizeof(real_t), st));
                hipLaunchKernelGGL(add_block_kernel<real_t>, grid1d((size_t)kc * kc), dim3(256), 0, st, CtC.ptr, kc, w, T1.ptr, kc, 0);
                T1.download(CtCw, (size_t)kc * kc, st);
                HIP_CHECK(hipStreamSynchronize(st));
            }
            if (BeTBe || BeTBeChol) {                                     // :9184-9243 / :10073-10110
                M.alloc((size_t)kq * kq);
                HIP_CHECK(hipMemsetAsync(M.ptr, 0, (size_t)kq * kq * sizeof(real_t), st));
                hipLaunchKernelGGL(add_block_kernel<real_t>, grid1d((size_t)kp * kp), dim3(256), 0, st, G.ptr, kp, (real_t)1, M.ptr, kq, m.k_user);
                // Quirk Q9 (reference behaviour, reproduced on purpose -- see DESIGN.md "Parity"): in the implicit model
                // whose last iteration ran CG (no finalize_chol) the epilogue scales the cached C^T C by w_user only `if (w_user == 1.)`
                // (src/collective.c:10077, the test is inverted), so for w_user != 1 the UNWEIGHTED C^T C ends up in
                // BeTBe / BeTBeChol.  With Cholesky (or w_user == 1) the weighted matrix comes out, as intended.
                const real_t w_betbe = (m.implicit && !last_step_cholesky) ? (real_t)1 : w;
                hipLaunchKernelGGL(add_block_kernel<real_t>, grid1d((size_t)kc * kc), dim3(256), 0, st, CtC.ptr, kc, w_betbe, M.ptr, kq, 0);
                if (m.implicit) {                                         // lam is already inside G; the k_user block gets its own
                    if (m.k_user) hipLaunchKernelGGL(add_diag_kernel<real_t>, grid1d(m.k_user), dim3(256), 0, st, M.ptr, kq, 0, m.k_user, lamA);
                } else {
                    const real_t mult = m.scale_lam_sideinfo ? (real_t)(m.p + m.n) : (scale_lam ? (real_t)m.n : (real_t)1);
                    hipLaunchKernelGGL(add_diag_kernel<real_t>, grid1d(kq), dim3(256), 0, st, M.ptr, kq, 0, kq, lamA * mult);
                    if (lam_bias_diff != 0)
                        hipLaunchKernelGGL(add_diag_kernel<real_t>, grid1d(1), dim3(256), 0, st, M.ptr, kq, kq - 1, kq, lam_bias_diff * mult);
                }
                if (BeTBe) M.download(BeTBe, (size_t)kq * kq, st);
                if (BeTBeChol) {
                    hipLaunchKernelGGL(potrf_upper_kernel<real_t>, dim3(1), dim3(256), 0, st, M.ptr, kq);
                    M.download(BeTBeChol, (size_t)kq * kq, st);
                }
            }
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(st));
        return 0;
    });
}

int cmfrec_hip_session_after_gather(cmfrec_hip_session *s, int which)
{
    return guarded([&]() {
        HIP_CHECK(hipSetDevice(s->dev.device));
        const cmfrec_hip_model &m = s->mdl;
        hipStream_t st = s->dev.stream;
        if (m.implicit || !s->has_bias) return 0;
        if (which == 'B') {
            if (m.item_bias)     // collective.c:8723-8725
                hipLaunchKernelGGL(col_extract_kernel<real_t>, grid1d(m.n), dim3(256), 0, st, s->B.ptr, s->ldB, m.n, s->k_totB, s->biasB.ptr);
            if (m.user_bias)     // collective.c:8728-8732
                hipLaunchKernelGGL(col_fill_kernel<real_t>, grid1d(m.n), dim3(256), 0, st, s->B.ptr, s->ldB, m.n, s->k_totB, (real_t)1);
        } else if (which == 'A') {
            if (m.user_bias)     // collective.c:8882-8884
                hipLaunchKernelGGL(col_extract_kernel<real_t>, grid1d(m.m), dim3(256), 0, st, s->A.ptr, s->ldA, m.m, s->k_totA, s->biasA.ptr);
            if (m.item_bias)     // collective.c:8538-8543 (done at the top of the next iteration in the reference)
                hipLaunchKernelGGL(col_fill_kernel<real_t>, grid1d(m.m), dim3(256), 0, st, s->A.ptr, s->ldA, m.m, s->k_totA, (real_t)1);
        }
        HIP_CHECK(hipGetLastError());
        return 0;
    });
}

int cmfrec_hip_session_update(cmfrec_hip_session *s, int which, int use_cholesky)
{
    return guarded([&]() {
        HIP_CHECK(hipSetDevice(s->dev.device));
        const bool isAB = (which == 'A' || which == 'B'), isImp = (which == 'a' || which == 'b');
        const bool nn = (isAB || isImp) ? s->nonneg : (which == 'C' ? s->nonneg_C : s->nonneg_D);
        // l1_lam_unique order: user bias, item bias, A, B, C, D (collective.c:8652-8654, :8823-8825, :8369-8423, :8472-8516)
        const bool ub = !s->mdl.implicit && s->mdl.user_bias, ib = !s->mdl.implicit && s->mdl.item_bias;
        const real_t l1 = which == 'A' ? s->l16[2] : which == 'B' ? s->l16[3] : which == 'a' ? s->l16[2] / s->w_implicit
                          : which == 'b' ? s->l16[3] / s->w_implicit
                          : (which == 'C' ? s->l16[4] / s->mdl.w_user : s->l16[5] / s->mdl.w_item);
        const real_t l1_last = which == 'A' ? s->l16[ub ? 0 : 2] : which == 'B' ? s->l16[ib ? 1 : 3] : l1;
        const bool chol = use_cholesky || !s->mdl.use_cg || nn || l1 != 0 || l1_last != 0;   // common.c:725, :2781, :3320: no CG with nonneg / L1
        struct SolveScope {                                           // the closed-form launches of this update
            const DeviceInfo &d;
            SolveScope(const DeviceInfo &d_, bool on, int steps, real_t l1_, real_t l1l, real_t sc) : d(d_)
            { d.nonneg_now = on; d.max_cd_steps = steps; d.l1_now = l1_; d.l1_last_now = l1l; d.l1_scale = sc; }
            ~SolveScope() { d.nonneg_now = false; d.l1_now = 0; d.l1_last_now = 0; d.l1_scale = 1; }
        } scope(s->dev, nn, s->max_cd_steps, l1, l1_last,
                // the dense C / D update scales lambda -- and the penalty -- by the number of rows of U / I (common.c:2832, :2882)
                // the Ai / Bi update (Case 3) by the rows of the fixed matrix (common.c:3131, :3182)
                ((which == 'C' || which == 'D' || isImp) && (s->mdl.scale_lam || s->mdl.scale_lam_sideinfo))
                    ? (real_t)(which == 'C' ? s->mdl.m_u : which == 'D' ? s->mdl.n_i : which == 'a' ? s->mdl.n : s->mdl.m)
                    : (real_t)1);
        if (which == 'A' || which == 'B') {
            EventPair ev{s->new_event(), s->new_event()};
            HIP_CHECK(hipEventRecord(ev.a, s->dev.stream));
            int rc = 0;
            if (which == 'A' && !s->XrParts.empty()) {
                for (int c = 0; c < (int)s->XrParts.size() && rc == 0; c++) {
                    rc = update_factor(s, true, chol, c);
                    HIP_CHECK(hipEventRecord(s->partEv[c], s->dev.stream));
                }
            } else {
                rc = update_factor(s, which == 'A', chol);
                if (rc == 0 && !chol && (which == 'A' ? s->has_cfA : s->has_cfB)) {
                    // rows of this half-step that the reference solves in closed form next to the CG rows
                    // (cmfrec_hip_session_set_closed_form_rows): CG results aside, closed form over all rows, CG rows back
                    const bool isA = which == 'A';
                    real_t *self = isA ? s->A.ptr : s->B.ptr;
                    const size_t ld = isA ? s->ldA : s->ldB, rows = (size_t)(isA ? s->mdl.m : s->mdl.n);
                    s->cf_keep.alloc_at_least(rows * ld);
                    HIP_CHECK(hipMemcpyAsync(s->cf_keep.ptr, self, rows * ld * sizeof(real_t), hipMemcpyDeviceToDevice, s->dev.stream));
                    rc = update_factor(s, isA, true);
                    if (rc != 0) {
                        // the closed-form pass failed part-way: put the CG results back whole rather than leave a mix of zeroed
                        // and half-solved rows behind the error code
                        HIP_CHECK(hipMemcpyAsync(self, s->cf_keep.ptr, rows * ld * sizeof(real_t), hipMemcpyDeviceToDevice, s->dev.stream));
                    } else {
                        hipLaunchKernelGGL(restore_rows_kernel<real_t>, grid1d(rows * ld), dim3(256), 0, s->dev.stream, self, s->cf_keep.ptr, ld, rows,
                                           (isA ? s->cfmaskA : s->cfmaskB).ptr);
                        HIP_CHECK(hipGetLastError());
                    }
                }
            }
            if (rc == 0 && (which == 'A' ? s->n_zrowsA : s->n_zrowsB) > 0) {
                // rows the reference does not solve under NA_as_zero_U / _I (cmfrec_hip_session_set_zero_rows): the unknowns of
                // the row, its own bias included; a bias column fixed to 1 stays
                const bool isA = which == 'A';
                const bool self_bias = !s->mdl.implicit && (isA ? s->mdl.user_bias : s->mdl.item_bias);
                const int ncols = (isA ? s->k_totA : s->k_totB) + (self_bias ? 1 : 0), cnt = isA ? s->n_zrowsA : s->n_zrowsB;
                hipLaunchKernelGGL(zero_rows_kernel<real_t>, grid1d((size_t)cnt * ncols), dim3(256), 0, s->dev.stream, isA ? s->A.ptr : s->B.ptr,
                                   isA ? s->ldA : s->ldB, ncols, (isA ? s->zrowsA : s->zrowsB).ptr, cnt);
                HIP_CHECK(hipGetLastError());
            }
            HIP_CHECK(hipEventRecord(ev.b, s->dev.stream));
            (which == 'A' ? s->evA : s->evB).push_back(ev);
            // long-lived sessions: the timing events are a window over the most recent updates, not an unbounded log
            trim_events(which == 'A' ? s->evA : s->evB);
            for (auto &v : (which == 'A' ? s->binA : s->binB).ev) trim_events(v);
            return rc;
        }
        if (which == 'C' || which == 'D') {
            if (s->side_local) {
                // the session holds the side information of its row block only: the whole-matrix update would read rows it
                // does not have -- the sharded form is sideinfo_partial + all-reduce + sideinfo_finish
                g_last_error = "cmfrec_hip_session_update: C / D of a session with local side information go through "
                               "cmfrec_hip_session_sideinfo_partial / _finish";
                return 2;
            }
            const bool isC = which == 'C';
            int rc = update_sideinfo(s, isC, chol);
            const bool *any = isC ? s->cfC_any : s->cfD_any;
            if (rc == 0 && !chol && (isC ? s->sparseU : s->sparseI) && (any[1] || any[2])) {
                // dense side information with NaN (cmfrec_hip_session_set_closed_form_rows 'C' / 'D'): next to the CG attributes the
                // reference solves some in closed form (mask 1) and some by CG from zero with k_side + k steps (mask 2)
                real_t *Cm = isC ? s->C.ptr : s->D.ptr;
                const int kc = (isC ? s->mdl.k_user : s->mdl.k_item) + s->mdl.k;
                const size_t rows = (size_t)(isC ? s->mdl.p : s->mdl.q), ld = (size_t)kc;
                const unsigned char *mask = (isC ? s->cfmaskC : s->cfmaskD).ptr;
                hipStream_t st = s->dev.stream;
                s->cf_keep.alloc_at_least(rows * ld);
                for (int value = 1; value <= 2 && rc == 0; value++) {
                    if (!any[value]) continue;
                    HIP_CHECK(hipMemcpyAsync(s->cf_keep.ptr, Cm, rows * ld * sizeof(real_t), hipMemcpyDeviceToDevice, st));
                    if (value == 2) HIP_CHECK(hipMemsetAsync(Cm, 0, rows * ld * sizeof(real_t), st));
                    rc = value == 1 ? update_sideinfo(s, isC, true) : update_sideinfo(s, isC, false, kc);
                    if (rc != 0) { HIP_CHECK(hipMemcpyAsync(Cm, s->cf_keep.ptr, rows * ld * sizeof(real_t), hipMemcpyDeviceToDevice, st)); break; }
                    hipLaunchKernelGGL(keep_rows_unless_kernel<real_t>, grid1d(rows * ld), dim3(256), 0, st, Cm, s->cf_keep.ptr, ld, rows, mask,
                                       (unsigned char)value);
                    HIP_CHECK(hipGetLastError());
                }
            }
            return rc;
        }
        if ((which == 'a' || which == 'b') && s->implicit_feats) return update_implicit_feats(s, which == 'a');
        g_last_error = "cmfrec_hip: unknown update target";
        return 2;
    });
}

int cmfrec_hip_session_iterate(cmfrec_hip_session *s, int niter, int finalize_chol)
{
    const cmfrec_hip_model &m = s->mdl;
    if (m.row_begin != 0 || m.row_end != m.m || m.col_begin != 0 || m.col_end != m.n) {
        g_last_error = "cmfrec_hip: iterate() needs a session that owns all rows (use update()+all-gather on shards)";
        return 2;
    }
    for (int it = 0; it < niter; it++) {
        // collective.c:8336-8340 / :9829-9830
        int chol = (finalize_chol && m.use_cg && it == niter - 1) ? 1 : 0;
        int rc;
        if (m.p > 0 && (rc = cmfrec_hip_session_update(s, 'C', chol))) return rc;
        if (m.q > 0 && (rc = cmfrec_hip_session_update(s, 'D', chol))) return rc;
        if (s->implicit_feats) {                                              // collective.c:8448-8534: Bi, then Ai
            if ((rc = cmfrec_hip_session_update(s, 'b', 1))) return rc;
            if ((rc = cmfrec_hip_session_update(s, 'a', 1))) return rc;
        }
        if ((rc = cmfrec_hip_session_update(s, 'B', chol))) return rc;
        if ((rc = cmfrec_hip_session_after_gather(s, 'B'))) return rc;
        if ((rc = cmfrec_hip_session_update(s, 'A', chol))) return rc;
        if ((rc = cmfrec_hip_session_after_gather(s, 'A'))) return rc;
    }
    return 0;
}

int cmfrec_hip_session_sync(cmfrec_hip_session *s)
{
    return guarded([&]() {
        HIP_CHECK(hipStreamSynchronize(s->dev.stream));
        return 0;
    });
}

void *cmfrec_hip_session_device_ptr(cmfrec_hip_session *s, int which, size_t *rows, size_t *ld)
{
    switch (which) {
        case 'A': if (rows) *rows = s->mdl.m; if (ld) *ld = s->ldA; return s->A.ptr;
        case 'B': if (rows) *rows = s->mdl.n; if (ld) *ld = s->ldB; return s->B.ptr;
        // (whoever asks for C / D may write to it: the cached eigenvectors are not trusted afterwards)
        case 'C': if (rows) *rows = s->mdl.p; if (ld) *ld = s->mdl.k_user + s->mdl.k; s->eigA.fresh = false; return s->C.ptr;
        case 'D': if (rows) *rows = s->mdl.q; if (ld) *ld = s->mdl.k_item + s->mdl.k; s->eigB.fresh = false; return s->D.ptr;
        case 'a': if (rows) *rows = s->mdl.m; if (ld) *ld = 1; return s->biasA.ptr;
        case 'b': if (rows) *rows = s->mdl.n; if (ld) *ld = 1; return s->biasB.ptr;
        case 'P': if (rows) *rows = s->side_part.n; if (ld) *ld = 1; return s->side_part.ptr;   // partial sums of the C / D update
    }
    return nullptr;
}

void *cmfrec_hip_session_stream(cmfrec_hip_session *s) { return (void *)s->dev.stream; }

int cmfrec_hip_session_kernel_time(cmfrec_hip_session *s, int which, double *ms, long *launches)
{
    return guarded([&]() {
        HIP_CHECK(hipStreamSynchronize(s->dev.stream));
        auto &v = (which == 'A') ? s->evA : s->evB;
        double tot = 0;
        for (auto &p : v) {
            float t = 0;
            HIP_CHECK(hipEventElapsedTime(&t, p.a, p.b));
            tot += t;
        }
        if (ms) *ms = tot;
        if (launches) *launches = (long)v.size();
        return 0;
    });
}

int cmfrec_hip_session_bin_stats(cmfrec_hip_session *s, int which, int bin, double *ms, long *launches,
                                 long *rows, unsigned long long *nnz)
{
    return guarded([&]() {
        if (bin < 0 || bin >= NBINS) return 2;
        HIP_CHECK(hipStreamSynchronize(s->dev.stream));
        BinTimers &bt = (which == 'A') ? s->binA : s->binB;
        const SparseShard &X = (which == 'A') ? s->Xr : s->Xc;
        double tot = 0;
        for (auto &p : bt.ev[bin]) {
            float t = 0;
            HIP_CHECK(hipEventElapsedTime(&t, p.a, p.b));
            tot += t;
        }
        if (ms) *ms = tot;
        // with row parts every half-step launches each bin once per part: report half-steps, not launches
        const long per_step = (which == 'A' && !s->XrParts.empty()) ? (long)s->XrParts.size() : 1;
        if (launches) *launches = (long)bt.ev[bin].size() / per_step;
        if (rows) *rows = X.bin_rows[bin];
        if (nnz) *nnz = X.bin_nnz[bin];
        return 0;
    });
}

int cmfrec_hip_session_bin_overlaps(cmfrec_hip_session *s, int which, int bin)
{
    const SparseShard &X = (which == 'A') ? s->Xr : s->Xc;
    if (which == 'A' && !s->XrParts.empty()) return 1;                    // the bins of a part alternate between two streams
    if (cg_bin_streams() > 1) return 1;                                    // the bins of a half-step run side by side (launch_cg_S)
    // (the Gramian path stays in line, launch_cg_S)
    const bool gram_in_line = cmfrec_hip_session_vh_mode(s, which) == 2 && true;
    return (bin == BIN_VHEAVY && X.vh_runs_aside(s->dev.num_cus) && !gram_in_line) ? 1 : 0;
}

// The most recent collective Cholesky half-step of the session: rows solved by the low-rank kernels (0: the path was not taken)
// and the eigen-decomposition behind them (3 tridiagonalisation + QL, 2 the one-workgroup Jacobi kernel)
int cmfrec_hip_session_lowrank_info(cmfrec_hip_session *s, int *rows, int *eig)
{
    if (rows) *rows = s->lr.last_rows;
    if (eig) *eig = s->lr.last_eig;
    return 0;
}

// rows of the CSR ('A') / CSC ('B') shard with at least this many entries are split rows (their entries are kept sorted by
// opposing index, SparseShard::vh_min)
int cmfrec_hip_session_vh_min(cmfrec_hip_session *s, int which)
{
    return ((which == 'A') ? s->Xr : s->Xc).vh_min;
}

int cmfrec_hip_session_vh_mode(cmfrec_hip_session *s, int which)
{
    const SparseShard &X = (which == 'A') ? s->Xr : s->Xc;
    if (X.bin_rows[BIN_VHEAVY] <= 0) return 0;
    const bool gram = (switches().vh != 0) ? switches().vh == 2 : X.prefer_gram((size_t)(s->mdl.k + s->mdl.k_main) * sizeof(real_t));
    return (gram && s->mdl.k + s->mdl.k_main <= 16 * GRAM_NTT) ? 2 : 1;
}

void cmfrec_hip_session_reset_timers(cmfrec_hip_session *s)
{
    (void)hipStreamSynchronize(s->dev.stream);
    s->binA.clear(); s->binB.clear();
    for (auto *v : {&s->evA, &s->evB}) {
        for (auto &p : *v) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        v->clear();
    }
}

// ============================ level 2: operators with host buffers ============================

int cmfrec_hip_optimizeA_implicit(real_t *A, size_t lda, const real_t *B, size_t ldb, int_t m, int_t n, int_t k,
                                  const size_t Xcsr_p[], const int_t Xcsr_i[], const real_t *Xcsr, real_t lam,
                                  bool use_cg, bool precondition_cg, int_t max_cg_steps, real_t *BtB_out)
{
    return guarded([&]() {
        DeviceInfo dev;
        init_device(dev, -1);
        DevBuf<real_t> dA, dB, dG;
        GramWorkspace gws;
        SparseShard X;
        dA.upload(A, (size_t)m * lda, dev.stream);
        dB.upload(B, (size_t)n * ldb, dev.stream);
        dG.alloc((size_t)k * k);
        X.opp_row_bytes_hint = (size_t)k * sizeof(real_t);
        shard_from_csr(X, m, Xcsr_p, Xcsr_i, Xcsr, n, dev.stream);
        launch_gram(dev, gws, dB.ptr, ldb, n, k, dG.ptr, (real_t)1, use_cg ? (real_t)0 : lam);
        int rc;
        if (use_cg) {
            CgCall c{dA.ptr, lda, dB.ptr, ldb, k, nullptr, dG.ptr, lam, lam, false, false, max_cg_steps, true, precondition_cg};
            rc = launch_cg_any(dev, c, X);
        } else {
            // common.c:3334 zeroes m*k - (lda-k) elements from A
            HIP_CHECK(hipMemsetAsync(dA.ptr, 0, ((size_t)m * lda - (lda - (size_t)k)) * sizeof(real_t), dev.stream));
            CholCall c{dA.ptr, lda, dB.ptr, ldb, k, 0, nullptr, dG.ptr, 0, 0, 0, lam, lam, false, false, false, CHOL_IMPLICIT};
            rc = launch_chol(dev, c, &X);
        }
        dA.download(A, (size_t)m * lda, dev.stream);
        if (BtB_out) dG.download(BtB_out, (size_t)k * k, dev.stream);
        HIP_CHECK(hipStreamSynchronize(dev.stream));
        return rc;
    });
}

int cmfrec_hip_optimizeA_explicit(real_t *A, size_t lda, const real_t *B, size_t ldb, int_t m, int_t n, int_t k,
                                  const size_t Xcsr_p[], const int_t Xcsr_i[], const real_t *Xcsr,
                                  const real_t *bias_sub, real_t lam, real_t lam_last, bool scale_lam,
                                  bool scale_bias_const, bool use_cg, bool precondition_cg, int_t max_cg_steps)
{
    return cmfrec_hip_optimizeA_explicit_weighted(A, lda, B, ldb, m, n, k, Xcsr_p, Xcsr_i, Xcsr, nullptr, nullptr, bias_sub, lam, lam_last,
                                                  scale_lam, scale_bias_const, use_cg, precondition_cg, max_cg_steps);
}

int cmfrec_hip_optimizeA_explicit_weighted(real_t *A, size_t lda, const real_t *B, size_t ldb, int_t m, int_t n, int_t k,
                                           const size_t Xcsr_p[], const int_t Xcsr_i[], const real_t *Xcsr, const real_t *weight,
                                           const real_t *wsum, const real_t *bias_sub, real_t lam, real_t lam_last, bool scale_lam,
                                           bool scale_bias_const, bool use_cg, bool precondition_cg, int_t max_cg_steps)
{
    return guarded([&]() {
        DeviceInfo dev;
        init_device(dev, -1);
        DevBuf<real_t> dA, dB, dbias;
        SparseShard X;
        dA.upload(A, (size_t)m * lda, dev.stream);
        dB.upload(B, (size_t)n * ldb, dev.stream);
        if (bias_sub) dbias.upload(bias_sub, n, dev.stream);
        X.opp_row_bytes_hint = (size_t)k * sizeof(real_t);
        shard_from_csr(X, m, Xcsr_p, Xcsr_i, Xcsr, n, dev.stream, weight);
        if (weight != nullptr && wsum != nullptr && X.weighted()) {      // the driver's multipliers instead of the rows' own sums
            X.wsum.upload(wsum, (size_t)m, dev.stream);
            HIP_CHECK(hipStreamSynchronize(dev.stream));
        }
        int rc;
        if (use_cg) {
            CgCall c{dA.ptr, lda, dB.ptr, ldb, k, bias_sub ? dbias.ptr : nullptr, nullptr, lam, lam_last, scale_lam,
                     scale_bias_const, max_cg_steps, false, precondition_cg};
            rc = launch_cg_any(dev, c, X);
        } else {
            CholCall c{dA.ptr, lda, dB.ptr, ldb, k, 0, bias_sub ? dbias.ptr : nullptr, nullptr, 0, 0, 0, lam, lam_last,
                       scale_lam, false, scale_bias_const, CHOL_EXPLICIT};
            rc = launch_chol(dev, c, &X);
        }
        dA.download(A, (size_t)m * lda, dev.stream);
        HIP_CHECK(hipStreamSynchronize(dev.stream));
        return rc;
    });
}

int cmfrec_hip_optimizeA_dense_full(real_t *A, size_t lda, const real_t *B, size_t ldb, int_t m, int_t n, int_t k,
                                    const real_t *Xfull, size_t ldX, bool do_B, real_t lam, real_t lam_last,
                                    bool scale_lam)
{
    return guarded([&]() {
        if (lam != lam_last) { g_last_error = "cmfrec_hip: dense_full supports lam_last == lam only"; return 2; }
        DeviceInfo dev;
        init_device(dev, -1);
        DevBuf<real_t> dA, dB, dX, dG;
        GramWorkspace gws;
        dA.alloc((size_t)m * lda);
        HIP_CHECK(hipMemsetAsync(dA.ptr, 0, dA.n * sizeof(real_t), dev.stream));
        dB.upload(B, (size_t)n * ldb, dev.stream);
        size_t xrows = do_B ? (size_t)n : (size_t)m;
        dX.upload(Xfull, xrows * ldX, dev.stream);
        dG.alloc((size_t)k * k);
        launch_gram(dev, gws, dB.ptr, ldb, n, k, dG.ptr, (real_t)1, scale_lam ? lam * (real_t)n : lam);
        if (do_B) launch_gemm<true>(dev, m, k, n, (real_t)1, dX.ptr, ldX, dB.ptr, ldb, dA.ptr, lda);
        else      launch_gemm<false>(dev, m, k, n, (real_t)1, dX.ptr, ldX, dB.ptr, ldb, dA.ptr, lda);
        CholCall c{dA.ptr, lda, nullptr, 0, k, 0, nullptr, dG.ptr, 0, 0, 0, 0, 0, false, false, false, CHOL_PREFILLED};
        int rc = launch_chol(dev, c, nullptr, m);
        // only the k solved columns are written back (padding columns are never touched by the reference)
        std::vector<real_t> tmp((size_t)m * lda);
        dA.download(tmp.data(), (size_t)m * lda, dev.stream);
        HIP_CHECK(hipStreamSynchronize(dev.stream));
        for (int r = 0; r < m; r++) memcpy(A + (size_t)r * lda, tmp.data() + (size_t)r * lda, (size_t)k * sizeof(real_t));
        return rc;
    });
}

int cmfrec_hip_topN_batch(const real_t *A, size_t lda, int_t nu, const real_t *B, size_t ldb, int_t n, int_t k,
                          const real_t *biasB, const size_t excl_p[], const int_t excl_i[], int_t n_top,
                          int_t *out_ids, real_t *out_scores)
{
    return guarded([&]() {
        if (nu <= 0 || n <= 0 || k <= 0 || n_top <= 0 || !A || !B || !out_ids) {
            g_last_error = "cmfrec_hip_topN_batch: invalid arguments";
            return 2;
        }
        if (k > TOPN_KMAX || n_top > TOPN_NMAX || n_top > n) {
            g_last_error = "cmfrec_hip_topN_batch: needs k <= 64 and n_top <= min(128, n)";
            return 2;
        }
        DeviceInfo dev;
        init_device(dev, -1);
        DevBuf<real_t> dA, dB, dbias, dsc;
        DevBuf<size_t> dep; DevBuf<int> dei, dids;
        dA.upload(A, (size_t)nu * lda, dev.stream);
        dB.upload(B, (size_t)n * ldb, dev.stream);
        if (biasB) dbias.upload(biasB, (size_t)n, dev.stream);
        if (excl_p) {
            dep.upload(excl_p, (size_t)nu + 1, dev.stream);
            dei.upload(excl_i, std::max<size_t>(excl_p[nu], 1), dev.stream);
        }
        dids.alloc((size_t)nu * n_top);
        if (out_scores) dsc.alloc((size_t)nu * n_top);
        TopnParams<real_t> P;
        P.A = dA.ptr; P.lda = lda; P.nu = nu; P.B = dB.ptr; P.ldb = ldb; P.n = n; P.k = k;
        P.biasB = biasB ? dbias.ptr : nullptr;
        P.excl_p = excl_p ? dep.ptr : nullptr; P.excl_i = excl_p ? dei.ptr : nullptr;
        P.n_top = n_top; P.out_ids = dids.ptr; P.out_scores = out_scores ? dsc.ptr : nullptr;
        const size_t smem = topn_lds_bytes(sizeof(real_t));
        HIP_CHECK(hipFuncSetAttribute((const void *)topn_kernel<real_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const int tiles = (nu + TOPN_UT - 1) / TOPN_UT;
        hipLaunchKernelGGL(topn_kernel<real_t>, dim3(std::min(tiles, dev.num_cus * 2)), dim3(TOPN_TH), smem, dev.stream, P);
        HIP_CHECK(hipGetLastError());
        dids.download(out_ids, (size_t)nu * n_top, dev.stream);
        if (out_scores) dsc.download(out_scores, (size_t)nu * n_top, dev.stream);
        HIP_CHECK(hipStreamSynchronize(dev.stream));
        return 0;
    });
}

// ---- the reference's stand-alone entry points for the prediction matrices and the ranking, under their own names and with
// ---- their own signatures (round 6; src/cmfrec.h:1922-1960, :2104-2127) ------------------------------------------------------
// Host buffers in and out like the fit entry points.  The Gramians run on the library's MFMA kernels, the solves with many
// right-hand sides on the row Cholesky kernel (CHOL_PREFILLED), the factorisation of the small block matrix on potrf_upper_kernel.
// Like the reference (syrk / potrf with one triangle referenced, src/collective.c:10345-10470) the symmetric outputs carry their
// UPPER triangle (row-major), the strictly lower part is zero.
static void zero_strictly_lower(real_t *M, int n)
{
    if (M == nullptr) return;
    for (int i = 1; i < n; i++) for (int j = 0; j < i; j++) M[(size_t)i * n + j] = 0;
}
// out[kd, kd] = scale * X[:, :kd]^T X[:, :kd] over `rows` rows (X on the host, leading dimension ldx, first used column at X)
static void host_gram(const DeviceInfo &dev, GramWorkspace &gws, const real_t *X, size_t ldx, int rows, int kd, real_t scale, DevBuf<real_t> &dX,
                      DevBuf<real_t> &out)
{
    dX.upload(X, (size_t)rows * ldx, dev.stream);
    out.alloc_at_least((size_t)kd * kd);
    launch_gram(dev, gws, dX.ptr, ldx, rows, kd, out.ptr, scale, (real_t)0);
}

int_t precompute_collective_implicit(
    real_t *B, int_t n, real_t *C, int_t p, real_t *U_colmeans, bool NA_as_zero_U,
    int_t k, int_t k_user, int_t k_item, int_t k_main,
    real_t lam, real_t w_main, real_t w_user, real_t w_main_multiplier,
    bool nonneg, bool extra_precision,
    real_t *BtB, real_t *BeTBe, real_t *BeTBeChol, real_t *CtUbias)
{
    (void)extra_precision;                                  // (the two summation orders of the reference; one here)
    return guarded([&]() {
        if (B == nullptr || BtB == nullptr || n <= 0 || k + k_main <= 0) { g_last_error = "precompute_collective_implicit: B, BtB and n > 0 are required"; return 2; }
        if (p > 0 && (C == nullptr || BeTBe == nullptr)) { g_last_error = "precompute_collective_implicit: p > 0 needs C and BeTBe"; return 2; }
        if (w_main_multiplier != (real_t)1) w_main *= w_main_multiplier;       // collective.c:10501-10507
        if (w_main != (real_t)1) { lam /= w_main; w_user /= w_main; }
        DeviceInfo dev;
        init_device(dev, -1);
        GramWorkspace gws;
        hipStream_t st = dev.stream;
        const int kk = k + k_main, ktB = k_item + kk, kc = k_user + k, kq = k_user + kk;
        DevBuf<real_t> dB, dC, G, CtC, M;
        dB.upload(B, (size_t)n * ktB, st);
        G.alloc((size_t)kk * kk);
        launch_gram(dev, gws, dB.ptr + k_item, (size_t)ktB, n, kk, G.ptr, (real_t)1, lam);      // B^T B + lam I (:10509-10515)
        G.download(BtB, (size_t)kk * kk, st);
        if (p > 0) {
            host_gram(dev, gws, C, (size_t)kc, p, kc, w_user, dC, CtC);                            // w C^T C (:10523-10543)
            M.alloc((size_t)kq * kq);
            HIP_CHECK(hipMemsetAsync(M.ptr, 0, (size_t)kq * kq * sizeof(real_t), st));
            hipLaunchKernelGGL(add_block_kernel<real_t>, grid1d((size_t)kk * kk), dim3(256), 0, st, G.ptr, kk, (real_t)1, M.ptr, kq, k_user);
            hipLaunchKernelGGL(add_block_kernel<real_t>, grid1d((size_t)kc * kc), dim3(256), 0, st, CtC.ptr, kc, (real_t)1, M.ptr, kq, 0);
            if (k_user) hipLaunchKernelGGL(add_diag_kernel<real_t>, grid1d(k_user), dim3(256), 0, st, M.ptr, kq, 0, k_user, lam);   // :10545-10546
            M.download(BeTBe, (size_t)kq * kq, st);
            if (BeTBeChol != nullptr && !nonneg) {                                                 // :10548-10555
                hipLaunchKernelGGL(potrf_upper_kernel<real_t>, dim3(1), dim3(256), 0, st, M.ptr, kq);
                M.download(BeTBeChol, (size_t)kq * kq, st);
            }
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(st));
        zero_strictly_lower(BtB, kk);
        if (p > 0) {
            zero_strictly_lower(BeTBe, kq);
            if (BeTBeChol != nullptr && !nonneg) zero_strictly_lower(BeTBeChol, kq);
            if (C != nullptr && CtUbias != nullptr && U_colmeans != nullptr && NA_as_zero_U)       // :10557-10564
                for (int f = 0; f < kc; f++) {
                    double acc = 0;
                    for (int_t j = 0; j < p; j++) acc += (double)C[(size_t)j * kc + f] * (double)U_colmeans[j];
                    CtUbias[f] = (real_t)(-(double)w_user * acc);
                }
        }
        return 0;
    });
}

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
    real_t *BtB, real_t *TransBtBinvBt, real_t *BtXbias, real_t *BeTBeChol, real_t *BiTBi,
    real_t *TransCtCinvCt, real_t *CtCw, real_t *CtUbias)
{
    return guarded([&]() {
        if (B == nullptr || n <= 0) { g_last_error = "precompute_collective_explicit: B and n > 0 are required"; return 2; }
        if (user_bias && B_plus_bias == nullptr) { g_last_error = "precompute_collective_explicit: user_bias needs the B_plus_bias buffer"; return 2; }
        if ((TransBtBinvBt != nullptr || BeTBeChol != nullptr || (Bi != nullptr && add_implicit_features)) && BtB == nullptr) {
            g_last_error = "precompute_collective_explicit: the matrices derived from BtB need the BtB buffer";
            return 2;
        }
        if (n_max == 0) n_max = n;                                                 // collective.c:10240-10241
        if (include_all_X) n = n_max;
        const int k_main_i = k_main;
        real_t lam_last = lam;
        if (lam_unique != nullptr) { lam_last = lam_unique[user_bias ? 0 : 2]; lam = lam_unique[2]; }    // :10245-10249
        if (w_main != (real_t)1) { lam /= w_main; lam_last /= w_main; w_user /= w_main; w_implicit /= w_main; }
        real_t lam_B = lam, lam_last_B = lam_last, lam_C = lam;
        if (scale_lam || scale_lam_sideinfo) {                                      // :10263-10277
            const real_t multiplier = (real_t)(n + (scale_lam_sideinfo ? p : 0));
            lam *= multiplier;
            lam_C *= (real_t)p;
            lam_last *= scale_bias_const ? scaling_biasA : multiplier;
            lam_B = lam; lam_last_B = lam_last;
        }
        const int ktB0 = k_item + k + k_main;
        const real_t *Bh = B;
        if (user_bias) {                                                            // append_ones_last_col, :10279-10300
            for (int_t c = 0; c < n_max; c++) {
                memcpy(B_plus_bias + (size_t)c * (ktB0 + 1), B + (size_t)c * ktB0, (size_t)ktB0 * sizeof(real_t));
                B_plus_bias[(size_t)c * (ktB0 + 1) + ktB0] = 1;
            }
            k_main++;
            Bh = B_plus_bias;
        }
        const int kk = k + k_main, ktB = k_item + kk, kc = k_user + k, kq = k_user + kk, kki = k + k_main_i;
        if (NA_as_zero_X && BtXbias != nullptr) {                                   // :10302-10342
            std::vector<double> acc((size_t)kk, 0.);
            if (n_max > n && glob_mean != (real_t)0)
                for (int_t c = n; c < n_max; c++) for (int f = 0; f < kk; f++) acc[f] -= (double)glob_mean * (double)Bh[(size_t)c * ktB + k_item + f];
            if (biasB != nullptr || glob_mean != (real_t)0)
                for (int_t c = 0; c < n; c++) {
                    const double coef = -((biasB != nullptr ? (double)biasB[c] : 0.) + (double)glob_mean);
                    for (int f = 0; f < kk; f++) acc[f] += coef * (double)Bh[(size_t)c * ktB + k_item + f];
                }
            for (int f = 0; f < kk; f++) BtXbias[f] = (real_t)acc[f];
        }
        DeviceInfo dev;
        init_device(dev, -1);
        GramWorkspace gws;
        hipStream_t st = dev.stream;
        DevBuf<real_t> dB, dBi, dC, G, Gi, CtC, M, Bp, Cc;
        if (BtB != nullptr) {                                                       // :10344-10351
            dB.upload(Bh, (size_t)n_max * ktB, st);
            G.alloc((size_t)kk * kk);
            launch_gram(dev, gws, dB.ptr + k_item, (size_t)ktB, n, kk, G.ptr, (real_t)1, (real_t)0);
            G.download(BtB, (size_t)kk * kk, st);
        }
        const bool with_bi = Bi != nullptr && add_implicit_features;
        if (with_bi) {                                                              // :10353-10360
            if (BiTBi == nullptr) { g_last_error = "precompute_collective_explicit: add_implicit_features needs the BiTBi buffer"; return 2; }
            host_gram(dev, gws, Bi, (size_t)kki, n, kki, w_implicit, dBi, Gi);
            Gi.download(BiTBi, (size_t)kki * kki, st);
        }
        if (TransBtBinvBt != nullptr && !nonneg && !add_implicit_features) {        // :10362-10386
            M.alloc((size_t)kk * kk); Bp.alloc((size_t)n * kk);
            HIP_CHECK(hipMemcpyAsync(M.ptr, G.ptr, (size_t)kk * kk * sizeof(real_t), hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(add_diag_kernel<real_t>, grid1d(kk), dim3(256), 0, st, M.ptr, kk, 0, kk - 1, lam_B);
            hipLaunchKernelGGL(add_diag_kernel<real_t>, grid1d(1), dim3(256), 0, st, M.ptr, kk, kk - 1, kk, lam_last_B);
            hipLaunchKernelGGL(copy_mat_kernel<real_t>, grid1d((size_t)n * kk), dim3(256), 0, st, dB.ptr + k_item, (size_t)ktB, Bp.ptr, (size_t)kk, (size_t)n, kk);
            CholCall c{Bp.ptr, (size_t)kk, nullptr, 0, kk, 0, nullptr, M.ptr, 0, 0, 0, 0, 0, false, false, false, CHOL_PREFILLED};
            const int rc = launch_chol(dev, c, nullptr, n);
            if (rc) return rc;
            Bp.download(TransBtBinvBt, (size_t)n * kk, st);
        }
        const bool with_c = p > 0 && C != nullptr;
        if (with_c && CtCw != nullptr) {                                            // :10388-10421
            host_gram(dev, gws, C, (size_t)kc, p, kc, (real_t)1, dC, CtC);
            if (TransCtCinvCt != nullptr && !add_implicit_features && !nonneg) {
                M.alloc((size_t)kc * kc); Cc.alloc((size_t)p * kc);
                HIP_CHECK(hipMemcpyAsync(M.ptr, CtC.ptr, (size_t)kc * kc * sizeof(real_t), hipMemcpyDeviceToDevice, st));
                HIP_CHECK(hipMemcpyAsync(Cc.ptr, dC.ptr, (size_t)p * kc * sizeof(real_t), hipMemcpyDeviceToDevice, st));
                hipLaunchKernelGGL(add_diag_kernel<real_t>, grid1d(kc), dim3(256), 0, st, M.ptr, kc, 0, kc, lam_C / w_user);
                CholCall c{Cc.ptr, (size_t)kc, nullptr, 0, kc, 0, nullptr, M.ptr, 0, 0, 0, 0, 0, false, false, false, CHOL_PREFILLED};
                const int rc = launch_chol(dev, c, nullptr, p);
                if (rc) return rc;
                Cc.download(TransCtCinvCt, (size_t)p * kc, st);
            }
            if (w_user != (real_t)1) {                                              // :10419-10420 (in place, stays on the device for the block matrix)
                DevBuf<real_t> T1;
                T1.alloc((size_t)kc * kc);
                HIP_CHECK(hipMemsetAsync(T1.ptr, 0, (size_t)kc * kc * sizeof(real_t), st));
                hipLaunchKernelGGL(add_block_kernel<real_t>, grid1d((size_t)kc * kc), dim3(256), 0, st, CtC.ptr, kc, w_user, T1.ptr, kc, 0);
                HIP_CHECK(hipMemcpyAsync(CtC.ptr, T1.ptr, (size_t)kc * kc * sizeof(real_t), hipMemcpyDeviceToDevice, st));
                HIP_CHECK(hipStreamSynchronize(st));
            }
            CtC.download(CtCw, (size_t)kc * kc, st);
        }
        if (BeTBeChol != nullptr && (C != nullptr || add_implicit_features) && !nonneg) {      // :10423-10461
            M.alloc((size_t)kq * kq);
            HIP_CHECK(hipMemsetAsync(M.ptr, 0, (size_t)kq * kq * sizeof(real_t), st));
            hipLaunchKernelGGL(add_block_kernel<real_t>, grid1d((size_t)kk * kk), dim3(256), 0, st, G.ptr, kk, (real_t)1, M.ptr, kq, k_user);
            if (with_c && CtCw != nullptr)
                hipLaunchKernelGGL(add_block_kernel<real_t>, grid1d((size_t)kc * kc), dim3(256), 0, st, CtC.ptr, kc, (real_t)1, M.ptr, kq, 0);
            else if (with_c) {
                host_gram(dev, gws, C, (size_t)kc, p, kc, w_user, dC, CtC);
                hipLaunchKernelGGL(add_block_kernel<real_t>, grid1d((size_t)kc * kc), dim3(256), 0, st, CtC.ptr, kc, (real_t)1, M.ptr, kq, 0);
            }
            if (with_bi) hipLaunchKernelGGL(add_block_kernel<real_t>, grid1d((size_t)kki * kki), dim3(256), 0, st, Gi.ptr, kki, (real_t)1, M.ptr, kq, k_user);
            hipLaunchKernelGGL(add_diag_kernel<real_t>, grid1d(kq), dim3(256), 0, st, M.ptr, kq, 0, kq - 1, lam);           // add_to_diag2
            hipLaunchKernelGGL(add_diag_kernel<real_t>, grid1d(1), dim3(256), 0, st, M.ptr, kq, kq - 1, kq, lam_last);
            hipLaunchKernelGGL(potrf_upper_kernel<real_t>, dim3(1), dim3(256), 0, st, M.ptr, kq);
            M.download(BeTBeChol, (size_t)kq * kq, st);
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(st));
        zero_strictly_lower(BtB, kk);
        if (with_bi) zero_strictly_lower(BiTBi, kki);
        if (with_c && CtCw != nullptr) zero_strictly_lower(CtCw, kc);
        if (BeTBeChol != nullptr && (C != nullptr || add_implicit_features) && !nonneg) zero_strictly_lower(BeTBeChol, kq);
        if (C != nullptr && CtUbias != nullptr && p > 0 && U_colmeans != nullptr && NA_as_zero_U)      // :10463-10470
            for (int f = 0; f < kc; f++) {
                double acc = 0;
                for (int_t j = 0; j < p; j++) acc += (double)C[(size_t)j * kc + f] * (double)U_colmeans[j];
                CtUbias[f] = (real_t)(-(double)w_user * acc);
            }
        return 0;
    });
}

// topN (src/common.c:5127-5380) for one user: scores of the candidate items on the device (one thread per item), a stable radix
// sort by descending score (rocPRIM; ties: the earlier candidate, i.e. the lower item id without an include list), the first n_top.
static int topn_one_user(const real_t *a_vec, int_t k_user, const real_t *B, int_t k_item, const real_t *biasB, real_t glob_mean, real_t biasA,
                         int_t k, int_t k_main, int_t *include_ix, int_t n_include, int_t *exclude_ix, int_t n_exclude, int_t *outp_ix,
                         real_t *outp_score, int_t n_top, int_t n)
{
    auto bad = [&](const char *msg) { g_last_error = msg; fprintf(stderr, "%s\n", msg); return 2; };
    if (a_vec == nullptr || B == nullptr || outp_ix == nullptr) return bad("topN: a_vec, B and outp_ix are required");
    if (include_ix != nullptr && exclude_ix != nullptr) return bad("Cannot pass both 'include_ix' and 'exclude_ix'.");
    if (n_top <= 0) return bad("'n_top' must be greater than zero.");
    if (exclude_ix != nullptr && n_exclude > n - n_top) return bad("Number of rankeable entities is less than 'n_top'");
    if (include_ix != nullptr && n_include > n) return bad("Number of entities to include is larger than 'n'.");
    if (include_ix != nullptr) for (int_t i = 0; i < n_include; i++) if (include_ix[i] < 0 || include_ix[i] >= n) return bad("'include_ix' contains invalid entries");
    if (exclude_ix != nullptr) for (int_t i = 0; i < n_exclude; i++) if (exclude_ix[i] < 0 || exclude_ix[i] >= n) return bad("'exclude_ix' contains invalid entries");
    for (int_t f = 0; f < k_user + k + k_main; f++) if (std::isnan((double)a_vec[f])) return bad("The latent factors contain NAN values");
    if (std::isnan((double)biasA)) return bad("The bias is a NAN value");
    const int k_pred = k + k_main, ktB = k_item + k + k_main;
    std::vector<int> cand;
    if (include_ix != nullptr) cand.assign(include_ix, include_ix + n_include);
    else {
        std::vector<char> out((size_t)n, 0);
        if (exclude_ix != nullptr) for (int_t i = 0; i < n_exclude; i++) out[(size_t)exclude_ix[i]] = 1;
        cand.reserve((size_t)n);
        for (int_t i = 0; i < n; i++) if (!out[(size_t)i]) cand.push_back((int)i);
    }
    const int ncand = (int)cand.size();
    if (ncand < n_top) return bad("Number of rankeable entities is less than 'n_top'");
    DeviceInfo dev;
    init_device(dev, -1);
    hipStream_t st = dev.stream;
    DevBuf<real_t> da, dB, dbias, sc_in, sc_out;
    DevBuf<int> id_in, id_out;
    DevBuf<unsigned char> tmp;
    da.upload(a_vec + k_user, (size_t)k_pred, st);
    dB.upload(B, (size_t)n * ktB, st);
    if (biasB != nullptr) dbias.upload(biasB, (size_t)n, st);
    id_in.upload(cand.data(), (size_t)ncand, st);
    sc_in.alloc((size_t)ncand); sc_out.alloc((size_t)ncand); id_out.alloc((size_t)ncand);
    hipLaunchKernelGGL(topn_one_user_scores_kernel<real_t>, grid1d((size_t)ncand), dim3(256), 0, st, da.ptr, k_pred, dB.ptr + k_item, (size_t)ktB,
                       biasB != nullptr ? dbias.ptr : nullptr, id_in.ptr, ncand, sc_in.ptr);
    HIP_CHECK(hipGetLastError());
    size_t bytes = 0;
    HIP_CHECK(rocprim::radix_sort_pairs_desc(nullptr, bytes, sc_in.ptr, sc_out.ptr, id_in.ptr, id_out.ptr, (size_t)ncand, 0, 8 * sizeof(real_t), st));
    tmp.alloc(std::max<size_t>(bytes, 1));
    HIP_CHECK(rocprim::radix_sort_pairs_desc(tmp.ptr, bytes, sc_in.ptr, sc_out.ptr, id_in.ptr, id_out.ptr, (size_t)ncand, 0, 8 * sizeof(real_t), st));
    std::vector<real_t> hs((size_t)n_top);
    std::vector<int> hi((size_t)n_top);
    id_out.download(hi.data(), (size_t)n_top, st);
    sc_out.download(hs.data(), (size_t)n_top, st);
    HIP_CHECK(hipStreamSynchronize(st));
    for (int_t i = 0; i < n_top; i++) {
        outp_ix[i] = hi[(size_t)i];
        if (outp_score != nullptr) outp_score[i] = hs[(size_t)i] + (glob_mean + biasA);          // common.c:5349-5358
    }
    return 0;
}

int_t topN_old_collective_explicit(
    real_t *a_vec, real_t a_bias, real_t *A, real_t *biasA, int_t row_index, real_t *B, real_t *biasB, real_t glob_mean,
    int_t k, int_t k_user, int_t k_item, int_t k_main, int_t *include_ix, int_t n_include, int_t *exclude_ix, int_t n_exclude,
    int_t *outp_ix, real_t *outp_score, int_t n_top, int_t n, int_t n_max, bool include_all_X, int nthreads)
{
    (void)nthreads;
    return guarded([&]() {
        if (include_all_X || n == 0) n = n_max;                                                   // collective.c:11560-11561
        if (a_vec != nullptr)
            return topn_one_user(a_vec, k_user, B, k_item, biasB, glob_mean, a_bias, k, k_main, include_ix, n_include, exclude_ix, n_exclude, outp_ix, outp_score, n_top, n);
        if (A == nullptr) { g_last_error = "topN_old_collective_explicit: a_vec or A is required"; return 2; }
        return topn_one_user(A + (size_t)row_index * (size_t)(k_user + k + k_main), k_user, B, k_item, biasB, glob_mean,
                             biasA == nullptr ? (real_t)0 : biasA[row_index], k, k_main, include_ix, n_include, exclude_ix, n_exclude, outp_ix,
                             outp_score, n_top, n);
    });
}

int_t topN_old_collective_implicit(
    real_t *a_vec, real_t *A, int_t row_index, real_t *B, int_t k, int_t k_user, int_t k_item, int_t k_main,
    int_t *include_ix, int_t n_include, int_t *exclude_ix, int_t n_exclude, int_t *outp_ix, real_t *outp_score, int_t n_top, int_t n, int nthreads)
{
    return topN_old_collective_explicit(a_vec, (real_t)0, A, nullptr, row_index, B, nullptr, (real_t)0, k, k_user, k_item, k_main, include_ix, n_include,
                                        exclude_ix, n_exclude, outp_ix, outp_score, n_top, n, n, false, nthreads);
}

int cmfrec_hip_optimizeA_collective(real_t *A, size_t lda, const real_t *B, size_t ldb, const real_t *C, int_t m,
                                    int_t m_u, int_t n, int_t p, int_t k, int_t k_main, int_t k_user, int_t k_item,
                                    const size_t Xcsr_p[], const int_t Xcsr_i[], const real_t *Xcsr,
                                    const real_t *bias_sub, const real_t *U, real_t lam, real_t w_user,
                                    real_t lam_last, bool scale_lam, bool scale_lam_sideinfo)
{
    return guarded([&]() {
        if (m_u > m) { g_last_error = "cmfrec_hip: m_u > m is not supported"; return 2; }
        DeviceInfo dev;
        init_device(dev, -1);
        const int kc = k_user + k, kt = k_user + k + k_main;
        DevBuf<real_t> dA, dB, dC, dU, dG, dbias;
        GramWorkspace gws;
        SparseShard X;
        dA.upload(A, (size_t)m * lda, dev.stream);
        dB.upload(B, (size_t)n * ldb, dev.stream);
        dC.upload(C, (size_t)p * kc, dev.stream);
        dU.upload(U, (size_t)m_u * p, dev.stream);
        if (bias_sub) dbias.upload(bias_sub, n, dev.stream);
        dG.alloc((size_t)kc * kc);
        X.opp_row_bytes_hint = (size_t)k * sizeof(real_t);
        shard_from_csr(X, m, Xcsr_p, Xcsr_i, Xcsr, n, dev.stream);
        launch_gram(dev, gws, dC.ptr, (size_t)kc, p, kc, dG.ptr, w_user, (real_t)0);
        // collective.c:4817-4822 zeroes max(m,m_u)*lda - (lda-k_totA) elements
        HIP_CHECK(hipMemsetAsync(dA.ptr, 0, ((size_t)m * lda - (lda - (size_t)kt)) * sizeof(real_t), dev.stream));
        launch_gemm<false>(dev, m_u, kc, p, w_user, dU.ptr, (size_t)p, dC.ptr, (size_t)kc, dA.ptr, lda);
        CholCall c{dA.ptr, lda, dB.ptr + k_item, ldb, kt, k_user, bias_sub ? dbias.ptr : nullptr, dG.ptr, kc, m_u, p,
                   lam, lam_last, (bool)(scale_lam || scale_lam_sideinfo), scale_lam_sideinfo, false, CHOL_COLLECTIVE};
        LowRankScratch lrs;
        int rc = (m_u >= m) ? launch_collective_lowrank(dev, lrs, c, X, dC.ptr, dU.ptr, p, w_user, k, n) : -1;
        if (rc < 0) rc = launch_chol(dev, c, &X);
        dA.download(A, (size_t)m * lda, dev.stream);
        HIP_CHECK(hipStreamSynchronize(dev.stream));
        return rc;
    });
}

// Collective half-step with SPARSE side information (U given as CSR over the m_u rows that have any): the branches of
// collective_closed_form_block (collective.c:1223-1847) / collective_closed_form_block_implicit (:1849-2131) with
// u_vec == NULL, u_vec_sp != NULL, !NA_as_zero_U: the row's present attributes add  w * C_j C_j^T  to the upper-left
// [k_user+k]^2 block (:1636-1653, :2003-2011) and  w * u_j C_j  to the right-hand side (:1719-1731, :2013-2021); under
// scale_lam_sideinfo lambda is also multiplied by their number (:1338-1346).  Both gathers -- rows of B through X, rows of
// C through U -- run through the same staging ring and matrix-core rank-1 updates of one kernel launch.
int cmfrec_hip_optimizeA_collective_sparse(real_t *A, size_t lda, const real_t *B, size_t ldb, const real_t *C, int_t m,
                                           int_t m_u, int_t n, int_t p, int_t k, int_t k_main, int_t k_user, int_t k_item,
                                           const size_t Xcsr_p[], const int_t Xcsr_i[], const real_t *Xcsr,
                                           const real_t *bias_sub, const size_t Ucsr_p[], const int_t Ucsr_i[],
                                           const real_t *Ucsr, real_t lam, real_t w_user, real_t lam_last, bool scale_lam,
                                           bool scale_lam_sideinfo, bool implicit)
{
    return guarded([&]() {
        if (m_u > m) { g_last_error = "cmfrec_hip: m_u > m is not supported here"; return 2; }
        DeviceInfo dev;
        init_device(dev, -1);
        const int kc = k_user + k, kk = k + k_main, kt = k_user + kk;
        DevBuf<real_t> dA, dB, dC, dG, dM, dbias;
        GramWorkspace gws;
        SparseShard X, Us;
        dA.alloc((size_t)m * lda);
        HIP_CHECK(hipMemsetAsync(dA.ptr, 0, dA.n * sizeof(real_t), dev.stream));          // collective.c:4817-4822, :6018-6019
        dB.upload(B, (size_t)n * ldb, dev.stream);
        dC.upload(C, (size_t)p * kc, dev.stream);
        if (bias_sub) dbias.upload(bias_sub, n, dev.stream);
        X.opp_row_bytes_hint = (size_t)k * sizeof(real_t);
        shard_from_csr(X, m, Xcsr_p, Xcsr_i, Xcsr, n, dev.stream);
        std::vector<size_t> up((size_t)m + 1);
        for (int r = 0; r <= m; r++) up[r] = Ucsr_p[std::min(r, m_u)];
        shard_from_csr(Us, m, up.data(), Ucsr_i, Ucsr, p, dev.stream);
        int rc;
        if (implicit) {
            dG.alloc((size_t)kk * kk); dM.alloc((size_t)kt * kt);
            launch_gram(dev, gws, dB.ptr + k_item, ldb, n, kk, dG.ptr, (real_t)1, lam);
            hipLaunchKernelGGL(betbe_base_kernel<real_t>, grid1d((size_t)kt * kt), dim3(256), 0, dev.stream, dG.ptr, kk, k_user, lam,
                               dM.ptr);
            CholCall c{dA.ptr, lda, dB.ptr + k_item, ldb, kt, k_user, nullptr, nullptr, kc, m_u, p, lam, lam, false, false, false,
                       CHOL_COLLECTIVE_IMPLICIT, dM.ptr};
            c.X2 = &Us; c.B2 = dC.ptr; c.ldb2 = (size_t)kc; c.kc2 = kc; c.w2 = w_user;
            rc = launch_chol(dev, c, &X);
        } else {
            CholCall c{dA.ptr, lda, dB.ptr + k_item, ldb, kt, k_user, bias_sub ? dbias.ptr : nullptr, nullptr, kc, m_u, p, lam,
                       lam_last, (bool)(scale_lam || scale_lam_sideinfo), scale_lam_sideinfo, false, CHOL_COLLECTIVE};
            c.X2 = &Us; c.B2 = dC.ptr; c.ldb2 = (size_t)kc; c.kc2 = kc; c.w2 = w_user;
            rc = launch_chol(dev, c, &X);
        }
        std::vector<real_t> tmp((size_t)m * lda);
        dA.download(tmp.data(), (size_t)m * lda, dev.stream);
        HIP_CHECK(hipStreamSynchronize(dev.stream));
        for (int r = 0; r < m; r++) memcpy(A + (size_t)r * lda, tmp.data() + (size_t)r * lda, (size_t)kt * sizeof(real_t));
        return rc;
    });
}

// Factors of rows that were not part of the fit, all of them in one pass (the step after the path):
// factors_collective_explicit_multiple (collective.c:10865-11174) / factors_collective_implicit_multiple
// (:11176-11340) restricted to sparse X (COO or CSR, values already transformed: minus the global mean, times
// alpha) and dense side information without missing values.  Per row (collective_factors_warm :3555-3964,
// collective_factors_cold :3309-3440, the *_implicit twins :3442-3553, :3966-4087) this is the closed-form row
// update of the fit with B (and C) fixed:
//   explicit, no U:       factors_closed_form on [B | 1]            -> Cholesky mode EXPLICIT
//   explicit, U, nnz > 0: collective_closed_form_block              -> mode COLLECTIVE
//   explicit, U, nnz = 0: "cold" (C^T C + (lam/w)(p if scale_lam_sideinfo) I)^-1 C^T u, bias 0; the last of the
//                         k_user+k unknowns keeps the unscaled lam/w (scale_bias_const := scale_lam_sideinfo in the
//                         call at :3397-3411); with TransCtCinvCt given it is u^T TransCtCinvCt (:3380-3386)
//   implicit:             factors_implicit_chol / collective_closed_form_block_implicit -> modes IMPLICIT /
//                         COLLECTIVE_IMPLICIT; lam_x is what sits on the diagonal of the X block (the reference adds
//                         the *unscaled* lam there when it builds BtB itself, :11270-11280, and lam / w_main on the
//                         k_user block), BtB_pre (lam included) replaces B^T B + lam_x I when given.
// l1_lam / l1_lam_bias: the L1 penalty of the row systems and of the bias unknown (solve_elasticnet instead of the Cholesky
// substitution, common.c:2228-2294; collective_factors_warm / _cold hand them down like lam / lam_bias, collective.c:3571-3931,
// :3321-3400), already divided by w_main.  Rows that only have side information take l1_lam / w_user (:3395).
static int factors_multiple_impl(real_t *A, real_t *biasA, int_t m_x, int_t m_u, int_t p, const real_t *U,
                                 const real_t *U_colmeans, const int_t ixA[], const int_t ixB[], const real_t *X,
                                 size_t nnz, const size_t Xcsr_p[], const int_t Xcsr_i[], const real_t *Xcsr,
                                 const real_t *B, int_t n, const real_t *C, const real_t *biasB, int_t k, int_t k_user,
                                 int_t k_item, int_t k_main, real_t lam, real_t lam_bias, real_t lam_x, real_t w_user,
                                 bool implicit, bool scale_lam, bool scale_lam_sideinfo, bool scale_bias_const,
                                 const real_t *BtB_pre, const real_t *TransCtCinvCt_pre,
                                 const int_t U_row[], const int_t U_col[], const real_t *U_sp, size_t nnz_U,
                                 const size_t U_csr_p[], const int_t U_csr_i[], const real_t *U_csr, bool nonneg,
                                 real_t l1_lam, real_t l1_lam_bias)
{
    return guarded([&]() {
        // sparse side information (COO or CSR over m_u rows, missing = absent): second gather source of the row kernel
        const bool spU = (U == nullptr && p > 0 && ((nnz_U > 0 && U_row && U_col && U_sp) || U_csr_p));
        const int m_max = std::max(m_x, (p > 0 && (U || spU)) ? m_u : 0);
        if (m_max <= 0) return 0;
        if (!A || !B || n <= 0 || k < 0 || (p > 0 && (U || spU) && !C) || (nnz > 0 && !Xcsr_p && (!ixA || !ixB || !X)) ||
            (implicit && biasA)) {
            g_last_error = "cmfrec_hip_factors_multiple: invalid arguments";
            return 2;
        }
        if (spU && !implicit && scale_lam_sideinfo) {
            g_last_error = "cmfrec_hip_factors_multiple: sparse side information with scale_lam_sideinfo is not supported "
                           "(the reference scales the rows without observations differently, collective.c:3397-3411)";
            return 2;
        }
        if (!(p > 0 && (U || spU))) { p = 0; m_u = 0; }
        // index validation on the host, before anything reaches the device (the fit entry points do the same,
        // fit.hip): a row id outside [0, m_x) or an item id outside [0, n) would corrupt device memory silently
        {
            auto bad = [](const char *what) { g_last_error = std::string("cmfrec_hip_factors_multiple: ") + what; return 2; };
            if (Xcsr_p) {
                const size_t nz = Xcsr_p[m_x];
                for (int r = 0; r < m_x; r++) if (Xcsr_p[r] > Xcsr_p[r + 1]) return bad("Xcsr_p is not non-decreasing");
                if (nz > 0 && (!Xcsr_i || !Xcsr)) return bad("Xcsr_i / Xcsr missing");
                for (size_t e = 0; e < nz; e++) if (Xcsr_i[e] < 0 || Xcsr_i[e] >= n) return bad("item index of X outside [0, n)");
            } else {
                for (size_t e = 0; e < nnz; e++)
                    if (ixA[e] < 0 || ixA[e] >= m_x || ixB[e] < 0 || ixB[e] >= n) return bad("row / item index of X outside [0, m_x) x [0, n)");
            }
            if (spU) {
                if (U_csr_p) {
                    const size_t nz = U_csr_p[m_u];
                    for (int r = 0; r < m_u; r++) if (U_csr_p[r] > U_csr_p[r + 1]) return bad("U_csr_p is not non-decreasing");
                    if (nz > 0 && (!U_csr_i || !U_csr)) return bad("U_csr_i / U_csr missing");
                    for (size_t e = 0; e < nz; e++) if (U_csr_i[e] < 0 || U_csr_i[e] >= p) return bad("attribute index of U outside [0, p)");
                } else {
                    for (size_t e = 0; e < nnz_U; e++)
                        if (U_row[e] < 0 || U_row[e] >= m_u || U_col[e] < 0 || U_col[e] >= p) return bad("row / attribute index of U outside [0, m_u) x [0, p)");
                }
            }
        }
        DeviceInfo dev;
        init_device(dev, -1);
        hipStream_t st = dev.stream;
        const int ub = biasA ? 1 : 0;
        // non-negative factors: every row system below goes through solve_nonneg with the reference's sweep limit for
        // new rows, 10 x the number of unknowns (collective.c:3401, :3800-3931, :4041-4054)
        dev.nonneg_now = nonneg;
        dev.max_cd_steps = 10 * (k_user + k + k_main + ub);
        const bool l1on = l1_lam != (real_t)0 || (ub && l1_lam_bias != (real_t)0);
        if (l1on && TransCtCinvCt_pre) {
            g_last_error = "cmfrec_hip_factors_multiple: TransCtCinvCt cannot be used with an L1 penalty (collective.c:3378)";
            return 2;
        }
        dev.l1_now = l1_lam;
        dev.l1_last_now = ub ? l1_lam_bias : l1_lam;
        dev.l1_scale = 1;
        const int kc = k_user + k, kk = k + k_main, kt = k_user + kk + ub, ktA = k_user + kk;
        const size_t ldb_host = (size_t)(k_item + kk), ldB = ldb_host + ub, ldA = (size_t)kt;
        DevBuf<real_t> dA, dB, dC, dU, dmeans, dbias, dG, dM, dCtC, dcold, dT;
        GramWorkspace gws;
        SparseShard Xs;
        dB.alloc((size_t)n * ldB);
        HIP_CHECK(hipMemcpy2DAsync(dB.ptr, ldB * sizeof(real_t), B, ldb_host * sizeof(real_t), ldb_host * sizeof(real_t),
                                   (size_t)n, hipMemcpyHostToDevice, st));
        if (ub) hipLaunchKernelGGL(col_fill_kernel<real_t>, grid1d(n), dim3(256), 0, st, dB.ptr, ldB, n, (int)ldb_host, (real_t)1);
        if (biasB) dbias.upload(biasB, (size_t)n, st);
        dA.alloc((size_t)m_max * ldA);
        HIP_CHECK(hipMemsetAsync(dA.ptr, 0, dA.n * sizeof(real_t), st));
        if (Xcsr_p) {
            std::vector<size_t> pp((size_t)m_max + 1);
            for (int r = 0; r <= m_max; r++) pp[r] = Xcsr_p[std::min(r, m_x)];
            shard_from_csr(Xs, m_max, pp.data(), Xcsr_i, Xcsr, n, st);
        } else if (nnz == 0) {
            std::vector<size_t> pp((size_t)m_max + 1, 0);
            shard_from_csr(Xs, m_max, pp.data(), nullptr, nullptr, n, st);
        } else {
            DevBuf<int> dr, dc; DevBuf<real_t> dv;
            dr.upload(ixA, std::max<size_t>(nnz, 1), st); dc.upload(ixB, std::max<size_t>(nnz, 1), st);
            dv.upload(X, std::max<size_t>(nnz, 1), st);
            shard_from_coo(Xs, m_max, n, dr.ptr, dc.ptr, dv.ptr, nnz, (real_t)0, (real_t)1, st);
            HIP_CHECK(hipStreamSynchronize(st));
        }
        SparseShard Us;
        if (spU) {
            dC.upload(C, (size_t)p * kc, st);
            if (U_csr_p) {
                std::vector<size_t> up((size_t)m_max + 1);
                for (int r = 0; r <= m_max; r++) up[r] = U_csr_p[std::min(r, m_u)];
                shard_from_csr(Us, m_max, up.data(), U_csr_i, U_csr, p, st);
            } else {
                DevBuf<int> dr, dc; DevBuf<real_t> dv;
                dr.upload(U_row, nnz_U, st); dc.upload(U_col, nnz_U, st); dv.upload(U_sp, nnz_U, st);
                shard_from_coo(Us, m_max, p, dr.ptr, dc.ptr, dv.ptr, nnz_U, (real_t)0, (real_t)1, st);
                HIP_CHECK(hipStreamSynchronize(st));
            }
        } else if (p > 0) {
            dC.upload(C, (size_t)p * kc, st);
            dU.upload(U, (size_t)m_u * p, st);
            if (U_colmeans) {
                dmeans.upload(U_colmeans, (size_t)p, st);
                hipLaunchKernelGGL(sub_colmeans_kernel<real_t>, grid1d((size_t)m_u * p), dim3(256), 0, st, dU.ptr, (size_t)m_u, p,
                                   dmeans.ptr);
            }
            dCtC.alloc((size_t)kc * kc);
        }
        const real_t *opp = dB.ptr + k_item;
        const real_t *bias_sub = biasB ? dbias.ptr : nullptr;
        int rc = 0;
        if (implicit) {
            dG.alloc((size_t)kk * kk);
            if (BtB_pre) dG.upload(BtB_pre, (size_t)kk * kk, st);
            else launch_gram(dev, gws, opp, ldB, n, kk, dG.ptr, (real_t)1, lam_x);
            if (p == 0) {
                CholCall c{dA.ptr + k_user, ldA, opp, ldB, kk, 0, nullptr, dG.ptr, 0, 0, 0, lam, lam, false, false, false,
                           CHOL_IMPLICIT};
                rc = launch_chol(dev, c, &Xs);
            } else if (spU) {
                dM.alloc((size_t)ktA * ktA);
                hipLaunchKernelGGL(betbe_base_kernel<real_t>, grid1d((size_t)ktA * ktA), dim3(256), 0, st, dG.ptr, kk, k_user, lam,
                                   dM.ptr);
                CholCall c{dA.ptr, ldA, opp, ldB, ktA, k_user, nullptr, nullptr, kc, m_u, p, lam, lam, false, false, false,
                           CHOL_COLLECTIVE_IMPLICIT, dM.ptr};
                c.X2 = &Us; c.B2 = dC.ptr; c.ldb2 = (size_t)kc; c.kc2 = kc; c.w2 = w_user;
                rc = launch_chol(dev, c, &Xs);
            } else {
                dM.alloc((size_t)ktA * ktA);
                hipLaunchKernelGGL(betbe_base_kernel<real_t>, grid1d((size_t)ktA * ktA), dim3(256), 0, st, dG.ptr, kk, k_user, lam,
                                   dM.ptr);
                launch_gram(dev, gws, dC.ptr, (size_t)kc, p, kc, dCtC.ptr, w_user, (real_t)0);
                launch_gemm<false>(dev, m_u, kc, p, w_user, dU.ptr, (size_t)p, dC.ptr, (size_t)kc, dA.ptr, ldA);
                CholCall c{dA.ptr, ldA, opp, ldB, ktA, k_user, nullptr, dCtC.ptr, kc, m_u, p, lam, lam, false, false, false,
                           CHOL_COLLECTIVE_IMPLICIT, dM.ptr};
                rc = launch_chol(dev, c, &Xs);
            }
        } else if (p == 0) {
            CholCall c{dA.ptr + k_user, ldA, opp, ldB, kk + ub, 0, bias_sub, nullptr, 0, 0, 0, lam, lam_bias,
                       (bool)(scale_lam || scale_lam_sideinfo), false, scale_bias_const, CHOL_EXPLICIT};
            rc = launch_chol(dev, c, &Xs);
        } else if (spU) {
            // rows with attributes but no observations come out of the same launch: without scale_lam_sideinfo the
            // "cold" solution (collective.c:3309-3440 with u_vec_sp) is the block system with an empty X part
            CholCall c{dA.ptr, ldA, opp, ldB, kt, k_user, bias_sub, nullptr, kc, m_u, p, lam, lam_bias, scale_lam, false,
                       scale_bias_const, CHOL_COLLECTIVE};
            c.X2 = &Us; c.B2 = dC.ptr; c.ldb2 = (size_t)kc; c.kc2 = kc; c.w2 = w_user;
            rc = launch_chol(dev, c, &Xs);
        } else {
            launch_gram(dev, gws, dC.ptr, (size_t)kc, p, kc, dCtC.ptr, w_user, (real_t)0);
            launch_gemm<false>(dev, m_u, kc, p, w_user, dU.ptr, (size_t)p, dC.ptr, (size_t)kc, dA.ptr, ldA);
            CholCall c{dA.ptr, ldA, opp, ldB, kt, k_user, bias_sub, dCtC.ptr, kc, m_u, p, lam, lam_bias,
                       (bool)(scale_lam || scale_lam_sideinfo), scale_lam_sideinfo, scale_bias_const, CHOL_COLLECTIVE};
            rc = launch_chol(dev, c, &Xs);
            if (rc == 0) {
                // cold rows
                dcold.alloc((size_t)m_u * kc);
                if (TransCtCinvCt_pre) {
                    dT.upload(TransCtCinvCt_pre, (size_t)p * kc, st);
                    launch_gemm<false>(dev, m_u, kc, p, (real_t)1, dU.ptr, (size_t)p, dT.ptr, (size_t)kc, dcold.ptr, (size_t)kc);
                } else {
                    dM.alloc((size_t)kc * kc);
                    const real_t lc = lam / w_user;
                    launch_gram(dev, gws, dC.ptr, (size_t)kc, p, kc, dM.ptr, (real_t)1, scale_lam_sideinfo ? lc * (real_t)p : lc);
                    if (scale_lam_sideinfo)
                        hipLaunchKernelGGL(add_diag_kernel<real_t>, dim3(1), dim3(64), 0, st, dM.ptr, kc, kc - 1, kc,
                                           lc - lc * (real_t)p);
                    launch_gemm<false>(dev, m_u, kc, p, (real_t)1, dU.ptr, (size_t)p, dC.ptr, (size_t)kc, dcold.ptr, (size_t)kc);
                    CholCall cc{dcold.ptr, (size_t)kc, nullptr, 0, kc, 0, nullptr, dM.ptr, 0, 0, 0, 0, 0, false, false, false,
                                CHOL_PREFILLED};
                    // factors_closed_form on C with l1_lam / w_user, scaled by p like lam except on the last unknown (collective.c:3385-3400)
                    dev.l1_last_now = l1_lam / w_user;
                    dev.l1_now = scale_lam_sideinfo ? dev.l1_last_now * (real_t)p : dev.l1_last_now;
                    rc = launch_chol(dev, cc, nullptr, m_u);
                    dev.l1_now = l1_lam; dev.l1_last_now = ub ? l1_lam_bias : l1_lam;
                }
                hipLaunchKernelGGL(cold_select_kernel<real_t>, dim3(m_u), dim3(64), 0, st, dA.ptr, ldA, kt, kc, dcold.ptr,
                                   Xs.p.ptr, m_u);
            }
        }
        HIP_CHECK(hipGetLastError());
        if (rc) return rc;
        HIP_CHECK(hipMemcpy2DAsync(A, (size_t)ktA * sizeof(real_t), dA.ptr, ldA * sizeof(real_t), (size_t)ktA * sizeof(real_t),
                                   (size_t)m_max, hipMemcpyDeviceToHost, st));
        if (ub) {
            DevBuf<real_t> dba;
            dba.alloc((size_t)m_max);
            hipLaunchKernelGGL(col_extract_kernel<real_t>, grid1d(m_max), dim3(256), 0, st, dA.ptr, ldA, m_max, ktA, dba.ptr);
            dba.download(biasA, (size_t)m_max, st);
            HIP_CHECK(hipStreamSynchronize(st));
        }
        HIP_CHECK(hipStreamSynchronize(st));
        return 0;
    });
}

int cmfrec_hip_factors_multiple(real_t *A, real_t *biasA, int_t m_x, int_t m_u, int_t p, const real_t *U,
                                const real_t *U_colmeans, const int_t ixA[], const int_t ixB[], const real_t *X,
                                size_t nnz, const size_t Xcsr_p[], const int_t Xcsr_i[], const real_t *Xcsr,
                                const real_t *B, int_t n, const real_t *C, const real_t *biasB, int_t k, int_t k_user,
                                int_t k_item, int_t k_main, real_t lam, real_t lam_bias, real_t lam_x, real_t w_user,
                                bool implicit, bool scale_lam, bool scale_lam_sideinfo, bool scale_bias_const,
                                const real_t *BtB_pre, const real_t *TransCtCinvCt_pre,
                                const int_t U_row[], const int_t U_col[], const real_t *U_sp, size_t nnz_U,
                                const size_t U_csr_p[], const int_t U_csr_i[], const real_t *U_csr, bool nonneg)
{
    return factors_multiple_impl(A, biasA, m_x, m_u, p, U, U_colmeans, ixA, ixB, X, nnz, Xcsr_p, Xcsr_i, Xcsr, B, n, C, biasB, k,
                                 k_user, k_item, k_main, lam, lam_bias, lam_x, w_user, implicit, scale_lam, scale_lam_sideinfo,
                                 scale_bias_const, BtB_pre, TransCtCinvCt_pre, U_row, U_col, U_sp, nnz_U, U_csr_p, U_csr_i, U_csr,
                                 nonneg, (real_t)0, (real_t)0);
}

int cmfrec_hip_factors_multiple_l1(real_t *A, real_t *biasA, int_t m_x, int_t m_u, int_t p, const real_t *U,
                                   const real_t *U_colmeans, const int_t ixA[], const int_t ixB[], const real_t *X,
                                   size_t nnz, const size_t Xcsr_p[], const int_t Xcsr_i[], const real_t *Xcsr,
                                   const real_t *B, int_t n, const real_t *C, const real_t *biasB, int_t k, int_t k_user,
                                   int_t k_item, int_t k_main, real_t lam, real_t lam_bias, real_t lam_x, real_t w_user,
                                   bool implicit, bool scale_lam, bool scale_lam_sideinfo, bool scale_bias_const,
                                   const real_t *BtB_pre, const real_t *TransCtCinvCt_pre,
                                   const int_t U_row[], const int_t U_col[], const real_t *U_sp, size_t nnz_U,
                                   const size_t U_csr_p[], const int_t U_csr_i[], const real_t *U_csr, bool nonneg,
                                   real_t l1_lam, real_t l1_lam_bias)
{
    return factors_multiple_impl(A, biasA, m_x, m_u, p, U, U_colmeans, ixA, ixB, X, nnz, Xcsr_p, Xcsr_i, Xcsr, B, n, C, biasB, k,
                                 k_user, k_item, k_main, lam, lam_bias, lam_x, w_user, implicit, scale_lam, scale_lam_sideinfo,
                                 scale_bias_const, BtB_pre, TransCtCinvCt_pre, U_row, U_col, U_sp, nnz_U, U_csr_p, U_csr_i, U_csr,
                                 nonneg, l1_lam, l1_lam_bias);
}

}  // extern "C"

// ---- on-device self-test of the cross-lane primitives (lanes.hpp) -----------------------------
namespace cmfhip {
__global__ void lanes_selftest_kernel(real_t *out)
{
    const int lane = threadIdx.x;
    const real_t x = (real_t)(lane * 3 + 1);
    const real_t y = (real_t)(1000 + lane);
    real_t *o = out + lane;
    o[0 * 64] = lanes::xor1(x);
    o[1 * 64] = lanes::xor2(x);
    o[2 * 64] = lanes::xor4(x);
    o[3 * 64] = lanes::xor8(x);
    o[4 * 64] = lanes::recv_xor4(x, y);
    o[5 * 64] = lanes::recv_xor8(x, y);
    o[6 * 64] = lanes::tswap32_add(x, y);
    o[7 * 64] = lanes::tswap16_add(x, y);
    o[8 * 64] = lanes::bcast8<0>(x);
    o[9 * 64] = lanes::bcast8<3>(x);
    o[10 * 64] = lanes::bcast8<5>(x);
    o[11 * 64] = lanes::bcast8<7>(x);
    o[12 * 64] = lanes::wave_sum(x);
    real_t v[8];
    for (int i = 0; i < 8; i++) v[i] = (real_t)(lane + 100 * i);
    o[13 * 64] = treduce8_low<real_t>(v, lane);
    o[14 * 64] = treduce8_high<real_t>(v, lane);
    o[15 * 64] = lanes::half_mirror(x);
    o[16 * 64] = (real_t)lanes::bcast8<6>(lane * 5 + 2);
}
}  // namespace cmfhip

// Timing probe of the dense contraction C[M, N] = op(A) B on the device (tools/microbench/gemm_probe.py): milliseconds per call of
// the library's own MFMA kernel and -- when librocblas can be loaded at run time; the shared objects are not linked against it since
// round 6 -- of rocBLAS on the same operands, and the largest difference between their results relative to the largest entry
// (ms_rocblas = -1 and the difference against a plain one-thread-per-output kernel when the library is not there).
// transa != 0: A is stored [K, M].
namespace cmfhip {
template <typename T>
__global__ void gemm_naive_kernel(int M, int N, int K, int transa, const T *__restrict__ A, size_t lda, const T *__restrict__ B, size_t ldb,
                                  T *__restrict__ C, size_t ldc)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)M * N) return;
    const int i = (int)(e / N), j = (int)(e % N);
    double acc = 0;
    for (int q = 0; q < K; q++) acc += (double)(transa ? A[(size_t)q * lda + i] : A[(size_t)i * lda + q]) * (double)B[(size_t)q * ldb + j];
    C[(size_t)i * ldc + j] = (T)acc;
}
struct RocBlasProbe {          // (measurement tool only)
    typedef int (*create_t)(void **);
    typedef int (*destroy_t)(void *);
    typedef int (*set_stream_t)(void *, hipStream_t);
    typedef int (*gemm_t)(void *, int, int, int, int, int, const real_t *, const real_t *, int, const real_t *, int, const real_t *, real_t *, int);
    create_t create = nullptr; destroy_t destroy = nullptr; set_stream_t set_stream = nullptr; gemm_t gemm = nullptr;
    RocBlasProbe()
    {
        void *lib = dlopen("librocblas.so.5", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("librocblas.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) return;
        create = (create_t)dlsym(lib, "rocblas_create_handle"); destroy = (destroy_t)dlsym(lib, "rocblas_destroy_handle");
        set_stream = (set_stream_t)dlsym(lib, "rocblas_set_stream");
        gemm = (gemm_t)dlsym(lib, sizeof(real_t) == 4 ? "rocblas_sgemm" : "rocblas_dgemm");
    }
    bool ok() const { return create && destroy && set_stream && gemm; }
};
}  // namespace cmfhip
extern "C" int cmfrec_hip_gemm_probe(int M, int N, int K, int transa, int reps, double *ms_own, double *ms_rocblas, double *max_rel_diff)
{
    return guarded([&]() {
        DeviceInfo dev;
        init_device(dev, -1);
        DevBuf<real_t> A, B, C1, C2;
        A.alloc((size_t)M * K); B.alloc((size_t)K * N); C1.alloc((size_t)M * N); C2.alloc((size_t)M * N);
        std::vector<real_t> ha((size_t)M * K), hb((size_t)K * N);
        unsigned long long x = 88172645463325252ull;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (real_t)((double)(x >> 11) / 9007199254740992.0 - 0.5); };
        for (auto &v : ha) v = rnd();
        for (auto &v : hb) v = rnd();
        A.upload(ha.data(), ha.size(), dev.stream); B.upload(hb.data(), hb.size(), dev.stream);
        const size_t lda = transa ? (size_t)M : (size_t)K;
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
        static const RocBlasProbe rb;
        void *h = nullptr;
        const bool have_lib = rb.ok() && rb.create(&h) == 0 && rb.set_stream(h, dev.stream) == 0;
        auto run = [&](bool own, real_t *C, double *ms) {
            const real_t one = 1, zero = 0;
            for (int r = 0; r < reps + 1; r++) {
                if (r == 1) HIP_CHECK(hipEventRecord(e0, dev.stream));
                if (!own)   // row-major C is the column-major C^T = B^T op(A)^T: first operand B, second operand A with the flag inverted
                    (void)rb.gemm(h, 111 /* none */, transa ? 112 /* transpose */ : 111, N, M, K, &one, B.ptr, N, A.ptr, (int)lda, &zero, C, N);
                else if (transa) launch_gemm<true>(dev, M, N, K, (real_t)1, A.ptr, lda, B.ptr, (size_t)N, C, (size_t)N);
                else launch_gemm<false>(dev, M, N, K, (real_t)1, A.ptr, lda, B.ptr, (size_t)N, C, (size_t)N);
            }
            HIP_CHECK(hipEventRecord(e1, dev.stream));
            HIP_CHECK(hipEventSynchronize(e1));
            float t = 0;
            HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
            if (ms) *ms = (double)t / std::max(reps, 1);
        };
        run(true, C1.ptr, ms_own);
        if (have_lib) run(false, C2.ptr, ms_rocblas);
        else {
            if (ms_rocblas) *ms_rocblas = -1.0;
            hipLaunchKernelGGL(gemm_naive_kernel<real_t>, grid1d((size_t)M * N), dim3(256), 0, dev.stream, M, N, K, transa, A.ptr, lda, B.ptr, (size_t)N,
                               C2.ptr, (size_t)N);
            HIP_CHECK(hipGetLastError());
        }
        std::vector<real_t> h1((size_t)M * N), h2((size_t)M * N);
        C1.download(h1.data(), h1.size(), dev.stream); C2.download(h2.data(), h2.size(), dev.stream);
        HIP_CHECK(hipStreamSynchronize(dev.stream));
        if (h) (void)rb.destroy(h);
        double mx = 0, df = 0;
        for (size_t e = 0; e < h1.size(); e++) { mx = std::max(mx, std::fabs((double)h2[e])); df = std::max(df, std::fabs((double)h1[e] - (double)h2[e])); }
        if (max_rel_diff) *max_rel_diff = df / std::max(mx, 1e-300);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        return 0;
    });
}

// The symmetric eigen-decomposition of the low-rank path on its own (tests/test_gpu_operators.py::test_sym_eig,
// tools/microbench/eig_probe.py): A [n, n] symmetric -> Q [n, n] (Q[i][c] = component i of eigenvector c), lam [n] (clamped at
// zero); method 0 = tridiagonalisation + QL (eig_kernels.hpp), 1 = the one-workgroup Jacobi kernel.  ms: device time of `reps` runs.
extern "C" int cmfrec_hip_sym_eig(int n, const real_t *A, real_t *Q, real_t *lam, int method, int reps, double *ms)
{
    return guarded([&]() {
        if (n < 2 || n > EIG_MAX_N) { g_last_error = "cmfrec_hip_sym_eig: 2 <= n <= 320"; return 2; }
        DeviceInfo dev;
        init_device(dev, -1);
        DevBuf<real_t> dA, dQ, dQt, dL;
        DevBuf<double> W, V, D, E, Tau;
        DevBuf<int> info;
        const size_t nn = (size_t)n * n;
        dA.upload(A, nn, dev.stream); dQ.alloc(nn); dQt.alloc(nn); dL.alloc((size_t)n);
        W.alloc(nn); V.alloc(nn); D.alloc((size_t)n); E.alloc((size_t)n); Tau.alloc((size_t)n); info.alloc(1);
        HIP_CHECK(hipMemsetAsync(info.ptr, 0, sizeof(int), dev.stream));
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
        const bool wide = n > 288;
        const int rows = wide ? 32 : 64;
        const size_t smem = ((size_t)n * rows + 4 * (size_t)n) * sizeof(double);
        auto kern = wide ? eig_ql_rows_kernel<real_t, 32> : eig_ql_rows_kernel<real_t, 64>;
        HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int r = 0; r < std::max(reps, 1) + 1; r++) {
            if (r == 1) HIP_CHECK(hipEventRecord(e0, dev.stream));
            if (method == 0 || method == 2) {        // (2: only the tridiagonalisation inside the timed runs -- where the time goes)
                hipLaunchKernelGGL(eig_tridiag_kernel<real_t>, dim3(1), dim3(EIG_TRIDIAG_THREADS), 0, dev.stream, dA.ptr, n, W.ptr, D.ptr, E.ptr, Tau.ptr);
                if (method == 0 || r == 0)
                    hipLaunchKernelGGL(kern, dim3((n + rows - 1) / rows), dim3(64), smem, dev.stream, n, W.ptr, D.ptr, E.ptr, Tau.ptr, dQ.ptr, dQt.ptr,
                                       (size_t)n, dL.ptr, info.ptr);
            } else
                hipLaunchKernelGGL(jacobi_eig_kernel<real_t>, dim3(1), dim3(1024), 0, dev.stream, dA.ptr, n, W.ptr, V.ptr, dQ.ptr, dQt.ptr, (size_t)n,
                                   dL.ptr, 30, sizeof(real_t) == 4 ? 1e-9 : 1e-13);
            HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(hipEventRecord(e1, dev.stream));
        HIP_CHECK(hipEventSynchronize(e1));
        float t = 0;
        HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
        if (ms) *ms = (double)t / std::max(reps, 1);
        int h_info = 0;
        HIP_CHECK(hipMemcpyAsync(&h_info, info.ptr, sizeof(int), hipMemcpyDeviceToHost, dev.stream));
        dQ.download(Q, nn, dev.stream); dL.download(lam, (size_t)n, dev.stream);
        HIP_CHECK(hipStreamSynchronize(dev.stream));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (h_info != 0) { g_last_error = "cmfrec_hip_sym_eig: a QL iteration did not converge"; return 4; }
        return 0;
    });
}

#ifdef CMF_CG_TICKS
// (instrumented builds only; not part of include/cmfrec_hip.h) copies the 32 tick sums to `out` and clears them
extern "C" int cmfrec_hip_debug_cg_ticks(unsigned long long *out)
{
    unsigned long long *buf = cg_ticks_buffer();
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out, buf, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemset(buf, 0, 32 * sizeof(unsigned long long)));
    return 0;
}
#endif

extern "C" int cmfrec_hip_selftest_lanes(void)
{
    int bad = -1;
    int rc = guarded([&]() {
        DeviceInfo dev;
        init_device(dev, -1);
        DevBuf<real_t> d;
        d.alloc(17 * 64);
        hipLaunchKernelGGL(lanes_selftest_kernel, dim3(1), dim3(64), 0, dev.stream, d.ptr);
        HIP_CHECK(hipGetLastError());
        std::vector<real_t> h(17 * 64);
        d.download(h.data(), h.size(), dev.stream);
        HIP_CHECK(hipStreamSynchronize(dev.stream));
        auto X = [](int l) { return (double)(l * 3 + 1); };
        auto Y = [](int l) { return (double)(1000 + l); };
        bad = 0;
        for (int l = 0; l < 64; l++) {
            double e[17];
            e[0] = X(l ^ 1); e[1] = X(l ^ 2); e[2] = X(l ^ 4); e[3] = X(l ^ 8);
            e[4] = (l & 4) ? Y(l ^ 4) : X(l ^ 4);
            e[5] = (l & 8) ? Y(l ^ 8) : X(l ^ 8);
            e[6] = (l < 32) ? X(l) + X(l + 32) : Y(l - 32) + Y(l);
            e[7] = (l & 16) ? Y(l - 16) + Y(l) : X(l) + X(l + 16);
            e[8] = X((l & ~7) | 0); e[9] = X((l & ~7) | 3); e[10] = X((l & ~7) | 5); e[11] = X((l & ~7) | 7);
            double s = 0; for (int j = 0; j < 64; j++) s += X(j);
            e[12] = s;
            { int b = l & 7; double t = 0; for (int j = 0; j < 8; j++) t += (double)(((l & ~7) | j) + 100 * b); e[13] = t; }
            { int b = l >> 3; double t = 0; for (int j = 0; j < 8; j++) t += (double)(((l & 7) | (j << 3)) + 100 * b); e[14] = t; }
            e[15] = X(l ^ 7);
            e[16] = (double)(((l & ~7) | 6) * 5 + 2);
            for (int c = 0; c < 17; c++)
                if ((double)h[c * 64 + l] != e[c]) {
                    if (bad < 8) fprintf(stderr, "lanes selftest: case %d lane %d got %g expected %g\n", c, l, (double)h[c * 64 + l], e[c]);
                    bad++;
                }
        }
        return 0;
    });
    return rc ? -rc : bad;
}
