// device.hpp -- host-side device plumbing for the ALS path: HBM buffers, CSR shards with the
// nnz-binned row schedule, and the launchers of the row-update kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/cmfrec_hip.h"
#include "cg_kernels.hpp"
#include "chol_kernels.hpp"
#include "dense_kernels.hpp"
#include "gram_cg_kernels.hpp"
#include "topn_kernels.hpp"

namespace cmfhip {

extern thread_local std::string g_last_error;

struct HipError {
    int code;       // C-ABI return code: 1 OOM, 4 HIP failure
};

inline void hip_check(hipError_t e, const char *what, const char *file, int line)
{
    if (e == hipSuccess) return;
    char buf[512];
    snprintf(buf, sizeof buf, "cmfrec_hip: %s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_last_error = buf;
    fprintf(stderr, "%s\n", buf);
    throw HipError{e == hipErrorOutOfMemory ? 1 : 4};
}
#define HIP_CHECK(x) ::cmfhip::hip_check((x), #x, __FILE__, __LINE__)

// Run-time switches: the CMFREC_HIP_* environment variables are read ONCE PER SESSION -- when a session is
// created (every level-1 / level-2 entry point creates its own) or when cmfrec_hip_reload_switches() is called -- into this
// process-wide struct, instead of a getenv per launch path.  Each is an A/B switch or an on-device cross-check (another kernel
// for the same row systems, exercised by tests/test_gpu_switches.py and the tests that name it), a test hook, or a deployment
// setting (devices, exchange, size limit); none selects another model.
struct Switches {
    bool poison_lds = false;        // CMFREC_HIP_POISON_LDS: NaN patterns in LDS and fresh buffers in front of the launches (test hook)
    int vh_min = 0;                 // CMFREC_HIP_VH_MIN: where the split rows begin (0: by precision and path)
    int vh = 0;                     // CMFREC_HIP_VH: split rows 1 = stream (launch pair per CG pass), 2 = gram (one gather, CG on the row's Gramian); 0: by shape
    bool gram_slice = false;        // CMFREC_HIP_GRAM_KERNEL=slice: LDS-staged workgroup kernel for the slice partials
    int bins_par = 2;               // CMFREC_HIP_BINS_PAR: streams the nnz bins of a half-step are spread over (1: in line)
    bool nt_split = true;           // CMFREC_HIP_NT_SPLIT=0: a double-precision length bin as one launch instead of two by tile size
    bool cg_generic = false;        // CMFREC_HIP_CG_KERNEL=generic: lane <-> unknown CG kernel everywhere
    int chol = 0;                   // CMFREC_HIP_CHOL: 1 = rows (workgroup-per-row kernel only), 2 = noslices
    int parts_coop = 1;             // CMFREC_HIP_PARTS_COOP: the rank-k update of those rows with ONE gather shared by the row's two wavefronts through LDS (chol_parts_coop_kernels.hpp; 3: three steps in flight instead of four); 0 = each wavefront gathers for itself (round 5)
    int chol_wg = 2;                // CMFREC_HIP_CHOL_WG: eight-block rows in double precision factorised by a workgroup of 2 (default) / 4 wavefronts per row (chol_wg_kernels.hpp); 0 = one wavefront per row (rounds 2-5)
    int gramk = -1;                 // CMFREC_HIP_GRAMK: 0 / 1 force the producer / consumer pair off / on (-1: by width)
    int gramk_batch = 0;            // CMFREC_HIP_GRAMK_BATCH: work items per batch (test hook: several batches on a small problem)
    int lowrank = -1;               // CMFREC_HIP_LOWRANK: 0 / 1 force the low-rank row kernel off / on (-1: by shape)
    bool eig_jacobi = false;        // CMFREC_HIP_EIG=jacobi: the one-workgroup Jacobi kernel instead of tridiagonalisation + QL (cross-check)
    int debug_skip = 0;             // CMFREC_HIP_CG_SKIP / _CHOL_SKIP / _WAVE_SKIP (timing builds only: -DCMF_CG_DEBUG / -DCMF_CHOL_DEBUG)
    bool debug_ticks = false;       // CMFREC_HIP_GRAM_TICKS / _CHOL_TICKS (timing builds only)
    void reload()
    {
        auto str = [](const char *n) -> const char * { const char *v = getenv(n); return (v != nullptr && v[0] != 0) ? v : nullptr; };
        auto num = [&](const char *n, int dflt) -> int { const char *v = str(n); return v ? atoi(v) : dflt; };
        poison_lds = str("CMFREC_HIP_POISON_LDS") != nullptr;
        vh_min = num("CMFREC_HIP_VH_MIN", 0);
        const char *v = str("CMFREC_HIP_VH");
        vh = !v ? 0 : strcmp(v, "stream") == 0 ? 1 : strcmp(v, "gram") == 0 ? 2 : 1;       // (any other value streams, as before)
        v = str("CMFREC_HIP_GRAM_KERNEL"); gram_slice = v && strcmp(v, "slice") == 0;
        bins_par = num("CMFREC_HIP_BINS_PAR", 2);
        nt_split = num("CMFREC_HIP_NT_SPLIT", 1) != 0;
        v = str("CMFREC_HIP_CG_KERNEL"); cg_generic = v && strcmp(v, "generic") == 0;
        v = str("CMFREC_HIP_CHOL"); chol = !v ? 0 : strcmp(v, "rows") == 0 ? 1 : strcmp(v, "noslices") == 0 ? 2 : 0;
        parts_coop = num("CMFREC_HIP_PARTS_COOP", 1);
        chol_wg = num("CMFREC_HIP_CHOL_WG", 2);
        if (chol_wg == 1) chol_wg = 2;
        gramk = num("CMFREC_HIP_GRAMK", -1);
        gramk_batch = num("CMFREC_HIP_GRAMK_BATCH", 0);
        lowrank = num("CMFREC_HIP_LOWRANK", -1);
        v = str("CMFREC_HIP_EIG"); eig_jacobi = v && strcmp(v, "jacobi") == 0;
        debug_skip = num("CMFREC_HIP_CG_SKIP", num("CMFREC_HIP_CHOL_SKIP", num("CMFREC_HIP_WAVE_SKIP", 0)));
        debug_ticks = str("CMFREC_HIP_GRAM_TICKS") != nullptr || str("CMFREC_HIP_CHOL_TICKS") != nullptr;
    }
};
inline Switches &switches_mut() { static Switches sw; static bool first = (sw.reload(), true); (void)first; return sw; }
inline const Switches &switches() { return switches_mut(); }

template <typename T>
struct DevBuf {
    T *ptr = nullptr;
    size_t n = 0;
    bool owned = true;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (ptr && owned) (void)hipFree(ptr);
        ptr = nullptr;
        n = 0;
    }
    void alloc(size_t count)
    {
        release();
        n = count;
        owned = true;
        if (count) HIP_CHECK(hipMalloc((void **)&ptr, count * sizeof(T)));
        if (count && switches().poison_lds) {          // test hook (poison_lds below): device buffers too
            HIP_CHECK(hipMemset(ptr, 0xFF, count * sizeof(T)));
            HIP_CHECK(hipDeviceSynchronize());
        }
    }
    void alloc_at_least(size_t count)
    {
        if (n < count) alloc(count);
    }
    void upload(const T *host, size_t count, hipStream_t st)
    {
        if (count > n) alloc(count);
        // (hipMemcpyDefault: the source may also be device memory -- side information generated shard-wise in HBM, bench.py)
        if (count) HIP_CHECK(hipMemcpyAsync(ptr, host, count * sizeof(T), hipMemcpyDefault, st));
    }
    void download(T *host, size_t count, hipStream_t st) const
    {
        if (count) HIP_CHECK(hipMemcpyAsync(host, ptr, count * sizeof(T), hipMemcpyDeviceToHost, st));
    }
};

// Rows are scheduled longest-first inside nnz bins; every bin is a persistent launch whose teams
// stride over the sorted list, which balances the heavy-tailed row lengths the reference handles
// with `omp schedule(dynamic)` (common.c:3259,3349).  A row of a bin with W waves per row keeps its
// gathered tiles in registers when it has <= 64*W non-zeros.
constexpr int MAX_DEVICES = 16;   // per-device caches of launch attributes
constexpr int NBINS = 6;
constexpr int BIN_VHEAVY = 0;   // > 1024 nnz : every CG pass split over many workgroups (vh_* kernels); measured on C2: 2048 -> 5.05, 1024 -> 4.93, 512 -> 5.11 ms
constexpr int BIN_HEAVY = 1;    // 257..1024  : 8 waves / row (double: register-resident up to 512 nnz, else re-streamed; single: two tiles per wave, resident)
constexpr int BIN_MED4 = 2;     // 129..256   : 4 waves / row
constexpr int BIN_MED2 = 3;     // 65..128    : 2 waves / row
constexpr int BIN_LIGHT = 4;    // 33..64     : 1 wave / row, 4 rows / workgroup
constexpr int BIN_TINY = 5;     // 1..32      : 1 wave / row, half-size tiles, double-buffered gather
constexpr int BIN_MIN_NNZ[NBINS] = {1025, 257, 129, 65, 33, 1};
// The boundary of the split rows is a property of the shard (SparseShard::vh_min).  Single precision: 1025 -- the 8-wave kernel
// keeps two tiles per wave and nothing else, so the boundary cannot move up.  Double precision: an 8-wave team keeps 512
// entries in registers and re-streams the second tile of longer rows on every pass (1.7x the algorithmic bytes measured on
// that bin in round 2); since the split rows' Gramian kernel does the last column block of k = 50 on the vector ALU (round 3),
// cutting at 513 is the faster split WHERE THE GRAMIAN PATH TAKES THE SPLIT ROWS: C2 4.36 -> 4.16 ms (513), 4.24 (769), 4.34
// (385) -- profiles/r03_c; where they are streamed (C1: small opposing matrices, many references) 1025 stays better (2.48
// against 2.65 ms).  CMFREC_HIP_VH_MIN overrides.
inline int vh_min_env()
{
    const int v = switches().vh_min;
    return v > 0 ? std::min(sizeof(real_t) == 4 ? BIN_MIN_NNZ[BIN_VHEAVY] : (1 << 30), std::max(258, v)) : 0;
}

struct SparseShard {
    int nrows = 0;
    size_t nnz = 0;
    int vh_min = BIN_MIN_NNZ[BIN_VHEAVY];       // rows of at least this many entries are split rows (build_bins)
    size_t opp_row_bytes_hint = 50 * sizeof(real_t);   // bytes of a gathered row (k x sizeof), for the choice of vh_min; set before the build
    int bin_of(long long l) const
    {
        if (l >= vh_min) return BIN_VHEAVY;
        for (int b = 1; b < NBINS; b++)
            if (l >= BIN_MIN_NNZ[b]) return b;
        return -1;   // empty row
    }
    static bool gram_pays(double vh_nnz, int n_other_, size_t opp_bytes_per_row)
    {
        // Round 6: always.  Rounds 3-5 streamed the split rows of double precision once per CG pass where they share their opposing
        // rows often enough to live in cache (more than 40 references per opposing row, or 3.5 with a matrix below 100 MB: C1).  Measured
        // again on C1 with the slice kernel as it stands (remainder columns on the vector ALU, the boundary at 513): 2.24 -> 1.86 ms per
        // iteration, its item step 1.31 -> 0.93 (profiles/r06/r06_n_c1_split_rows_sweep.txt); the streamed path stays behind
        // CMFREC_HIP_VH=stream, for k > 64 and for block systems.
        (void)vh_nnz; (void)opp_bytes_per_row;
        return sizeof(real_t) == 4 || n_other_ > 0;
    }
    DevBuf<size_t> p;
    DevBuf<int> i;
    DevBuf<real_t> v;
    // observation weights of the explicit model (empty: none): one per entry in the order of `v`, and per row the multiplier
    // of lambda under scale_lam -- the sum of the row's weights, 1 for a row without entries (wsumA / wsumB of the driver,
    // collective.c:7978-8008; summed in double, entry by entry, as there)
    DevBuf<real_t> w, wsum;
    // NA_as_zero_X with observation weights: the multipliers of that model -- sum of the row's weights + number of its ABSENT entries
    // (collective.c:7991-8022) -- in a buffer of their own, rebuilt by the session whenever X or the flag changes (round 6; they
    // used to overwrite wsum in place, which a second upload of X or switching the flag off left stale)
    DevBuf<real_t> wsum_naz;
    bool weighted() const { return w.ptr != nullptr && nnz > 0; }
    DevBuf<int> order;       // row ids sorted by nnz descending: [bin 0 | bin 1 | ... | bin 4 | empty rows]
    DevBuf<RowDesc> desc;    // same order: {row, nnz, CSR offset}
    int bin_rows[NBINS] = {0, 0, 0, 0, 0, 0};
    int bin_first[NBINS] = {0, 0, 0, 0, 0, 0};
    size_t bin_nnz[NBINS] = {0, 0, 0, 0, 0, 0};
    int n_nonempty = 0, n_empty = 0;
    int max_nnz = 0;
    int n_long = 0;          // rows with more than LONG_ROW entries (they lead the processing order)
    int n_gt96 = 0;          // rows with more than 96 entries (the six-block low-rank build of double precision, session.hip)
    int n_gt16 = 0;          // rows with more than 16 entries: the rest of the tiny bin goes two rows per wavefront
    int n_gt512 = 0;         // rows with more than 512 entries (single precision: the 257..512 part of the heavy bin runs on 4-wave teams)
    int n_gt_low[4] = {0, 0, 0, 0};   // rows with more than CG_NT_LOW * 8 W entries, W = 1, 2, 4, 8 (48 / 96 / 192 / 384): where a length bin's launch by tile size begins
    bool is_part = false;    // one of several parts of a block that are updated one after the other (session.hip)
    int n_other = 0;         // rows of the opposing matrix the entries refer to
    // Split rows: read their gathered rows once and run the CG on the row's own Gramian (gram_cg_kernels.hpp, one wavefront
    // per slice: c4shard's item rows 2.31 -> 1.57 ms, BASELINE config 4 on one GPU 40.6 -> 21.8 ms, C2's items 0.89 ->
    // 0.73 ms), or stream them once per CG pass (sorted entries, one slice of the opposing matrix per XCD).  Single
    // precision always takes the Gramian (the matrix cores outrun any gather).  In double precision v_mfma_f64_16x16x4 is no
    // faster than the VALU, and streaming wins once the split rows of a launch share their opposing rows often enough to live
    // in cache: C2's items (15 references per opposing row, a 144 MB opposing matrix) still favour the Gramian; C1
    // (MovieLens-10M-shaped: 4 / 28 MB opposing matrices, 100+ references) did not in round 3 -- 1.29 against 1.06 ms for its A-step --
    // and does since (gram_pays above: round 6).
    // `opp_bytes_per_row` = k x sizeof(real_t) of the launch.  CMFREC_HIP_VH=gram / stream force one.
    bool prefer_gram(size_t opp_bytes_per_row) const { return gram_pays((double)bin_nnz[0], n_other, opp_bytes_per_row); }
    // few split rows (less than about one round of workgroups per CG pass): their launch sequence is a chain of
    // latencies and runs on the second stream beside the other bins
    bool vh_runs_aside(int num_cus) const
    {
        return bin_rows[0] > 0 && n_chunks <= 4 * num_cus;
    }
    static constexpr int LONG_ROW = 1024;
    // rows (they lead the processing order) with more than `maxlen` entries, for the nnz-bin boundaries 32 .. 1024
    int rows_longer_than(int maxlen, int cap) const
    {
        int n;
        if (maxlen >= max_nnz) n = 0;
        else if (maxlen >= LONG_ROW) n = n_long;
        else if (maxlen >= 256) n = bin_first[BIN_MED4];
        else if (maxlen >= 128) n = bin_first[BIN_MED2];
        else if (maxlen >= 96) n = n_gt96;
        else if (maxlen >= 64) n = bin_first[BIN_LIGHT];
        else if (maxlen >= 32) n = bin_first[BIN_TINY];
        else n = n_nonempty;
        return std::min(n, cap);
    }
    // split-row work list and CG state of the very heavy rows (cg_kernels.hpp, VhState)
    int n_chunks = 0;
    DevBuf<int> vh_chunk_row, vh_chunk_start, vh_chunk_cnt, vh_chunk_off, vh_launch, vh_done;
    DevBuf<VhWork> vh_work;
    int n_launch = 0;
    DevBuf<real_t> vh_r, vh_p, vh_r_old, vh_part;
    // Gramian path of the very heavy rows (gram_cg_kernels.hpp): slices of <= GRAM_SLICE non-zeros
    static constexpr int GRAM_SLICE = 2048;
    int slice_len = GRAM_SLICE;          // entries per slice of this shard (build_bins)
    int n_slices = 0;
    DevBuf<int> sl_vrow, sl_first, sl_count, row_sl_off;
    std::vector<int> h_row_sl_off;       // host copy of row_sl_off (batching of the two-kernel Cholesky mode)
    DevBuf<real_t> gram_part;
    mutable DevBuf<real_t> chol_part;    // partial normal matrices of the slices (wave-per-row Cholesky kernel), sized on first use

    void upload(int nrows_, const size_t *hp, const int *hi, const real_t *hv, hipStream_t st, const real_t *hw = nullptr)
    {
        nrows = nrows_;
        nnz = hp[nrows] - hp[0];
        std::vector<size_t> p0(nrows + 1);
        for (int r = 0; r <= nrows; r++) p0[r] = hp[r] - hp[0];
        p.upload(p0.data(), nrows + 1, st);
        i.upload(hi + hp[0], nnz, st);
        v.upload(hv + hp[0], nnz, st);
        w.release(); wsum.release();
        if (hw != nullptr && nnz > 0) {
            w.upload(hw + hp[0], nnz, st);
            std::vector<real_t> ws((size_t)nrows);
            for (int r = 0; r < nrows; r++) {
                double acc_w = 0;
                for (size_t e = hp[r]; e < hp[r + 1]; e++) acc_w += hw[e];
                ws[r] = (hp[r + 1] > hp[r]) ? (real_t)acc_w : (real_t)1;
            }
            wsum.upload(ws.data(), (size_t)nrows, st);
            HIP_CHECK(hipStreamSynchronize(st));
        }
        std::vector<int> ord(nrows);
        std::iota(ord.begin(), ord.end(), 0);
        auto len = [&](int r) { return (long long)(p0[r + 1] - p0[r]); };
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return len(a) > len(b); });
        std::vector<RowDesc> dsc(nrows);
        std::vector<unsigned> lens(nrows);
        for (int q = 0; q < nrows; q++) {
            const int r = ord[q];
            dsc[q].row = r; dsc[q].nnz = (int)len(r); dsc[q].st = (unsigned long long)p0[r];
            lens[q] = (unsigned)len(r);
        }
        order.upload(ord.data(), nrows, st);
        desc.upload(dsc.data(), nrows, st);
        build_bins(lens.data(), st);
        HIP_CHECK(hipStreamSynchronize(st));   // host staging vectors go out of scope
    }

    // split-row work list: chunks (row-contiguous, the order their partials are added in) and the
    // workgroup -> chunk map of the pass kernel (empty: identity)
    void set_vh_chunks(const std::vector<int> &c_row, const std::vector<int> &c_start, const std::vector<int> &c_cnt,
                       const std::vector<int> &c_off, std::vector<int> launch, hipStream_t st)
    {
        n_chunks = (int)c_row.size();
        if (launch.empty()) { launch.resize(n_chunks); std::iota(launch.begin(), launch.end(), 0); }
        n_launch = (int)launch.size();
        vh_chunk_row.upload(c_row.data(), c_row.size(), st);
        vh_chunk_start.upload(c_start.data(), c_start.size(), st);
        vh_chunk_cnt.upload(c_cnt.data(), c_cnt.size(), st);
        vh_chunk_off.upload(c_off.data(), c_off.size(), st);
        vh_launch.upload(launch.data(), launch.size(), st);
        vh_part.alloc((size_t)n_chunks * 64);
        // resolved work items of the pass kernel (the split rows lead the processing order: desc[vi] is row vi of them)
        const int nvh = (int)c_off.size() - 1;
        std::vector<RowDesc> hd((size_t)std::max(nvh, 1));
        HIP_CHECK(hipMemcpyAsync(hd.data(), desc.ptr, (size_t)nvh * sizeof(RowDesc), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        std::vector<VhWork> work(launch.size());
        for (size_t b = 0; b < launch.size(); b++) {
            VhWork &w = work[b];
            const int c = launch[b];
            w.pad_ = 0;
            if (c < 0) { w.vi = -1; w.row = 0; w.cnt = 0; w.chunk = 0; w.st = 0; continue; }
            w.vi = c_row[c]; w.row = hd[w.vi].row; w.cnt = c_cnt[c]; w.chunk = c;
            w.st = hd[w.vi].st + (unsigned long long)c_start[c];
        }
        vh_work.upload(work.data(), work.size(), st);
        HIP_CHECK(hipStreamSynchronize(st));
    }

    // nnz bins, split-row work list and CG state from the row lengths in processing order (descending)
    void build_bins(const unsigned *lens_sorted, hipStream_t st)
    {
        for (int b = 0; b < NBINS; b++) { bin_rows[b] = 0; bin_nnz[b] = 0; }
        n_empty = 0; n_long = 0; n_gt16 = 0; n_gt96 = 0; n_gt512 = 0; n_slices = 0; h_row_sl_off.assign(1, 0);
        n_gt_low[0] = n_gt_low[1] = n_gt_low[2] = n_gt_low[3] = 0;
        std::vector<int> c_row, c_first, c_cnt, c_off(1, 0);
        std::vector<int> s_row, s_first, s_count, s_off(1, 0);
        // few split rows (C2's users: 50 rows, 65 k entries): short slices, so that their Gramian kernels -- which run in line
        // with the other bins, see launch_cg_S -- have a few hundred wavefronts to spread over instead of a few dozen
        // where the split rows begin (see vh_min_env above): double precision cuts at 513 when the rows above that go through
        // their Gramian, at 1025 otherwise
        vh_min = BIN_MIN_NNZ[BIN_VHEAVY];
        if (vh_min_env() > 0) vh_min = vh_min_env();
        else if (sizeof(real_t) == 8) {
            double nnz513 = 0;
            for (int q = 0; q < nrows && lens_sorted[q] >= 513u; q++) nnz513 += (double)lens_sorted[q];
            const bool gram = (switches().vh != 0) ? switches().vh == 2 : gram_pays(nnz513, n_other, opp_row_bytes_hint);
            // (Round 5: 385 / 321 / 258 for a shard whose split rows hold 40 % of the entries -- C2's items -- move the eight-wave bin's rows
            //  to the slice kernel, 87 against 115-126 ps per entry: in line the two bins gain 0.035 ms together, side by side on two streams
            //  the half-step does not change, 3.285 / 3.298 against 3.293 / 3.286 ms; profiles/r05/r05_l_split_boundary.txt, r05_m_*)
            if (gram && opp_row_bytes_hint <= 16 * 4 * sizeof(real_t)) vh_min = 513;
        }
        long long vh_total = 0;
        for (int q = 0; q < nrows && (long long)lens_sorted[q] >= vh_min; q++) vh_total += (long long)lens_sorted[q];
        slice_len = (vh_total < (long long)GRAM_SLICE * 1024) ? 256 : GRAM_SLICE;
        for (int q = 0; q < nrows; q++) {
            const long long l = (long long)lens_sorted[q];
            if (l > LONG_ROW) n_long++;
            if (l > 16) n_gt16++;
            if (l > 96) n_gt96++;
            if (l > 512) n_gt512++;
            for (int w = 0; w < 4; w++) if (l > (long long)CG_NT_LOW * 8 * (1 << w)) n_gt_low[w]++;
            const int b = bin_of(l);
            if (b < 0) { n_empty++; continue; }
            if (b == BIN_VHEAVY) {
                const int CHN = TILE * VH_CHUNK_TILES;
                for (long long f = 0; f < l; f += CHN) {
                    c_row.push_back(bin_rows[b]); c_first.push_back((int)f); c_cnt.push_back((int)std::min<long long>(CHN, l - f));
                }
                c_off.push_back((int)c_row.size());
                // equal slices, multiples of 16 non-zeros (one staging round)
                const int nsl = (int)((l + slice_len - 1) / slice_len);
                const int per = (int)((((l + nsl - 1) / nsl) + 15) / 16 * 16);
                for (long long f = 0; f < l; f += per) {
                    s_row.push_back(bin_rows[b]); s_first.push_back((int)f); s_count.push_back((int)std::min<long long>(per, l - f));
                }
                s_off.push_back((int)s_row.size());
            }
            bin_rows[b]++; bin_nnz[b] += (size_t)l;
        }
        int acc = 0;
        for (int b = 0; b < NBINS; b++) { bin_first[b] = acc; acc += bin_rows[b]; }
        n_nonempty = acc;
        max_nnz = nrows ? (int)lens_sorted[0] : 0;
        if (bin_rows[BIN_VHEAVY]) {
            const int nvh = bin_rows[BIN_VHEAVY];
            set_vh_chunks(c_row, c_first, c_cnt, c_off, std::vector<int>(), st);      // row order; finalize_vheavy() refines it
            vh_done.alloc(nvh); vh_r_old.alloc(nvh);
            vh_r.alloc((size_t)nvh * 64); vh_p.alloc((size_t)nvh * 64);
            n_slices = (int)s_row.size();
            sl_vrow.upload(s_row.data(), s_row.size(), st);
            sl_first.upload(s_first.data(), s_first.size(), st);
            sl_count.upload(s_count.data(), s_count.size(), st);
            row_sl_off.upload(s_off.data(), s_off.size(), st);
            h_row_sl_off = s_off;
            gram_part.alloc((size_t)n_slices * GRAM_PART);
        }
        HIP_CHECK(hipStreamSynchronize(st));
    }
};

struct DeviceInfo {
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    DevBuf<real_t> cg_gfull, cg_rconst;   // block systems on the tiled CG kernels: weighted Gramian (k x k) and per-row constants
    DevBuf<real_t> tile_init;       // initial matrices of a Cholesky launch in tile-linear layout (chol_wave_kernels.hpp, tile_pack_kernel)
    DevBuf<int> row_counter;        // work counters of the dynamically scheduled row kernels (zeroed before each launch)
    DevBuf<real_t> potrs_inv, potrs_tmp;   // launch_potrs_rows: L^-1 and M^-1 of the shared matrix; the product before it replaces the rows
    DevBuf<real_t> potrs_ref;              // ... single precision: the matrix itself and the residuals of the refinement step
    DevBuf<real_t> gemm_ws;         // partial products of the split-K GEMMs (session.hip, launch_gemm)
    // second stream + events for two row kernels that may overlap (heavy-row teams beside light-row teams); created on
    // first use, owned here
    hipStream_t aux_stream = nullptr;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    // third stream for the eigen-decomposition of the low-rank path (one workgroup, overlaps the row kernels of both other streams)
    hipStream_t eig_stream_ = nullptr;
    hipEvent_t eig_fork = nullptr;
    hipStream_t eig_stream()
    {
        if (!eig_stream_) HIP_CHECK(hipStreamCreateWithFlags(&eig_stream_, hipStreamNonBlocking));
        return eig_stream_;
    }
    // solver option of the update in progress (set by the session around a half-step): closed-form rows are solved by
    // the non-negative coordinate descent instead of the Cholesky factorisation
    mutable bool nonneg_now = false;
    mutable int max_cd_steps = 100;
    mutable real_t l1_now = 0;      // L1 penalty of the update in progress (already divided by w_user / w_item for C / D)
    mutable real_t l1_last_now = 0; // ... of the last unknown (the bias' own penalty under l1_lam_unique)
    mutable real_t l1_scale = 1;    // ... times this for the launches whose lambda is scaled by a row count on the host
    DeviceInfo() = default;
    DeviceInfo(const DeviceInfo &) = delete;
    DeviceInfo &operator=(const DeviceInfo &) = delete;
    ~DeviceInfo()
    {
        if (eig_fork) (void)hipEventDestroy(eig_fork);
        if (eig_stream_) (void)hipStreamDestroy(eig_stream_);
        if (fork_ev) (void)hipEventDestroy(fork_ev);
        if (join_ev) (void)hipEventDestroy(join_ev);
        for (int i = 0; i < MAX_BIN_STREAMS; i++) {
            if (bin_join[i]) (void)hipEventDestroy(bin_join[i]);
            if (i > 0 && bin_streams[i]) (void)hipStreamDestroy(bin_streams[i]);
        }
        if (aux_stream) (void)hipStreamDestroy(aux_stream);
        if (stream) (void)hipStreamDestroy(stream);
    }
    void ensure_aux()
    {
        if (aux_stream) return;
        HIP_CHECK(hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&join_ev, hipEventDisableTiming));
    }
    // further streams for the nnz bins of a half-step running side by side (launch_cg_S): bin_streams[0] is aux_stream
    static constexpr int MAX_BIN_STREAMS = 5;
    hipStream_t bin_streams[MAX_BIN_STREAMS] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t bin_join[MAX_BIN_STREAMS] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    void ensure_bin_streams(int n)
    {
        ensure_aux();
        bin_streams[0] = aux_stream;
        if (!bin_join[0]) HIP_CHECK(hipEventCreateWithFlags(&bin_join[0], hipEventDisableTiming));
        for (int i = 1; i < n && i < MAX_BIN_STREAMS; i++) {
            if (!bin_streams[i]) HIP_CHECK(hipStreamCreateWithFlags(&bin_streams[i], hipStreamNonBlocking));
            if (!bin_join[i]) HIP_CHECK(hipEventCreateWithFlags(&bin_join[i], hipEventDisableTiming));
        }
    }
};

// HIP-event pairs around the launches of one nnz bin (0 heavy, 1 medium, 2 light) on the stream
// the kernel runs on; read back by cmfrec_hip_session_kernel_time.
struct EventPair { hipEvent_t a, b; };
struct BinTimers {
    std::vector<EventPair> ev[NBINS]; // per nnz bin (bin 0 = whole split-row sequence of the very heavy rows)
    void clear()
    {
        for (auto &v : ev) {
            for (auto &p : v) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
            v.clear();
        }
    }
};

// ------------------------------------------------------------------------------------------
// dense contractions
template <bool TRANSA>
inline void launch_gemm(const DeviceInfo &dev, int M, int N, int K, real_t alpha, const real_t *A, size_t lda,
                        const real_t *B, size_t ldb, real_t *C, size_t ldc)
{
    if (M <= 0 || N <= 0) return;
    // C[M, N] = alpha * op(A) B, all row-major: the dense contractions of the side-information path (w U C, U^T A, I D, the
    // rotations of the low-rank path).  Round 3: the library's own MFMA kernel (dense_kernels.hpp, gemm_mfma_kernel) -- the
    // matrix cores under an LDS-staged 128 x 128 tile, split over K with an ordered reduction where M x N alone would leave
    // CUs idle (U^T A: 512 x 256 outputs over 1.5 M rows).  Round 6: the only GEMM of the library -- the rocBLAS alternative
    // (CMFREC_HIP_GEMM_OWN=0, rounds 3-5) went with the link against it; cmfrec_hip_gemm_probe still times rocBLAS beside it
    // when that library can be loaded at run time (a measurement tool, not a path).
    {
        DeviceInfo &d = const_cast<DeviceInfo &>(dev);
        const int bm = (M + GEMM_BM - 1) / GEMM_BM, bn = (N + GEMM_BN - 1) / GEMM_BN;
        int nsplit = 1;
        if ((long long)bm * bn < 2LL * dev.num_cus && K >= 4096)
            // (chunks of at least 256: config 3's I^T B -- 64 x 128 outputs over 10,677 rows -- ran as 11 workgroups of 1024 rows each,
            //  204 us per call on 11 of 256 CUs; round 5: 42 workgroups)
            nsplit = (int)std::min<long long>((K + 255) / 256, std::max<long long>(1, (4LL * dev.num_cus) / ((long long)bm * bn)));
        int kchunk = (std::max(K, 1) + nsplit - 1) / nsplit;
        kchunk = (kchunk + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
        nsplit = (std::max(K, 1) + kchunk - 1) / kchunk;
        const dim3 grid(bn, bm, nsplit);
        if (nsplit == 1) {
            hipLaunchKernelGGL((gemm_mfma_kernel<real_t, TRANSA>), grid, dim3(256), 0, dev.stream, M, N, K, kchunk, alpha, A, lda, B, ldb, C, ldc,
                               (size_t)0);
        } else {
            const size_t ss = (size_t)M * N;
            d.gemm_ws.alloc_at_least(ss * nsplit);
            hipLaunchKernelGGL((gemm_mfma_kernel<real_t, TRANSA>), grid, dim3(256), 0, dev.stream, M, N, K, kchunk, alpha, A, lda, B, ldb,
                               d.gemm_ws.ptr, (size_t)N, ss);
            hipLaunchKernelGGL(gemm_splitk_reduce_kernel<real_t>, dim3((unsigned)((ss + 255) / 256)), dim3(256), 0, dev.stream, d.gemm_ws.ptr, ss,
                               nsplit, M, N, C, ldc);
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
}

// ------------------------------------------------------------------------------------------
// Gramian  out[k,k] = scale * B[:, :k]^T B[:, :k] + add_diag * I
struct GramWorkspace {
    DevBuf<real_t> partial;
};

inline void launch_gram(const DeviceInfo &dev, GramWorkspace &ws, const real_t *B, size_t ldb, int n, int k,
                        real_t *out, real_t scale, real_t add_diag)
{
    if (k <= 64) {
        // two workgroups per CU: 27.9 + 7.1 us for the two stages on C2's 359 k x 50 / 160 k x 50 matrices, against 35.8 + 5.1 (one),
        // 33.0 + 12.3 (four), 36.5 + 17.6 (eight) -- profiles/r03/r03_bo_gram_blocks.txt
        int nblocks = std::max(1, std::min(dev.num_cus * 2, (n + 127) / 128));
        int rpb = (n + nblocks - 1) / nblocks;
        nblocks = (n + rpb - 1) / rpb;
        if (nblocks < 1) nblocks = 1;
        size_t need = (size_t)nblocks * k * k;
        if (ws.partial.n < need) ws.partial.alloc(need);
        // double precision, 48 < k <= 52: the live columns of the last block on the vector ALU (dense_kernels.hpp)
        const int rem = (sizeof(real_t) == 8 && k > 48 && k <= 52) ? k - 48 : 0;
        switch (rem) {
#define CMF_GRAM_REM(R) case R: hipLaunchKernelGGL((gram_mfma_partial_kernel<real_t, R>), dim3(nblocks), dim3(256), 0, dev.stream, B, ldb, n, k, rpb, ws.partial.ptr); break;
            CMF_GRAM_REM(1) CMF_GRAM_REM(2) CMF_GRAM_REM(3) CMF_GRAM_REM(4)
#undef CMF_GRAM_REM
            default: hipLaunchKernelGGL((gram_mfma_partial_kernel<real_t, 0>), dim3(nblocks), dim3(256), 0, dev.stream, B, ldb, n, k, rpb, ws.partial.ptr); break;
        }
        hipLaunchKernelGGL(gram_reduce_kernel<real_t>, dim3((k * k + GRAM_RED_ENT - 1) / GRAM_RED_ENT), dim3(256), 0, dev.stream,
                           ws.partial.ptr, nblocks, k * k, out, scale, add_diag, k);
    } else {
        // k > 64: out = scale * B^T B through the split-K GEMM (both operands the same matrix: 4 x the triangle's flops in
        // the rare k > 64 set-ups, no extra kernel), then the diagonal shift; both triangles are filled
        launch_gemm<true>(dev, k, k, n, scale, B, ldb, B, ldb, out, (size_t)k);
        if (add_diag != 0)
            hipLaunchKernelGGL(add_diag_kernel<real_t>, dim3((k + 255) / 256), dim3(256), 0, dev.stream, out, k, 0, k, add_diag);
    }
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// CG row updates
struct CgCall {
    real_t *A; size_t lda;
    const real_t *B; size_t ldb;
    int k;
    const real_t *bias_sub;
    const real_t *BtB;       // implicit only
    real_t lam, lam_last;
    bool scale_lam, scale_bias_const;
    int max_cg_steps;
    bool implicit;
    bool precond = false;
    // block system with dense side information (A points at the first unknown, koff of them precede the X block)
    int koff = 0, kc = 0;
    const real_t *CtC = nullptr, *UC = nullptr;
    real_t w_side = 0;
    int rows_with_u = 0, p_side = 0;
    bool scale_lam_sideinfo = false;
    // ... or sparse side information: the row's attributes (X2, CSR over the same rows) gather rows of C2[*, kc]
    const SparseShard *X2 = nullptr;
    const real_t *C2 = nullptr;
    // implicit features of the explicit model: Bi [*, ki], BiTBi = Bi^T Bi (unweighted), weight w_imp
    const real_t *Bi = nullptr, *BiTBi = nullptr;
    int ki = 0;
    real_t w_imp = 0;
    const real_t *gsum = nullptr;     // [rows, ki]: sum of the rows of Bi at each row's observed positions (segmented gather-sum), or null
    int skip_first = 0;               // generic kernel: the first positions of the processing order are solved elsewhere (session.hip, launch_cg_wide)
    // explicit model on the tiled kernels with a matrix every row shares and a per-row constant in the first residual (GRAMX builds):
    // NA_as_zero_X with observation weights (session.hip, update_factor_naz_weighted) -- Gx [k, k], rconst_x [rows, ldr_x], and the
    // entries' values / weights read from these arrays (CSR order of X) instead of the shard's own
    const real_t *Gx = nullptr, *rconst_x = nullptr;
    size_t ldr_x = 0;
    const real_t *values_override = nullptr, *weights_override = nullptr;
    bool gx_all_rows = false;         // ... with the preconditioner (generic kernel): the rows without entries are part of the launch
    const real_t *wsum_override = nullptr;   // lambda's per-row multipliers under scale_lam instead of the rows' own counts / sums
};

enum class CgVariant { Auto, Generic };
CgVariant cg_variant_from_env();

// layout of DeviceInfo::row_counter: [0, 64) the Cholesky kernels' counters, then CG_NCOUNTERS padded counters per nnz bin
constexpr int NCOUNTER_SETS = 2 * NBINS + 2;   // one set per nnz bin + spares for bins that run as two launches (NBINS, NBINS + 1; NBINS + 2 + bin)
constexpr size_t ROW_COUNTER_INTS = 64 + (size_t)NCOUNTER_SETS * CG_NCOUNTERS * CG_COUNTER_STRIDE;
inline size_t cg_counter_offset(int bin) { return 64 + (size_t)bin * CG_NCOUNTERS * CG_COUNTER_STRIDE; }

// Test hook (CMFREC_HIP_POISON_LDS=1, tests/test_gpu_operators.py): fill the LDS of every CU with NaN patterns in front of the
// launches of the CG row update, so that a kernel that reads LDS it has not written -- found once in gram_cg_kernel: 0 x stale
// LDS, NaN only when the previous tenant left a NaN pattern there, i.e. on some boxes in some runs -- fails every time.
__global__ void __launch_bounds__(256) poison_lds_kernel(int words)
{
    extern __shared__ unsigned int poison_words[];
    volatile unsigned int *w = poison_words;
    for (int e = threadIdx.x; e < words; e += 256) w[e] = 0xFFFFFFFFu;       // NaN as float and as either half of a double
}
inline void poison_lds(hipStream_t st, int num_cus)
{
    if (!switches().poison_lds) return;
    constexpr int BYTES = 80 * 1024;                                         // two workgroups cover the 160 KB of a CU
    static thread_local bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK(hipFuncSetAttribute((const void *)poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(poison_lds_kernel, dim3(num_cus * 4), dim3(256), BYTES, st, BYTES / 4);
    HIP_CHECK(hipGetLastError());
}

template <int S, bool IMPLICIT, int W, int RPB, bool GRAMX = false, int NRES_ = 0, int NTSEL = 0>
inline void launch_cg_bin(const DeviceInfo &dev, CgParams<real_t> P, int first, int count, BinTimers *tm, int bin, hipStream_t st, int counter_set = -1)
{
    if (count <= 0) return;
    EventPair ev{nullptr, nullptr};
    if (tm) {
        HIP_CHECK(hipEventCreate(&ev.a));
        HIP_CHECK(hipEventCreate(&ev.b));
        HIP_CHECK(hipEventRecord(ev.a, st));
    }
    poison_lds(st, dev.num_cus);
    P.order += first;
    P.desc += first;
    P.nrows = count;
    P.counter = dev.row_counter.ptr + cg_counter_offset(counter_set >= 0 ? counter_set : bin);          // zeroed by launch_cg_S
    constexpr int threads = 64 * W * RPB;
    size_t smem = (((IMPLICIT || GRAMX) ? (size_t)gram_elems<real_t>(S) : 0) + (size_t)RPB * 2 * W * 64) * sizeof(real_t);
    auto kern = cg_rows_kernel<real_t, S, IMPLICIT, W, RPB, GRAMX, NRES_, NTSEL>;
    // per device: the dynamic-LDS attribute and the occupancy belong to the device the kernel was loaded on
    static thread_local int bpc_dev[MAX_DEVICES] = {0};
    int &blocks_per_cu = bpc_dev[std::min(std::max(dev.device, 0), MAX_DEVICES - 1)];
    if (blocks_per_cu == 0) {
        if (smem > 48 * 1024)
            HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        int nb = 0;
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, smem));
        blocks_per_cu = std::max(1, nb);
    }
    int teams_needed = (count + RPB - 1) / RPB;
    int grid = std::min(teams_needed, dev.num_cus * blocks_per_cu);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), smem, st, P);
    HIP_CHECK(hipGetLastError());
    if (tm) {
        HIP_CHECK(hipEventRecord(ev.b, st));
        tm->ev[bin].push_back(ev);
    }
}

// The rows of at most 32 entries: one row per wavefront (cg_rows_tiny_kernel).  Two rows per wavefront (round 5, cg_rows_pair_kernel)
// issued 15 % fewer vector instructions per row but read its Gramian from LDS in every pass and a pair cost its longer row -- C2 3.375
// against 3.348 ms (profiles/r05/r05_e_*, r05_w_*); removed in round 6, the sources are in the history (git log -- cmfrec_amd/csrc/cg_pair_kernels.hpp).
template <int S, bool IMPLICIT, bool GRAMX = false>
inline void launch_cg_tiny(const DeviceInfo &dev, CgParams<real_t> P, int first, int count, BinTimers *tm, hipStream_t st, int n_gt16)
{
    if (count <= 0) return;
    EventPair ev{nullptr, nullptr};
    if (tm) {
        HIP_CHECK(hipEventCreate(&ev.a));
        HIP_CHECK(hipEventCreate(&ev.b));
        HIP_CHECK(hipEventRecord(ev.a, st));
    }
    poison_lds(st, dev.num_cus);
    // One launch for the bin (a second launch for the rows of <= 16 entries costs more in launch tails beside the other bins than
    // the shorter tile saves, profiles/r04/r04_z2); the kernels take the 16-slot tile where the rows allow it.
    constexpr bool tiny16_on = true;
    const int count_le16 = std::min(count, std::max(0, first + count - std::max(first, n_gt16)));
    size_t smem = ((IMPLICIT || GRAMX) ? (size_t)gram_elems<real_t>(S) : 0) * sizeof(real_t);
    const int di = std::min(std::max(dev.device, 0), MAX_DEVICES - 1);
    CgParams<real_t> P1 = P;
    P1.order += first;
    P1.desc += first;
    P1.nrows = count;
    P1.counter = dev.row_counter.ptr + cg_counter_offset(BIN_TINY);
    {
        const bool mixed = tiny16_on && count_le16 > 0;
        auto kern = mixed ? cg_rows_tiny_kernel<real_t, S, IMPLICIT, GRAMX, 0> : cg_rows_tiny_kernel<real_t, S, IMPLICIT, GRAMX, 4>;
        static thread_local int bpc_dev[MAX_DEVICES][2] = {{0}};
        int &blocks_per_cu = bpc_dev[di][mixed ? 1 : 0];
        if (blocks_per_cu == 0) {
            int nb = 0;
            HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, smem));
            blocks_per_cu = std::max(1, nb);
        }
        const int grid = std::min((count + 3) / 4, dev.num_cus * blocks_per_cu);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, P1);
    }
    HIP_CHECK(hipGetLastError());
    if (tm) {
        HIP_CHECK(hipEventRecord(ev.b, st));
        tm->ev[BIN_TINY].push_back(ev);
    }
}

// (gram_wave_tu.hip)
void launch_gram_wave(dim3 grid, hipStream_t st, const CgParams<real_t> &P, const GramParams<real_t> &G, bool implicit, int rem);

// very heavy rows: one (pass, update) launch pair per CG pass
// split rows of this launch on the Gramian path?  (CMFREC_HIP_VH=stream / gram force one; k > 64 and block systems stream)
template <bool GRAMX>
inline bool vh_takes_gram(int k, const SparseShard &X)
{
    if (GRAMX || k > 16 * GRAM_NTT) return false;
    return (switches().vh != 0) ? switches().vh == 2 : X.prefer_gram((size_t)k * sizeof(real_t));
}

template <int S, bool IMPLICIT, bool GRAMX = false>
inline void launch_cg_vheavy(const DeviceInfo &dev_, CgParams<real_t> P, const SparseShard &X, BinTimers *tm, hipStream_t st)
{
    const int nvh = X.bin_rows[BIN_VHEAVY];
    if (nvh <= 0) return;
    struct { hipStream_t stream; int num_cus; } dev{st, dev_.num_cus};      // everything below launches on `st`
    EventPair ev{nullptr, nullptr};
    if (tm) {
        HIP_CHECK(hipEventCreate(&ev.a));
        HIP_CHECK(hipEventCreate(&ev.b));
        HIP_CHECK(hipEventRecord(ev.a, dev.stream));
    }
    // Default: read the gathered rows once and run the CG on the row's own Gramian (gram_cg_kernels.hpp); CMFREC_HIP_VH=stream
    // takes the launch pair per CG pass below (also the on-device cross-check of the Gramian path, and the only path for
    // k > 64 and for block systems).
    if (vh_takes_gram<GRAMX>(P.k, X)) {
        // one gather: Gramian slices on the matrix cores, then CG on the k x k system (gram_cg_kernels.hpp)
        GramParams<real_t> G;
        G.sl_vrow = X.sl_vrow.ptr; G.sl_first = X.sl_first.ptr; G.sl_count = X.sl_count.ptr; G.row_sl_off = X.row_sl_off.ptr;
        G.part = X.gram_part.ptr; G.n_slices = X.n_slices; G.nvh = nvh;
        P.nrows = nvh;
#ifdef CMF_CG_DEBUG
        if (switches().debug_ticks) {
            static unsigned long long *d_t = nullptr;
            unsigned long long h[4];
            if (d_t == nullptr) { HIP_CHECK(hipMalloc((void **)&d_t, sizeof(h))); HIP_CHECK(hipMemset(d_t, 0, sizeof(h))); }
            else {
                HIP_CHECK(hipDeviceSynchronize());
                HIP_CHECK(hipMemcpy(h, d_t, sizeof(h), hipMemcpyDeviceToHost));
                if (h[1] > 0)
                    fprintf(stderr, "gram_wave: %.0f ticks per wave, %llu waves, %llu slices, %llu nnz: %.1f ticks per nnz and wave\n",
                            (double)h[0] / h[1], h[1], h[2], h[3], (double)h[0] / (double)h[3]);
                HIP_CHECK(hipMemset(d_t, 0, sizeof(h)));
            }
            G.ticks = d_t;
        }
#endif
        // slice partials: one wavefront per slice straight from the gather (gram_wave_kernel), or the LDS-staged workgroup
        // kernel (CMFREC_HIP_GRAM_KERNEL=slice)
        poison_lds(dev.stream, dev.num_cus);
        const bool slice_kernel = switches().gram_slice;
        if (slice_kernel)
            hipLaunchKernelGGL((gram_slice_kernel<real_t, IMPLICIT>), dim3(std::min(X.n_slices, dev.num_cus * 2)), dim3(64 * GRAM_NW), 0,
                           dev.stream, P, G);
        else {
            // double precision, 48 < k <= 52 (k = 50): the two to four columns of the last block as vector products instead of
            // four matrix-core tiles per slab
            const dim3 gw(std::min((X.n_slices + 3) / 4, dev.num_cus * 4));
            const int rem = (sizeof(real_t) == 8) ? P.k - 16 * (GRAM_NTT - 1) : 0;
            launch_gram_wave(gw, dev.stream, P, G, IMPLICIT, rem);       // gram_wave_tu.hip: built with its own scheduler
        }
        poison_lds(dev.stream, dev.num_cus);
        hipLaunchKernelGGL((gram_cg_kernel<real_t, IMPLICIT>), dim3(std::min(nvh, dev.num_cus * 8)), dim3(256), 0, dev.stream, P, G);
        HIP_CHECK(hipGetLastError());
        if (tm) {
            HIP_CHECK(hipEventRecord(ev.b, dev.stream));
            tm->ev[BIN_VHEAVY].push_back(ev);
        }
        return;
    }
    VhState<real_t> V;
    V.r = X.vh_r.ptr; V.p = X.vh_p.ptr; V.r_old = X.vh_r_old.ptr; V.done = X.vh_done.ptr; V.part = X.vh_part.ptr;
    V.chunk_row = X.vh_chunk_row.ptr; V.chunk_start = X.vh_chunk_start.ptr; V.chunk_cnt = X.vh_chunk_cnt.ptr;
    V.chunk_off = X.vh_chunk_off.ptr; V.launch = X.vh_launch.ptr; V.work = X.vh_work.ptr;
    V.nvh = nvh; V.nchunks = X.n_chunks; V.nlaunch = X.n_launch;
    P.nrows = nvh;
    poison_lds(dev.stream, dev.num_cus);
    const dim3 gp(X.n_launch), bp(64 * VH_CHUNK_TILES), gu(nvh), bu(64 * VH_UPD_WAVES);
    hipLaunchKernelGGL((vh_pass_kernel<real_t, S, IMPLICIT, 0>), gp, bp, 0, dev.stream, P, V);
    hipLaunchKernelGGL((vh_update_kernel<real_t, IMPLICIT, 0, GRAMX>), gu, bu, 0, dev.stream, P, V);
    for (int step = 0; step < P.max_cg_steps; step++) {
        hipLaunchKernelGGL((vh_pass_kernel<real_t, S, IMPLICIT, 1>), gp, bp, 0, dev.stream, P, V);
        hipLaunchKernelGGL((vh_update_kernel<real_t, IMPLICIT, 1, GRAMX>), gu, bu, 0, dev.stream, P, V);
    }
    HIP_CHECK(hipGetLastError());
    if (tm) {
        HIP_CHECK(hipEventRecord(ev.b, dev.stream));
        tm->ev[BIN_VHEAVY].push_back(ev);
    }
}

// number of streams the nnz bins of a half-step are spread over (CMFREC_HIP_BINS_PAR; 1 = one after the other)
inline int cg_bin_streams()
{
    const int n = switches().bins_par;  // C2: 4.19 (1) / 3.99 (2) / 4.21 (3) / 4.11 (4) / 4.30 (6) ms per iteration, profiles/r03_d
    return std::min(std::max(n, 1), DeviceInfo::MAX_BIN_STREAMS + 1);
}

// A length bin of double-precision rows as two launches by tile size (cg_kernels.hpp, NTSEL): the rows of more than 48 W entries lead
// the bin's processing order and keep two wavefronts per SIMD, the others run on the build that fits three.  One event pair around
// both; the second launch on its own counter set.  CMFREC_HIP_NT_SPLIT=0: one launch (A/B switch).
template <int S, bool IMPLICIT, int W, int RPB, bool GRAMX>
inline void launch_cg_bin_by_tile(const DeviceInfo &dev, const CgParams<real_t> &P, const SparseShard &X, int first, int count, BinTimers *tm, int bin,
                                  hipStream_t st)
{
    // (double precision: not the two-wave teams -- a workgroup of two wavefronts with its own copy of the Gramian, 28 KB, fits five
    //  times into the 160 KB of LDS, not the six that three wavefronts per SIMD would be: measured 4-7 % slower as two launches, r05_u)
#ifdef CMFREC_HIP_FLOAT
    constexpr bool by_tile = S >= CMF_CG_NT_MIN_S && CMF_CG_NT_F32 && W <= 4;
#else
    // (round 6: the eight-wave teams too -- their rows of 257..384 entries on the build for 5 / 6 entries per lane group, 168 registers:
    //  the workgroup no longer owns every register of its CU, the neighbouring stream's workgroups fit beside it)
    constexpr bool by_tile = S >= CMF_CG_NT_MIN_S && S <= CG_NTSEL_MAX_S && (W == 1 || W == 4 || W == 8);
#endif
    if (count <= 0) return;
    if constexpr (by_tile) {
        if (switches().nt_split) {
            const int n_hi = std::min(count, std::max(0, X.n_gt_low[W == 1 ? 0 : W == 2 ? 1 : W == 4 ? 2 : 3] - first));
            EventPair ev{nullptr, nullptr};
            if (tm) { HIP_CHECK(hipEventCreate(&ev.a)); HIP_CHECK(hipEventCreate(&ev.b)); HIP_CHECK(hipEventRecord(ev.a, st)); }
            launch_cg_bin<S, IMPLICIT, W, RPB, GRAMX, 0, 2>(dev, P, first, n_hi, nullptr, bin, st);
            launch_cg_bin<S, IMPLICIT, W, RPB, GRAMX, 0, 1>(dev, P, first + n_hi, count - n_hi, nullptr, bin, st, NBINS + 2 + bin);
            if (tm) { HIP_CHECK(hipEventRecord(ev.b, st)); tm->ev[bin].push_back(ev); }
            return;
        }
    }
    launch_cg_bin<S, IMPLICIT, W, RPB, GRAMX>(dev, P, first, count, tm, bin, st);
}

// one of the register-tiled bins (8 / 4 / 2 / 1 wavefronts per row)
template <int S, bool IMPLICIT, bool GRAMX>
inline void launch_cg_any_bin(const DeviceInfo &dev, const CgParams<real_t> &P, const SparseShard &X, BinTimers *tm, int bin, hipStream_t st)
{
    const int first = X.bin_first[bin], count = X.bin_rows[bin];
    switch (bin) {
        case BIN_HEAVY:
#ifdef CMFREC_HIP_FLOAT
            {
                // single precision: rows of 257..512 entries on four waves with two resident tiles each (cg_rows_kernel, NRES_),
                // the longer ones on eight; one event pair around both launches, the second on a spare counter set
                const int n_long8 = std::min(count, std::max(0, X.n_gt512 - first));
                EventPair ev{nullptr, nullptr};
                if (tm) { HIP_CHECK(hipEventCreate(&ev.a)); HIP_CHECK(hipEventCreate(&ev.b)); HIP_CHECK(hipEventRecord(ev.a, st)); }
                launch_cg_bin<S, IMPLICIT, 8, 1, GRAMX>(dev, P, first, n_long8, nullptr, bin, st);
                launch_cg_bin<S, IMPLICIT, 4, 1, GRAMX, 2>(dev, P, first + n_long8, count - n_long8, nullptr, bin, st, NBINS);
                if (tm) { HIP_CHECK(hipEventRecord(ev.b, st)); tm->ev[bin].push_back(ev); }
                break;
            }
            // (double precision: rows of 257..384 entries on six-wave teams were measured slower -- 0.245 -> 0.27 ms for C2's items,
            //  0.184 -> 0.203 for its users, profiles/r04/r04_u -- a second launch per bin costs more than the idle waves)
#endif
            launch_cg_bin_by_tile<S, IMPLICIT, 8, 1, GRAMX>(dev, P, X, first, count, tm, bin, st); break;
        case BIN_MED4: launch_cg_bin_by_tile<S, IMPLICIT, 4, 1, GRAMX>(dev, P, X, first, count, tm, bin, st); break;
        case BIN_MED2: launch_cg_bin_by_tile<S, IMPLICIT, 2, 1, GRAMX>(dev, P, X, first, count, tm, bin, st); break;
        default: launch_cg_bin_by_tile<S, IMPLICIT, 1, 4, GRAMX>(dev, P, X, first, count, tm, bin, st); break;
    }
}

template <int S, bool IMPLICIT, bool GRAMX = false>
inline void launch_cg_S(const DeviceInfo &dev, const CgParams<real_t> &P, const SparseShard &X, BinTimers *tm)
{
    // Few split rows (less than about one round of workgroups per pass: the users of C2) make their launch sequence --
    // 2 kernels per CG pass -- a chain of latencies, 0.08 ms with next to no work; it then runs on the second stream
    // beside the other bins.  Many split rows (the items of C2) fill the chip by themselves and stay in line.
    if (dev.row_counter.n < ROW_COUNTER_INTS) const_cast<DeviceInfo &>(dev).row_counter.alloc(ROW_COUNTER_INTS);
    HIP_CHECK(hipMemsetAsync(dev.row_counter.ptr, 0, ROW_COUNTER_INTS * sizeof(int), dev.stream));   // work counters of the bins
    // (The Gramian path is two launches, not two per pass, and its slices are short when the rows are few: it stays in line.)
    const bool vh_aside = X.bin_rows[BIN_VHEAVY] > 0 && X.vh_runs_aside(dev.num_cus) && !vh_takes_gram<GRAMX>(P.k, X);
    // A shard that is one of several parts of a block (multi-GPU overlap, session.hip) has a quarter of the rows per
    // launch, so ramp-up and tail of every bin weigh four times as much: its bins alternate between the two streams,
    // the next bin fills the CUs the previous one is vacating.  (Not for whole blocks: there the per-bin event timings
    // feed the roofline report and must not overlap.)
    const bool alt = X.is_part;
    DeviceInfo &d = const_cast<DeviceInfo &>(dev);
    // Round 3: the bins of a half-step side by side.  Every bin is a persistent launch whose teams claim rows from a counter, so
    // a launch that shares the chip with its neighbours just takes its rows as workgroups become resident; what disappears are
    // the ramp-up and the tail of six launches per half-step, during which most CUs idle (C2: 4.36 -> 4.09 ms with two streams).
    // CMFREC_HIP_BINS_PAR = number of streams (1: in line as before; default cg_bin_streams()).  The per-bin events then
    // measure overlapping spans.
    const int npar = cg_bin_streams();
    if (npar > 1 && !(vh_aside || alt)) {
        d.ensure_bin_streams(npar - 1);
        hipStream_t ss[DeviceInfo::MAX_BIN_STREAMS + 1];
        ss[0] = dev.stream;
        for (int i = 1; i < npar; i++) ss[i] = d.bin_streams[i - 1];
        HIP_CHECK(hipEventRecord(d.fork_ev, dev.stream));
        for (int i = 1; i < npar; i++) HIP_CHECK(hipStreamWaitEvent(ss[i], d.fork_ev, 0));
        // Bins onto streams: longest estimated bin first, each to the stream with the least work so far (LPT), so that the
        // streams end together and no bin's tail runs alone.  Estimate = entries + a per-row share in entry equivalents, fitted
        // to the in-line times of the bins of C2 (profiles/r03_f, both half-steps: ~10.5 M entries per ms everywhere, plus ~12 ns
        // per row of an 8-wave team, ~1 ns per row of the tiny bin, < 1 ns elsewhere).
        static const double row_equiv[NBINS] = {0.0, 126.0, 7.0, 7.0, 0.0, 11.5};
        double cost[NBINS], load[DeviceInfo::MAX_BIN_STREAMS + 1] = {0, 0, 0, 0, 0, 0};
        int order_b[NBINS];
        for (int b = 0; b < NBINS; b++) {
            cost[b] = (double)X.bin_nnz[b] + row_equiv[b] * (double)X.bin_rows[b];
            order_b[b] = b;
        }
        std::sort(order_b, order_b + NBINS, [&](int a, int b2) { return cost[a] > cost[b2]; });
        for (int q = 0; q < NBINS; q++) {
            const int b = order_b[q];
            if (X.bin_rows[b] <= 0) continue;
            int si = 0;
            for (int i = 1; i < npar; i++) if (load[i] < load[si]) si = i;
            load[si] += cost[b];
            if (b == BIN_VHEAVY) launch_cg_vheavy<S, IMPLICIT, GRAMX>(dev, P, X, tm, ss[si]);
            else if (b == BIN_TINY) launch_cg_tiny<S, IMPLICIT, GRAMX>(dev, P, X.bin_first[BIN_TINY], X.bin_rows[BIN_TINY], tm, ss[si], X.n_gt16);
            else launch_cg_any_bin<S, IMPLICIT, GRAMX>(dev, P, X, tm, b, ss[si]);
        }
        for (int i = 1; i < npar; i++) {
            HIP_CHECK(hipEventRecord(d.bin_join[i - 1], ss[i]));
            HIP_CHECK(hipStreamWaitEvent(dev.stream, d.bin_join[i - 1], 0));
        }
        return;
    }
    if (vh_aside || alt) {
        d.ensure_aux();
        HIP_CHECK(hipEventRecord(d.fork_ev, dev.stream));
        HIP_CHECK(hipStreamWaitEvent(d.aux_stream, d.fork_ev, 0));
    }
    hipStream_t s0 = dev.stream, s1 = alt ? d.aux_stream : dev.stream;
    launch_cg_vheavy<S, IMPLICIT, GRAMX>(dev, P, X, tm, (vh_aside || (alt && X.is_part)) ? d.aux_stream : dev.stream);
    // then longest rows first: they are the longest-running teams
    launch_cg_any_bin<S, IMPLICIT, GRAMX>(dev, P, X, tm, BIN_HEAVY, s0);
    launch_cg_any_bin<S, IMPLICIT, GRAMX>(dev, P, X, tm, BIN_MED4, s1);
    launch_cg_any_bin<S, IMPLICIT, GRAMX>(dev, P, X, tm, BIN_MED2, s0);
    launch_cg_any_bin<S, IMPLICIT, GRAMX>(dev, P, X, tm, BIN_LIGHT, s1);
    launch_cg_tiny<S, IMPLICIT, GRAMX>(dev, P, X.bin_first[BIN_TINY], X.bin_rows[BIN_TINY], tm, s0, X.n_gt16);
    if (vh_aside || alt) {
        HIP_CHECK(hipEventRecord(d.join_ev, d.aux_stream));
        HIP_CHECK(hipStreamWaitEvent(dev.stream, d.join_ev, 0));
    }
}

template <int NF, bool IMPLICIT>
inline void launch_cg_generic(const DeviceInfo &dev, CgParams<real_t> P, const SparseShard &X, int first = 0, bool all_rows = false)
{
    int count = (P.kc > 0 || P.Bi != nullptr || all_rows) ? X.nrows : X.n_nonempty;   // rows without entries still have side information / get zeroed
    if (count <= first) return;
    // rows of 129 non-zeros and more (they lead the processing order): a workgroup per row -- sixteen wavefronts for the
    // rows beyond 1024 non-zeros (the longest row is the critical path of the launch: C1's items with implicit features
    // 31 -> 13 ms), four for the others; the rest: a wavefront per row
    const int nteam = std::min(count, X.bin_first[BIN_MED2]);
    const int nvh = std::min(nteam, X.bin_rows[BIN_VHEAVY]);
    poison_lds(dev.stream, dev.num_cus);
    if (nvh > first) {
        P.row_first = first; P.nrows = nvh;
        hipLaunchKernelGGL((cg_rows_generic_kernel<real_t, NF, IMPLICIT, 16>), dim3(std::min(nvh - first, dev.num_cus * 2)), dim3(1024), 0,
                           dev.stream, P);
    }
    if (nteam > std::max(nvh, first)) {
        P.row_first = std::max(nvh, first); P.nrows = nteam;
        hipLaunchKernelGGL((cg_rows_generic_kernel<real_t, NF, IMPLICIT, 4>), dim3(std::min(nteam - P.row_first, dev.num_cus * 8)), dim3(256), 0,
                           dev.stream, P);
    }
    if (count > std::max(nteam, first)) {
        P.row_first = std::max(nteam, first); P.nrows = count;
        int grid = std::min((count - P.row_first + 3) / 4, dev.num_cus * 8);
        hipLaunchKernelGGL((cg_rows_generic_kernel<real_t, NF, IMPLICIT, 1>), dim3(grid), dim3(256), 0, dev.stream, P);
    }
    HIP_CHECK(hipGetLastError());
}

#ifdef CMF_CG_TICKS
// 8 kernel families x 4 sums (cg_kernels.hpp, CgParams::ticks); one buffer per process, read and cleared by cmfrec_hip_debug_cg_ticks
inline unsigned long long *cg_ticks_buffer()
{
    static unsigned long long *buf = nullptr;
    if (buf == nullptr) {
        HIP_CHECK(hipMalloc((void **)&buf, 32 * sizeof(unsigned long long)));
        HIP_CHECK(hipMemset(buf, 0, 32 * sizeof(unsigned long long)));
    }
    return buf;
}
#endif

inline int launch_cg(const DeviceInfo &dev, const CgCall &c, const SparseShard &X, BinTimers *tm = nullptr)
{
    CgParams<real_t> P;
    P.A = c.A; P.lda = c.lda; P.B = c.B; P.ldb = c.ldb; P.k = c.k;
    P.indptr = X.p.ptr; P.indices = X.i.ptr; P.values = X.v.ptr;
    P.bias_sub = c.bias_sub; P.order = X.order.ptr; P.desc = X.desc.ptr; P.nrows = 0; P.BtB = c.BtB;
    if (!c.implicit && X.weighted()) { P.weights = X.w.ptr; P.wsum = X.wsum.ptr; }
    P.lam = c.lam; P.lam_last = c.lam_last;
    P.scale_lam = c.scale_lam; P.scale_bias_const = c.scale_bias_const;
    P.max_cg_steps = c.max_cg_steps;
    P.precond = c.precond ? 1 : 0;
    P.koff = c.koff; P.kc = c.kc; P.CtC = c.CtC; P.UC = c.UC; P.w_side = c.w_side;
    P.rows_with_u = c.rows_with_u; P.p_side = c.p_side; P.scale_lam_sideinfo = c.scale_lam_sideinfo ? 1 : 0;
    if (c.X2) { P.indptr2 = c.X2->p.ptr; P.indices2 = c.X2->i.ptr; P.values2 = c.X2->v.ptr; P.C2 = c.C2; }
    P.Bi = c.Bi; P.BiTBi = c.BiTBi; P.ki = c.ki; P.w_imp = c.w_imp;
#ifdef CMF_CG_DEBUG
    P.dbg = switches().debug_skip;
#endif
#ifdef CMF_CG_TICKS
    P.ticks = cg_ticks_buffer();
#endif
    const int S = (c.k + 7) / 8;
    if (c.values_override != nullptr) P.values = c.values_override;
    if (c.weights_override != nullptr) { P.weights = c.weights_override; P.wsum = X.wsum_naz.ptr; }     // (NA_as_zero_X with weights only)
    if (c.wsum_override != nullptr) P.wsum = c.wsum_override;
    if (c.Gx != nullptr && (c.kc > 0 || c.X2 != nullptr || c.Bi != nullptr)) {
        // block systems whose X block is a matrix every row shares (NA_as_zero_X: collective_block_cg's NA_as_zero_X branches with the
        // precomputed B^T B, collective.c:2430-2445, :2700-2760): the lane <-> unknown kernel with CgParams::gx, every row of the launch
        if (c.implicit || c.skip_first != 0) {
            g_last_error = "cmfrec_hip: block CG with a shared matrix: the explicit model";
            return 2;
        }
        P.BtB = c.Gx; P.rconst = c.rconst_x; P.ldr = c.ldr_x; P.gx = c.gx_all_rows ? 2 : 1;   // 2: the constant of the biases / the mean exists
        const int NFb = (c.koff + c.k + 63) / 64;
        switch (NFb) {
            case 1: launch_cg_generic<1, false>(dev, P, X, 0, true); return 0;
            case 2: launch_cg_generic<2, false>(dev, P, X, 0, true); return 0;
            case 3: launch_cg_generic<3, false>(dev, P, X, 0, true); return 0;
            default: break;
        }
        g_last_error = "cmfrec_hip: block CG with a shared matrix: at most 192 unknowns per row";
        return 2;
    }
    if (c.Gx != nullptr && c.precond) {
        // ... with the Jacobi preconditioner (factors_explicit_pcg_NA_as_zero_weighted, common.c:1443-1613): the lane <-> unknown kernel
        // with the shared matrix and the per-row constant (CgParams::gx), any width it takes
        if (c.implicit || c.kc > 0 || c.Bi != nullptr || c.koff != 0 || c.skip_first != 0 || c.X2 != nullptr) {
            g_last_error = "cmfrec_hip: preconditioned CG with a shared matrix: plain explicit rows";
            return 2;
        }
        P.BtB = c.Gx; P.rconst = c.rconst_x; P.ldr = c.ldr_x; P.gx = 1;
        const int NFx = (c.k + 63) / 64;
        switch (NFx) {
            case 1: launch_cg_generic<1, false>(dev, P, X, 0, c.gx_all_rows); return 0;
            case 2: launch_cg_generic<2, false>(dev, P, X, 0, c.gx_all_rows); return 0;
            case 3: launch_cg_generic<3, false>(dev, P, X, 0, c.gx_all_rows); return 0;
            default: break;
        }
        g_last_error = "cmfrec_hip: preconditioned CG with a shared matrix: at most 192 unknowns per row";
        return 2;
    }
    if (c.Gx != nullptr) {
        // shared matrix + per-row constant on the GRAMX builds (no block structure: the explicit model's own lambda rules apply)
        if (c.implicit || c.precond || S > 8 || c.kc > 0 || c.Bi != nullptr || c.koff != 0 || c.skip_first != 0) {
            g_last_error = "cmfrec_hip: CG with a shared matrix: plain explicit rows of at most 64 unknowns, no preconditioner";
            return 2;
        }
        P.BtB = c.Gx; P.rconst = c.rconst_x; P.ldr = c.ldr_x;
#define CMF_XCASE(SS) case SS: launch_cg_S<SS, false, true>(dev, P, X, tm); break;
        switch (S) {
            CMF_XCASE(1) CMF_XCASE(2) CMF_XCASE(3) CMF_XCASE(4)
            CMF_XCASE(5) CMF_XCASE(6) CMF_XCASE(7) CMF_XCASE(8)
        }
#undef CMF_XCASE
        HIP_CHECK(hipGetLastError());
        return 0;
    }
    // Block systems (dense side information on EVERY row of the launch and / or implicit features, no k_user offset) on the
    // tiled kernels: the weighted Gramian w C^T C + w_i Bi^T Bi acts on the unknowns like the implicit model's B^T B, the row's
    // constant w (U C)_row + w_i sum Bi_j joins the first residual (GRAMX builds, cg_kernels.hpp).  Rows without entries
    // (still solved from their side information / zeroed) go through the generic kernel afterwards.
    const bool block = (c.kc > 0 || c.Bi != nullptr);
    const bool tiled_block = block && cg_variant_from_env() != CgVariant::Generic && !c.implicit && c.koff == 0 && !c.precond && c.skip_first == 0 &&
                             S <= 8 && c.X2 == nullptr && (c.kc == 0 || (c.rows_with_u >= X.nrows && c.CtC != nullptr && c.UC != nullptr)) &&
                             (c.Bi == nullptr || (c.gsum != nullptr && c.BiTBi != nullptr)) && c.kc <= c.k && c.ki <= c.k;
    if (tiled_block) {
        DeviceInfo &d = const_cast<DeviceInfo &>(dev);
        d.cg_gfull.alloc_at_least((size_t)c.k * c.k);
        d.cg_rconst.alloc_at_least((size_t)std::max(X.nrows, 1) * c.k);
        hipLaunchKernelGGL(block_gram_kernel<real_t>, dim3((c.k * c.k + 255) / 256), dim3(256), 0, dev.stream, c.CtC, c.kc, c.w_side, c.BiTBi,
                           c.ki, c.w_imp, c.k, d.cg_gfull.ptr);
        hipLaunchKernelGGL(block_rconst_kernel<real_t>, dim3((unsigned)(((size_t)X.nrows * c.k + 255) / 256)), dim3(256), 0, dev.stream, c.UC, c.kc,
                           c.w_side, c.gsum, c.ki, c.w_imp, c.k, (size_t)X.nrows, d.cg_rconst.ptr);
        P.BtB = d.cg_gfull.ptr; P.rconst = d.cg_rconst.ptr; P.ldr = (size_t)c.k;
#define CMF_BCASE(SS) case SS: launch_cg_S<SS, false, true>(dev, P, X, tm); break;
        switch (S) {
            CMF_BCASE(1) CMF_BCASE(2) CMF_BCASE(3) CMF_BCASE(4)
            CMF_BCASE(5) CMF_BCASE(6) CMF_BCASE(7) CMF_BCASE(8)
        }
#undef CMF_BCASE
        if (X.nrows > X.n_nonempty) {
            // rows without entries: solved from their side information alone / zeroed (collective.c:1258-1268), one wavefront per row
            CgParams<real_t> Pe = P;
            Pe.BtB = c.BtB; Pe.rconst = nullptr;
            Pe.row_first = X.n_nonempty; Pe.nrows = X.nrows;
            const int cnt = X.nrows - X.n_nonempty;
            const int NFe = (c.koff + c.k + 63) / 64;
            const int grid = std::min((cnt + 3) / 4, dev.num_cus * 8);
            switch (NFe) {
                case 1: hipLaunchKernelGGL((cg_rows_generic_kernel<real_t, 1, false, 1>), dim3(grid), dim3(256), 0, dev.stream, Pe); break;
                default: hipLaunchKernelGGL((cg_rows_generic_kernel<real_t, 2, false, 1>), dim3(grid), dim3(256), 0, dev.stream, Pe); break;
            }
        }
        HIP_CHECK(hipGetLastError());
        return 0;
    }
    // the Jacobi-preconditioned variants (not a default anywhere in the reference) and the remaining block systems run on the
    // generic kernel
    const bool generic = cg_variant_from_env() == CgVariant::Generic || S > 8 || c.precond || c.kc > 0 || c.Bi != nullptr || c.skip_first > 0;
    if (!generic) {
#define CMF_CASE(SS)                                                        \
    case SS:                                                                \
        if (c.implicit) launch_cg_S<SS, true>(dev, P, X, tm);               \
        else            launch_cg_S<SS, false>(dev, P, X, tm);              \
        return 0;
        switch (S) {
            CMF_CASE(1) CMF_CASE(2) CMF_CASE(3) CMF_CASE(4)
            CMF_CASE(5) CMF_CASE(6) CMF_CASE(7) CMF_CASE(8)
        }
#undef CMF_CASE
    }
    const int NF = (c.koff + c.k + 63) / 64;
#define CMF_GCASE(NN)                                                       \
    case NN:                                                                \
        if (c.implicit) launch_cg_generic<NN, true>(dev, P, X, c.skip_first);  \
        else            launch_cg_generic<NN, false>(dev, P, X, c.skip_first); \
        return 0;
    switch (NF) {
        CMF_GCASE(1) CMF_GCASE(2) CMF_GCASE(3) CMF_GCASE(4) CMF_GCASE(5)
    }
#undef CMF_GCASE
    g_last_error = "cmfrec_hip: CG path supports k <= 320 per solved matrix";
    return 2;
}

}  // namespace cmfhip
