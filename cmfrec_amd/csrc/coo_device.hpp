// coo_device.hpp -- COO -> CSR / CSC on the device.
//
// Device-side replacement of coo_to_csr_and_csc (/root/reference/src/helpers.c:1375-1491): a counting
// sort that is *stable in input order* -- the entries of a row keep their COO order, which fixes the
// floating-point summation order of every per-row reduction downstream.  Here: histogram (atomics on
// integers only) + exclusive scan for indptr, a stable LSD radix sort of (key, position) pairs for the
// permutation (rocPRIM; stable by contract), one gather for indices / values (alpha folded in: a single
// IEEE multiply, bit-identical to the host's), a stable descending sort of the row lengths for the
// processing order and the 16-byte row descriptors.  Integer / index work: bit-exact against the host path
// (tests/test_gpu_operators.py::test_coo_device_matches_host).
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <vector>

#include "device.hpp"

namespace cmfhip {

__global__ void coo_count_kernel(const int *__restrict__ key, size_t nnz, unsigned *__restrict__ counts)
{
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < nnz; e += (size_t)gridDim.x * blockDim.x)
        atomicAdd(&counts[key[e]], 1u);
}

__global__ void coo_iota_kernel(unsigned *__restrict__ out, size_t n)
{
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
        out[e] = (unsigned)e;
}

// wt / w_out: the observation weights travel with the values (null: none)
__global__ void coo_gather_kernel(const unsigned *__restrict__ perm, const int *__restrict__ other,
                                  const real_t *__restrict__ val, real_t subtract, real_t alpha,
                                  int *__restrict__ i_out, real_t *__restrict__ v_out, size_t nnz,
                                  const real_t *__restrict__ wt = nullptr, real_t *__restrict__ w_out = nullptr)
{
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < nnz; e += (size_t)gridDim.x * blockDim.x) {
        const unsigned src = perm[e];
        i_out[e] = other[src];
        real_t x = val[src];
        if (subtract != real_t(0)) x -= subtract;          // centring, common.c:3603-3613
        if (alpha != real_t(1)) x *= alpha;                // collective.c:9606-9611
        v_out[e] = x;
        if (wt != nullptr) w_out[e] = wt[src];
    }
}

// lambda multiplier of every row under scale_lam with observation weights: the sum of its weights in double, entry by entry
// in CSR order; 1 for a row without entries (wsumA / wsumB, collective.c:7978-8008)
__global__ void row_weight_sum_kernel(const size_t *__restrict__ p, const real_t *__restrict__ w, int nrows, real_t *__restrict__ wsum)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    double acc = 0;
    for (size_t e = p[r]; e < p[r + 1]; e++) acc += (double)w[e];
    wsum[r] = (p[r + 1] > p[r]) ? (real_t)acc : (real_t)1;
}

// The same sweep with observation weights (initialize_biases_onesided / _twosided, weighted branches without NA_as_zero:
// common.c:4180-4205, :4672-4692, :4826-4847): a weighted running mean in double, then the shrinkage
// wsum / (wsum + lam * mult), mult = the driver's wsumA / wsumB under scale_lam (SparseShard::wsum), else 1.  Rows beyond
// LONG_ROW entries: weighted sums over the lanes of a wavefront (the same quantity up to rounding, as in the unweighted sweep).
__device__ __forceinline__ double bias_scale_weighted(double wsum_run, bool nonempty, real_t lam_b, const real_t *wsum_drv, int r, int onesided)
{
#pragma clang fp contract(off)
    const double mult = (wsum_drv != nullptr) ? (double)wsum_drv[r] : 1.;
    if (onesided) {                       // common.c:4196-4203
        const double ws = nonempty ? wsum_run : 0.;
        return ws / (ws + (double)lam_b * mult);
    }
    return nonempty ? wsum_run / (wsum_run + (double)lam_b * mult) : 1.;
}

__global__ void bias_sweep_weighted_kernel(const size_t *__restrict__ p, const int *__restrict__ idx, const real_t *__restrict__ v,
                                           const real_t *__restrict__ w, const real_t *__restrict__ other, const int *__restrict__ order,
                                           int first_q, int rows, real_t lam_b, const real_t *__restrict__ wsum_drv, int onesided,
                                           real_t *__restrict__ bias)
{
#pragma clang fp contract(off)
    const int q = first_q + blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= rows) return;
    const int r = order[q];
    const size_t st = p[r], en = p[r + 1];
    double bm = 0, ws = 2.220446049250313e-16;      // DBL_EPSILON
    if (other != nullptr) {
        for (size_t e = st; e < en; e++) { ws += (double)w[e]; bm += ((double)w[e] * ((double)(v[e] - other[idx[e]]) - bm)) / ws; }
    } else {
        for (size_t e = st; e < en; e++) { ws += (double)w[e]; bm += ((((double)v[e]) - bm) * (double)w[e]) / ws; }
    }
    bm *= bias_scale_weighted(ws, en > st, lam_b, wsum_drv, r, onesided);
    bias[r] = (real_t)bm;
}

__global__ void __launch_bounds__(64)
bias_sweep_weighted_long_kernel(const size_t *__restrict__ p, const int *__restrict__ idx, const real_t *__restrict__ v,
                                const real_t *__restrict__ w, const real_t *__restrict__ other, const int *__restrict__ order, int n_long,
                                real_t lam_b, const real_t *__restrict__ wsum_drv, int onesided, real_t *__restrict__ bias)
{
#pragma clang fp contract(off)
    const int q = blockIdx.x;
    if (q >= n_long) return;
    const int r = order[q];
    const size_t st = p[r], en = p[r + 1];
    double sum = 0, ws = 0;
    if (other != nullptr) {
        for (size_t e = st + threadIdx.x; e < en; e += 64) { sum += (double)w[e] * (double)(v[e] - other[idx[e]]); ws += (double)w[e]; }
    } else {
        for (size_t e = st + threadIdx.x; e < en; e += 64) { sum += (double)w[e] * (double)v[e]; ws += (double)w[e]; }
    }
    sum = lanes::wave_sum(sum); ws = lanes::wave_sum(ws) + 2.220446049250313e-16;
    if (threadIdx.x == 0) bias[r] = (real_t)((sum / ws) * bias_scale_weighted(ws, en > st, lam_b, wsum_drv, r, onesided));
}

// One sweep of the bias start values over the rows of one orientation (initialize_biases_twosided /
// _onesided: src/common.c:4643-4669, 4799-4825, 4265-4289).
//   other == nullptr: one-sided.   user_rule: scale only rows with entries, by cnt (users of the two-sided
//   sweep); else by max(cnt, 1) (items, one-sided).
// Rows are taken in processing order (longest first).  Rows up to LONG_ROW entries: one thread per row, the
// entries strictly in CSR order with a double running mean -- the host arithmetic.  Longer rows (first_q..):
// one wavefront per row, lane-strided double sums + a fixed butterfly, mean = sum / cnt: the same quantity up
// to rounding (the strictly sequential form costs ~100 ns per entry on a GPU thread: 90 ms for the ten sweeps
// of a MovieLens-10M-shaped matrix).
// extra: attributes counted on top of the row's entries under scale_lam_sideinfo (wsumA / wsumB, collective.c:8071-8104)
__device__ __forceinline__ double bias_scale(size_t cnt, real_t lam_b, int scale_lam, int user_rule, int extra = 0)
{
#pragma clang fp contract(off)
    // unfused on purpose: the reference's own binary (gcc -ffp-contract=fast) fuses cnt + lam*cnt into an FMA in the
    // item sweep but not in the user sweep -- a lowering accident, not semantics (differences of 1 ulp)
    const double sc = (user_rule ? (double)cnt : (double)(cnt > 1 ? cnt : 1)) + (double)extra;
    const double den = (double)cnt + (double)lam_b * (scale_lam ? sc : 1.);
    return (!user_rule || cnt > 0) ? (double)cnt / den : 1.;
}

__global__ void bias_sweep_kernel(const size_t *__restrict__ p, const int *__restrict__ idx, const real_t *__restrict__ v,
                                  const real_t *__restrict__ other, const int *__restrict__ order, int first_q, int rows,
                                  real_t lam_b, int scale_lam, int user_rule, real_t *__restrict__ bias, int extra, int extra_rows)
{
#pragma clang fp contract(off)
    const int q = first_q + blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= rows) return;
    const int r = order[q];                   // rows of similar length share a wavefront
    const size_t st = p[r], en = p[r + 1], cnt = en - st;
    double bm = 0;
    if (other != nullptr) {
        for (size_t e = st; e < en; e++) bm += (v[e] - other[idx[e]] - bm) / (double)(e - st + 1);
    } else {
        for (size_t e = st; e < en; e++) bm += (v[e] - bm) / (double)(e - st + 1);
    }
    bm *= bias_scale(cnt, lam_b, scale_lam, user_rule, r < extra_rows ? extra : 0);
    bias[r] = (real_t)bm;
}

__global__ void __launch_bounds__(64)
bias_sweep_long_kernel(const size_t *__restrict__ p, const int *__restrict__ idx, const real_t *__restrict__ v,
                       const real_t *__restrict__ other, const int *__restrict__ order, int n_long,
                       real_t lam_b, int scale_lam, int user_rule, real_t *__restrict__ bias, int extra, int extra_rows)
{
#pragma clang fp contract(off)
    const int q = blockIdx.x;
    if (q >= n_long) return;
    const int r = order[q];
    const size_t st = p[r], en = p[r + 1], cnt = en - st;
    double sum = 0;
    if (other != nullptr) {
        for (size_t e = st + threadIdx.x; e < en; e += 64) sum += (double)(v[e] - other[idx[e]]);
    } else {
        for (size_t e = st + threadIdx.x; e < en; e += 64) sum += (double)v[e];
    }
    sum = lanes::wave_sum(sum);
    if (threadIdx.x == 0) bias[r] = (real_t)((sum / (double)cnt) * bias_scale(cnt, lam_b, scale_lam, user_rule, r < extra_rows ? extra : 0));
}

__global__ void coo_desc_kernel(const unsigned *__restrict__ ord, const unsigned *__restrict__ len_sorted,
                                const size_t *__restrict__ p, RowDesc *__restrict__ desc, int *__restrict__ order, int nrows)
{
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nrows; q += gridDim.x * blockDim.x) {
        const int r = (int)ord[q];
        order[q] = r;
        desc[q].row = r; desc[q].nnz = (int)len_sorted[q]; desc[q].st = (unsigned long long)p[r];
    }
}

struct u32_to_size {
    __host__ __device__ size_t operator()(unsigned c) const { return (size_t)c; }
};

// d_key / d_other / d_val: the COO triplet in HBM (key = the index that becomes the row of this shard)
inline void finalize_vheavy(SparseShard &S, int n_other, hipStream_t st);

inline void shard_from_coo(SparseShard &S, int nrows, int n_other, const int *d_key, const int *d_other, const real_t *d_val,
                           size_t nnz, real_t subtract, real_t alpha, hipStream_t st, const real_t *d_wt = nullptr)
{
    S.nrows = nrows; S.nnz = nnz; S.n_other = n_other;
    const int grid_e = (int)std::min<size_t>(4096, (nnz + 255) / 256 + 1), grid_r = std::min(2048, (nrows + 255) / 256 + 1);
    DevBuf<unsigned> counts; counts.alloc((size_t)nrows + 1);
    HIP_CHECK(hipMemsetAsync(counts.ptr, 0, ((size_t)nrows + 1) * sizeof(unsigned), st));
    if (nnz) hipLaunchKernelGGL(coo_count_kernel, dim3(grid_e), dim3(256), 0, st, d_key, nnz, counts.ptr);
    S.p.alloc((size_t)nrows + 1);
    DevBuf<unsigned char> tmp;
    {
        auto in = rocprim::make_transform_iterator(counts.ptr, u32_to_size());
        size_t bytes = 0;
        HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, in, S.p.ptr, (size_t)0, (size_t)nrows + 1, rocprim::plus<size_t>(), st));
        tmp.alloc(bytes + 16);
        HIP_CHECK(rocprim::exclusive_scan(tmp.ptr, bytes, in, S.p.ptr, (size_t)0, (size_t)nrows + 1, rocprim::plus<size_t>(), st));
    }
    S.i.alloc(nnz); S.v.alloc(nnz);
    S.w.release(); S.wsum.release();
    if (d_wt != nullptr && nnz) { S.w.alloc(nnz); S.wsum.alloc((size_t)nrows); }
    if (nnz) {
        DevBuf<int> keys_out; DevBuf<unsigned> pos, perm;
        keys_out.alloc(nnz); pos.alloc(nnz); perm.alloc(nnz);
        hipLaunchKernelGGL(coo_iota_kernel, dim3(grid_e), dim3(256), 0, st, pos.ptr, nnz);
        unsigned bits = 1;
        while (bits < 32 && (1ull << bits) < (unsigned long long)nrows) bits++;
        size_t bytes = 0;
        HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, d_key, keys_out.ptr, pos.ptr, perm.ptr, nnz, 0u, bits, st));
        DevBuf<unsigned char> tmp2; tmp2.alloc(bytes + 16);
        HIP_CHECK(rocprim::radix_sort_pairs(tmp2.ptr, bytes, d_key, keys_out.ptr, pos.ptr, perm.ptr, nnz, 0u, bits, st));
        hipLaunchKernelGGL(coo_gather_kernel, dim3(grid_e), dim3(256), 0, st, perm.ptr, d_other, d_val, subtract, alpha, S.i.ptr, S.v.ptr, nnz,
                           d_wt, S.w.ptr);
        if (S.w.ptr != nullptr)       // (before finalize_vheavy re-orders the split rows: the driver sums in CSR order)
            hipLaunchKernelGGL(row_weight_sum_kernel, dim3(grid_r), dim3(256), 0, st, S.p.ptr, S.w.ptr, nrows, S.wsum.ptr);
        HIP_CHECK(hipStreamSynchronize(st));       // temporaries are released here
    }
    // processing order: rows by length, descending, ties by row id (stable)
    DevBuf<unsigned> len_sorted, rows, ord;
    len_sorted.alloc(nrows); rows.alloc(nrows); ord.alloc(nrows);
    hipLaunchKernelGGL(coo_iota_kernel, dim3(grid_r), dim3(256), 0, st, rows.ptr, (size_t)nrows);
    {
        size_t bytes = 0;
        HIP_CHECK(rocprim::radix_sort_pairs_desc(nullptr, bytes, counts.ptr, len_sorted.ptr, rows.ptr, ord.ptr, (size_t)nrows, 0u, 32u, st));
        DevBuf<unsigned char> tmp3; tmp3.alloc(bytes + 16);
        HIP_CHECK(rocprim::radix_sort_pairs_desc(tmp3.ptr, bytes, counts.ptr, len_sorted.ptr, rows.ptr, ord.ptr, (size_t)nrows, 0u, 32u, st));
        S.order.alloc(nrows); S.desc.alloc(nrows);
        hipLaunchKernelGGL(coo_desc_kernel, dim3(grid_r), dim3(256), 0, st, ord.ptr, len_sorted.ptr, S.p.ptr, S.desc.ptr, S.order.ptr, nrows);
        std::vector<unsigned> hl(nrows);
        HIP_CHECK(hipMemcpyAsync(hl.data(), len_sorted.ptr, (size_t)nrows * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        S.build_bins(hl.data(), st);
    }
    finalize_vheavy(S, n_other, st);
}

// ---- very heavy rows: XCD-aware split-row schedule ---------------------------------------------
// The split-row path re-gathers the opposing rows of a very heavy row once per CG pass.  Popular items
// share their users, so the passes of all very heavy rows touch the same opposing rows over and over:
// if the entries of these rows are ordered by opposing index and cut at the same NX index boundaries,
// chunks of one index range read one 1/NX slice of the opposing matrix.  Workgroup b runs on XCD b % 8
// (observed placement, used for speed only), so range x is launched at positions 8q + x and swept in
// index order: the slice streams through that XCD's 4 MiB L2 once per pass instead of every chunk
// pulling its rows over the fabric.  Sorting a row's entries is a (stable) permutation of its sums.
constexpr int VH_NX_MAX = 64;
constexpr int VH_XCDS = 8;

__global__ void vh_partition_kernel(const RowDesc *__restrict__ desc, const int *__restrict__ idx, int nvh, int n_other,
                                    int VH_NX, int *__restrict__ off /* [nvh][VH_NX + 1] */)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nvh * (VH_NX + 1)) return;
    const int vi = t / (VH_NX + 1), x = t % (VH_NX + 1);
    const RowDesc d = desc[vi];
    const long long bound = ((long long)n_other * x + VH_NX - 1) / VH_NX;     // first index of range x
    int lo = 0, hi = d.nnz;
    while (lo < hi) {                                                            // lower_bound in the sorted row
        const int mid = (lo + hi) >> 1;
        if (idx[d.st + mid] < bound) lo = mid + 1; else hi = mid;
    }
    off[t] = lo;
}

inline void finalize_vheavy(SparseShard &S, int n_other, hipStream_t st)
{
    const int nvh = S.bin_rows[BIN_VHEAVY];
    S.n_other = n_other;
    if (nvh <= 0 || n_other <= 0) return;
    const int VH_NX = 8;                            // index ranges (a multiple of the XCD count)
    std::vector<RowDesc> hd(nvh);
    HIP_CHECK(hipMemcpyAsync(hd.data(), S.desc.ptr, (size_t)nvh * sizeof(RowDesc), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    // 1. entries of every very heavy row by opposing index (stable)
    {
        std::vector<unsigned long long> hb(nvh), he(nvh);
        unsigned long long lo = ~0ull, hi = 0;
        for (int v = 0; v < nvh; v++) {
            hb[v] = hd[v].st; he[v] = hd[v].st + (unsigned long long)hd[v].nnz;
            lo = std::min(lo, hb[v]); hi = std::max(hi, he[v]);
        }
        DevBuf<unsigned long long> db, de; db.upload(hb.data(), nvh, st); de.upload(he.data(), nvh, st);
        DevBuf<int> ki; DevBuf<real_t> kv;
        ki.alloc(S.nnz); kv.alloc(S.nnz);
        HIP_CHECK(hipMemcpyAsync(ki.ptr, S.i.ptr, S.nnz * sizeof(int), hipMemcpyDeviceToDevice, st));
        HIP_CHECK(hipMemcpyAsync(kv.ptr, S.v.ptr, S.nnz * sizeof(real_t), hipMemcpyDeviceToDevice, st));
        unsigned bits = 1;
        while (bits < 32 && (1ull << bits) < (unsigned long long)n_other) bits++;
        size_t bytes = 0;
        HIP_CHECK(rocprim::segmented_radix_sort_pairs(nullptr, bytes, ki.ptr, S.i.ptr, kv.ptr, S.v.ptr, S.nnz, (unsigned)nvh,
                                                      db.ptr, de.ptr, 0u, bits, st));
        DevBuf<unsigned char> tmp; tmp.alloc(bytes + 16);
        HIP_CHECK(rocprim::segmented_radix_sort_pairs(tmp.ptr, bytes, ki.ptr, S.i.ptr, kv.ptr, S.v.ptr, S.nnz, (unsigned)nvh,
                                                      db.ptr, de.ptr, 0u, bits, st));
        if (S.w.ptr != nullptr) {
            // the weights take the same (stable) permutation: the same sort once more, on the saved keys
            DevBuf<int> ki2; ki2.alloc(S.nnz);
            HIP_CHECK(hipMemcpyAsync(kv.ptr, S.w.ptr, S.nnz * sizeof(real_t), hipMemcpyDeviceToDevice, st));
            HIP_CHECK(rocprim::segmented_radix_sort_pairs(tmp.ptr, bytes, ki.ptr, ki2.ptr, kv.ptr, S.w.ptr, S.nnz, (unsigned)nvh,
                                                          db.ptr, de.ptr, 0u, bits, st));
            HIP_CHECK(hipStreamSynchronize(st));
        }
        HIP_CHECK(hipStreamSynchronize(st));
    }
    // 2. where the index ranges start inside each row
    std::vector<int> off((size_t)nvh * (VH_NX + 1));
    {
        DevBuf<int> doff; doff.alloc(off.size());
        hipLaunchKernelGGL(vh_partition_kernel, dim3((int)(off.size() + 255) / 256), dim3(256), 0, st, S.desc.ptr, S.i.ptr, nvh,
                           n_other, VH_NX, doff.ptr);
        doff.download(off.data(), off.size(), st);
        HIP_CHECK(hipStreamSynchronize(st));
    }
    // 3. chunks (row-contiguous: the order their partials are added in) and the launch map
    const int CHN = TILE * VH_CHUNK_TILES;
    std::vector<int> c_row, c_start, c_cnt, c_off(1, 0);
    struct Key { double pos; int chunk; };
    std::vector<Key> per_x[VH_XCDS];
    for (int v = 0; v < nvh; v++) {
        for (int x = 0; x < VH_NX; x++) {
            const int a = off[(size_t)v * (VH_NX + 1) + x], b = off[(size_t)v * (VH_NX + 1) + x + 1], len = b - a;
            if (len <= 0) continue;
            const int nch = (len + CHN - 1) / CHN;
            const int per = (((len + nch - 1) / nch) + 7) / 8 * 8;
            for (int f = 0; f < len; f += per) {
                per_x[x % VH_XCDS].push_back(Key{(double)(x / VH_XCDS) + (double)f / (double)len, (int)c_row.size()});   // range x -> XCD x % 8, ranges in turn
                c_row.push_back(v); c_start.push_back(a + f); c_cnt.push_back(std::min(per, len - f));
            }
        }
        c_off.push_back((int)c_row.size());
    }
    size_t longest = 0;
    for (int x = 0; x < VH_XCDS; x++) {
        std::stable_sort(per_x[x].begin(), per_x[x].end(), [](const Key &a, const Key &b) { return a.pos < b.pos; });
        longest = std::max(longest, per_x[x].size());
    }
    std::vector<int> launch(longest * VH_XCDS, -1);
    for (int x = 0; x < VH_XCDS; x++)
        for (size_t q = 0; q < per_x[x].size(); q++) launch[q * VH_XCDS + x] = per_x[x][q].chunk;
    S.set_vh_chunks(c_row, c_start, c_cnt, c_off, launch, st);
}

// host CSR -> shard (SparseShard::upload) + the very-heavy-row schedule
inline void shard_from_csr(SparseShard &S, int nrows, const size_t *hp, const int *hi, const real_t *hv, int n_other,
                           hipStream_t st, const real_t *hw = nullptr)
{
    S.n_other = n_other;
    S.upload(nrows, hp, hi, hv, st, hw);
    finalize_vheavy(S, n_other, st);
}

}  // namespace cmfhip
