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

__global__ void coo_gather_kernel(const unsigned *__restrict__ perm, const int *__restrict__ other,
                                  const real_t *__restrict__ val, real_t subtract, real_t alpha,
                                  int *__restrict__ i_out, real_t *__restrict__ v_out, size_t nnz)
{
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < nnz; e += (size_t)gridDim.x * blockDim.x) {
        const unsigned src = perm[e];
        i_out[e] = other[src];
        real_t x = val[src];
        if (subtract != real_t(0)) x -= subtract;          // centring, common.c:3603-3613
        if (alpha != real_t(1)) x *= alpha;                // collective.c:9606-9611
        v_out[e] = x;
    }
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
__device__ __forceinline__ double bias_scale(size_t cnt, real_t lam_b, int scale_lam, int user_rule)
{
#pragma clang fp contract(off)
    // unfused on purpose: the reference's own binary (gcc -ffp-contract=fast) fuses cnt + lam*cnt into an FMA in the
    // item sweep but not in the user sweep -- a lowering accident, not semantics (differences of 1 ulp)
    const double sc = user_rule ? (double)cnt : (double)(cnt > 1 ? cnt : 1);
    const double den = (double)cnt + (double)lam_b * (scale_lam ? sc : 1.);
    return (!user_rule || cnt > 0) ? (double)cnt / den : 1.;
}

__global__ void bias_sweep_kernel(const size_t *__restrict__ p, const int *__restrict__ idx, const real_t *__restrict__ v,
                                  const real_t *__restrict__ other, const int *__restrict__ order, int first_q, int rows,
                                  real_t lam_b, int scale_lam, int user_rule, real_t *__restrict__ bias)
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
    bm *= bias_scale(cnt, lam_b, scale_lam, user_rule);
    bias[r] = (real_t)bm;
}

__global__ void __launch_bounds__(64)
bias_sweep_long_kernel(const size_t *__restrict__ p, const int *__restrict__ idx, const real_t *__restrict__ v,
                       const real_t *__restrict__ other, const int *__restrict__ order, int n_long,
                       real_t lam_b, int scale_lam, int user_rule, real_t *__restrict__ bias)
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
    if (threadIdx.x == 0) bias[r] = (real_t)((sum / (double)cnt) * bias_scale(cnt, lam_b, scale_lam, user_rule));
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
inline void shard_from_coo(SparseShard &S, int nrows, const int *d_key, const int *d_other, const real_t *d_val,
                           size_t nnz, real_t subtract, real_t alpha, hipStream_t st)
{
    S.nrows = nrows; S.nnz = nnz;
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
        hipLaunchKernelGGL(coo_gather_kernel, dim3(grid_e), dim3(256), 0, st, perm.ptr, d_other, d_val, subtract, alpha, S.i.ptr, S.v.ptr, nnz);
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
}

}  // namespace cmfhip
