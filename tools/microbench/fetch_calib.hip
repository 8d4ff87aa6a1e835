// Calibration of rocprofv3's FETCH_SIZE on the access pattern of the CG row kernels: the
// micro-architecture guide calibrates the gfx950 "x2" correction only for 16 B / lane coalesced reads.  Two kernels with a
// KNOWN byte count, each launched alone (run under `rocprofv3 --pmc FETCH_SIZE`):
//   calib_stream_kernel : 16 B per lane, fully coalesced, every byte of a 2 GiB buffer once
//   calib_gather_kernel : the gather of cg_kernels.hpp -- rows of k = 50 doubles (400 B) at distinct random positions of
//                         a 3.2 GB matrix, read as 8-lane x 8-byte (64 B) segments, 7 segments per row, each row once
// The host prints, per kernel, the logical bytes and the bytes of all 64 B sectors / 128 B lines the rows touch; tools/
// fetch_calibration.py divides them by the counter.   hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

__global__ void calib_stream_kernel(const double2 *__restrict__ src, size_t n16, double *__restrict__ sink)
{
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = src[i];
        acc += v.x + v.y;
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

// one wavefront per 8 rows: lane = jj * 8 + ll reads row idx[8 w + jj], columns ll + 8 s (the CG kernels' load_tile)
__global__ void calib_gather_kernel(const double *__restrict__ B, size_t ldb, int k, const int *__restrict__ idx, size_t nrows,
                                    double *__restrict__ sink)
{
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63, jj = lane >> 3, ll = lane & 7;
    const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    double acc = 0;
    for (size_t g = wave; g * 8 < nrows; g += nw) {
        const size_t r = g * 8 + jj;
        if (r < nrows) {
            const double *rp = B + (size_t)idx[r] * ldb + ll;
#pragma unroll
            for (int s = 0; s < 7; s++)
                if (ll + 8 * s < k) acc += rp[8 * s];
        }
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

// one wavefront per 4 rows: lane = kc * 16 + lm reads row idx[4 w + kc], columns lm + 16 cb (gram_wave_kernel's slab:
// 16 lanes x 8 B = 128 contiguous bytes per row and instruction)
__global__ void calib_gather128_kernel(const double *__restrict__ B, size_t ldb, int k, const int *__restrict__ idx, size_t nrows,
                                       double *__restrict__ sink)
{
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63, kc = lane >> 4, lm = lane & 15;
    const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    double acc = 0;
    for (size_t g = wave; g * 4 < nrows; g += nw) {
        const size_t r = g * 4 + kc;
        if (r < nrows) {
            const double *rp = B + (size_t)idx[r] * ldb;
#pragma unroll
            for (int cb = 0; cb < 4; cb++) acc += rp[min(lm + 16 * cb, k - 1)];
        }
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

int main()
{
    const size_t stream_bytes = (size_t)2 << 30;
    const int k = 50;
    const size_t R = 8000000, N = 4000000;            // matrix rows, gathered rows (distinct)
    double *dS, *dB, *sink; int *dIdx;
    hipMalloc(&dS, stream_bytes); hipMalloc(&dB, R * k * sizeof(double)); hipMalloc(&sink, 64); hipMalloc(&dIdx, N * sizeof(int));
    hipMemset(dS, 0, stream_bytes); hipMemset(dB, 0, R * k * sizeof(double));
    std::vector<int> perm(R); std::iota(perm.begin(), perm.end(), 0);
    std::mt19937_64 rng(7); std::shuffle(perm.begin(), perm.end(), rng);
    perm.resize(N);
    hipMemcpy(dIdx, perm.data(), N * sizeof(int), hipMemcpyHostToDevice);
    // bytes of the sectors / lines the gathered rows touch (rows are 400 B: not aligned to either)
    std::vector<char> s64((R * k * 8 + 63) / 64, 0), s128((R * k * 8 + 127) / 128, 0);
    for (size_t i = 0; i < N; i++) {
        const size_t b0 = (size_t)perm[i] * k * 8, b1 = b0 + (size_t)k * 8 - 1;
        for (size_t s = b0 / 64; s <= b1 / 64; s++) s64[s] = 1;
        for (size_t s = b0 / 128; s <= b1 / 128; s++) s128[s] = 1;
    }
    size_t n64 = 0, n128 = 0;
    for (char c : s64) n64 += c;
    for (char c : s128) n128 += c;
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(calib_stream_kernel, dim3(4096), dim3(256), 0, 0, (const double2 *)dS, stream_bytes / 16, sink);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(calib_gather_kernel, dim3(4096), dim3(256), 0, 0, dB, (size_t)k, k, dIdx, N, sink);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(calib_gather128_kernel, dim3(4096), dim3(256), 0, 0, dB, (size_t)k, k, dIdx, N, sink);
        hipDeviceSynchronize();
    }
    printf("{\"stream_bytes\": %zu, \"gather_logical_bytes\": %zu, \"gather_sector64_bytes\": %zu, \"gather_line128_bytes\": %zu, \"launches_each\": 3}\n",
           stream_bytes, N * (size_t)k * 8, n64 * 64, n128 * 128);
    return 0;
}
