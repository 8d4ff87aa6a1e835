// Speed of light of the CG kernels' gather on one MI355X: rows of k doubles (ld = k, unaligned 400-byte rows for k = 50) at
// random positions, each wavefront fetching 64-row tiles into registers the way load_tile does, at the occupancy of the row
// kernels (8 wavefronts per CU) and at full occupancy.  Variants: 8 lanes x 8 B per row segment (global_load_dwordx2, the
// kernels' layout) against 4 lanes x 16 B (global_load_dwordx4).  Prints useful GB/s (rows x k x 8 bytes).
//   hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

template <int W>   // W = 8: dwordx2, 8 lanes per row; W = 4: dwordx4, 4 lanes per row
__global__ void __launch_bounds__(256) gather_kernel(const double *__restrict__ B, unsigned ldb_bytes, int k, const int *__restrict__ idx,
                                                     size_t ntiles, double *__restrict__ sink, int passes)
{
    extern __shared__ double pad[];
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    double acc = 0;
    for (size_t tl = wave; tl < ntiles; tl += nw) {
        const int my = idx[tl * 64 + lane];
        if (W == 8) {
            const int jj = lane >> 3, ll = lane & 7;
            const int col_last = min(ll + 48, k - 1) - ll;
            double v[8][7];
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const unsigned it = (unsigned)__shfl(my, jj * 8 + t);
                const double *rp = reinterpret_cast<const double *>(reinterpret_cast<const char *>(B + ll) + (unsigned long long)it * ldb_bytes);
#pragma unroll
                for (int s = 0; s < 7; s++) v[t][s] = rp[s < 6 ? 8 * s : col_last];
            }
            for (int p = 0; p < passes; p++)
#pragma unroll
                for (int t = 0; t < 8; t++)
#pragma unroll
                    for (int s = 0; s < 7; s++) acc = fma(v[t][s], acc, v[t][s]);
        } else {
            const int jj = lane >> 2, ll = lane & 3;          // 16 row groups x 4 lanes, a lane owns columns 2 ll, 2 ll + 1 (+ 8 s)
            const int col_last = min(2 * ll + 48, k - 2) - 2 * ll;
            double2 v[4][7];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const unsigned it = (unsigned)__shfl(my, jj * 4 + t);
                const double *rp = reinterpret_cast<const double *>(reinterpret_cast<const char *>(B + 2 * ll) + (unsigned long long)it * ldb_bytes);
#pragma unroll
                for (int s = 0; s < 7; s++) v[t][s] = *reinterpret_cast<const double2 *>(rp + (s < 6 ? 8 * s : col_last));
            }
            for (int p = 0; p < passes; p++)
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int s = 0; s < 7; s++) { acc = fma(v[t][s].x, acc, v[t][s].y); acc = fma(v[t][s].y, acc, v[t][s].x); }
        }
    }
    if (acc == 1.2345e300) sink[0] = acc + pad[0];
}

int main()
{
    const int k = 50;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double *sink; hipMalloc(&sink, 64);
    hipFuncSetAttribute(reinterpret_cast<const void *>(gather_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(gather_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (size_t R : {(size_t)358868, (size_t)8000000}) {
        const size_t N = 64 * 65536 * 2;                   // gathered rows per launch (8.4 M x 400 B = 3.4 GB)
        double *dB; int *dIdx;
        hipMalloc(&dB, R * k * sizeof(double) + 256); hipMalloc(&dIdx, N * sizeof(int));
        hipMemset(dB, 0, R * k * sizeof(double) + 256);
        std::vector<int> h(N); std::mt19937_64 rng(7);
        for (size_t i = 0; i < N; i++) h[i] = (int)(rng() % R);
        hipMemcpy(dIdx, h.data(), N * sizeof(int), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int width : {8, 4})
            for (int occ : {8, 16, 32})                    // wavefronts per CU
                for (int passes : {0, 4}) {
                    // 256-thread workgroups; LDS request limits the workgroups per CU
                    const int wg_per_cu = occ / 4;
                    const size_t lds = (size_t)(160 * 1024 / wg_per_cu) - 1024;
                    const dim3 grid(cus * wg_per_cu);
                    float best = 1e30f;
                    for (int rep = 0; rep < 4; rep++) {
                        hipEventRecord(e0);
                        if (width == 8) hipLaunchKernelGGL(gather_kernel<8>, grid, dim3(256), lds, 0, dB, (unsigned)(k * 8), k, dIdx, N / 64, sink, passes);
                        else hipLaunchKernelGGL(gather_kernel<4>, grid, dim3(256), lds, 0, dB, (unsigned)(k * 8), k, dIdx, N / 64, sink, passes);
                        hipEventRecord(e1); hipEventSynchronize(e1);
                        if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
                        float ms; hipEventElapsedTime(&ms, e0, e1);
                        if (rep > 0 && ms < best) best = ms;
                    }
                    printf("rows=%zu (%.0f MB) lanes/row=%d waves/CU=%d fma_passes=%d : %.3f ms  %.2f TB/s useful\n", R, R * k * 8 / 1e6, width, occ,
                           passes, best, (double)N * k * 8 / (best * 1e-3) / 1e12);
                }
        hipFree(dB); hipFree(dIdx);
    }
    return 0;
}
