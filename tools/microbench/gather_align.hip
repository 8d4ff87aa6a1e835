// Does the alignment / leading dimension of the gathered rows bound the CG kernels' gather?  Rows of k = 50 doubles at random
// positions of a matrix with leading dimension ld in {50 (400 B, 16-byte aligned: the layout today), 52, 56 (448 B, 64-byte
// aligned), 64 (512 B, 128-byte aligned)}; 64-row register tiles, 8 lanes x 8 B per row segment exactly as load_tile does,
// 8 wavefronts per CU.  Prints useful TB/s (rows x 400 bytes) -- the same quantity as gather_rate.hip.
//   hipcc --offload-arch=gfx950 -O3 gather_align.hip -o gather_align && ./gather_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

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
    }
    if (acc == 1.2345e300) sink[0] = acc + pad[0];
}

int main()
{
    const int k = 50;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double *sink; hipMalloc(&sink, 64);
    hipFuncSetAttribute(reinterpret_cast<const void *>(gather_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (size_t R : {(size_t)160112, (size_t)358868, (size_t)8000000}) {
        const size_t N = 64 * 65536 * 2;
        int *dIdx; hipMalloc(&dIdx, N * sizeof(int));
        std::vector<int> h(N); std::mt19937_64 rng(7);
        for (size_t i = 0; i < N; i++) h[i] = (int)(rng() % R);
        hipMemcpy(dIdx, h.data(), N * sizeof(int), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int ld : {50, 52, 56, 64}) {
            double *dB; hipMalloc(&dB, R * ld * sizeof(double) + 512); hipMemset(dB, 0, R * ld * sizeof(double) + 512);
            for (int passes : {1, 4}) {
                const int wg_per_cu = 2;
                const size_t lds = (size_t)(160 * 1024 / wg_per_cu) - 1024;
                const dim3 grid(cus * wg_per_cu);
                float best = 1e30f;
                for (int rep = 0; rep < 5; rep++) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(gather_kernel, grid, dim3(256), lds, 0, dB, (unsigned)(ld * 8), k, dIdx, N / 64, sink, passes);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep > 0 && ms < best) best = ms;
                }
                printf("rows=%zu (%.0f MB at ld=%d) fma_passes=%d : %.3f ms  %.2f TB/s useful\n", R, R * ld * 8 / 1e6, ld, passes, best,
                       (double)N * k * 8 / (best * 1e-3) / 1e12);
            }
            hipFree(dB);
        }
        hipFree(dIdx);
    }
    return 0;
}
