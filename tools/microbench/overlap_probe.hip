// Do a gather phase and an FP64 product phase of the same wavefront overlap across the wavefronts of a CU, or add up?
// Each wavefront loops over 64-row tiles: gather the tile into registers (load_tile's pattern, or constants when LOAD = 0),
// then PASSES x (phase 1 + phase 2 of tile_pass without the cross-lane part: 112 independent FMAs per pass on 8 + 7
// accumulators).  Reports time for load only, compute only and both, at 8 wavefronts per CU, plus the shader clock seen by
// the kernel (s_memtime ticks per 100 MHz s_memrealtime tick).
//   hipcc --offload-arch=gfx950 -O3 overlap_probe.hip -o overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>

template <int LOAD, int PASSES, int CNT = 64, int PRED = 0>
__global__ void __launch_bounds__(256, 2) probe_kernel(const double *__restrict__ B, unsigned ldb_bytes, int k, const int *__restrict__ idx,
                                                       size_t ntiles, double *__restrict__ sink, unsigned long long *__restrict__ clk)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int jj = lane >> 3, ll = lane & 7;
    const int col_last = min(ll + 48, k - 1) - ll;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    double res = 0;
    for (size_t tl = wave; tl < ntiles; tl += nw) {
        const int my = idx[tl * 64 + lane];
        double v[8][7];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const bool ok = (jj * 8 + t) < CNT;
            const unsigned it = (unsigned)__shfl(my, ok || PRED ? jj * 8 + t : 0);
            const double *rp = reinterpret_cast<const double *>(reinterpret_cast<const char *>(B + ll) + (unsigned long long)it * ldb_bytes);
#pragma unroll
            for (int s = 0; s < 7; s++) {
                if (PRED) v[t][s] = ok ? rp[s < 6 ? 8 * s : col_last] : 0.0;
                else v[t][s] = LOAD ? rp[s < 6 ? 8 * s : col_last] : (double)(it & 7) * 1e-3;
            }
        }
        double vrep[7];
#pragma unroll
        for (int s = 0; s < 7; s++) vrep[s] = 1e-3 * (s + lane);
#pragma unroll 1
        for (int p = 0; p < PASSES; p++) {
            double c[8], out[7];
#pragma unroll
            for (int t = 0; t < 8; t++) {
                double a = 0;
#pragma unroll
                for (int s = 0; s < 7; s++) a = fma(v[t][s], vrep[s], a);
                c[t] = a;
            }
#pragma unroll
            for (int s = 0; s < 7; s++) out[s] = 0;
#pragma unroll
            for (int t = 0; t < 8; t++)
#pragma unroll
                for (int s = 0; s < 7; s++) out[s] = fma(c[t], v[t][s], out[s]);
#pragma unroll
            for (int s = 0; s < 7; s++) vrep[s] = out[s] * 1e-3;
        }
#pragma unroll
        for (int s = 0; s < 7; s++) res += vrep[s];
#pragma unroll
        for (int t = 0; t < 8; t++) res += v[t][0];
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (res == 1.2345e300) sink[0] = res;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int LOAD, int PASSES, int CNT = 64, int PRED = 0>
static void run(const char *what, int cus, const double *dB, int k, const int *dIdx, size_t ntiles, double *sink, unsigned long long *dclk)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f; unsigned long long hc[2] = {0, 1};
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe_kernel<LOAD, PASSES, CNT, PRED>), dim3(cus * 2), dim3(256), 0, 0, dB, (unsigned)(k * 8), k, dIdx, ntiles, sink, dclk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) { best = ms; hipMemcpy(hc, dclk, 16, hipMemcpyDeviceToHost); }
    }
    printf("%-28s %.3f ms   shader clock %.0f MHz\n", what, best, 100.0 * (double)hc[0] / (double)hc[1]);
}

int main()
{
    const int k = 50;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double *sink; hipMalloc(&sink, 64);
    unsigned long long *dclk; hipMalloc(&dclk, 16);
    const size_t R = 358868, N = 64 * 65536 * 2;
    double *dB; int *dIdx;
    hipMalloc(&dB, R * k * sizeof(double) + 256); hipMalloc(&dIdx, N * sizeof(int));
    hipMemset(dB, 0, R * k * sizeof(double) + 256);
    std::vector<int> h(N); std::mt19937_64 rng(7);
    for (size_t i = 0; i < N; i++) h[i] = (int)(rng() % R);
    hipMemcpy(dIdx, h.data(), N * sizeof(int), hipMemcpyHostToDevice);
    run<1, 0>("gather only", cus, dB, k, dIdx, N / 64, sink, dclk);
    run<0, 4>("4 passes only", cus, dB, k, dIdx, N / 64, sink, dclk);
    run<1, 4>("gather + 4 passes", cus, dB, k, dIdx, N / 64, sink, dclk);
    run<0, 8>("8 passes only", cus, dB, k, dIdx, N / 64, sink, dclk);
    run<1, 8>("gather + 8 passes", cus, dB, k, dIdx, N / 64, sink, dclk);
    run<1, 4, 46, 0>("gather 46/64 padded + 4", cus, dB, k, dIdx, N / 64, sink, dclk);
    run<1, 4, 46, 1>("gather 46/64 masked + 4", cus, dB, k, dIdx, N / 64, sink, dclk);
    run<1, 4, 32, 0>("gather 32/64 padded + 4", cus, dB, k, dIdx, N / 64, sink, dclk);
    run<1, 4, 32, 1>("gather 32/64 masked + 4", cus, dB, k, dIdx, N / 64, sink, dclk);
    run<1, 4, 16, 0>("gather 16/64 padded + 4", cus, dB, k, dIdx, N / 64, sink, dclk);
    run<1, 4, 16, 1>("gather 16/64 masked + 4", cus, dB, k, dIdx, N / 64, sink, dclk);
    run<0, 16>("16 passes only", cus, dB, k, dIdx, N / 64, sink, dclk);
    run<1, 16>("gather + 16 passes", cus, dB, k, dIdx, N / 64, sink, dclk);
    return 0;
}
