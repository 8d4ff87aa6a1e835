// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 and LDS read latency on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k_f64(double *out, int iters)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k_f32(float *out, int iters)
{
    f4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// v_fma_f64 / v_fma_f32 issue rate: NACC independent chains per lane
template <typename T, int NACC>
__global__ void __launch_bounds__(256) k_fma(T *out, int iters)
{
    T acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = T(threadIdx.x + i);
    T a = T(1.0) + T(threadIdx.x) * T(1e-9), b = T(1e-3);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = acc[i] * a + b;
    }
    T s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// dependent chain of LDS reads: latency per ds_read_b64
__global__ void __launch_bounds__(64) k_lds(int *out, int iters)
{
    __shared__ int next[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) next[i] = (i * 17 + 5) & 1023;
    __syncthreads();
    int p = threadIdx.x;
    for (int it = 0; it < iters; it++) p = next[p];
    out[threadIdx.x] = p;
}
// dependent chain of global loads over a large buffer: latency per load
__global__ void __launch_bounds__(64) k_glb(const int *next, int *out, int iters)
{
    int p = threadIdx.x * 4099;
    for (int it = 0; it < iters; it++) p = next[p];
    out[threadIdx.x] = p;
}
int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    double clk = prop.clockRate * 1e3;   // Hz
    printf("device %s, CUs %d, clock %.0f MHz\n", prop.name, prop.multiProcessorCount, clk / 1e6);
    double *o; hipMalloc(&o, 4096 * 512 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto launch, const char *name, double mfma_per_wave, int waves_per_simd) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double cyc = ms * 1e-3 * clk;
        printf("%-28s %8.3f ms  -> %.1f cycles per MFMA per SIMD (%d waves/SIMD)\n", name, ms, cyc / (mfma_per_wave * waves_per_simd), waves_per_simd);
    };
    const int iters = 20000;
    int ncu = prop.multiProcessorCount;
    run([&] { hipLaunchKernelGGL(k_f64<8>, dim3(ncu), dim3(256), 0, 0, o, iters); }, "f64 16x16x4, 1 wave/SIMD", 8.0 * iters, 1);
    run([&] { hipLaunchKernelGGL(k_f64<8>, dim3(2 * ncu), dim3(256), 0, 0, o, iters); }, "f64 16x16x4, 2 waves/SIMD", 8.0 * iters, 2);
    run([&] { hipLaunchKernelGGL(k_f64<1>, dim3(ncu), dim3(256), 0, 0, o, iters); }, "f64 dependent chain", 1.0 * iters, 1);
    run([&] { hipLaunchKernelGGL(k_f32<8>, dim3(ncu), dim3(256), 0, 0, (float *)o, iters); }, "f32 16x16x4, 1 wave/SIMD", 8.0 * iters, 1);
    run([&] { hipLaunchKernelGGL(k_f32<1>, dim3(ncu), dim3(256), 0, 0, (float *)o, iters); }, "f32 dependent chain", 1.0 * iters, 1);
    auto runf = [&](auto launch, const char *name, double fma_per_wave, int waves_per_simd) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double cyc = ms * 1e-3 * clk;
        printf("%-28s %8.3f ms  -> %.2f cycles per wave-FMA per SIMD (%d waves/SIMD) = %.1f TFLOP/s\n", name, ms,
               cyc / (fma_per_wave * waves_per_simd), waves_per_simd, fma_per_wave * waves_per_simd * 4 * ncu * 128 / (ms * 1e-3) / 1e12);
    };
    runf([&] { hipLaunchKernelGGL((k_fma<double, 16>), dim3(ncu), dim3(256), 0, 0, o, iters); }, "v_fma_f64, 1 wave/SIMD", 16.0 * iters, 1);
    runf([&] { hipLaunchKernelGGL((k_fma<double, 16>), dim3(4 * ncu), dim3(256), 0, 0, o, iters); }, "v_fma_f64, 4 waves/SIMD", 16.0 * iters, 4);
    runf([&] { hipLaunchKernelGGL((k_fma<double, 1>), dim3(ncu), dim3(256), 0, 0, o, iters); }, "v_fma_f64 dependent", 1.0 * iters, 1);
    runf([&] { hipLaunchKernelGGL((k_fma<float, 16>), dim3(ncu), dim3(256), 0, 0, (float *)o, iters); }, "v_fma_f32, 1 wave/SIMD", 16.0 * iters, 1);
    runf([&] { hipLaunchKernelGGL((k_fma<float, 16>), dim3(4 * ncu), dim3(256), 0, 0, (float *)o, iters); }, "v_fma_f32, 4 waves/SIMD", 16.0 * iters, 4);
    {
        int *oi = (int *)o;
        hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, oi, 100000); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, oi, 100000); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("LDS dependent read latency: %.1f cycles (%.1f ns)\n", ms * 1e-3 * clk / 100000, ms * 1e6 / 100000);
    }
    for (size_t mb : {1, 16, 128, 1024}) {
        size_t n = mb * 1024 * 1024 / 4;
        int *h = (int *)malloc(n * 4);
        // random cyclic-ish permutation walk
        unsigned long long x = 88172645463325252ull;
        for (size_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (int)(x % n); }
        int *d; hipMalloc(&d, n * 4); hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
        int *oi = (int *)o; const int it2 = 20000;
        hipLaunchKernelGGL(k_glb, dim3(1), dim3(64), 0, 0, d, oi, 100); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k_glb, dim3(1), dim3(64), 0, 0, d, oi, it2); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("global dependent load latency, %4zu MB footprint: %.0f ns\n", mb, ms * 1e6 / it2);
        hipFree(d); free(h);
    }
    return 0;
}
