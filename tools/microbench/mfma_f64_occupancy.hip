// Microbenchmark (round 5): v_mfma_f64_16x16x4 rate against the wavefronts per SIMD that issue them, NACC independent accumulators
// per wavefront, the whole chip busy.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_occupancy mfma_f64_occupancy.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int WPS>
__global__ void __launch_bounds__(256, WPS) k_f64(double *out, int iters)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        a += 1e-9;
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    double *o; hipMalloc(&o, (size_t)8 * ncu * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, int nacc, int wps, int iters) {
        hipLaunchKernelGGL(kern, dim3(wps * ncu), dim3(256), 0, 0, o, iters); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(wps * ncu), dim3(256), 0, 0, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (double)nacc * iters * wps * 4 * ncu;
        printf("f64 16x16x4  %2d accumulators  %d waves/SIMD  %8.3f ms  %6.1f TFLOP/s  %6.1f ns per MFMA and SIMD\n", nacc, wps, ms,
               mfma * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)nacc * iters * wps));
    };
    const int iters = 40000;
    run(k_f64<9, 1>, 9, 1, iters); run(k_f64<9, 2>, 9, 2, iters); run(k_f64<9, 3>, 9, 3, iters); run(k_f64<9, 4>, 9, 4, iters);
    run(k_f64<32, 1>, 32, 1, iters / 2); run(k_f64<18, 2>, 18, 2, iters / 2); run(k_f64<4, 4>, 4, 4, iters); run(k_f64<4, 8>, 4, 8, iters);
    return 0;
}
