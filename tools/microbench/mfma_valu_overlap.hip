// Microbenchmark (round 4): do v_mfma_{f64,f32}_16x16x4 and the vector ALU's FMAs of the same precision run SIDE BY SIDE on a
// SIMD of gfx950, or do they share the datapath?  Four kernels per precision, four wavefronts per SIMD, everything in registers:
//   valu   : every wavefront issues NV independent FMAs per iteration
//   mfma   : every wavefront issues NM independent MFMAs per iteration
//   same   : every wavefront issues both, interleaved in program order
//   split  : wavefronts 0, 1 of a SIMD issue 2 NV FMAs, wavefronts 2, 3 issue 2 NM MFMAs (same totals per SIMD)
// If the pipes were independent, same / split would take max(valu, mfma); if they share the datapath, valu + mfma.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip ; run: ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <typename T> struct Op;
template <> struct Op<double> {
    typedef d4 acc; typedef double val;
    static __device__ __forceinline__ acc mma(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ val fma(val x, val a, val b) { return x * a + b; }
    static __device__ __forceinline__ val init(int i) { return 1.0 + i; }
    static __device__ __forceinline__ double sum(val x) { return x; }
    static constexpr double flop_per_valu = 2.0 * 64, flop_per_mfma = 2.0 * 16 * 16 * 4;
};
template <> struct Op<float> {
    typedef f4 acc; typedef f2 val;      // packed FMAs: the vector ALU's single-precision peak
    static __device__ __forceinline__ acc mma(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ val fma(val x, val a, val b) { return x * a + b; }
    static __device__ __forceinline__ val init(int i) { return f2{1.0f + i, 2.0f + i}; }
    static __device__ __forceinline__ double sum(val x) { return x[0] + x[1]; }
    static constexpr double flop_per_valu = 4.0 * 64, flop_per_mfma = 2.0 * 16 * 16 * 4;
};

// MODE 0 valu, 1 mfma, 2 same, 3 split
template <typename T, int MODE, int NV, int NM>
__global__ void __launch_bounds__(1024, 4) k(double *out, int iters)
{
    typedef Op<T> O;
    typename O::val x[16];
    typename O::acc acc[4];
    for (int i = 0; i < 16; i++) x[i] = O::init(i + (threadIdx.x & 3));
    for (int i = 0; i < 4; i++) acc[i] = typename O::acc{0, 0, 0, 0};
    const typename O::val a = O::init(0) * (T)1e-3 + (T)1, b = O::init(1) * (T)1e-6;
    const T ma = (T)(threadIdx.x * 1e-3), mb = (T)(1.0 + threadIdx.x * 1e-4);
    const int wave = threadIdx.x >> 6;
    // wavefront w of a 16-wave workgroup sits on SIMD w % 4: waves {0..7} are the first two of every SIMD
    const bool do_v = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 8);
    const bool do_m = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 8);
    constexpr int RV = (MODE == 3) ? 2 : 1;
    for (int it = 0; it < iters; it++) {
        if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < NM; j++) {
                acc[j & 3] = O::mma(ma, mb, acc[j & 3]);
#pragma unroll
                for (int i = 0; i < NV / NM; i++) x[(j * (NV / NM) + i) & 15] = O::fma(x[(j * (NV / NM) + i) & 15], a, b);
            }
        } else {
            if (do_v) {
#pragma unroll
                for (int i = 0; i < RV * NV; i++) x[i & 15] = O::fma(x[i & 15], a, b);
            }
            if (do_m) {
#pragma unroll
                for (int j = 0; j < RV * NM; j++) acc[j & 3] = O::mma(ma, mb, acc[j & 3]);
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 16; i++) s += O::sum(x[i]);
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename T, int MODE, int NV, int NM>
float run(double *out, int cus, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<T, MODE, NV, NM>), dim3(cus), dim3(1024), 0, 0, out, iters / 10);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<T, MODE, NV, NM>), dim3(cus), dim3(1024), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <typename T, int NV, int NM>
void series(const char *name, double *out, int cus, int iters)
{
    typedef Op<T> O;
    const float v = run<T, 0, NV, NM>(out, cus, iters), m = run<T, 1, NV, NM>(out, cus, iters);
    const float s = run<T, 2, NV, NM>(out, cus, iters), p = run<T, 3, NV, NM>(out, cus, iters);
    const double waves = (double)cus * 16, per_it_v = NV * O::flop_per_valu, per_it_m = NM * O::flop_per_mfma;
    printf("%s  %d FMA + %d MFMA per wavefront and iteration, 4 wavefronts per SIMD, %d iterations\n", name, NV, NM, iters);
    printf("   valu only  %8.3f ms  (%.1f TFLOP/s)\n", v, waves * iters * per_it_v / (v * 1e9));
    printf("   mfma only  %8.3f ms  (%.1f TFLOP/s)\n", m, waves * iters * per_it_m / (m * 1e9));
    printf("   same wave  %8.3f ms  = %.2f x (valu + mfma), %.2f x max(valu, mfma)\n", s, s / (v + m), s / (v > m ? v : m));
    printf("   split      %8.3f ms  = %.2f x (valu + mfma), %.2f x max(valu, mfma)\n", p, p / (v + m), p / (v > m ? v : m));
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double *out;
    hipMalloc(&out, (size_t)cus * 1024 * sizeof(double));
    printf("device %s, %d CUs\n", prop.gcnArchName, cus);
    series<double, 64, 4>("fp64", out, cus, 20000);     // 64 FMAs ~ 4 MFMAs in pipe time if the rates are 78.6 / 78.6 TFLOP/s (x 16 flop ratio)
    series<double, 128, 4>("fp64", out, cus, 20000);
    series<float, 64, 8>("fp32 (v_pk_fma_f32)", out, cus, 20000);
    series<float, 128, 8>("fp32 (v_pk_fma_f32)", out, cus, 20000);
    return 0;
}
