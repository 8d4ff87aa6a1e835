// slot_probe.hip -- does a small launch on a second stream run BESIDE a persistent launch that leaves workgroup slots open?
// A spin kernel with 256 VGPRs per lane (two workgroups of 256 threads per CU, as gramk_producer_kernel) holds `grid` workgroups
// for ~5 ms; 50 dependent tiny launches go to another stream right behind it.  Printed: when the tiny chain finished relative to
// the spin kernel.      hipcc --offload-arch=gfx950 -O3 -o slot_probe slot_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef NACC
#define NACC 250
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256, 2) spin_kernel(float *out, long long ticks)
{
    // NACC live accumulators keep the register count at the bound
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (float)(threadIdx.x + i);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = acc[i] * 1.0001f + 0.5f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
    if (s == 12345.678f) out[0] = s;
}
__global__ void tiny_kernel(int *p) { if (threadIdx.x == 0) atomicAdd(p, 1); }
__global__ void __launch_bounds__(256) small_kernel(int *p, float *o)
{
    float a[48];
#pragma unroll
    for (int i = 0; i < 48; i++) a[i] = (float)(threadIdx.x * i);
    for (int r = 0; r < 20; r++)
#pragma unroll
        for (int i = 0; i < 48; i++) a[i] = a[i] * 1.01f + 1.f;
    float s = 0;
#pragma unroll
    for (int i = 0; i < 48; i++) s += a[i];
    if (s == 1.2345f) o[0] = s;
    if (threadIdx.x == 0) atomicAdd(p, 1);
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    float *out; int *cnt; CK(hipMalloc(&out, 4)); CK(hipMalloc(&cnt, 4)); CK(hipMemset(cnt, 0, 4));
    hipEvent_t a0, a1, b0, b1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    const long long ticks = 500000;     // 100 MHz wall clock: 5 ms
    // warm-up
    hipLaunchKernelGGL(spin_kernel, dim3(8), dim3(256), 0, sa, out, 1000LL);
    hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, sb, cnt);
    CK(hipDeviceSynchronize());
    printf("%d CUs; spin kernel holds its workgroups 5 ms; 50 dependent tiny launches on a second stream behind it\n", cus);
    const int grids[] = {2 * cus, 2 * cus - 8, 2 * cus - 16, 2 * cus - 32, 2 * cus - 64, cus + cus / 2, cus, cus / 2};
    for (int small = 0; small < 2; small++)
    for (int g : grids) {
        CK(hipEventRecord(a0, sa));
        hipLaunchKernelGGL(spin_kernel, dim3(g), dim3(256), 0, sa, out, ticks);
        CK(hipEventRecord(a1, sa));
        CK(hipEventRecord(b0, sb));
        for (int i = 0; i < 50; i++) {
            if (small) hipLaunchKernelGGL(small_kernel, dim3(4), dim3(256), 0, sb, cnt, out);
            else hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, sb, cnt);
        }
        CK(hipEventRecord(b1, sb));
        CK(hipDeviceSynchronize());
        float ta, tb, tab; CK(hipEventElapsedTime(&ta, a0, a1)); CK(hipEventElapsedTime(&tb, b0, b1)); CK(hipEventElapsedTime(&tab, a0, b1));
        printf("%s grid %4d (%+4d of 2 x CUs): spin %.3f ms, tiny chain %.3f ms, chain done %.3f ms after the spin kernel's start -> %s\n", small ? "4 x 256 threads, ~50 VGPRs:" : "1 x 64 threads:", g,
               g - 2 * cus, ta, tb, tab, tab < 0.8f * ta ? "beside" : "behind");
    }
    return 0;
}
