// What bounds the gather of 400-byte rows at ~6.0 TB/s (tools/microbench/gather_rate.hip)?  If it is the number of cache lines a CU's
// vector L1 can have in flight (a 64-row tile touches 256 lines = the whole 32 KB L1), loads that do not allocate there -- the
// cache-policy bits of gfx940+: sc0 / sc1 / nt -- or wider loads may move the ceiling.  Same access pattern as load_tile (8 lanes x
// 8 B per 64-byte row segment, 56 loads per 64-row tile), the load instruction issued with each policy.
//   hipcc --offload-arch=gfx950 -O3 gather_policy.hip -o gather_policy && ./gather_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

#define LOAD2(POL) asm volatile("global_load_dwordx2 %0, %1, off " POL : "=v"(v[t][s]) : "v"(rp + (s < 6 ? 8 * s : col_last)) : "memory")

template <int POLICY>
__global__ void __launch_bounds__(256) gather_kernel(const double *__restrict__ B, unsigned ldb_bytes, int k, const int *__restrict__ idx,
                                                     size_t ntiles, double *__restrict__ sink)
{
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
            for (int s = 0; s < 7; s++) {
                if (POLICY == 0) LOAD2("");
                else if (POLICY == 1) LOAD2("nt");
                else if (POLICY == 2) LOAD2("sc0");
                else if (POLICY == 3) LOAD2("sc1");
                else if (POLICY == 4) LOAD2("sc0 sc1");
                else if (POLICY == 5) LOAD2("sc0 nt");
                else if (POLICY == 6) LOAD2("sc1 nt");
                else LOAD2("sc0 sc1 nt");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int s = 0; s < 7; s++) acc = fma(v[t][s], acc, v[t][s]);
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

template <int POLICY>
static void run(const char *name, const double *dB, int k, const int *dIdx, size_t N, double *sink, int cus, size_t R)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wg_per_cu : {2, 4}) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(gather_kernel<POLICY>, dim3(cus * wg_per_cu), dim3(256), 0, 0, dB, (unsigned)(k * 8), k, dIdx, N / 64, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("rows=%zu policy %-12s waves/CU=%2d : %.3f ms  %.2f TB/s useful\n", R, name, wg_per_cu * 4, best, (double)N * k * 8 / (best * 1e-3) / 1e12);
    }
}

int main()
{
    const int k = 50;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double *sink; hipMalloc(&sink, 64);
    for (size_t R : {(size_t)160112, (size_t)358868, (size_t)8000000}) {
        const size_t N = 64 * 65536 * 2;
        int *dIdx; hipMalloc(&dIdx, N * sizeof(int));
        std::vector<int> h(N); std::mt19937_64 rng(7);
        for (size_t i = 0; i < N; i++) h[i] = (int)(rng() % R);
        hipMemcpy(dIdx, h.data(), N * sizeof(int), hipMemcpyHostToDevice);
        double *dB; hipMalloc(&dB, R * k * sizeof(double) + 512); hipMemset(dB, 0, R * k * sizeof(double) + 512);
        run<0>("default", dB, k, dIdx, N, sink, cus, R);
        run<1>("nt", dB, k, dIdx, N, sink, cus, R);
        run<2>("sc0", dB, k, dIdx, N, sink, cus, R);
        run<3>("sc1", dB, k, dIdx, N, sink, cus, R);
        run<4>("sc0 sc1", dB, k, dIdx, N, sink, cus, R);
        run<5>("sc0 nt", dB, k, dIdx, N, sink, cus, R);
        run<6>("sc1 nt", dB, k, dIdx, N, sink, cus, R);
        run<7>("sc0 sc1 nt", dB, k, dIdx, N, sink, cus, R);
        hipFree(dB); hipFree(dIdx);
    }
    return 0;
}
