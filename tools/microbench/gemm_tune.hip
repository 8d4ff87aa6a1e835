// Tuning harness of the tall dense contraction C[M, N] = A[M, K] B[K, N] (row-major, the w U C / I D products of the
// side-information path) on the matrix cores: variants of gemm_mfma_kernel timed on configuration 5's shape.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cmfrec_amd/csrc -o gemm_tune tools/microbench/gemm_tune.hip ; run: ./gemm_tune
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "dense_kernels.hpp"
using namespace cmfhip;

template <typename T>
static void run(int M, int N, int K)
{
    T *A, *B, *C;
    hipMalloc(&A, sizeof(T) * (size_t)M * K); hipMalloc(&B, sizeof(T) * (size_t)K * N); hipMalloc(&C, sizeof(T) * (size_t)M * N);
    std::vector<T> ha((size_t)M * K), hb((size_t)K * N);
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (T)((double)(x >> 11) / 9007199254740992.0 - 0.5); };
    for (auto &v : ha) v = rnd();
    for (auto &v : hb) v = rnd();
    hipMemcpy(A, ha.data(), sizeof(T) * ha.size(), hipMemcpyHostToDevice);
    hipMemcpy(B, hb.data(), sizeof(T) * hb.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid((N + GEMM_BN - 1) / GEMM_BN, (M + GEMM_BM - 1) / GEMM_BM, 1);
    auto timeit = [&](auto launch, const char *name) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; r++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        // spot check of 64 entries against a host dot product
        std::vector<T> hc((size_t)M * N);
        hipMemcpy(hc.data(), C, sizeof(T) * hc.size(), hipMemcpyDeviceToHost);
        double worst = 0;
        for (int s = 0; s < 64; s++) {
            const int i = (int)((size_t)s * 7919 * 13 % M), j = (s * 31) % N;
            double ref = 0;
            for (int kk = 0; kk < K; kk++) ref += (double)ha[(size_t)i * K + kk] * (double)hb[(size_t)kk * N + j];
            worst = std::max(worst, std::fabs(ref - (double)hc[(size_t)i * N + j]));
        }
        printf("%-44s %8.3f ms  %6.1f TFLOP/s   max abs err of 64 samples %.2e\n", name, ms, 2.0 * M * N * K / ms / 1e9, worst);
    };
    timeit([&] { hipLaunchKernelGGL((gemm_mfma_kernel<T, false>), grid, dim3(256), 0, 0, M, N, K, ((K + 15) / 16) * 16, (T)1, A, (size_t)K, B, (size_t)N, C, (size_t)N, (size_t)0); },
           sizeof(T) == 4 ? "f32 gemm_mfma_kernel<NN>" : "f64 gemm_mfma_kernel<NN>");
    hipFree(A); hipFree(B); hipFree(C);
}

int main()
{
    run<float>(1562500 / 4, 256, 512);
    run<double>(200000, 128, 64);
    return 0;
}
