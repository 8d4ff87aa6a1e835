// Timing of the symmetric eigen-decomposition needed by the low-rank path (n = 256): rocSOLVER syevd / syevj against the
// one-workgroup Jacobi kernel of lowrank_kernels.hpp.  hipcc --offload-arch=gfx950 -O3 -std=c++17 eig_bench.hip -lrocsolver -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
#include "../../cmfrec_amd/csrc/lowrank_kernels.hpp"
using namespace cmfhip;
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 256, p = argc > 2 ? atoi(argv[2]) : 512;
    std::mt19937 rng(1); std::normal_distribution<double> nd(0, 0.1);
    std::vector<double> C((size_t)p * n), A((size_t)n * n, 0.0);
    for (auto &v : C) v = nd(rng);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = 0; for (int r = 0; r < p; r++) s += C[(size_t)r * n + i] * C[(size_t)r * n + j]; A[(size_t)i * n + j] = 0.5 * s; }
    double *dA, *dW, *dV, *dQ, *dQt, *dL, *dD, *dE, *dA2; int *dinfo;
    hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&dA2, sizeof(double) * n * n); hipMalloc(&dW, sizeof(double) * n * n); hipMalloc(&dV, sizeof(double) * n * n);
    hipMalloc(&dQ, sizeof(double) * n * n); hipMalloc(&dQt, sizeof(double) * n * n); hipMalloc(&dL, sizeof(double) * n); hipMalloc(&dD, sizeof(double) * n);
    hipMalloc(&dE, sizeof(double) * n); hipMalloc(&dinfo, sizeof(int));
    hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    rocblas_handle h; rocblas_create_handle(&h);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto check = [&](const double *Qrm_host, const double *lam, const char *name) {
        double err = 0, orth = 0;
        for (int i = 0; i < n; i += 7) for (int j = 0; j < n; j += 5) {
            double s = 0, o = 0;
            for (int c = 0; c < n; c++) { s += Qrm_host[(size_t)i * n + c] * lam[c] * Qrm_host[(size_t)j * n + c]; o += Qrm_host[(size_t)i * n + c] * Qrm_host[(size_t)j * n + c]; }
            err = fmax(err, fabs(s - A[(size_t)i * n + j])); orth = fmax(orth, fabs(o - (i == j)));
        }
        printf("%s: |Q L Q^T - A| = %.2e, |Q Q^T - I| = %.2e\n", name, err, orth);
    };
    std::vector<double> Q((size_t)n * n), L(n);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(jacobi_eig_kernel<double>, dim3(1), dim3(1024), 0, 0, dA, n, dW, dV, dQ, dQt, (size_t)n, dL, 30, 1e-13);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("own jacobi (1 workgroup): %.3f ms\n", ms);
    }
    hipMemcpy(Q.data(), dQ, sizeof(double) * n * n, hipMemcpyDeviceToHost); hipMemcpy(L.data(), dL, sizeof(double) * n, hipMemcpyDeviceToHost);
    check(Q.data(), L.data(), "own jacobi");
    for (int rep = 0; rep < 3; rep++) {
        hipMemcpy(dA2, dA, sizeof(double) * n * n, hipMemcpyDeviceToDevice);
        hipEventRecord(e0);
        rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_upper, n, dA2, n, dD, dE, dinfo);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rocsolver_dsyevd: %.3f ms\n", ms);
    }
    {   // column-major eigenvectors: memory [c][i] -> Q_rm[i][c]
        std::vector<double> Vc((size_t)n * n); hipMemcpy(Vc.data(), dA2, sizeof(double) * n * n, hipMemcpyDeviceToHost); hipMemcpy(L.data(), dD, sizeof(double) * n, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; i++) for (int c = 0; c < n; c++) Q[(size_t)i * n + c] = Vc[(size_t)c * n + i];
        check(Q.data(), L.data(), "syevd");
    }
    for (int rep = 0; rep < 3; rep++) {
        hipMemcpy(dA2, dA, sizeof(double) * n * n, hipMemcpyDeviceToDevice);
        double *resid; int *nsweeps; hipMalloc(&resid, 8); hipMalloc(&nsweeps, 4);
        hipEventRecord(e0);
        rocsolver_dsyevj(h, rocblas_esort_none, rocblas_evect_original, rocblas_fill_upper, n, dA2, n, 1e-13, resid, 30, nsweeps, dD, dinfo);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int ns; hipMemcpy(&ns, nsweeps, 4, hipMemcpyDeviceToHost);
        printf("rocsolver_dsyevj: %.3f ms (%d sweeps)\n", ms, ns);
    }
    return 0;
}
