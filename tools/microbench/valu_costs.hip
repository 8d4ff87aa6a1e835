// Microbenchmark: issue cost (shader cycles per wave-instruction and SIMD) of the instruction kinds the CG row kernels are
// made of, on gfx950: f64 / f32 FMAs, DPP moves, permlane swaps, selects, LDS reads / writes / bpermute.  One block of 16
// independent instructions per asm statement, REP statements per loop trip; timed with s_memtime inside the kernel
// (ticks = shader cycles) for 1, 2 and 4 wavefronts per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_costs valu_costs.hip ; run: ./valu_costs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum Kind { FMA64, ADD64, MUL64, FMA32, PKFMA32, DPP_QP, DPP_ROW, PERM32, PERM16, CNDMASK, BPERM, LDS_R64, LDS_R128, LDS_R64B,
            LDS_W64, LDS_W128, READLANE, MOV32, FMA64_SGPR, DPP64_BCAST, LDS_R2_64, CNDMASK_E64, FMA64_LDS_MIX, NKINDS };
static const char *kind_name[NKINDS] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_fma_f32", "v_pk_fma_f32", "v_mov_b32 dpp quad_perm",
                                        "v_mov_b32 dpp row_shr:4", "v_permlane32_swap", "v_permlane16_swap", "v_cndmask_b32",
                                        "ds_bpermute_b32", "ds_read_b64 (lane-linear)", "ds_read_b128 (lane-linear)",
                                        "ds_read_b64 (8-lane broadcast)", "ds_write_b64", "ds_write_b128", "v_readlane_b32",
                                        "v_mov_b32", "v_fma_f64 (sgpr operand)", "v_mov_b64_dpp row_newbcast", "ds_read2_b64 (counted as 1)",
                                        "v_cndmask_b32_e64 (sgpr mask)", "v_fma_f64 + ds_read_b64 1:1 (per pair)"};

template <int KIND>
__global__ void __launch_bounds__(256) k_cost(unsigned long long *ticks, double *sink, int iters)
{
    __shared__ __attribute__((aligned(16))) double lds[4096];
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 4096; e += 256) lds[e] = e * 0.5;
    __syncthreads();
    double d[16];
    float f[32];
    int iv[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { d[i] = 1.0 + lane * 1e-3 + i; iv[i] = lane * 17 + i; }
#pragma unroll
    for (int i = 0; i < 32; i++) f[i] = 1.0f + lane * 1e-3f + i;
    double a = 1.0 + 1e-9 * lane, b = 1e-3;
    float af = 1.0f + 1e-6f * lane, bf = 1e-3f;
    const unsigned addr_lin8 = (threadIdx.x * 8) & 0x7ff8;          // 8 B per lane, lane-linear
    const unsigned addr_lin16 = (threadIdx.x * 16) & 0x7ff0;
    const unsigned addr_b8 = ((threadIdx.x >> 3) * 8 * 57) & 0x7ff8;   // same address inside an 8-lane group
    const unsigned bp_addr = ((lane ^ 9) * 4);
    double dd[16][2];
#pragma unroll
    for (int i = 0; i < 16; i++) { dd[i][0] = i; dd[i][1] = -i; }
    const double sa = __builtin_amdgcn_readfirstlane((int)(iters & 7)) * 1e-3 + 1.0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
            if constexpr (KIND == FMA64) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(a), "v"(b));
                R16(X)
#undef X
            } else if constexpr (KIND == FMA64_SGPR) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "s"(sa), "v"(b));
                R16(X)
#undef X
            } else if constexpr (KIND == ADD64) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(b));
                R16(X)
#undef X
            } else if constexpr (KIND == MUL64) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(a));
                R16(X)
#undef X
            } else if constexpr (KIND == FMA32) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(af), "v"(bf));
                R16(X)
#undef X
            } else if constexpr (KIND == PKFMA32) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[i]) : "v"(a), "v"(b));
                R16(X)
#undef X
            } else if constexpr (KIND == DPP_QP) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(iv[i]) : "v"(iv[(i + 1) & 15]));
                R16(X)
#undef X
            } else if constexpr (KIND == DPP_ROW) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(iv[i]) : "v"(iv[(i + 1) & 15]));
                R16(X)
#undef X
            } else if constexpr (KIND == PERM32) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(iv[i]), "+v"(iv[(i + 8) & 15]));
                X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
            } else if constexpr (KIND == PERM16) {
#define X(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(iv[i]), "+v"(iv[(i + 8) & 15]));
                X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
            } else if constexpr (KIND == CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iv[i]) : "v"(iv[(i + 1) & 15]) : );
                R16(X)
#undef X
            } else if constexpr (KIND == DPP64_BCAST) {
#define X(i) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0x3" : "+v"(d[i]) : "v"(d[(i + 1) & 15]));
                R16(X)
#undef X
            } else if constexpr (KIND == CNDMASK_E64) {
                const unsigned long long msk = 0x5555aaaa3333ccccull ^ (unsigned long long)iters;
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(iv[i]) : "v"(iv[(i + 1) & 15]), "v"(iv[(i + 2) & 15]), "s"(msk));
                R16(X)
#undef X
            } else if constexpr (KIND == LDS_R2_64) {
#define X(i) asm volatile("ds_read2_b64 %0, %1 offset0:" #i "*2 offset1:" #i "*2+8" : "=v"(*(double __attribute__((ext_vector_type(2))) *)dd[i]) : "v"(addr_lin8));
                R16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if constexpr (KIND == FMA64_LDS_MIX) {
                double e[16];
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:" #i "*8" : "=v"(e[i]) : "v"(addr_b8));
                R16(X)
#undef X
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(a), "v"(b));
                R16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
#define X(i) asm volatile("" ::"v"(e[i]));
                R16(X)
#undef X
            } else if constexpr (KIND == MOV32) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(iv[i]) : "v"(iv[(i + 1) & 15]));
                R16(X)
#undef X
            } else if constexpr (KIND == READLANE) {
                int s[16];
#define X(i) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s[i]) : "v"(iv[i]));
                R16(X)
#undef X
#define X(i) asm volatile("" ::"s"(s[i]));
                R16(X)
#undef X
            } else if constexpr (KIND == BPERM) {
#define X(i) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(iv[i]) : "v"(bp_addr), "v"(iv[(i + 1) & 15]));
                R16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if constexpr (KIND == LDS_R64) {
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:" #i "*8" : "=v"(d[i]) : "v"(addr_lin8));
                R16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if constexpr (KIND == LDS_R64B) {
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:" #i "*8" : "=v"(d[i]) : "v"(addr_b8));
                R16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if constexpr (KIND == LDS_R128) {
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:" #i "*16" : "=v"(*(double __attribute__((ext_vector_type(2))) *)dd[i]) : "v"(addr_lin16));
                R16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if constexpr (KIND == LDS_W64) {
#define X(i) asm volatile("ds_write_b64 %0, %1 offset:" #i "*8" : : "v"(addr_lin8), "v"(d[i]));
                R16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if constexpr (KIND == LDS_W128) {
#define X(i) asm volatile("ds_write_b128 %0, %1 offset:" #i "*16" : : "v"(addr_lin16), "v"(*(double __attribute__((ext_vector_type(2))) *)dd[i]));
                R16(X)
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)");
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += d[i] + iv[i] + f[i] + f[i + 16] + dd[i][0] + dd[i][1];
    sink[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
    if (lane == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run_kind(unsigned long long *d_t, double *d_s, int ncu)
{
    const int iters = 2000;
    for (int wps : {1, 2, 4}) {
        const int grid = ncu * wps;
        hipLaunchKernelGGL(k_cost<KIND>, dim3(grid), dim3(256), 0, 0, d_t, d_s, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_cost<KIND>, dim3(grid), dim3(256), 0, 0, d_t, d_s, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long h[4096 * 4];
        hipMemcpy(h, d_t, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost);
        double sum = 0;
        for (int i = 0; i < grid * 4; i++) sum += (double)h[i];
        const double per_wave = sum / (grid * 4) / ((double)iters * 64);        // ticks per instruction as one wave sees it
        printf("%-32s %d waves/SIMD: %7.2f ticks/instr/wave -> %6.2f ticks per instr per SIMD   (%.3f ms)\n", kind_name[KIND], wps, per_wave,
               per_wave / wps, ms);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    printf("device %s, CUs %d (s_memtime ticks at 100 MHz if constant clock; compare with the ms column)\n", prop.name, ncu);
    unsigned long long *d_t; double *d_s;
    hipMalloc(&d_t, sizeof(unsigned long long) * 4096 * 4);
    hipMalloc(&d_s, sizeof(double) * 4096 * 256);
    run_kind<FMA64>(d_t, d_s, ncu);
    run_kind<FMA64_SGPR>(d_t, d_s, ncu);
    run_kind<ADD64>(d_t, d_s, ncu);
    run_kind<MUL64>(d_t, d_s, ncu);
    run_kind<FMA32>(d_t, d_s, ncu);
    run_kind<PKFMA32>(d_t, d_s, ncu);
    run_kind<MOV32>(d_t, d_s, ncu);
    run_kind<DPP_QP>(d_t, d_s, ncu);
    run_kind<DPP_ROW>(d_t, d_s, ncu);
    run_kind<PERM32>(d_t, d_s, ncu);
    run_kind<PERM16>(d_t, d_s, ncu);
    run_kind<CNDMASK>(d_t, d_s, ncu);
    run_kind<READLANE>(d_t, d_s, ncu);
    run_kind<BPERM>(d_t, d_s, ncu);
    run_kind<LDS_R64>(d_t, d_s, ncu);
    run_kind<LDS_R64B>(d_t, d_s, ncu);
    run_kind<LDS_R128>(d_t, d_s, ncu);
    run_kind<LDS_W64>(d_t, d_s, ncu);
    run_kind<LDS_W128>(d_t, d_s, ncu);
    run_kind<DPP64_BCAST>(d_t, d_s, ncu);
    run_kind<CNDMASK_E64>(d_t, d_s, ncu);
    run_kind<LDS_R2_64>(d_t, d_s, ncu);
    run_kind<FMA64_LDS_MIX>(d_t, d_s, ncu);
    return 0;
}
