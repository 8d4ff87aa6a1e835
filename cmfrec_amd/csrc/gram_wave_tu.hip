// gram_wave_tu.hip -- the split rows' slice kernel (gram_wave_kernel, gram_cg_kernels.hpp) in a translation unit of its own,
// because it wants another instruction scheduler than the row kernels: it is a stream of matrix-pipe instructions with the
// next group's loads and ~30 vector instructions per slab to be placed between them, and the back end's ILP-first strategy
// (-mllvm -amdgpu-sched-strategy=max-ilp, set for this file's double-precision object in the Makefile) does that better than
// the occupancy-first default: 0.681 -> 0.634 ms for the item step's split rows of C2, while the same flag on the whole library
// makes the 33..64 bin's kernel 20 % slower (profiles/r03/r03_bj_bench_lines.txt).  Same sources, same arithmetic.
#include <hip/hip_runtime.h>
#include "../../include/cmfrec_hip.h"
#include "gram_cg_kernels.hpp"

namespace cmfhip {

// launches gram_wave_kernel<real_t, implicit, rem> on `st`; rem = live columns of the last column block that go through the
// vector ALU (double precision, 48 < k <= 52), 0 = all blocks on the matrix pipe
void launch_gram_wave(dim3 grid, hipStream_t st, const CgParams<real_t> &P, const GramParams<real_t> &G, bool implicit, int rem)
{
#define CMF_GW(IMPL, R) hipLaunchKernelGGL((gram_wave_kernel<real_t, IMPL, R>), grid, dim3(256), 0, st, P, G)
    if (rem == 0 && P.k == 16 * GRAM_NTT) {           // every column block complete (k = 64): compile-time column offsets
        if (implicit) hipLaunchKernelGGL((gram_wave_kernel<real_t, true, 0, true>), grid, dim3(256), 0, st, P, G);
        else hipLaunchKernelGGL((gram_wave_kernel<real_t, false, 0, true>), grid, dim3(256), 0, st, P, G);
        return;
    }
    if (implicit) {
        switch (rem) {
            case 1: CMF_GW(true, 1); break;
            case 2: CMF_GW(true, 2); break;
            case 3: CMF_GW(true, 3); break;
            case 4: CMF_GW(true, 4); break;
            default: CMF_GW(true, 0); break;
        }
    } else {
        switch (rem) {
            case 1: CMF_GW(false, 1); break;
            case 2: CMF_GW(false, 2); break;
            case 3: CMF_GW(false, 3); break;
            case 4: CMF_GW(false, 4); break;
            default: CMF_GW(false, 0); break;
        }
    }
#undef CMF_GW
}

}  // namespace cmfhip
