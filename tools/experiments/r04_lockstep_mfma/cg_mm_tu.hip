// cg_mm_tu.hip -- the lock-step short-row kernels with the Gramian product on the matrix pipe (cg_mm_kernels.hpp), in a
// translation unit of their own (they are independent of the other row kernels' instantiations and compile in seconds).
#include <hip/hip_runtime.h>
#include <algorithm>
#include "../../include/cmfrec_hip.h"
#include "cg_mm_kernels.hpp"

namespace cmfhip {

// implicit model, one row per wavefront, sixteen rows per workgroup; `P.desc` / `P.nrows` / `P.counter` already point at the
// rows of this launch.  Returns false when there is no instantiation for the width (k > 64).
bool launch_cg_tiny_mm(int num_cus, hipStream_t st, const CgParams<real_t> &P)
{
    const int S = (P.k + 7) / 8;
#ifndef CMF_MM_NW
#define CMF_MM_NW 16
#endif
    constexpr int NW = CMF_MM_NW;
    const int nbatches = (P.nrows + NW - 1) / NW;
    if (nbatches <= 0) return true;
    const dim3 grid(std::min(nbatches, num_cus * (16 / NW))), block(64 * NW);
#define CMF_MM(SS) case SS: hipLaunchKernelGGL((cg_rows_tiny_mm_kernel<real_t, SS, NW>), grid, block, (mm_smem_bytes<real_t, NW>()), st, P); return true;
    switch (S) {
        CMF_MM(1) CMF_MM(2) CMF_MM(3) CMF_MM(4) CMF_MM(5) CMF_MM(6) CMF_MM(7) CMF_MM(8)
    }
#undef CMF_MM
    return false;
}

}  // namespace cmfhip
