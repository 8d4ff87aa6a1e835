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
    const int nbatches = (P.nrows + MM_ROWS - 1) / MM_ROWS;
    if (nbatches <= 0) return true;
    const dim3 grid(std::min(nbatches, num_cus)), block(64 * MM_ROWS);
#define CMF_MM(SS) case SS: hipLaunchKernelGGL((cg_rows_tiny_mm_kernel<real_t, SS>), grid, block, (mm_smem_bytes<real_t, MM_ROWS>()), st, P); return true;
    switch (S) {
        CMF_MM(1) CMF_MM(2) CMF_MM(3) CMF_MM(4) CMF_MM(5) CMF_MM(6) CMF_MM(7) CMF_MM(8)
    }
#undef CMF_MM
    return false;
}

}  // namespace cmfhip
