// chol_wg_tu.hip -- the four-wavefront factorisation of the eight-block closed-form rows (chol_wg_kernels.hpp) in a translation unit
// of its own: the kernel is four instantiations of a long row function, and a unit of its own keeps it out of session.hip's
// four-minute compile (an experiment on it rebuilds in seconds).
#include <hip/hip_runtime.h>
#include <algorithm>
#include "../../include/cmfrec_hip.h"
#include "chol_wg_kernels.hpp"
#include "chol_parts_coop_kernels.hpp"

namespace cmfhip {

// launches chol_wg8_kernel<real_t, border> over the rows [W.row_first, W.nrows) of the processing order on `st`; returns the HIP status
// of the launch.  Double precision only (the single-precision build keeps its own kernels for these widths).
hipError_t launch_chol_wg8(int num_cus, bool border, int waves_per_row, hipStream_t st, const CholParams<real_t> &W, const RowDesc *desc, const CholSlices<real_t> &SL)
{
#ifdef CMFREC_HIP_FLOAT
    (void)num_cus; (void)border; (void)waves_per_row; (void)st; (void)W; (void)desc; (void)SL;
    return hipErrorNotSupported;
#else
    // CMFREC_HIP_CHOL_WG: 1 / 4 = four wavefronts per row (two rows per CU), 2 = two wavefronts per row (four rows per CU)
    const int nw = (waves_per_row == 4) ? 4 : 2;
    auto kern = (nw == 4) ? (border ? chol_wg8_kernel<real_t, true, 4> : chol_wg8_kernel<real_t, false, 4>)
                          : (border ? chol_wg8_kernel<real_t, true, 2> : chol_wg8_kernel<real_t, false, 2>);
    static thread_local int bpc[2][2] = {{0, 0}, {0, 0}};
    int &blocks_per_cu = bpc[nw == 4 ? 1 : 0][border ? 1 : 0];
    if (blocks_per_cu == 0) {
        int nb = 0;
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 64 * nw, 0);
        if (e != hipSuccess) return e;
        blocks_per_cu = std::max(1, nb);
    }
    const int rows = W.nrows - W.row_first;
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(std::min(rows, num_cus * blocks_per_cu)), dim3(64 * nw), 0, st, W, desc, SL);
    return hipGetLastError();
#endif
}

// launches chol_parts_coop_kernel<real_t, border> over the work items [W.row_first, W.nrows) on `st` (chol_parts_coop_kernels.hpp)
hipError_t launch_chol_parts_coop(int num_cus, bool border, int depth, hipStream_t st, const CholParams<real_t> &W, const RowDesc *desc, const CholSlices<real_t> &SL)
{
#ifdef CMFREC_HIP_FLOAT
    (void)num_cus; (void)border; (void)depth; (void)st; (void)W; (void)desc; (void)SL;
    return hipErrorNotSupported;
#else
    // depth: steps of four entries in flight in registers (3: no spilled register in either build; 4: the build with the border column spills)
    // explicit-feedback weights (every mode but the implicit model's): the build without per-step weights and selects
    const bool expl = !(W.mode == CHOL_IMPLICIT || W.mode == CHOL_COLLECTIVE_IMPLICIT);
    auto kern = expl ? ((depth == 4) ? (border ? chol_parts_coop_kernel<real_t, true, 4, true> : chol_parts_coop_kernel<real_t, false, 4, true>)
                                     : (border ? chol_parts_coop_kernel<real_t, true, 3, true> : chol_parts_coop_kernel<real_t, false, 3, true>))
                     : ((depth == 4) ? (border ? chol_parts_coop_kernel<real_t, true, 4, false> : chol_parts_coop_kernel<real_t, false, 4, false>)
                                     : (border ? chol_parts_coop_kernel<real_t, true, 3, false> : chol_parts_coop_kernel<real_t, false, 3, false>));
    const int items = W.nrows - W.row_first;
    if (items <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(std::min(items, num_cus * 4)), dim3(128), 0, st, W, desc, SL);
    return hipGetLastError();
#endif
}

}  // namespace cmfhip
