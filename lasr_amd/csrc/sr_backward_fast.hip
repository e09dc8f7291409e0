// sr_backward_fast.hip -- backward raster kernel for LASR's mode combination (euclidean / softmax / prod / vertex attributes /
// double sided), 3, 6 and 9 attribute channels.  Same template as the generic instantiation in sr_raster.hip (sr_backward.h),
// but THIS translation unit is compiled with -ffp-contract=fast-honor-pragmas (see the Makefile): the backward pass is VALU-issue bound and
// its gradient bar is relative 1e-3, so v_mul_f32 + v_add_f32 pairs fuse into v_fma_f32.
#include <hip/hip_runtime.h>

#include "sr_backward.h"

namespace lasr {

void launch_backward_fast(int nch, dim3 grid, hipStream_t st, const RasterArgs& A, const float* colors, const float* aggrs,
                          const float* gcolors, float* gfaces, float* gtex)
{
    if (nch == 9)
        hipLaunchKernelGGL((sr_backward_kernel<true, 9>), grid, dim3(BWD_THREADS), 0, st, A, colors, aggrs, gcolors, gfaces, gtex);
    else if (nch == 6)
        hipLaunchKernelGGL((sr_backward_kernel<true, 6>), grid, dim3(BWD_THREADS), 0, st, A, colors, aggrs, gcolors, gfaces, gtex);
    else
        hipLaunchKernelGGL((sr_backward_kernel<true, 3>), grid, dim3(BWD_THREADS), 0, st, A, colors, aggrs, gcolors, gfaces, gtex);
}

}  // namespace lasr
