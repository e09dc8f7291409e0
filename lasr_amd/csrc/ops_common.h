// ops_common.h -- helpers shared by the operator translation units (ops.hip, fused.hip).
#pragma once
#include "host_common.h"
#include "sr_device.h"

namespace lasr {

// Sum over a 256-thread block; every thread gets the total.  `red` = 4 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red)
{
    v = wave_sum_to_lane63(v);
    __syncthreads();                       // protect `red` from the previous use
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace lasr

// Launch on `st` (a hipStream_t in scope), bracketed by profiling events when lasr_prof_enable(1) is in effect.
#define LASR_LAUNCH(ID, KERNEL, GRID, BLOCK, LDS, ...)                                   \
    do {                                                                                 \
        ProfScope ps_(ID, st);                                                           \
        hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, st, __VA_ARGS__);                   \
    } while (0)
