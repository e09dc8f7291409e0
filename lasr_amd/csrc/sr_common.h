// sr_common.h -- what the raster translation units share: the kernel argument block and the block -> work mapping.
#pragma once
#include <hip/hip_runtime.h>

#include "sr_device.h"

namespace lasr {

struct RasterArgs {
    const float* __restrict__ recs;      // [N*F, REC]
    const short4* __restrict__ rects;    // [N*F] exact pixel rectangle (x0,x1,row0,row1) of the bbox test
    const short4* __restrict__ grects;   // [N, ceil(F/64)] union of the pixel rects of 64 consecutive faces
    const float* __restrict__ textures;  // [N,F,T,3]
    int N, F, T, res, IS;
    float near, far, eps, sigma, gamma, thr;
    const float* __restrict__ near_far_dev;   // optional: {near, far} read on the device (no host sync)
    Modes m;
    int overwrite_grads;                      // backward, vertex attributes: store the face's gradients instead of adding to them
    int use_bg;                               // forward: the background colour comes from bg[] instead of the pre-filled soft_colors
    const int* __restrict__ choice;           // forward, optional: device word written by sr_choose_kernel; a forward kernel whose
                                              // id (CHOICE_*) differs returns at once (both candidates are launched)
    int choice_max;                           // < 0: *choice is the id; else *choice is the launch's number of non-empty tiles
                                              // (sr_order_kernel) and the id is CHOICE_COOP iff it is at most this
    const int* __restrict__ order;            // forward, optional: block -> (image, 8x8 tile) table written by sr_order_kernel
    float bg[9];
    // launch constants of the backward pass, computed once on the host (IEEE division / square root: the bits the device
    // expansions gave, without ~30 VALU instructions per face): 1 / IS, the pixel-centre fma coefficients 2 / IS, (1 - IS) / IS,
    // (IS - 1) / IS, and stage 1's reject distance -sqrt(1.05 thr)
    float inv_is, cx_a, cx_b, cy_b, far_t;
};

constexpr int CHOICE_ONE_WAVE = 0, CHOICE_COOP = 1;
__device__ __forceinline__ int chosen_kernel(const RasterArgs& A)
{
    const int w = *A.choice;
    return A.choice_max < 0 ? w : (w <= A.choice_max ? CHOICE_COOP : CHOICE_ONE_WAVE);
}

// Block -> (image, tile) with all tiles of an image kept on one XCD (block b runs
// on XCD b % 8; an image's records are then fetched into a single L2).
__device__ __forceinline__ int xcd_remap(int b, int total)
{
    const int per = total >> 3;
    if ((total & 7) == 0) return (b & 7) * per + (b >> 3);
    return b;
}

}  // namespace lasr
