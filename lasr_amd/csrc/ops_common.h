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

// (x, y, z, w) quaternion -> rotation matrix, normalising first (kornia 0.5.3's quaternion_to_rotation_matrix, call sites
// nnutils/mesh_net.py:232,250,265), and its gradient; shared by fused.hip (quat / skinning kernels) and glue.hip (pose chain)
struct Quat { float x, y, z, w, inv_n; };

__device__ __forceinline__ Quat load_quat(const float* q)
{
    Quat r;
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    r.inv_n = 1.f / fmaxf(n, 1e-12f);
    r.x = q[0] * r.inv_n; r.y = q[1] * r.inv_n; r.z = q[2] * r.inv_n; r.w = q[3] * r.inv_n;
    return r;
}

__device__ __forceinline__ void quat_matrix(const Quat& q, float* m)
{
    const float tx = 2.f * q.x, ty = 2.f * q.y, tz = 2.f * q.z;
    m[0] = 1.f - (ty * q.y + tz * q.z); m[1] = tx * q.y - tz * q.w;         m[2] = tx * q.z + ty * q.w;
    m[3] = tx * q.y + tz * q.w;         m[4] = 1.f - (tx * q.x + tz * q.z); m[5] = ty * q.z - tx * q.w;
    m[6] = tx * q.z - ty * q.w;         m[7] = ty * q.z + tx * q.w;         m[8] = 1.f - (tx * q.x + ty * q.y);
}

// gradient w.r.t. the raw quaternion given the gradient of the 9 matrix entries
__device__ __forceinline__ void quat_matrix_backward(const Quat& q, const float* g, float* gq)
{
    const float x = q.x, y = q.y, z = q.z, w = q.w;
    const float gx = 2.f * (y * (g[1] + g[3]) + z * (g[2] + g[6]) + w * (g[7] - g[5]) - 2.f * x * (g[4] + g[8]));
    const float gy = 2.f * (x * (g[1] + g[3]) + z * (g[5] + g[7]) + w * (g[2] - g[6]) - 2.f * y * (g[0] + g[8]));
    const float gz = 2.f * (x * (g[2] + g[6]) + y * (g[5] + g[7]) + w * (g[3] - g[1]) - 2.f * z * (g[0] + g[4]));
    const float gw = 2.f * (z * (g[3] - g[1]) + y * (g[2] - g[6]) + x * (g[7] - g[5]));
    // through q / max(|q|, eps): (g - qhat (qhat . g)) / |q| ; with the eps clamp active the norm is a constant
    const bool clamped = q.inv_n >= 1e12f;
    const float d = clamped ? 0.f : (x * gx + y * gy + z * gz + w * gw);
    gq[0] = (gx - x * d) * q.inv_n; gq[1] = (gy - y * d) * q.inv_n;
    gq[2] = (gz - z * d) * q.inv_n; gq[3] = (gw - w * d) * q.inv_n;
}

}  // namespace lasr

// Launch on `st` (a hipStream_t in scope), bracketed by profiling events when lasr_prof_enable(1) is in effect.
#define LASR_LAUNCH(ID, KERNEL, GRID, BLOCK, LDS, ...)                                   \
    do {                                                                                 \
        ProfScope ps_(ID, st);                                                           \
        hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, st, __VA_ARGS__);                   \
    } while (0)
