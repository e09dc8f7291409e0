// fused.hip -- the small operators between the kernels of ops.hip / sr_raster.hip (include/lasr_ops.h), gfx950 / wave64.
//
// In the reference each of these is a chain of 20-100 eager elementwise launches per call (forward and again in
// autograd's backward); at LASR's sizes every one of them is launch-bound, so each chain becomes one kernel pair:
//   flow reprojection   nnutils/mesh_net.py:93-104   (background fill, two pinhole reprojections, difference, detach rules)
// Reductions are deterministic (fixed tree inside a block, fixed-order fold of block partials; no float atomics).
#include <hip/hip_runtime.h>

#include "../../include/lasr_ops.h"
#include "ops_common.h"

namespace lasr {

constexpr int FUSED_PX_PER_BLOCK = 2048;     // 256 threads x 8 pixels
__host__ __device__ inline int px_chunks(int P) { int n = (P + FUSED_PX_PER_BLOCK - 1) / FUSED_PX_PER_BLOCK; return n < 1 ? 1 : (n > 64 ? 64 : n); }

// ===========================================================================
// Flow reprojection, mesh_net.py:93-104.  px = the 6-attribute render [N,7,P]: planes 0-2 camera-space position of
// frame t at each pixel, 3-5 of frame t', 6 alpha.  Background (either depth < 1e-9) becomes the point (10,10,10);
// flow = proj(p1; pp1, fl1) - proj(p0; pp0, fl0), proj(p) = pp + (p.xy * fl) / p.z.  Gradient reaches p1, pp1, fl1 at
// foreground pixels only (p0's projection and every background pixel are detached, :102-103).
// ===========================================================================
__global__ __launch_bounds__(256) void flow_reproject_forward_kernel(const float* __restrict__ px, const float* __restrict__ pp0,
                                                                     const float* __restrict__ pp1, const float* __restrict__ fl0,
                                                                     const float* __restrict__ fl1, float2* __restrict__ flow,
                                                                     unsigned char* __restrict__ bg, int P, int nch)
{
    const int n = blockIdx.x;
    const float* q = px + (size_t)n * 7 * P;
    const float c0x = pp0[2 * n], c0y = pp0[2 * n + 1], c1x = pp1[2 * n], c1y = pp1[2 * n + 1], f0 = fl0[n], f1 = fl1[n];
    const int per = (P + nch - 1) / nch, p0 = blockIdx.y * per, p1 = min(P, p0 + per);
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
        float x0 = q[p], y0 = q[P + p], z0 = q[2 * (size_t)P + p];
        float x1 = q[3 * (size_t)P + p], y1 = q[4 * (size_t)P + p], z1 = q[5 * (size_t)P + p];
        const bool b = (z0 < 1e-9f) | (z1 < 1e-9f);
        if (b) x0 = y0 = z0 = x1 = y1 = z1 = 10.f;
        const float u0 = c0x + (x0 * f0) / z0, v0 = c0y + (y0 * f0) / z0;
        const float u1 = c1x + (x1 * f1) / z1, v1 = c1y + (y1 * f1) / z1;
        flow[(size_t)n * P + p] = make_float2(u1 - u0, v1 - v0);
        bg[(size_t)n * P + p] = b ? 1 : 0;
    }
}

// grad_px [N,7,P] is written whole (zeros where no gradient flows); partial sums of the intrinsics' gradients go to
// part[n][chunk][4] = (d pp1.x, d pp1.y, d fl1, -)
__global__ __launch_bounds__(256) void flow_reproject_backward_kernel(const float* __restrict__ px, const float* __restrict__ fl1,
                                                                      const float2* __restrict__ gflow, float* __restrict__ gpx,
                                                                      float* __restrict__ part, int P, int nch)
{
    __shared__ float red[4];
    const int n = blockIdx.x;
    const float* q = px + (size_t)n * 7 * P;
    float* g = gpx + (size_t)n * 7 * P;
    const float f1 = fl1[n];
    const int per = (P + nch - 1) / nch, p0 = blockIdx.y * per, p1 = min(P, p0 + per);
    float sx = 0.f, sy = 0.f, sf = 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
        const float z0 = q[2 * (size_t)P + p];
        const float x1 = q[3 * (size_t)P + p], y1 = q[4 * (size_t)P + p], z1 = q[5 * (size_t)P + p];
        const bool b = (z0 < 1e-9f) | (z1 < 1e-9f);
        float gx1 = 0.f, gy1 = 0.f, gz1 = 0.f;
        if (!b) {
            const float2 gf = gflow[(size_t)n * P + p];
            const float ax = gf.x / z1, ay = gf.y / z1;             // d/d(x1*f1), d/d(y1*f1)
            gx1 = ax * f1; gy1 = ay * f1;
            gz1 = -(ax * ((x1 * f1) / z1) + ay * ((y1 * f1) / z1));
            sx += gf.x; sy += gf.y; sf += ax * x1 + ay * y1;
        }
        g[p] = 0.f; g[P + p] = 0.f; g[2 * (size_t)P + p] = 0.f;
        g[3 * (size_t)P + p] = gx1; g[4 * (size_t)P + p] = gy1; g[5 * (size_t)P + p] = gz1;
        g[6 * (size_t)P + p] = 0.f;
    }
    sx = block_sum(sx, red); sy = block_sum(sy, red); sf = block_sum(sf, red);
    if (threadIdx.x == 0) {
        float* o = part + ((size_t)n * nch + blockIdx.y) * 4;
        o[0] = sx; o[1] = sy; o[2] = sf; o[3] = 0.f;
    }
}

__global__ __launch_bounds__(256) void flow_reproject_fold_kernel(const float* __restrict__ part, float* __restrict__ gpp1,
                                                                  float* __restrict__ gfl1, int N, int nch)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float a = 0.f, b = 0.f, c = 0.f;
    for (int k = 0; k < nch; k++) {
        const float* o = part + ((size_t)n * nch + k) * 4;
        a += o[0]; b += o[1]; c += o[2];
    }
    gpp1[2 * n] = a; gpp1[2 * n + 1] = b; gfl1[n] = c;
}

}  // namespace lasr

// ===========================================================================
// C ABI
// ===========================================================================
using namespace lasr;

extern "C" size_t lasr_flow_reproject_scratch_floats(int N, int P)
{
    if (N < 0 || P < 0) return 0;
    return (size_t)N * px_chunks(P) * 4 + 4;
}

extern "C" int lasr_flow_reproject_forward(const float* px, const float* pp0, const float* pp1, const float* fl0,
                                           const float* fl1, float* flow, unsigned char* bgmask, int N, int P,
                                           void* hip_stream)
{
    if (N < 0 || P < 0) return LASR_E_BADARG;
    if (N == 0 || P == 0) return LASR_OK;
    if (!px || !pp0 || !pp1 || !fl0 || !fl1 || !flow || !bgmask) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int nch = px_chunks(P);
    LASR_LAUNCH(K_FLOW_REPROJECT_FORWARD, flow_reproject_forward_kernel, dim3(N, nch), dim3(256), 0, px, pp0, pp1, fl0, fl1,
                (float2*)flow, bgmask, P, nch);
    return launch_ok();
}

extern "C" int lasr_flow_reproject_backward(const float* px, const float* fl1, const float* grad_flow, float* grad_px,
                                            float* grad_pp1, float* grad_fl1, float* scratch, int N, int P,
                                            void* hip_stream)
{
    if (N < 0 || P < 0) return LASR_E_BADARG;
    if (N == 0) return LASR_OK;
    if (!px || !fl1 || !grad_flow || !grad_px || !grad_pp1 || !grad_fl1 || !scratch) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int nch = px_chunks(P);
    LASR_LAUNCH(K_FLOW_REPROJECT_BACKWARD, flow_reproject_backward_kernel, dim3(N, nch), dim3(256), 0, px, fl1,
                (const float2*)grad_flow, grad_px, scratch, P, nch);
    int rc = launch_ok();
    if (rc) return rc;
    LASR_LAUNCH(K_FLOW_REPROJECT_BACKWARD, flow_reproject_fold_kernel, dim3((N + 255) / 256), dim3(256), 0, scratch, grad_pp1,
                grad_fl1, N, nch);
    return launch_ok();
}
