// glue.hip -- the small-tensor arithmetic of LASR.forward between its big operators, as single kernels.
//
// In the reference these are runs of eager elementwise / slice / cat / mean ops on tensors of a few hundred floats
// (nnutils/mesh_net.py:204-217 intrinsics bookkeeping, :259-283 bone-transform fix-up, :506-522 rotation distance,
// :374-530 the weighted sum of loss means): ~150 launches of 2-4 us per optimisation step and as many autograd nodes.
// Each kernel here is one workgroup (the data fits a few wavefronts); the point is launch count, not bandwidth.
#include <hip/hip_runtime.h>

#include "../../include/lasr_ops.h"
#include "mesh_losses.h"
#include "ops_common.h"

namespace lasr {

// ---- rotation distance (third_party/ext_utils/util_rot.py:27-37) ------------------------------------------------------
// cos = (trace(m1 m2^T) - 1) / 2, angle = acos(cos) where |cos| < 1, else 0 (cos >= 1) or pi (cos <= -1).
__device__ __forceinline__ float geodesic_cos(const float* a, const float* b)
{
    const float r0 = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];      // (m1 m2^T)[0,0]
    const float r1 = a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
    const float r2 = a[6] * b[6] + a[7] * b[7] + a[8] * b[8];
    return (r0 + r1 + r2 - 1.f) / 2.f;
}

__global__ __launch_bounds__(256) void geodesic_forward_kernel(const float* __restrict__ m1, const float* __restrict__ m2,
                                                               float* __restrict__ angle, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float c = geodesic_cos(m1 + 9 * (size_t)i, m2 + 9 * (size_t)i);
    angle[i] = fabsf(c) < 1.f ? acosf(c) : (c > 0.f ? 0.f : 3.14159265358979323846f);
}

__global__ __launch_bounds__(256) void geodesic_backward_kernel(const float* __restrict__ m1, const float* __restrict__ m2,
                                                                const float* __restrict__ gangle, float* __restrict__ g1,
                                                                float* __restrict__ g2, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* a = m1 + 9 * (size_t)i;
    const float* b = m2 + 9 * (size_t)i;
    const float c = geodesic_cos(a, b);
    // d acos / d cos = -1 / sqrt(1 - cos^2); outside (-1, 1) the angle is a constant (zero gradient)
    const float gc = fabsf(c) < 1.f ? -gangle[i] / sqrtf(1.f - c * c) * 0.5f : 0.f;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        g1[9 * (size_t)i + k] = gc * b[k];
        g2[9 * (size_t)i + k] = gc * a[k];
    }
}

// ---- total = sum_i w_i * mean(x_i), with per-group partial totals ----------------------------------------------------
struct MeansArgs {
    const float* x[LASR_MEANS_MAX_TERMS];
    int n[LASR_MEANS_MAX_TERMS];
    float w[LASR_MEANS_MAX_TERMS];
    int g[LASR_MEANS_MAX_TERMS];
    int terms, groups;
};

// One workgroup of 1024 threads.  Every thread strides over every term (all the loads of all the terms are in flight
// together), the 16 wave totals of each term meet in LDS, ONE barrier, then thread t folds term t's wave totals in wave order
// and thread 0 accumulates the weighted means in term order: out[0..groups) = group totals, out[groups] = total.  The order
// of every sum is the one of the round-1..4 kernel (which walked the terms one at a time, two barriers each: 10 us for LASR's
// dozen tiny tensors), so the numbers are bit-identical.
__global__ __launch_bounds__(1024) void weighted_means_kernel(MeansArgs A, float* __restrict__ out)
{
    __shared__ float red[LASR_MEANS_MAX_TERMS][16];
    __shared__ float val[LASR_MEANS_MAX_TERMS];
    const int tid = threadIdx.x;
    for (int t = 0; t < A.terms; t++) {
        const float* __restrict__ x = A.x[t];
        float s = 0.f;
        for (int i = tid; i < A.n[t]; i += 1024) s += x[i];
        s = wave_sum_to_lane63(s);
        if ((tid & 63) == 63) red[t][tid >> 6] = s;
    }
    __syncthreads();
    if (tid < A.terms) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) tot += red[tid][k];
        val[tid] = A.w[tid] * (A.n[tid] > 0 ? tot / (float)A.n[tid] : 0.f);
    }
    __syncthreads();
    if (tid <= A.groups) {                                   // thread g: its group's terms in term order; thread `groups`: all
        float acc = 0.f;
        for (int t = 0; t < A.terms; t++)
            if (tid == A.groups || A.g[t] == tid) acc += val[t];
        out[tid] = acc;
    }
}

struct MeansCoef { float c[LASR_MEANS_MAX_TERMS]; int terms; };
__global__ __launch_bounds__(64) void weighted_means_backward_kernel(MeansCoef C, const float* __restrict__ gtotal,
                                                                     float* __restrict__ coef)
{
    const int t = threadIdx.x;
    if (t < C.terms) coef[t] = gtotal[0] * C.c[t];
}

// ---- intrinsics bookkeeping (nnutils/mesh_net.py:204-217) ------------------------------------------------------------
// cams[:, 0] = crop scale a of each image (rows 0..B-1: frames t, rows B..2B-1: frames t'), pp = crop centre offsets.
//   scale_out = a * scale;   depth_out[:, 0] = a * depth[:, 0], other columns unchanged
//   ppoint_out[:B] = ppoint[:B];   ppoint_out[B + i] = (ppoint[i] + a_i pp_i / h + 1) * (a'_i / a_i) - a'_i pp'_i / h - 1
__global__ __launch_bounds__(256) void intrinsics_forward_kernel(const float* __restrict__ cams, int cam_stride,
                                                                 const float* __restrict__ pp, const float* __restrict__ scale,
                                                                 const float* __restrict__ depth,
                                                                 const float* __restrict__ ppoint, float* __restrict__ scale_out,
                                                                 float* __restrict__ depth_out, float* __restrict__ ppoint_out,
                                                                 int B, int H, int K, float half)
{
    const int n2 = 2 * B;
    for (int i = threadIdx.x; i < n2 * H; i += 256) scale_out[i] = cams[(i / H) * cam_stride] * scale[i];
    for (int i = threadIdx.x; i < n2 * K; i += 256) {
        const float d = depth[i];
        depth_out[i] = (i % K == 0) ? cams[(i / K) * cam_stride] * d : d;
    }
    for (int i = threadIdx.x; i < n2 * 2; i += 256) {
        const int r = i >> 1, c = i & 1;
        if (r < B) { ppoint_out[i] = ppoint[i]; continue; }
        const int r0 = r - B;
        const float a0 = cams[r0 * cam_stride], a1 = cams[r * cam_stride];
        const float ppb1 = a0 * pp[2 * r0 + c] / half, ppb2 = a1 * pp[2 * r + c] / half;
        const float ppa1 = ppoint[2 * r0 + c] + ppb1 + 1.f;
        ppoint_out[i] = ppa1 * (a1 / a0) - ppb2 - 1.f;
    }
}

__global__ __launch_bounds__(256) void intrinsics_backward_kernel(const float* __restrict__ cams, int cam_stride,
                                                                  const float* __restrict__ gscale_out,
                                                                  const float* __restrict__ gdepth_out,
                                                                  const float* __restrict__ gppoint_out, float* __restrict__ gscale,
                                                                  float* __restrict__ gdepth, float* __restrict__ gppoint, int B,
                                                                  int H, int K)
{
    const int n2 = 2 * B;
    for (int i = threadIdx.x; i < n2 * H; i += 256) gscale[i] = cams[(i / H) * cam_stride] * gscale_out[i];
    for (int i = threadIdx.x; i < n2 * K; i += 256) {
        const float g = gdepth_out[i];
        gdepth[i] = (i % K == 0) ? cams[(i / K) * cam_stride] * g : g;
    }
    for (int i = threadIdx.x; i < n2 * 2; i += 256) {
        const int r = i >> 1, c = i & 1;
        if (r >= B) { gppoint[i] = 0.f; continue; }                  // the predicted principal point of frame t' is not used
        const float a0 = cams[r * cam_stride], a1 = cams[(r + B) * cam_stride];
        gppoint[i] = gppoint_out[i] + gppoint_out[2 * (r + B) + c] * (a1 / a0);
    }
}

// ---- bone-transform fix-up (nnutils/mesh_net.py:259-283) --------------------------------------------------------------
// Per (image-hypothesis m, bone k): Q = the 3x3 the pose head predicts (row-major), t = (trans_x, trans_y, depth).
//   root (k = 0):   R' = Q^T,  T' = t
//   bones (k >= 1): R' = Q,    T' = t + c - Q^T c     with c = rest_ts[h, k-1] (rotate about the joint)
// (the reference transposes every Q, applies -R c + T + c with R = Q^T, then transposes the bones back.)
// pair_angle (optional, [M*K/2]): the rotation distance between matrix i of the first half of the batch (frame t) and matrix i of the
// second half (frame t'), nnutils/mesh_net.py:514-516 -- the values lasr_geodesic_forward gives for (quat[:half], quat[half:]),
// without a launch of its own (the fix-up reads every matrix anyway).
__global__ __launch_bounds__(256) void bone_fixup_forward_kernel(const float* __restrict__ quat, const float* __restrict__ trans,
                                                                 const float* __restrict__ depth, const float* __restrict__ rest,
                                                                 float* __restrict__ Rout, float* __restrict__ Tout,
                                                                 float* __restrict__ pair_angle, int M, int H, int K)
{
    const int i = blockIdx.x * 256 + threadIdx.x;                    // i = (m * K + k),  m = image * H + h
    if (i >= M * K) return;
    if (pair_angle && i < (M * K) / 2) {
        const float c = geodesic_cos(quat + 9 * (size_t)i, quat + 9 * ((size_t)i + (M * K) / 2));
        pair_angle[i] = fabsf(c) < 1.f ? acosf(c) : (c > 0.f ? 0.f : 3.14159265358979323846f);
    }
    const int k = i % K, h = (i / K) % H;
    const float* q = quat + 9 * (size_t)i;
    float* R = Rout + 9 * (size_t)i;
    float tx = trans[2 * (size_t)i], ty = trans[2 * (size_t)i + 1], tz = depth[i];
    if (k == 0) {
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) R[3 * r + c] = q[3 * c + r];
    } else {
#pragma unroll
        for (int j = 0; j < 9; j++) R[j] = q[j];
        const float* c = rest + 3 * ((size_t)h * (K - 1) + (k - 1));
        // (Q^T c)_r = sum_j Q[j][r] c_j ; summed j = 0,1,2 like the reference's matmul row
        const float r0 = q[0] * c[0] + q[3] * c[1] + q[6] * c[2];
        const float r1 = q[1] * c[0] + q[4] * c[1] + q[7] * c[2];
        const float r2 = q[2] * c[0] + q[5] * c[1] + q[8] * c[2];
        tx = -r0 + tx + c[0]; ty = -r1 + ty + c[1]; tz = -r2 + tz + c[2];
    }
    Tout[3 * (size_t)i] = tx; Tout[3 * (size_t)i + 1] = ty; Tout[3 * (size_t)i + 2] = tz;
}

// grad_quat / grad_trans / grad_depth per (m, k); grad_rest[h, k-1] = sum over the images of (g - Q g) in image order.
// gangle (optional, [M*K/2]): the upstream gradient of the forward's pair_angle; its part of grad_quat (lasr_geodesic_backward's
// expressions) is added to the fix-up's part -- the one addition autograd performs on the two gradients of `quat`.
__global__ __launch_bounds__(256) void bone_fixup_backward_kernel(const float* __restrict__ quat, const float* __restrict__ rest,
                                                                  const float* __restrict__ gR, const float* __restrict__ gT,
                                                                  const float* __restrict__ gangle,
                                                                  float* __restrict__ gquat, float* __restrict__ gtrans,
                                                                  float* __restrict__ gdepth, float* __restrict__ grest, int M,
                                                                  int H, int K)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int total = M * K;
    if (i < total) {
        const int k = i % K, h = (i / K) % H;
        const float* g = gR + 9 * (size_t)i;
        const float* t = gT + 3 * (size_t)i;
        float* gq = gquat + 9 * (size_t)i;
        gtrans[2 * (size_t)i] = t[0]; gtrans[2 * (size_t)i + 1] = t[1]; gdepth[i] = t[2];
        float v[9];
        if (k == 0) {
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) v[3 * c + r] = g[3 * r + c];
        } else {
            const float* c = rest + 3 * ((size_t)h * (K - 1) + (k - 1));
            // T'_r = t_r + c_r - sum_j Q[j][r] c_j   =>   dT'_r / dQ[j][r] = -c_j
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int r = 0; r < 3; r++) v[3 * j + r] = g[3 * j + r] - c[j] * t[r];
        }
        if (gangle) {
            const int half = total / 2, first = i < half ? i : i - half;
            const float* a = quat + 9 * (size_t)first;
            const float* b = quat + 9 * ((size_t)first + half);
            const float cs = geodesic_cos(a, b);
            const float gc = fabsf(cs) < 1.f ? -gangle[first] / sqrtf(1.f - cs * cs) * 0.5f : 0.f;
            const float* other = i < half ? b : a;
#pragma unroll
            for (int j = 0; j < 9; j++) v[j] = v[j] + gc * other[j];
        }
#pragma unroll
        for (int j = 0; j < 9; j++) gq[j] = v[j];
    }
    // rest_ts: one thread per (h, k-1, component) sums its images in order
    const int nrest = H * (K - 1) * 3;
    if (i < nrest) {
        const int comp = i % 3, kb = (i / 3) % (K - 1), h = i / (3 * (K - 1));
        const int images = M / H;
        float s = 0.f;
        for (int im = 0; im < images; im++) {
            const size_t e = ((size_t)(im * H + h)) * K + (kb + 1);
            const float* q = quat + 9 * e;
            const float* t = gT + 3 * e;
            // dT'_r / dc_j = delta_rj - Q[j][r]
            s += t[comp] - (q[3 * comp + 0] * t[0] + q[3 * comp + 1] * t[1] + q[3 * comp + 2] * t[2]);
        }
        grest[i] = s;
    }
}

// ---- joints and control points into the image (nnutils/mesh_net.py:285-288 + :302: two obj_to_cam calls with an identity "skin" and a
// pinhole_cam) ----------------------------------------------------------------------------------------------------------------
// Point j of hypothesis h (j < K-1: joint centre rest_ts[h, j]; else control point ctl_ts[h, j - (K-1)]) rides on part bone
// b = j mod (K-1): p' = p R[m, b+1] + T[m, b+1], then the body transform R[m, 0], T[m, 0], then the pinhole projection with image
// m's intrinsics -- for every (image, hypothesis) m = img * H + h.  The reference builds the repeated point tensor, the one-hot
// skin, two LBS calls and a projection (~10 launches each way); the sums below associate exactly like lbs_forward_kernel's
// row folds followed by pinhole_forward_kernel, so the values are the ones those kernels gave.  Transforms and intrinsics are
// constants here (detached in the reference); only the points receive gradient, summed over the images in image order.
__global__ __launch_bounds__(256) void project_points_forward_kernel(const float* __restrict__ rest, const float* __restrict__ ctl,
                                                                     const float* __restrict__ Rmat, const float* __restrict__ Tmat,
                                                                     const float* __restrict__ pp, const float* __restrict__ fl,
                                                                     float4* __restrict__ proj, int M, int H, int K)
{
    const int nb = K - 1, Pn = 2 * nb;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * Pn) return;
    const int m = i / Pn, j = i - m * Pn, h = m % H, b = j % nb;
    const float* p = (j < nb ? rest : ctl) + ((size_t)h * nb + b) * 3;
    const float* R = Rmat + ((size_t)m * K + b + 1) * 9;
    const float* T = Tmat + ((size_t)m * K + b + 1) * 3;
    const float* R0 = Rmat + (size_t)m * K * 9;
    const float* T0 = Tmat + (size_t)m * K * 3;
    float vs[3], c[3];
#pragma unroll
    for (int d = 0; d < 3; d++) vs[d] = (T[d] + p[2] * R[6 + d]) + (p[1] * R[3 + d] + p[0] * R[d]);     // lbs_forward_kernel's fold order
#pragma unroll
    for (int d = 0; d < 3; d++) c[d] = vs[0] * R0[d] + vs[1] * R0[3 + d] + vs[2] * R0[6 + d] + T0[d];
    const int img = m / H;
    const float f = fl[m];
    proj[i] = make_float4(pp[2 * img] + c[0] * f / c[2], pp[2 * img + 1] + c[1] * f / c[2], c[2], 1.f);          // geom_utils.py:32-33
}

__global__ __launch_bounds__(256) void project_points_backward_kernel(const float* __restrict__ rest, const float* __restrict__ ctl,
                                                                      const float* __restrict__ Rmat, const float* __restrict__ Tmat,
                                                                      const float* __restrict__ fl, const float4* __restrict__ gproj,
                                                                      float* __restrict__ grest, float* __restrict__ gctl, int M, int H,
                                                                      int K)
{
    const int nb = K - 1, Pn = 2 * nb;
    const int i = blockIdx.x * 256 + threadIdx.x;           // (h, point)
    if (i >= H * Pn) return;
    const int h = i / Pn, j = i - h * Pn, b = j % nb;
    const float* p = (j < nb ? rest : ctl) + ((size_t)h * nb + b) * 3;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int img = 0; img < M / H; img++) {                  // image order: deterministic
        const int m = img * H + h;
        const float* R = Rmat + ((size_t)m * K + b + 1) * 9;
        const float* T = Tmat + ((size_t)m * K + b + 1) * 3;
        const float* R0 = Rmat + (size_t)m * K * 9;
        const float* T0 = Tmat + (size_t)m * K * 3;
        float vs[3], c[3];
#pragma unroll
        for (int d = 0; d < 3; d++) vs[d] = (T[d] + p[2] * R[6 + d]) + (p[1] * R[3 + d] + p[0] * R[d]);
#pragma unroll
        for (int d = 0; d < 3; d++) c[d] = vs[0] * R0[d] + vs[1] * R0[3 + d] + vs[2] * R0[6 + d] + T0[d];
        const float4 g = gproj[(size_t)m * Pn + j];
        const float f = fl[m], iz = 1.f / c[2], xz = c[0] * iz, yz = c[1] * iz;              // == pinhole_backward_kernel
        const float gc[3] = {g.x * f * iz, g.y * f * iz, g.z - (g.x * xz + g.y * yz) * f * iz};
        float gv[3];
#pragma unroll
        for (int d = 0; d < 3; d++) gv[d] = gc[0] * R0[3 * d] + gc[1] * R0[3 * d + 1] + gc[2] * R0[3 * d + 2];      // g_vs = g_c R0^T
#pragma unroll
        for (int d = 0; d < 3; d++) acc[d] += gv[0] * R[3 * d] + gv[1] * R[3 * d + 1] + gv[2] * R[3 * d + 2];       // g_p = g_vs R^T
    }
    float* o = (j < nb ? grest : gctl) + ((size_t)h * nb + b) * 3;
    o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
}

// ---- the pose chain: intrinsics -> quaternion matrices -> bone fix-up -> joint / control-point projection, ONE launch each way ----
// The four operators above (and fused.hip's quat kernels) in the order LASR.forward runs them (nnutils/mesh_net.py:204-217, :232,
// :259-289, :302), as phases of one 256-thread workgroup separated by barriers: the tensors hold a few hundred floats, every
// launch of the chain costs ~4.7 us of which ~0.3 us is work (1024 threads: every phase is one pass at LASR's sizes).  Each phase is the expression sequence of its stand-alone kernel,
// so values and gradients are the ones the separate launches give (tests/test_step_fusions_gpu.py); the H-fold broadcast of
// trans / depth (`.repeat(1, H, 1, 1)`) and its gradient (the sum over hypotheses, in hypothesis order) happen inside.
//   quat4 [M*K,4] unit quaternions (x, y, z, w), trans [2B*K,2], depth [2B,K], scale [2B,H], ppoint [2B,2], M = 2B*H
struct PoseChain {
    const float* cams; const float* pp; const float* scale; const float* depth; const float* ppoint;
    const float* quat4; const float* trans; const float* rest; const float* ctl;
    float* scale_out; float* depth_out; float* ppoint_out; float* trans_rep; float* depth_rep;
    float* Rmat; float* Tmat; float* pair_angle; float4* proj;
    int cam_stride, B, H, K; float half;
};

__global__ __launch_bounds__(1024) void pose_chain_forward_kernel(PoseChain A)
{
    const int B = A.B, H = A.H, K = A.K, n2 = 2 * B, M = n2 * H, MK = M * K;
    // phase 1: intrinsics (intrinsics_forward_kernel)
    for (int i = threadIdx.x; i < n2 * H; i += 1024) A.scale_out[i] = A.cams[(i / H) * A.cam_stride] * A.scale[i];
    for (int i = threadIdx.x; i < n2 * K; i += 1024) {
        const float d = A.depth[i];
        A.depth_out[i] = (i % K == 0) ? A.cams[(i / K) * A.cam_stride] * d : d;
    }
    for (int i = threadIdx.x; i < n2 * 2; i += 1024) {
        const int r = i >> 1, c = i & 1;
        if (r < B) { A.ppoint_out[i] = A.ppoint[i]; continue; }
        const int r0 = r - B;
        const float a0 = A.cams[r0 * A.cam_stride], a1 = A.cams[r * A.cam_stride];
        const float ppb1 = a0 * A.pp[2 * r0 + c] / A.half, ppb2 = a1 * A.pp[2 * r + c] / A.half;
        const float ppa1 = A.ppoint[2 * r0 + c] + ppb1 + 1.f;
        A.ppoint_out[i] = ppa1 * (a1 / a0) - ppb2 - 1.f;
    }
    __syncthreads();
    // phase 2: quaternion -> matrix (quat_forward_kernel), rotation distance of the pair and bone fix-up (bone_fixup_forward_kernel)
    for (int i = threadIdx.x; i < MK; i += 1024) {
        float q[9];
        quat_matrix(load_quat(A.quat4 + 4 * (size_t)i), q);
        if (A.pair_angle && i < MK / 2) {
            float o[9];
            quat_matrix(load_quat(A.quat4 + 4 * ((size_t)i + MK / 2)), o);
            const float c = geodesic_cos(q, o);
            A.pair_angle[i] = fabsf(c) < 1.f ? acosf(c) : (c > 0.f ? 0.f : 3.14159265358979323846f);
        }
        const int k = i % K, m = i / K, h = m % H, img = m / H;
        float* R = A.Rmat + 9 * (size_t)i;
        const float tx0 = A.trans[2 * ((size_t)img * K + k)], ty0 = A.trans[2 * ((size_t)img * K + k) + 1], tz0 = A.depth_out[img * K + k];
        A.trans_rep[2 * (size_t)i] = tx0; A.trans_rep[2 * (size_t)i + 1] = ty0; A.depth_rep[i] = tz0;
        float tx = tx0, ty = ty0, tz = tz0;
        if (k == 0) {
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) R[3 * r + c] = q[3 * c + r];
        } else {
#pragma unroll
            for (int j = 0; j < 9; j++) R[j] = q[j];
            const float* c = A.rest + 3 * ((size_t)h * (K - 1) + (k - 1));
            const float r0 = q[0] * c[0] + q[3] * c[1] + q[6] * c[2];
            const float r1 = q[1] * c[0] + q[4] * c[1] + q[7] * c[2];
            const float r2 = q[2] * c[0] + q[5] * c[1] + q[8] * c[2];
            tx = -r0 + tx + c[0]; ty = -r1 + ty + c[1]; tz = -r2 + tz + c[2];
        }
        A.Tmat[3 * (size_t)i] = tx; A.Tmat[3 * (size_t)i + 1] = ty; A.Tmat[3 * (size_t)i + 2] = tz;
    }
    if (K < 2) return;
    __syncthreads();
    // phase 3: joints and control points into the image (project_points_forward_kernel; fl = scale_out, pp = ppoint_out)
    const int nb = K - 1, Pn = 2 * nb;
    for (int i = threadIdx.x; i < M * Pn; i += 1024) {
        const int m = i / Pn, j = i - m * Pn, h = m % H, b = j % nb;
        const float* p = (j < nb ? A.rest : A.ctl) + ((size_t)h * nb + b) * 3;
        const float* R = A.Rmat + ((size_t)m * K + b + 1) * 9;
        const float* T = A.Tmat + ((size_t)m * K + b + 1) * 3;
        const float* R0 = A.Rmat + (size_t)m * K * 9;
        const float* T0 = A.Tmat + (size_t)m * K * 3;
        float vs[3], c[3];
#pragma unroll
        for (int d = 0; d < 3; d++) vs[d] = (T[d] + p[2] * R[6 + d]) + (p[1] * R[3 + d] + p[0] * R[d]);
#pragma unroll
        for (int d = 0; d < 3; d++) c[d] = vs[0] * R0[d] + vs[1] * R0[3 + d] + vs[2] * R0[6 + d] + T0[d];
        const int img = m / H;
        const float f = A.scale_out[m];
        A.proj[i] = make_float4(A.ppoint_out[2 * img] + c[0] * f / c[2], A.ppoint_out[2 * img + 1] + c[1] * f / c[2], c[2], 1.f);
    }
}

struct PoseChainGrad {
    const float* cams; const float* quat4; const float* rest; const float* ctl; const float* Rmat; const float* Tmat; const float* fl;
    const float* g_scale_out; const float* g_ppoint_out; const float* g_trans_rep; const float* g_depth_rep;
    const float* gR; const float* gT; const float* g_angle; const float4* g_proj;
    float* g_scale; float* g_depth; float* g_ppoint; float* g_quat4; float* g_trans; float* g_rest; float* g_ctl;
    float* scratch;                                   // [M*K, 3]: gradient of the repeated (trans_x, trans_y, depth) per (m, k)
    int cam_stride, B, H, K;
};

// (1024 threads: every phase is one pass at LASR's sizes -- a second pass is a second chain of load round trips, and the launch is
// nothing but such chains; the per-(m, k) gradient of the repeated trans / depth stays in LDS when it fits, `lds_scratch`)
constexpr int POSE_CHAIN_THREADS = 1024;
constexpr int POSE_CHAIN_LDS_FLOATS = 12288;
__global__ __launch_bounds__(POSE_CHAIN_THREADS) void pose_chain_backward_kernel(PoseChainGrad A, int lds_scratch)
{
    extern __shared__ float pose_lds[];
    constexpr int NT = POSE_CHAIN_THREADS;
    float* const scratch = lds_scratch ? pose_lds : A.scratch;
    const int B = A.B, H = A.H, K = A.K, n2 = 2 * B, M = n2 * H, MK = M * K, nb = K - 1, Pn = 2 * nb;
    // phase 1: projection (project_points_backward_kernel): gradient of the points, summed over the images in image order
    if (K > 1) {
        for (int i = threadIdx.x; i < H * Pn; i += NT) {
            const int h = i / Pn, j = i - h * Pn, b = j % nb;
            const float* p = (j < nb ? A.rest : A.ctl) + ((size_t)h * nb + b) * 3;
            float acc[3] = {0.f, 0.f, 0.f};
            if (A.g_proj)
            for (int img = 0; img < n2; img++) {
                const int m = img * H + h;
                const float* R = A.Rmat + ((size_t)m * K + b + 1) * 9;
                const float* T = A.Tmat + ((size_t)m * K + b + 1) * 3;
                const float* R0 = A.Rmat + (size_t)m * K * 9;
                const float* T0 = A.Tmat + (size_t)m * K * 3;
                float vs[3], c[3];
#pragma unroll
                for (int d = 0; d < 3; d++) vs[d] = (T[d] + p[2] * R[6 + d]) + (p[1] * R[3 + d] + p[0] * R[d]);
#pragma unroll
                for (int d = 0; d < 3; d++) c[d] = vs[0] * R0[d] + vs[1] * R0[3 + d] + vs[2] * R0[6 + d] + T0[d];
                const float4 g = A.g_proj[(size_t)m * Pn + j];
                const float f = A.fl[m], iz = 1.f / c[2], xz = c[0] * iz, yz = c[1] * iz;
                const float gc[3] = {g.x * f * iz, g.y * f * iz, g.z - (g.x * xz + g.y * yz) * f * iz};
                float gv[3];
#pragma unroll
                for (int d = 0; d < 3; d++) gv[d] = gc[0] * R0[3 * d] + gc[1] * R0[3 * d + 1] + gc[2] * R0[3 * d + 2];
#pragma unroll
                for (int d = 0; d < 3; d++) acc[d] += gv[0] * R[3 * d] + gv[1] * R[3 * d + 1] + gv[2] * R[3 * d + 2];
            }
            float* o = (j < nb ? A.g_rest : A.g_ctl) + ((size_t)h * nb + b) * 3;
            o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
        }
        __syncthreads();
    }
    // phase 2: bone fix-up + rotation distance (bone_fixup_backward_kernel), then through the quaternion (quat_backward_kernel)
    for (int i = threadIdx.x; i < MK; i += NT) {
        const int k = i % K, h = (i / K) % H;
        const float* g = A.gR + 9 * (size_t)i;
        const float* t = A.gT + 3 * (size_t)i;
        scratch[3 * (size_t)i] = t[0] + (A.g_trans_rep ? A.g_trans_rep[2 * (size_t)i] : 0.f);
        scratch[3 * (size_t)i + 1] = t[1] + (A.g_trans_rep ? A.g_trans_rep[2 * (size_t)i + 1] : 0.f);
        scratch[3 * (size_t)i + 2] = t[2] + (A.g_depth_rep ? A.g_depth_rep[i] : 0.f);
        float v[9];
        if (k == 0) {
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) v[3 * c + r] = g[3 * r + c];
        } else {
            const float* c = A.rest + 3 * ((size_t)h * (K - 1) + (k - 1));
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int r = 0; r < 3; r++) v[3 * j + r] = g[3 * j + r] - c[j] * t[r];
        }
        const Quat qi = load_quat(A.quat4 + 4 * (size_t)i);
        if (A.g_angle) {
            const int half = MK / 2, first = i < half ? i : i - half;
            float a[9], b[9];
            quat_matrix(load_quat(A.quat4 + 4 * (size_t)first), a);
            quat_matrix(load_quat(A.quat4 + 4 * ((size_t)first + half)), b);
            const float cs = geodesic_cos(a, b);
            const float gc = fabsf(cs) < 1.f ? -A.g_angle[first] / sqrtf(1.f - cs * cs) * 0.5f : 0.f;
            const float* other = i < half ? b : a;
#pragma unroll
            for (int j = 0; j < 9; j++) v[j] = v[j] + gc * other[j];
        }
        float o[4];
        quat_matrix_backward(qi, v, o);
#pragma unroll
        for (int j = 0; j < 4; j++) A.g_quat4[4 * (size_t)i + j] = o[j];
    }
    if (K > 1) {
        // rest_ts: one thread per (h, k-1, component) sums its images in order, then adds the projection's part (phase 1)
        for (int i = threadIdx.x; i < H * nb * 3; i += NT) {
            const int comp = i % 3, kb = (i / 3) % nb, h = i / (3 * nb);
            float s = 0.f;
            for (int im = 0; im < n2; im++) {
                const size_t e = ((size_t)(im * H + h)) * K + (kb + 1);
                float q[9];
                quat_matrix(load_quat(A.quat4 + 4 * e), q);
                const float* t = A.gT + 3 * e;
                s += t[comp] - (q[3 * comp + 0] * t[0] + q[3 * comp + 1] * t[1] + q[3 * comp + 2] * t[2]);
            }
            A.g_rest[i] = s + A.g_rest[i];
        }
    }
    __syncthreads();
    // phase 3: the H-fold broadcast of trans / depth backwards (hypothesis order), then intrinsics (intrinsics_backward_kernel)
    for (int i = threadIdx.x; i < n2 * K; i += NT) {
        const int img = i / K, k = i - img * K;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int h = 0; h < H; h++) {
            const float* t = scratch + 3 * ((size_t)(img * H + h) * K + k);
            sx += t[0]; sy += t[1]; sz += t[2];
        }
        A.g_trans[2 * (size_t)i] = sx; A.g_trans[2 * (size_t)i + 1] = sy;
        A.g_depth[i] = k == 0 ? A.cams[img * A.cam_stride] * sz : sz;
    }
    for (int i = threadIdx.x; i < n2 * H; i += NT)
        A.g_scale[i] = A.g_scale_out ? A.cams[(i / H) * A.cam_stride] * A.g_scale_out[i] : 0.f;
    for (int i = threadIdx.x; i < n2 * 2; i += NT) {
        const int r = i >> 1, c = i & 1;
        if (r >= B || !A.g_ppoint_out) { A.g_ppoint[i] = 0.f; continue; }
        const float a0 = A.cams[r * A.cam_stride], a1 = A.cams[(r + B) * A.cam_stride];
        A.g_ppoint[i] = A.g_ppoint_out[i] + A.g_ppoint_out[2 * (r + B) + c] * (a1 / a0);
    }
}

// ---- symmetric Chamfer distance of two small point sets (pytorch3d.loss.chamfer_distance as used at nnutils/mesh_net.py:503) --
// a [N,P,3], b [N,Q,3] -> out[n] = mean_i min_j |a_i - b_j|^2 + mean_j min_i |b_j - a_i|^2, nearest indices kept for the
// backward.  One workgroup per batch item (the control-point sets hold <= 35 points; larger sets loop).
__global__ __launch_bounds__(256) void chamfer_forward_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              float* __restrict__ out, int* __restrict__ nn_ab,
                                                              int* __restrict__ nn_ba, int P, int Q)
{
    __shared__ float red[4];
    const int n = blockIdx.x;
    const float v = chamfer_forward_item(a + (size_t)n * P * 3, b + (size_t)n * Q * 3, nn_ab + (size_t)n * P, nn_ba + (size_t)n * Q, P, Q, red);
    if (threadIdx.x == 0) out[n] = v;
}

// grad_a[n,i] = g[n] * (2/P (a_i - b_nn(i)) + 2/Q sum_{j: nn'(j) = i} (a_i - b_j)), grad_b likewise (gather form: deterministic)
__global__ __launch_bounds__(256) void chamfer_backward_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                               const int* __restrict__ nn_ab, const int* __restrict__ nn_ba,
                                                               const float* __restrict__ g, float* __restrict__ ga,
                                                               float* __restrict__ gb, int P, int Q)
{
    const int n = blockIdx.x;
    chamfer_backward_item(a + (size_t)n * P * 3, b + (size_t)n * Q * 3, nn_ab + (size_t)n * P, nn_ba + (size_t)n * Q, g[n],
                          ga + (size_t)n * P * 3, gb + (size_t)n * Q * 3, P, Q);
}

// ---- mean shape of the batch (third_party/ext_nnutils/mesh_net.py:128-149, 171-185) ------------------------------------------
// mean_v / tex [H,Vp,3] hold the independent + right-half vertices; the full mesh appends the mirror images of the last S
// vertices (x flip) and pins the symmetry-plane vertices with a 0/1 mask; every (image, hypothesis) of the batch gets a copy,
// the colours go through a sigmoid.  out_v / out_tex [R*H, Vp+S, 3].  S = 0: no symmetry (plain tiling).
__global__ __launch_bounds__(256) void mean_shape_forward_kernel(const float* __restrict__ mean_v, const float* __restrict__ tex,
                                                                 const float* __restrict__ flip, const float* __restrict__ mask,
                                                                 float* __restrict__ out_v, float* __restrict__ out_tex, int R,
                                                                 int H, int Vp, int S)
{
    const int V = Vp + S;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R * H * V * 3) return;
    const int c = i % 3, v = (i / 3) % V, h = (i / (3 * V)) % H;
    const int src = v < Vp ? v : v - S;
    float x = mean_v[((size_t)h * Vp + src) * 3 + c];
    if (v >= Vp) x = flip[c] * x;
    if (mask) x = x * mask[3 * v + c];
    out_v[i] = x;
    const float t = tex[((size_t)h * Vp + src) * 3 + c];
    out_tex[i] = 1.f / (1.f + expf(-t));
}

__global__ __launch_bounds__(256) void mean_shape_backward_kernel(const float* __restrict__ tex, const float* __restrict__ flip,
                                                                  const float* __restrict__ mask, const float* __restrict__ g_v,
                                                                  const float* __restrict__ g_tex, float* __restrict__ gm,
                                                                  float* __restrict__ gt, int R, int H, int Vp, int S)
{
    const int V = Vp + S;
    const int i = blockIdx.x * 256 + threadIdx.x;              // (h, u, c) of the parameters
    if (i >= H * Vp * 3) return;
    const int c = i % 3, u = (i / 3) % Vp, h = i / (3 * Vp);
    const bool mirrored = S > 0 && u >= Vp - S;
    const int v2 = u + S;
    float sv = 0.f, sv2 = 0.f, st = 0.f;
    for (int n = 0; n < R; n++) {                              // images in order: deterministic
        const size_t row = ((size_t)n * H + h) * V;
        if (g_v) {
            sv += g_v[(row + u) * 3 + c];
            if (mirrored) sv2 += g_v[(row + v2) * 3 + c];
        }
        if (g_tex) {
            st += g_tex[(row + u) * 3 + c];
            if (mirrored) st += g_tex[(row + v2) * 3 + c];
        }
    }
    if (gm) {
        float a = mask ? sv * mask[3 * u + c] : sv;
        if (mirrored) a += (mask ? sv2 * mask[3 * v2 + c] : sv2) * flip[c];
        gm[i] = a;
    }
    if (gt) {
        const float sg = 1.f / (1.f + expf(-tex[i]));
        gt[i] = st * sg * (1.f - sg);
    }
}

// ---- observed images for the texture losses (nnutils/mesh_net.py:364-366, 436) ------------------------------------------------
// fg = masks > 0;  out[i] = imgs[i] * fg (object on black),  out[n + i] = 1 - fg + imgs[i] * fg (object on white);
// imgs [n,3,P], masks [n,P] -> out [2n,3,P] (the pair the perceptual network sees, and the two L1 targets).
__global__ __launch_bounds__(256) void obs_pair_kernel(const float* __restrict__ imgs, const float* __restrict__ masks,
                                                       float* __restrict__ out, int n, int P)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)n * 3 * P;
    if (i >= total) return;
    const size_t im = i / ((size_t)3 * P), p = i % P;
    const float fg = masks[im * P + p] > 0.f ? 1.f : 0.f;
    const float obs = imgs[i] * fg;
    out[i] = obs;
    out[total + i] = 1.f - fg + obs;
}

// ---- plumbing around the raster calls -------------------------------------------------------------------------------------------
// Background fill: dst [N, C, P], plane c of every image = values[c] (soft_rasterize.py:50-53 fills soft_colors with the background
// colour and alpha slot 1 before the forward kernel); 16-B stores, one workgroup per 4096 floats of a plane.
struct PlaneValues { float v[LASR_FILL_MAX_PLANES]; };
__global__ __launch_bounds__(256) void fill_planes_kernel(float* __restrict__ dst, PlaneValues V, int C, long long P)
{
    const long long chunks = (P + 4095) / 4096;
    const long long plane = blockIdx.x / chunks;             // n * C + c
    const float val = V.v[plane % C];
    float* __restrict__ d = dst + plane * P;
    const long long base = (blockIdx.x - plane * chunks) * 4096;   // this workgroup's 4096 floats: 4 rounds of 256 lanes x 16 B
    if (base + 4096 <= P && (((size_t)(d + base)) & 15) == 0) {
        const float4 q = make_float4(val, val, val, val);
        float4* o = (float4*)(d + base);
#pragma unroll
        for (int j = 0; j < 4; j++) o[j * 256 + threadIdx.x] = q;          // consecutive lanes, consecutive 16 B: coalesced
    } else {
        for (long long i = base + threadIdx.x; i < base + 4096 && i < P; i += 256) d[i] = val;
    }
}

}  // namespace lasr

using namespace lasr;

extern "C" int lasr_geodesic_forward(const float* m1, const float* m2, float* angle, int n, void* hip_stream)
{
    if (n < 0) return LASR_E_BADARG;
    if (n == 0) return LASR_OK;
    if (!m1 || !m2 || !angle) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_GEODESIC_FORWARD, geodesic_forward_kernel, dim3((n + 255) / 256), dim3(256), 0, m1, m2, angle, n);
    return launch_ok();
}

extern "C" int lasr_geodesic_backward(const float* m1, const float* m2, const float* grad_angle, float* grad_m1, float* grad_m2,
                                      int n, void* hip_stream)
{
    if (n < 0) return LASR_E_BADARG;
    if (n == 0) return LASR_OK;
    if (!m1 || !m2 || !grad_angle || !grad_m1 || !grad_m2) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_GEODESIC_BACKWARD, geodesic_backward_kernel, dim3((n + 255) / 256), dim3(256), 0, m1, m2, grad_angle, grad_m1,
                grad_m2, n);
    return launch_ok();
}

extern "C" int lasr_weighted_means_forward(const float* const* terms, const int* numels, const float* weights, const int* groups,
                                           int n_terms, int n_groups, float* out, void* hip_stream)
{
    if (n_terms < 0 || n_terms > LASR_MEANS_MAX_TERMS || n_groups < 1 || n_groups > LASR_MEANS_MAX_TERMS) return LASR_E_BADARG;
    if (!out || (n_terms > 0 && (!terms || !numels || !weights || !groups))) return LASR_E_BADARG;
    MeansArgs A;
    A.terms = n_terms; A.groups = n_groups;
    for (int t = 0; t < n_terms; t++) {
        if (numels[t] < 0 || groups[t] < 0 || groups[t] >= n_groups || (numels[t] > 0 && !terms[t])) return LASR_E_BADARG;
        A.x[t] = terms[t]; A.n[t] = numels[t]; A.w[t] = weights[t]; A.g[t] = groups[t];
    }
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_WEIGHTED_MEANS, weighted_means_kernel, dim3(1), dim3(1024), 0, A, out);
    return launch_ok();
}

extern "C" int lasr_weighted_means_backward(const int* numels, const float* weights, int n_terms, const float* grad_total,
                                            float* coef, void* hip_stream)
{
    if (n_terms < 0 || n_terms > LASR_MEANS_MAX_TERMS) return LASR_E_BADARG;
    if (n_terms == 0) return LASR_OK;
    if (!numels || !weights || !grad_total || !coef) return LASR_E_BADARG;
    MeansCoef C;
    C.terms = n_terms;
    for (int t = 0; t < n_terms; t++) C.c[t] = numels[t] > 0 ? weights[t] / (float)numels[t] : 0.f;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_WEIGHTED_MEANS, weighted_means_backward_kernel, dim3(1), dim3(64), 0, C, grad_total, coef);
    return launch_ok();
}

extern "C" int lasr_intrinsics_forward(const float* cams, int cam_stride, const float* pp, const float* scale, const float* depth,
                                       const float* ppoint, float* scale_out, float* depth_out, float* ppoint_out, int B, int H,
                                       int K, float half_size, void* hip_stream)
{
    if (B < 0 || H < 1 || K < 1 || cam_stride < 1 || !(half_size > 0.f)) return LASR_E_BADARG;
    if (B == 0) return LASR_OK;
    if (!cams || !pp || !scale || !depth || !ppoint || !scale_out || !depth_out || !ppoint_out) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_INTRINSICS, intrinsics_forward_kernel, dim3(1), dim3(256), 0, cams, cam_stride, pp, scale, depth, ppoint,
                scale_out, depth_out, ppoint_out, B, H, K, half_size);
    return launch_ok();
}

extern "C" int lasr_intrinsics_backward(const float* cams, int cam_stride, const float* grad_scale_out,
                                        const float* grad_depth_out, const float* grad_ppoint_out, float* grad_scale,
                                        float* grad_depth, float* grad_ppoint, int B, int H, int K, void* hip_stream)
{
    if (B < 0 || H < 1 || K < 1 || cam_stride < 1) return LASR_E_BADARG;
    if (B == 0) return LASR_OK;
    if (!cams || !grad_scale_out || !grad_depth_out || !grad_ppoint_out || !grad_scale || !grad_depth || !grad_ppoint)
        return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_INTRINSICS, intrinsics_backward_kernel, dim3(1), dim3(256), 0, cams, cam_stride, grad_scale_out, grad_depth_out,
                grad_ppoint_out, grad_scale, grad_depth, grad_ppoint, B, H, K);
    return launch_ok();
}

extern "C" int lasr_bone_fixup_forward(const float* quat, const float* trans, const float* depth, const float* rest_ts,
                                       float* rmat, float* tmat, int M, int H, int K, void* hip_stream)
{
    if (M < 0 || H < 1 || K < 1 || M % H != 0) return LASR_E_BADARG;
    if (M == 0) return LASR_OK;
    if (!quat || !trans || !depth || !rmat || !tmat || (K > 1 && !rest_ts)) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_BONE_FIXUP, bone_fixup_forward_kernel, dim3((M * K + 255) / 256), dim3(256), 0, quat, trans, depth, rest_ts,
                rmat, tmat, (float*)nullptr, M, H, K);
    return launch_ok();
}

extern "C" int lasr_bone_fixup_pair_forward(const float* quat, const float* trans, const float* depth, const float* rest_ts,
                                            float* rmat, float* tmat, float* pair_angle, int M, int H, int K, void* hip_stream)
{
    if (M < 0 || H < 1 || K < 1 || M % H != 0 || M % 2 != 0) return LASR_E_BADARG;
    if (M == 0) return LASR_OK;
    if (!quat || !trans || !depth || !rmat || !tmat || !pair_angle || (K > 1 && !rest_ts)) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_BONE_FIXUP, bone_fixup_forward_kernel, dim3((M * K + 255) / 256), dim3(256), 0, quat, trans, depth, rest_ts,
                rmat, tmat, pair_angle, M, H, K);
    return launch_ok();
}

extern "C" int lasr_bone_fixup_backward(const float* quat, const float* rest_ts, const float* grad_rmat, const float* grad_tmat,
                                        float* grad_quat, float* grad_trans, float* grad_depth, float* grad_rest, int M, int H,
                                        int K, void* hip_stream)
{
    if (M < 0 || H < 1 || K < 1 || M % H != 0) return LASR_E_BADARG;
    if (M == 0) return LASR_OK;
    if (!quat || !grad_rmat || !grad_tmat || !grad_quat || !grad_trans || !grad_depth || (K > 1 && (!rest_ts || !grad_rest)))
        return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int work = M * K > H * (K - 1) * 3 ? M * K : H * (K - 1) * 3;
    LASR_LAUNCH(K_BONE_FIXUP, bone_fixup_backward_kernel, dim3((work + 255) / 256), dim3(256), 0, quat, rest_ts, grad_rmat,
                grad_tmat, (const float*)nullptr, grad_quat, grad_trans, grad_depth, grad_rest, M, H, K);
    return launch_ok();
}

extern "C" int lasr_bone_fixup_pair_backward(const float* quat, const float* rest_ts, const float* grad_rmat, const float* grad_tmat,
                                             const float* grad_pair_angle, float* grad_quat, float* grad_trans, float* grad_depth,
                                             float* grad_rest, int M, int H, int K, void* hip_stream)
{
    if (M < 0 || H < 1 || K < 1 || M % H != 0 || M % 2 != 0) return LASR_E_BADARG;
    if (M == 0) return LASR_OK;
    if (!quat || !grad_rmat || !grad_tmat || !grad_pair_angle || !grad_quat || !grad_trans || !grad_depth ||
        (K > 1 && (!rest_ts || !grad_rest)))
        return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int work = M * K > H * (K - 1) * 3 ? M * K : H * (K - 1) * 3;
    LASR_LAUNCH(K_BONE_FIXUP, bone_fixup_backward_kernel, dim3((work + 255) / 256), dim3(256), 0, quat, rest_ts, grad_rmat,
                grad_tmat, grad_pair_angle, grad_quat, grad_trans, grad_depth, grad_rest, M, H, K);
    return launch_ok();
}

extern "C" int lasr_project_points_forward(const float* rest_ts, const float* ctl_ts, const float* Rmat, const float* Tmat,
                                           const float* pp, const float* fl, float* proj, int M, int H, int K, void* hip_stream)
{
    if (M < 0 || H < 1 || K < 2 || M % H != 0) return LASR_E_BADARG;
    if (M == 0) return LASR_OK;
    if (!rest_ts || !ctl_ts || !Rmat || !Tmat || !pp || !fl || !proj) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int total = M * 2 * (K - 1);
    LASR_LAUNCH(K_PROJECT_POINTS, project_points_forward_kernel, dim3((total + 255) / 256), dim3(256), 0, rest_ts, ctl_ts, Rmat, Tmat, pp,
                fl, (float4*)proj, M, H, K);
    return launch_ok();
}

extern "C" int lasr_project_points_backward(const float* rest_ts, const float* ctl_ts, const float* Rmat, const float* Tmat,
                                            const float* fl, const float* grad_proj, float* grad_rest, float* grad_ctl, int M, int H,
                                            int K, void* hip_stream)
{
    if (M < 0 || H < 1 || K < 2 || M % H != 0) return LASR_E_BADARG;
    if (!rest_ts || !ctl_ts || !grad_rest || !grad_ctl || (M > 0 && (!Rmat || !Tmat || !fl || !grad_proj))) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int total = H * 2 * (K - 1);
    LASR_LAUNCH(K_PROJECT_POINTS, project_points_backward_kernel, dim3((total + 255) / 256), dim3(256), 0, rest_ts, ctl_ts, Rmat, Tmat, fl,
                (const float4*)grad_proj, grad_rest, grad_ctl, M, H, K);
    return launch_ok();
}

extern "C" int lasr_pose_chain_forward(const float* cams, int cam_stride, const float* pp, const float* scale, const float* depth,
                                       const float* ppoint, const float* quat4, const float* trans, const float* rest_ts,
                                       const float* ctl_ts, float* scale_out, float* depth_out, float* ppoint_out, float* trans_rep,
                                       float* depth_rep, float* rmat, float* tmat, float* pair_angle, float* proj, int B, int H, int K,
                                       float half_size, void* hip_stream)
{
    if (B < 0 || H < 1 || K < 1 || cam_stride < 1 || !(half_size > 0.f)) return LASR_E_BADARG;
    if (B == 0) return LASR_OK;
    if (!cams || !pp || !scale || !depth || !ppoint || !quat4 || !trans || !scale_out || !depth_out || !ppoint_out || !trans_rep ||
        !depth_rep || !rmat || !tmat || (K > 1 && (!rest_ts || !ctl_ts || !proj)))
        return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    PoseChain A{cams, pp, scale, depth, ppoint, quat4, trans, rest_ts, ctl_ts, scale_out, depth_out, ppoint_out, trans_rep, depth_rep,
                rmat, tmat, pair_angle, (float4*)proj, cam_stride, B, H, K, half_size};
    LASR_LAUNCH(K_POSE_CHAIN, pose_chain_forward_kernel, dim3(1), dim3(1024), 0, A);
    return launch_ok();
}

extern "C" int lasr_pose_chain_backward(const float* cams, int cam_stride, const float* quat4, const float* rest_ts, const float* ctl_ts,
                                        const float* rmat, const float* tmat, const float* scale_out, const float* grad_scale_out,
                                        const float* grad_ppoint_out, const float* grad_trans_rep, const float* grad_depth_rep,
                                        const float* grad_rmat, const float* grad_tmat, const float* grad_pair_angle,
                                        const float* grad_proj, float* grad_scale, float* grad_depth, float* grad_ppoint,
                                        float* grad_quat4, float* grad_trans, float* grad_rest, float* grad_ctl, float* scratch, int B,
                                        int H, int K, void* hip_stream)
{
    if (B < 0 || H < 1 || K < 1 || cam_stride < 1) return LASR_E_BADARG;
    if (B == 0) return LASR_OK;
    if (!cams || !quat4 || !grad_rmat || !grad_tmat || !grad_scale || !grad_depth || !grad_ppoint || !grad_quat4 || !grad_trans ||
        !scratch || (K > 1 && (!rest_ts || !ctl_ts || !rmat || !tmat || !scale_out || !grad_rest || !grad_ctl)))
        return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    PoseChainGrad A{cams, quat4, rest_ts, ctl_ts, rmat, tmat, scale_out, grad_scale_out, grad_ppoint_out, grad_trans_rep, grad_depth_rep,
                    grad_rmat, grad_tmat, grad_pair_angle, (const float4*)grad_proj, grad_scale, grad_depth, grad_ppoint, grad_quat4,
                    grad_trans, grad_rest, grad_ctl, scratch, cam_stride, B, H, K};
    const size_t floats = (size_t)2 * B * H * K * 3;
    const int in_lds = floats <= (size_t)POSE_CHAIN_LDS_FLOATS;
    LASR_LAUNCH(K_POSE_CHAIN, pose_chain_backward_kernel, dim3(1), dim3(POSE_CHAIN_THREADS), in_lds ? floats * sizeof(float) : 0, A, in_lds);
    return launch_ok();
}

extern "C" int lasr_chamfer_forward(const float* a, const float* b, float* out, int* nn_ab, int* nn_ba, int N, int P, int Q,
                                    void* hip_stream)
{
    if (N < 0 || P < 1 || Q < 1) return LASR_E_BADARG;
    if (N == 0) return LASR_OK;
    if (!a || !b || !out || !nn_ab || !nn_ba) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_CHAMFER, chamfer_forward_kernel, dim3(N), dim3(256), 0, a, b, out, nn_ab, nn_ba, P, Q);
    return launch_ok();
}

extern "C" int lasr_chamfer_backward(const float* a, const float* b, const int* nn_ab, const int* nn_ba, const float* grad_out,
                                     float* grad_a, float* grad_b, int N, int P, int Q, void* hip_stream)
{
    if (N < 0 || P < 1 || Q < 1) return LASR_E_BADARG;
    if (N == 0) return LASR_OK;
    if (!a || !b || !nn_ab || !nn_ba || !grad_out || !grad_a || !grad_b) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_CHAMFER, chamfer_backward_kernel, dim3(N), dim3(256), 0, a, b, nn_ab, nn_ba, grad_out, grad_a, grad_b, P, Q);
    return launch_ok();
}

extern "C" int lasr_mean_shape_forward(const float* mean_v, const float* tex, const float* flip, const float* mask, float* out_v,
                                       float* out_tex, int R, int H, int Vp, int S, void* hip_stream)
{
    if (R < 0 || H < 0 || Vp < 0 || S < 0 || S > Vp) return LASR_E_BADARG;
    if (R == 0 || H == 0 || Vp == 0) return LASR_OK;
    if (!mean_v || !tex || !out_v || !out_tex || (S > 0 && !flip)) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int total = R * H * (Vp + S) * 3;
    LASR_LAUNCH(K_MEAN_SHAPE, mean_shape_forward_kernel, dim3((total + 255) / 256), dim3(256), 0, mean_v, tex, flip, mask, out_v,
                out_tex, R, H, Vp, S);
    return launch_ok();
}

extern "C" int lasr_mean_shape_backward(const float* tex, const float* flip, const float* mask, const float* grad_v,
                                        const float* grad_tex, float* grad_mean_v, float* grad_tex_param, int R, int H, int Vp,
                                        int S, void* hip_stream)
{
    if (R < 0 || H < 0 || Vp < 0 || S < 0 || S > Vp) return LASR_E_BADARG;
    if (H == 0 || Vp == 0) return LASR_OK;
    if ((grad_mean_v && !grad_v && R > 0) || (grad_tex_param && ((!grad_tex && R > 0) || !tex)) || (S > 0 && !flip))
        return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int total = H * Vp * 3;
    LASR_LAUNCH(K_MEAN_SHAPE, mean_shape_backward_kernel, dim3((total + 255) / 256), dim3(256), 0, tex, flip, mask, grad_v, grad_tex,
                grad_mean_v, grad_tex_param, R, H, Vp, S);
    return launch_ok();
}

extern "C" int lasr_obs_pair(const float* imgs, const float* masks, float* out, int n, int P, void* hip_stream)
{
    if (n < 0 || P < 0) return LASR_E_BADARG;
    if (n == 0 || P == 0) return LASR_OK;
    if (!imgs || !masks || !out) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t total = (size_t)n * 3 * P;
    LASR_LAUNCH(K_OBS_PAIR, obs_pair_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, imgs, masks, out, n, P);
    return launch_ok();
}

// ---- a training batch as ONE row gather (SURVEY section 8 row f2 + the input side of a1) --------------------------------------
// Every distinct frame pair of a sequence lives in HBM as one row of `table` [pairs, W]: the model's batch dictionary of the
// pair (/root/reference/nnutils/train_utils.py:164-178), key after key, each key's segment holding frame t then frame t'.
// A batch of B pairs is `out`, key-major: key k's segment is [B, len_k] (pair-major = the interleaved layout set_input
// produces, train_utils.py:179-180) at out_off[k].  The reference collates B samples on the host and copies ~15 tensors per
// iteration; the torch version of this gather was 15 index_select launches + 15 copies into the HIP graph's static inputs.
struct GatherKeys { int n; long long seg_off[LASR_GATHER_MAX_KEYS], seg_len[LASR_GATHER_MAX_KEYS], out_off[LASR_GATHER_MAX_KEYS];
                    int phase[LASR_GATHER_MAX_KEYS]; };   // phase[k] = (units of the keys before k) % (gridDim.y * 256), from the host

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, long long W, const long long* __restrict__ ids,
                                                          GatherKeys K, float* __restrict__ out, int pairs)
{
    const int b = blockIdx.x;
    long long row = ids[b];
    row = row < 0 ? 0 : (row >= pairs ? pairs - 1 : row);          // ids are produced by the loader itself; never read out of bounds
    const float* __restrict__ src = table + row * W;
    // the row's keys form one run of 16-byte quads (or of floats when a segment is not 16-byte aligned); block y of gridDim.y
    // takes every gridDim.y-th group of 256 of them, across the key boundaries, so the blocks stay evenly loaded whatever the
    // mix of segment lengths (image planes next to 2-float principal points)
    const int step = (int)gridDim.y * 256, first = (int)blockIdx.y * 256 + threadIdx.x;
    for (int k = 0; k < K.n; k++) {
        const long long len = K.seg_len[k];
        const float* __restrict__ s = src + K.seg_off[k];
        float* __restrict__ d = out + K.out_off[k] + (long long)b * len;
        const bool quad = ((K.seg_off[k] | len | K.out_off[k] | W) & 3) == 0;
        const long long units = quad ? len >> 2 : len;
        int i0 = first - K.phase[k];                   // first unit of this key owned by this thread (no 64-bit division here)
        if (i0 < 0) i0 += step;
        if (quad) for (long long i = i0; i < units; i += step) reinterpret_cast<float4*>(d)[i] = reinterpret_cast<const float4*>(s)[i];
        else      for (long long i = i0; i < units; i += step) d[i] = s[i];
    }
}

extern "C" int lasr_gather_rows(const float* table, long long W, int pairs, const long long* ids, int B, int n_keys,
                                const long long* seg_off, const long long* seg_len, const long long* out_off, float* out,
                                void* hip_stream)
{
    if (B < 0 || n_keys < 0 || n_keys > LASR_GATHER_MAX_KEYS || W < 0 || pairs < 0) return LASR_E_BADARG;
    if (B == 0 || n_keys == 0) return LASR_OK;
    if (!table || !ids || !seg_off || !seg_len || !out_off || !out || pairs == 0) return LASR_E_BADARG;
    GatherKeys K;
    K.n = n_keys;
    for (int k = 0; k < n_keys; k++) {
        if (seg_off[k] < 0 || seg_len[k] < 0 || out_off[k] < 0 || seg_off[k] + seg_len[k] > W) return LASR_E_BADARG;
        K.seg_off[k] = seg_off[k]; K.seg_len[k] = seg_len[k]; K.out_off[k] = out_off[k];
    }
    hipStream_t st = (hipStream_t)hip_stream;
    // one 16-byte quad per thread and pass; at least ~4 blocks per CU when the rows are long enough (rounds 1-4 capped the
    // grid at 64 blocks per row: 22 us for the 6.6 MB batch of the spot3 configuration, 0.3 TB/s)
    long long quads = 0;
    for (int k = 0; k < n_keys; k++) quads += (seg_len[k] + 3) / 4;
    long long want = (quads + 256 * 4 - 1) / (256 * 4);                  // ~4 quads per thread
    const long long cap = (4 * 256 + B - 1) / B > 1 ? (4 * 256 + B - 1) / B : 1;
    want = want < 1 ? 1 : (want > cap ? cap : want);
    const unsigned chunks = (unsigned)want;
    long long base = 0;
    for (int k = 0; k < n_keys; k++) {
        const bool quad = ((seg_off[k] | seg_len[k] | out_off[k] | W) & 3) == 0;
        K.phase[k] = (int)(base % ((long long)chunks * 256));
        base += quad ? seg_len[k] >> 2 : seg_len[k];
    }
    LASR_LAUNCH(K_GATHER_ROWS, gather_rows_kernel, dim3((unsigned)B, chunks), dim3(256), 0, table, W, ids, K, out, pairs);
    return launch_ok();
}

extern "C" int lasr_fill_planes(float* dst, const float* values, int n_values, int N, long long plane_elems, void* hip_stream)
{
    if (N < 0 || plane_elems < 0 || n_values < 1 || n_values > LASR_FILL_MAX_PLANES) return LASR_E_BADARG;
    if (N == 0 || plane_elems == 0) return LASR_OK;
    if (!dst || !values) return LASR_E_BADARG;
    PlaneValues V;
    for (int k = 0; k < n_values; k++) V.v[k] = values[k];
    hipStream_t st = (hipStream_t)hip_stream;
    const long long blocks = ((plane_elems + 4095) / 4096) * (long long)N * n_values;
    if (blocks > 0x7fffffffLL) return LASR_E_BADARG;
    LASR_LAUNCH(K_FILL_PLANES, fill_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, dst, V, n_values, plane_elems);
    return launch_ok();
}
