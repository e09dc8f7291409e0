// fused.hip -- the small operators between the kernels of ops.hip / sr_raster.hip (include/lasr_ops.h), gfx950 / wave64.
//
// In the reference each of these is a chain of 20-100 eager elementwise launches per call (forward and again in
// autograd's backward); at LASR's sizes every one of them is launch-bound, so each chain becomes one kernel pair:
//   flow reprojection   nnutils/mesh_net.py:93-104   (background fill, two pinhole reprojections, difference, detach rules)
//   quaternion -> R     kornia 0.5.3 quaternion_to_rotation_matrix, call sites mesh_net.py:232,250,265, net_blocks.py:359
//   GMM skin weights    nnutils/mesh_net.py:264-271   (Mahalanobis distance to every bone, softmax over bones)
//   flatten loss        third_party/ext_nnutils/loss_utils.py:110-152   (dihedral cosine over interior edges)
//   face gather         third_party/softras/soft_renderer/functional/face_vertices.py:4-22 (+ index_add_ in backward)
//   nearest neighbours  third_party/chamfer3D/chamfer3D.cu (idx1) and pytorch3d chamfer_distance, mesh_net.py:477,503
//   point <-> mesh      pytorch3d.loss.point_mesh_face_distance, mesh_net.py:470-471
//   perceptual reduce   third_party/PerceptualSimilarity/util/util.py:71-83 (feature normalisation, cosine, spatial mean)
// Reductions are deterministic (fixed tree inside a block, fixed-order fold of block partials; no float atomics).
#include <hip/hip_runtime.h>

#include "../../include/lasr_ops.h"
#include "ops_common.h"
#include "mesh_losses.h"

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
                                                                     unsigned char* __restrict__ bg, int P, int nch,
                                                                     long long in_stride)
{
    const int n = blockIdx.x;
    const float* q = px + (size_t)n * in_stride;     // 7 P for a [N,7,P] render; larger when the planes sit inside a wider render
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
                                                                      float* __restrict__ part, int P, int nch,
                                                                      long long in_stride, int out_planes)
{
    __shared__ float red[4];
    const int n = blockIdx.x;
    const float* q = px + (size_t)n * in_stride;
    float* g = gpx + (size_t)n * out_planes * P;     // 7 planes (with the alpha plane's zeros) or the 6 position planes only
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
        if (out_planes == 7) g[6 * (size_t)P + p] = 0.f;
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

// ===========================================================================
// Quaternion (x,y,z,w) -> rotation matrix with normalisation (eps 1e-12 like F.normalize).
// ===========================================================================
__global__ __launch_bounds__(256) void quat_forward_kernel(const float* __restrict__ q, float* __restrict__ R, int M)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    float m[9];
    quat_matrix(load_quat(q + 4 * (size_t)i), m);
#pragma unroll
    for (int k = 0; k < 9; k++) R[9 * (size_t)i + k] = m[k];
}

__global__ __launch_bounds__(256) void quat_backward_kernel(const float* __restrict__ q, const float* __restrict__ gR,
                                                            float* __restrict__ gq, int M)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    float g[9], o[4];
#pragma unroll
    for (int k = 0; k < 9; k++) g[k] = gR[9 * (size_t)i + k];
    quat_matrix_backward(load_quat(q + 4 * (size_t)i), g, o);
#pragma unroll
    for (int k = 0; k < 4; k++) gq[4 * (size_t)i + k] = o[k];
}

// ===========================================================================
// GMM skinning weights, mesh_net.py:264-271:
//   skin[h,k,v] = softmax_k( -10 * sum_d exp(log_ctl[h,k,d]) * ((ctl_ts[h,k] - verts[h,v]) R(ctl_rs[h,k]))_d^2 )
// ctl_ts, log_ctl [H*J,3], ctl_rs [H*J,4] (x,y,z,w), verts [H,V,3] (constant) -> skin [H,J,V].
// ===========================================================================
constexpr int SKIN_MAX_BONES = 64;

struct Bone { float t[3]; float R[9]; float w[3]; };

__device__ __forceinline__ float bone_logit(const Bone& b, float vx, float vy, float vz, float* r)
{
    const float d0 = b.t[0] - vx, d1 = b.t[1] - vy, d2 = b.t[2] - vz;
    r[0] = d0 * b.R[0] + d1 * b.R[3] + d2 * b.R[6];
    r[1] = d0 * b.R[1] + d1 * b.R[4] + d2 * b.R[7];
    r[2] = d0 * b.R[2] + d1 * b.R[5] + d2 * b.R[8];
    return -10.f * (b.w[0] * (r[0] * r[0]) + b.w[1] * (r[1] * r[1]) + b.w[2] * (r[2] * r[2]));
}

__device__ __forceinline__ void load_bones(Bone* bones, const float* ts, const float* rs, const float* lc, int h, int J)
{
    for (int k = threadIdx.x; k < J; k += blockDim.x) {
        const size_t i = (size_t)h * J + k;
        Bone b;
        for (int d = 0; d < 3; d++) { b.t[d] = ts[3 * i + d]; b.w[d] = expf(lc[3 * i + d]); }
        quat_matrix(load_quat(rs + 4 * i), b.R);
        bones[k] = b;
    }
    __syncthreads();
}

// One wave per block (V = 642 x 8 hypotheses: 88 blocks instead of 24 -- the launch is a latency chain, not bandwidth); the
// logits of all bones are evaluated ONCE into registers when they fit (J <= MAXJ; LASR: 20 or 35 part bones), the max / sum /
// normalise passes then run on registers.  Same expressions in the same order as the three-pass form (MAXJ = 0, kept for
// larger bone counts): bit-identical weights.
template <int MAXJ>
__global__ __launch_bounds__(64) void skin_forward_kernel(const float* __restrict__ ts, const float* __restrict__ rs,
                                                          const float* __restrict__ lc, const float* __restrict__ verts,
                                                          float* __restrict__ skin, int V, int J)
{
    __shared__ Bone bones[SKIN_MAX_BONES];
    const int h = blockIdx.y, v = blockIdx.x * 64 + threadIdx.x;
    load_bones(bones, ts, rs, lc, h, J);
    if (v >= V) return;
    const float* p = verts + ((size_t)h * V + v) * 3;
    const float vx = p[0], vy = p[1], vz = p[2];
    float r[3], mx = -INFINITY;
    if (MAXJ > 0) {
        float lg[MAXJ > 0 ? MAXJ : 1];
#pragma unroll
        for (int k = 0; k < MAXJ; k++) if (k < J) { lg[k] = bone_logit(bones[k], vx, vy, vz, r); mx = fmaxf(mx, lg[k]); }
        float z = 0.f;
#pragma unroll
        for (int k = 0; k < MAXJ; k++) if (k < J) { lg[k] = expf(lg[k] - mx); z += lg[k]; }
#pragma unroll
        for (int k = 0; k < MAXJ; k++) if (k < J) skin[((size_t)h * J + k) * V + v] = lg[k] / z;
        return;
    }
    for (int k = 0; k < J; k++) mx = fmaxf(mx, bone_logit(bones[k], vx, vy, vz, r));
    float z = 0.f;
    for (int k = 0; k < J; k++) z += expf(bone_logit(bones[k], vx, vy, vz, r) - mx);
    for (int k = 0; k < J; k++) skin[((size_t)h * J + k) * V + v] = expf(bone_logit(bones[k], vx, vy, vz, r) - mx) / z;
}

// backward: one block per (h, bone): sums over the vertices, then the quaternion / log-scale chain.  The softmax Jacobian's
// common term dot[h,v] = sum_k skin * gskin is formed by each block for the vertices it walks (J coalesced loads per vertex
// from L2, bones in order: the value the former separate skin_backward_dot_kernel launch wrote, 10 us of launch saved).
__global__ __launch_bounds__(256) void skin_backward_kernel(const float* __restrict__ ts, const float* __restrict__ rs,
                                                            const float* __restrict__ lc, const float* __restrict__ verts,
                                                            const float* __restrict__ skin, const float* __restrict__ gskin,
                                                            float* __restrict__ gts,
                                                            float* __restrict__ grs, float* __restrict__ glc, int V, int J)
{
    __shared__ float red[4];
    const int hk = blockIdx.x, h = hk / J;
    Bone b;
    for (int d = 0; d < 3; d++) { b.t[d] = ts[3 * (size_t)hk + d]; b.w[d] = expf(lc[3 * (size_t)hk + d]); }
    const Quat q = load_quat(rs + 4 * (size_t)hk);
    quat_matrix(q, b.R);
    float acc[15];                                   // g_ts[3] | g_lc[3] | g_R[9]
#pragma unroll
    for (int i = 0; i < 15; i++) acc[i] = 0.f;
    constexpr int VPT = 4;                           // vertices per thread and pass: 2 x VPT x 5 independent loads in flight
    for (int vb = threadIdx.x; vb < V; vb += 256 * VPT) {
        float dot[VPT];
        int vc[VPT];                                 // clamped vertex: every load is unconditional, so the compiler keeps all of
#pragma unroll                                       // an unrolled round in flight (guarded loads compiled to a branch + wait each)
        for (int u = 0; u < VPT; u++) { dot[u] = 0.f; vc[u] = min(vb + 256 * u, V - 1); }
#pragma unroll 5
        for (int k = 0; k < J; k++) {                // bones in order: the value of the former dot kernel
            const size_t row = ((size_t)h * J + k) * V;
#pragma unroll
            for (int u = 0; u < VPT; u++) dot[u] += skin[row + vc[u]] * gskin[row + vc[u]];
        }
#pragma unroll
        for (int u = 0; u < VPT; u++) {
            const int v = vb + 256 * u;
            if (v >= V) continue;
            const float* p = verts + ((size_t)h * V + v) * 3;
            const size_t i = (size_t)hk * V + v;
            const float ge = -10.f * (skin[i] * (gskin[i] - dot[u]));                     // d loss / d (sum_d w_d r_d^2)
            float r[3];
            const float d0 = b.t[0] - p[0], d1 = b.t[1] - p[1], d2 = b.t[2] - p[2];
            r[0] = d0 * b.R[0] + d1 * b.R[3] + d2 * b.R[6];
            r[1] = d0 * b.R[1] + d1 * b.R[4] + d2 * b.R[7];
            r[2] = d0 * b.R[2] + d1 * b.R[5] + d2 * b.R[8];
            float gr[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                acc[3 + d] += ge * b.w[d] * (r[d] * r[d]);           // d/d log_ctl: w = exp(log_ctl)
                gr[d] = 2.f * ge * b.w[d] * r[d];
            }
            acc[0] += gr[0] * b.R[0] + gr[1] * b.R[1] + gr[2] * b.R[2];
            acc[1] += gr[0] * b.R[3] + gr[1] * b.R[4] + gr[2] * b.R[5];
            acc[2] += gr[0] * b.R[6] + gr[1] * b.R[7] + gr[2] * b.R[8];
#pragma unroll
            for (int c = 0; c < 3; c++) { acc[6 + c] += d0 * gr[c]; acc[9 + c] += d1 * gr[c]; acc[12 + c] += d2 * gr[c]; }
        }
    }
#pragma unroll
    for (int i = 0; i < 15; i++) acc[i] = block_sum(acc[i], red);
    if (threadIdx.x == 0) {
        float gq[4];
        quat_matrix_backward(q, acc + 6, gq);
        for (int d = 0; d < 3; d++) { gts[3 * (size_t)hk + d] = acc[d]; glc[3 * (size_t)hk + d] = acc[3 + d]; }
        for (int d = 0; d < 4; d++) grs[4 * (size_t)hk + d] = gq[d];
    }
}

// ===========================================================================
// Flatten loss, ext_nnutils/loss_utils.py:110-152: for every listed interior edge (v0,v1) with opposite vertices v2, v3
//   loss[n] = sum_e (cos_e + 1)^2, cos_e = the cosine between the components of (v2-v0), (v3-v0) orthogonal to (v1-v0)
// quads [E,4] int32 (v0,v1,v2,v3), x [N,V,3] -> loss [N].
// ===========================================================================
__global__ __launch_bounds__(256) void flatten_forward_kernel(const float* __restrict__ x, const int* __restrict__ quads,
                                                              float* __restrict__ loss, int V, int E)
{
    __shared__ float red[4];
    const int n = blockIdx.x;
    const float acc = flatten_forward_block(x + (size_t)n * V * 3, quads, E, red);
    if (threadIdx.x == 0) loss[n] = acc;
}

// per-edge gradients w.r.t. its four vertices -> gedge [N,E,4,3]
__global__ __launch_bounds__(256) void flatten_backward_edge_kernel(const float* __restrict__ x, const int* __restrict__ quads,
                                                                    const float* __restrict__ gloss, float* __restrict__ gedge,
                                                                    int V, int E)
{
    const int n = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    float g[12];
    flatten_edge_gradient(x + (size_t)n * V * 3, quads + 4 * (size_t)e, gloss[n], g);
    float* o = gedge + ((size_t)n * E + e) * 12;
#pragma unroll
    for (int d = 0; d < 12; d++) o[d] = g[d];
}

// vertex-centric gather of the edge gradients: inc_ptr [V+1], inc [nnz] = edge * 4 + slot, ascending (deterministic)
__global__ __launch_bounds__(256) void flatten_backward_vertex_kernel(const float* __restrict__ gedge, const int* __restrict__ inc_ptr,
                                                                      const int* __restrict__ inc, float* __restrict__ gx,
                                                                      int V, int E)
{
    const int n = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const float* g = gedge + (size_t)n * E * 12;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int i = inc_ptr[v]; i < inc_ptr[v + 1]; i++) {
        const float* o = g + 3 * (size_t)inc[i];
        a0 += o[0]; a1 += o[1]; a2 += o[2];
    }
    float* out = gx + ((size_t)n * V + v) * 3;
    out[0] = a0; out[1] = a1; out[2] = a2;
}

// ===========================================================================
// Per-face gather of per-vertex attributes, face_vertices.py:4-22: out[n,f,c,:] = attr[n, faces[n,f,c], :].
// Backward: a vertex-centric sum without a precomputed incidence structure and without float atomics.  One block owns
// GATHER_VERTS vertices of one mesh: all threads scan the mesh's 3F corner indices once and file the corners that point into the
// block's vertex range into per-vertex LDS lists; each list (<= GATHER_CAP entries, mesh valence) is sorted so that the
// summation order -- ascending corner index -- does not depend on the order the scan happened to fill it in.
// ===========================================================================
// GV vertices per block: 64 when the launch needs the blocks to fill the chip (LASR's 16 meshes of 642 vertices: 13 us vs 16),
// 256 when there are plenty (256 meshes of 1212 vertices: 5 instead of 19 scans of each mesh's corner list, 27 us vs 51)
constexpr int GATHER_CAP = 30;

__global__ __launch_bounds__(256) void face_gather_forward_kernel(const float* __restrict__ attr, const long long* __restrict__ faces,
                                                                  float* __restrict__ out, int V, int F3, int C)
{
    const int n = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= F3) return;
    const long long vi = faces[(size_t)n * F3 + c];
    const float* src = attr + ((size_t)n * V + (size_t)vi) * C;
    float* dst = out + ((size_t)n * F3 + c) * C;
    for (int k = 0; k < C; k++) dst[k] = src[k];
}

template <int GATHER_VERTS, int BT = 256>
__global__ __launch_bounds__(BT) void face_gather_backward_kernel(const float* __restrict__ gout, const long long* __restrict__ faces,
                                                                  float* __restrict__ gattr, int V, int F3, int C)
{
    __shared__ int cnt[GATHER_VERTS];
    __shared__ int lists[GATHER_VERTS][GATHER_CAP];
    const int n = blockIdx.y, v0 = blockIdx.x * GATHER_VERTS, tid = threadIdx.x;
    const long long* fn = faces + (size_t)n * F3;
    if (tid < GATHER_VERTS) cnt[tid] = 0;
    __syncthreads();
    for (int c = tid; c < F3; c += BT) {
        const long long d = fn[c] - v0;
        if (d >= 0 && d < GATHER_VERTS) {
            const int slot = atomicAdd(&cnt[(int)d], 1);
            if (slot < GATHER_CAP) lists[(int)d][slot] = c;
        }
    }
    __syncthreads();
    if (tid < GATHER_VERTS && cnt[tid] <= GATHER_CAP) {           // insertion sort of a handful of corner ids
        int* l = lists[tid];
        const int m = cnt[tid];
        for (int i = 1; i < m; i++) {
            const int key = l[i];
            int j = i - 1;
            while (j >= 0 && l[j] > key) { l[j + 1] = l[j]; j--; }
            l[j + 1] = key;
        }
    }
    __syncthreads();
    const float* g = gout + (size_t)n * F3 * C;
    for (int item = tid; item < GATHER_VERTS * C; item += BT) {
        const int vi = item / C, ch = item - vi * C;
        if (v0 + vi >= V) continue;
        float acc = 0.f;
        const int m = cnt[vi];
        if (m <= GATHER_CAP) {
            for (int i = 0; i < m; i++) acc += g[(size_t)lists[vi][i] * C + ch];
        } else {                                                   // valence beyond the list: ordered rescan
            for (int c = 0; c < F3; c++)
                if (fn[c] == v0 + vi) acc += g[(size_t)c * C + ch];
        }
        gattr[((size_t)n * V + v0 + vi) * C + ch] = acc;
    }
}

// The same sums over a CSR vertex -> corner incidence the caller built once for the connectivity (fused_ops.face_incidence: corner
// ids 3 f + c grouped by vertex, ascending inside a vertex = the order of the sorted lists above, so the result has the same
// bits): no scan of the face tensor, no LDS filing.  One thread per (vertex, channel); a round keeps four gathers in flight
// (clamped indices: a guarded load would cost a branch and a full wait each).
__global__ __launch_bounds__(256) void face_gather_backward_csr_kernel(const float* __restrict__ gout, const int* __restrict__ inc_ptr,
                                                                       const int* __restrict__ inc, float* __restrict__ gattr,
                                                                       int V, int F3, int C, int shared)
{
    const int n = blockIdx.y, item = blockIdx.x * 256 + threadIdx.x;
    if (item >= V * C) return;
    const int v = item / C, ch = item - v * C;
    const int* ptr = inc_ptr + (shared ? 0 : (size_t)n * (V + 1));
    const int* lst = inc + (shared ? 0 : (size_t)n * F3);
    const float* g = gout + (size_t)n * F3 * C + ch;
    const int b = ptr[v], e = ptr[v + 1];
    float acc = 0.f;
    for (int i = b; i < e; i += 4) {
        const int last = e - 1;
        const int c0 = lst[i], c1 = lst[min(i + 1, last)], c2 = lst[min(i + 2, last)], c3 = lst[min(i + 3, last)];
        const float x0 = g[(size_t)c0 * C], x1 = g[(size_t)c1 * C], x2 = g[(size_t)c2 * C], x3 = g[(size_t)c3 * C];
        acc += x0;
        if (i + 1 < e) acc += x1;
        if (i + 2 < e) acc += x2;
        if (i + 3 < e) acc += x3;
    }
    gattr[(size_t)n * V * C + item] = acc;
}

// ===========================================================================
// Brute-force nearest neighbour: for each a[b,p] the nearest b[b,q] (squared distance, lowest index on ties).
// The sets are the <= 35 control points (Chamfer term, mesh_net.py:503) or the V <= ~1.3k mesh vertices (:477).
// ===========================================================================
// A block owns 16 points x 16 sub-ranges of the target set (one thread each): 81 blocks for the 1282-vertex mesh of the camel
// schedule instead of the 6 of a thread-per-point scan over all of b (55 us per call there).  The sub-ranges are ascending index
// ranges and both the scan and the merge keep the first minimum, so the lowest index still wins ties.
__global__ __launch_bounds__(256) void nearest_point_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            float* __restrict__ d2, int* __restrict__ idx, int P, int Q)
{
    const int n = blockIdx.y, p = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const bool live = p < P;
    float x = 0.f, y = 0.f, z = 0.f;
    if (live) { const float* s = a + ((size_t)n * P + p) * 3; x = s[0]; y = s[1]; z = s[2]; }
    const int per = (Q + 15) >> 4, q0 = sub * per, q1 = min(Q, q0 + per);
    const float* __restrict__ bn = b + (size_t)n * Q * 3;
    float best = INFINITY; int arg = 0x7fffffff;
    // four candidates per round with all twelve loads in flight (clamped indices: a repeated last candidate can never win the strict
    // `<` against itself); one at a time, each round waited for its own loads: 24 us for 1282 x 1282 points
    for (int q = q0; q < q1; q += 4) {
        const int last = q1 - 1;
        const int i0 = q, i1 = min(q + 1, last), i2 = min(q + 2, last), i3 = min(q + 3, last);
        const float ax = bn[3 * i0], ay = bn[3 * i0 + 1], az = bn[3 * i0 + 2];
        const float bx = bn[3 * i1], by = bn[3 * i1 + 1], bz = bn[3 * i1 + 2];
        const float cx = bn[3 * i2], cy = bn[3 * i2 + 1], cz = bn[3 * i2 + 2];
        const float ex = bn[3 * i3], ey = bn[3 * i3 + 1], ez = bn[3 * i3 + 2];
        float dx = x - ax, dy = y - ay, dz = z - az;
        float d = dx * dx + dy * dy + dz * dz;
        if (d < best) { best = d; arg = i0; }
        dx = x - bx; dy = y - by; dz = z - bz; d = dx * dx + dy * dy + dz * dz;
        if (d < best) { best = d; arg = i1; }
        dx = x - cx; dy = y - cy; dz = z - cz; d = dx * dx + dy * dy + dz * dz;
        if (d < best) { best = d; arg = i2; }
        dx = x - ex; dy = y - ey; dz = z - ez; d = dx * dx + dy * dy + dz * dz;
        if (d < best) { best = d; arg = i3; }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {                       // merge the 16 sub-ranges of a point (16 consecutive lanes)
        const float ob = __shfl_xor(best, o);
        const int oa = __shfl_xor(arg, o);
        if (ob < best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (live && sub == 0) { d2[(size_t)n * P + p] = best; idx[(size_t)n * P + p] = arg == 0x7fffffff ? 0 : arg; }
}

// ===========================================================================
// Point <-> triangle-mesh distance (pytorch3d point_mesh_face_distance semantics, squared Euclidean):
//   loss = mean_b [ mean_p min_f d2(p, tri_f) + mean_f min_p d2(p, tri_f) ]
// The closest point on a triangle is the interior projection when it falls inside, otherwise the nearest of the
// three clamped edge projections (Ericson, Real-Time Collision Detection 5.1.5, written as a min over candidates so
// that degenerate triangles fall back to their edges).  bary = barycentric weights of the closest point.
// ===========================================================================
struct Closest { float d2; float w[3]; };

__device__ __forceinline__ float dot3(const float* u, const float* v) { return u[0] * v[0] + u[1] * v[1] + u[2] * v[2]; }

__device__ __forceinline__ Closest point_triangle(const float* p, const float* a, const float* b, const float* c)
{
    float ab[3], ac[3], ap[3], bp[3], cp[3];
#pragma unroll
    for (int d = 0; d < 3; d++) { ab[d] = b[d] - a[d]; ac[d] = c[d] - a[d]; ap[d] = p[d] - a[d]; bp[d] = p[d] - b[d]; cp[d] = p[d] - c[d]; }
    const float d1 = dot3(ab, ap), d2 = dot3(ac, ap), d3 = dot3(ab, bp), d4 = dot3(ac, bp), d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    const float va = d3 * d6 - d5 * d4, vb = d5 * d2 - d1 * d6, vc = d1 * d4 - d3 * d2;
    const float eps = 1e-12f;
    Closest best;
    best.d2 = INFINITY; best.w[0] = 1.f; best.w[1] = 0.f; best.w[2] = 0.f;
    auto consider = [&](float w0, float w1, float w2) {
        float e = 0.f;
#pragma unroll
        for (int d = 0; d < 3; d++) { const float q = w0 * a[d] + w1 * b[d] + w2 * c[d] - p[d]; e += q * q; }
        if (e < best.d2) { best.d2 = e; best.w[0] = w0; best.w[1] = w1; best.w[2] = w2; }
    };
    if (va >= 0.f && vb >= 0.f && vc >= 0.f) {
        const float den = fmaxf(va + vb + vc, eps);
        const float v = vb / den, w = vc / den;
        consider(1.f - v - w, v, w);
    }
    const float tab = fminf(fmaxf(d1 / fmaxf(d1 - d3, eps), 0.f), 1.f);
    const float tac = fminf(fmaxf(d2 / fmaxf(d2 - d6, eps), 0.f), 1.f);
    const float tbc = fminf(fmaxf((d4 - d3) / fmaxf((d4 - d3) + (d5 - d6), eps), 0.f), 1.f);
    consider(1.f - tab, tab, 0.f);
    consider(1.f - tac, 0.f, tac);
    consider(0.f, 1.f - tbc, tbc);
    return best;
}

__device__ __forceinline__ void load_tri(const float* verts, const long long* f, float* a, float* b, float* c)
{
#pragma unroll
    for (int d = 0; d < 3; d++) { a[d] = verts[3 * f[0] + d]; b[d] = verts[3 * f[1] + d]; c[d] = verts[3 * f[2] + d]; }
}

// Forward, two launches.  Rounds 1-4 ran one thread per point over ALL faces and one thread per face over ALL points: 6 and 10
// workgroups for the 1282-point / 2560-face mesh of the camel schedule's last stage, 3.4 ms per step.  Now the other set is cut
// into chunks of PMF tile size and a block owns (256 points, one face chunk): hundreds of blocks, the same candidate order inside a
// chunk; per-chunk minima (of every point over the chunk's faces, of every face over the block's points) go to scratch and
// pmf_fold_kernel takes the minimum over the chunks in chunk order with a strict `<` -- the lowest index wins ties exactly as the
// single scan did.
// minimum over the wave in lane 63: six DPP steps in registers (a shuffle tree through the LDS crossbar puts ~600 cycles of
// dependent latency into every face of the pair loop below)
__device__ __forceinline__ float wave_min_to_lane63(float v)
{
    const int inf = 0x7f800000;
#define LASR_DPP_MIN(ctrl, rmask, bmask) \
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(v), ctrl, rmask, bmask, false)))
    LASR_DPP_MIN(0x111, 0xf, 0xf);   // row_shr:1
    LASR_DPP_MIN(0x112, 0xf, 0xf);   // row_shr:2
    LASR_DPP_MIN(0x114, 0xf, 0xe);   // row_shr:4
    LASR_DPP_MIN(0x118, 0xf, 0xc);   // row_shr:8
    LASR_DPP_MIN(0x142, 0xa, 0xf);   // row_bcast:15
    LASR_DPP_MIN(0x143, 0xc, 0xf);   // row_bcast:31
#undef LASR_DPP_MIN
    return v;
}

__host__ __device__ inline int pmf_tile(int n) { int t = 32; while ((n + t - 1) / t > 64) t *= 2; return t; }

struct PmfArgs {
    const float* verts; const long long* faces; const float* pts;
    float* part_pd; int* part_pa;       // [N][FC][P]  per face chunk: nearest face of every point
    float* part_fd; int* part_fa;       // [N][PC][F]  per point chunk: nearest point of every face
    int V, F, P, FT, PT, FC, PC;
};

// (Round 5, late: every (point, face) pair is evaluated ONCE.  A block owns 256 points and one chunk of faces; a thread keeps the
// nearest face of its point as before, and for every face of the chunk the block also reduces the 256 distances it has just computed
// to the nearest point among its 256 -- a wave minimum, the lowest lane among equals (= the lowest point index), then the four waves
// in order -- which is that face's partial for this block of points.  The separate faces-against-points blocks, which evaluated
// every pair a second time, are gone: point_triangle() is ~150 instructions, the wave minimum ~20.)
__global__ __launch_bounds__(256) void pmf_partial_kernel(PmfArgs A)
{
    __shared__ float tile[128 * 9];
    __shared__ float wmin_d[4][128];
    __shared__ int wmin_p[4][128];
    const int n = blockIdx.z, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* vn = A.verts + (size_t)n * A.V * 3;
    const int p = blockIdx.x * 256 + tid;
    const int fbeg = blockIdx.y * A.FT, fend = min(A.F, fbeg + A.FT);
    float q[3] = {0.f, 0.f, 0.f};
    if (p < A.P) { const float* s = A.pts + ((size_t)n * A.P + p) * 3; q[0] = s[0]; q[1] = s[1]; q[2] = s[2]; }
    float best = INFINITY; int barg = fbeg;
    for (int f0 = fbeg; f0 < fend; f0 += 128) {
        const int m = min(128, fend - f0);
        __syncthreads();
        for (int i = tid; i < m * 9; i += 256) {
            const int f = i / 9, r = i - 9 * f;
            tile[i] = vn[3 * A.faces[(size_t)(f0 + f) * 3 + r / 3] + r % 3];
        }
        __syncthreads();
        for (int j = 0; j < m; j++) {
            const Closest c = point_triangle(q, tile + 9 * j, tile + 9 * j + 3, tile + 9 * j + 6);
            if (c.d2 < best) { best = c.d2; barg = f0 + j; }
            // the face's nearest point among this wave's 64: minimum, then the lowest lane that has it
            const float mine = p < A.P ? c.d2 : INFINITY;
            const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_min_to_lane63(mine)), 63));
            const unsigned long long eq = __ballot(mine == w);
            if (lane == 0) { wmin_d[wave][j] = w; wmin_p[wave][j] = blockIdx.x * 256 + wave * 64 + (eq ? __ffsll((long long)eq) - 1 : 0); }
        }
        __syncthreads();
        if (tid < m) {                                              // the four waves in order, strict `<`: the lowest point index wins ties
            float d = wmin_d[0][tid]; int a = wmin_p[0][tid];
#pragma unroll
            for (int k = 1; k < 4; k++)
                if (wmin_d[k][tid] < d) { d = wmin_d[k][tid]; a = wmin_p[k][tid]; }
            const size_t o = ((size_t)n * A.PC + blockIdx.x) * A.F + f0 + tid;
            A.part_fd[o] = d; A.part_fa[o] = a;
        }
    }
    if (p < A.P) {
        const size_t o = ((size_t)n * A.FC + blockIdx.y) * A.P + p;
        A.part_pd[o] = best; A.part_pa[o] = barg;
    }
}

__global__ __launch_bounds__(256) void pmf_fold_kernel(PmfArgs A, float* __restrict__ dmin_p, int* __restrict__ arg_p,
                                                       float* __restrict__ dmin_f, int* __restrict__ arg_f)
{
    const int n = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    const bool face_side = blockIdx.y == 1;
    const int count = face_side ? A.F : A.P, chunks = face_side ? A.PC : A.FC;
    if (i >= count) return;
    const float* pd = face_side ? A.part_fd : A.part_pd;
    const int* pa = face_side ? A.part_fa : A.part_pa;
    float best = INFINITY; int barg = 0;
    // chunk order + strict `<`: the lowest index wins ties.  Four chunks per round, values AND indices loaded unconditionally (a
    // conditional index load is a branch and a full wait per chunk: 40 chunks x an L2 round trip = 12 us per launch)
    for (int k = 0; k < chunks; k += 4) {
        float d[4]; int a[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t o = ((size_t)n * chunks + min(k + u, chunks - 1)) * count + i;
            d[u] = pd[o]; a[u] = pa[o];
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (d[u] < best) { best = d[u]; barg = a[u]; }            // (a repeated last chunk cannot win the strict `<` against itself)
    }
    (face_side ? dmin_f : dmin_p)[(size_t)n * count + i] = best;
    (face_side ? arg_f : arg_p)[(size_t)n * count + i] = barg;
}

// Backward.  A WAVE per face (per point): its lanes scan the other set's nearest-index table 64 entries at a time and the matches
// of a ballot are processed in ascending order by the whole wave (uniform arithmetic, lane 0 stores) -- the contributions and
// their order are those of the single-thread scans of rounds 1-4 (own nearest element first, then ascending index), which took
// 10 and 6 workgroups and 0.8 ms per step at the camel sizes.
// Face side: gtri[n,f,corner,:] = its own nearest point (weight gf) + every point whose nearest face it is (weight gp).
__global__ __launch_bounds__(256) void pmf_backward_face_kernel(const float* __restrict__ verts, const long long* __restrict__ faces,
                                                                const float* __restrict__ pts, const int* __restrict__ arg_p,
                                                                const int* __restrict__ arg_f, float gp, float gf,
                                                                float* __restrict__ gtri, int V, int F, int P)
{
    const int n = blockIdx.y, f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (f >= F) return;
    float a[3], b[3], c[3], acc[9];
    load_tri(verts + (size_t)n * V * 3, faces + (size_t)f * 3, a, b, c);
#pragma unroll
    for (int i = 0; i < 9; i++) acc[i] = 0.f;
    auto add = [&](int p, float wgt) {
        const float* q = pts + ((size_t)n * P + p) * 3;
        const Closest cl = point_triangle(q, a, b, c);
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const float r = q[d] - (cl.w[0] * a[d] + cl.w[1] * b[d] + cl.w[2] * c[d]);        // p - closest point
            acc[d] -= 2.f * wgt * cl.w[0] * r; acc[3 + d] -= 2.f * wgt * cl.w[1] * r; acc[6 + d] -= 2.f * wgt * cl.w[2] * r;
        }
    };
    if (P > 0) add(arg_f[(size_t)n * F + f], gf);
    for (int p0 = 0; p0 < P; p0 += 256) {              // (four rounds of the table per step, loads in flight together)
        int av[4];
#pragma unroll
        for (int u = 0; u < 4; u++) av[u] = arg_p[(size_t)n * P + min(p0 + 64 * u + lane, P - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int p = p0 + 64 * u + lane;
            unsigned long long hits = __ballot(p < P && av[u] == f);
            while (hits) {
                const int j = __ffsll((long long)hits) - 1;
                hits &= hits - 1;
                add(p0 + 64 * u + j, gp);
            }
        }
    }
    if (lane == 0) {
        float* o = gtri + ((size_t)n * F + f) * 9;
#pragma unroll
        for (int i = 0; i < 9; i++) o[i] = acc[i];
    }
}

// Point side: gpts[n,p,:] = own nearest face (weight gp) + every face whose nearest point it is (weight gf)
__global__ __launch_bounds__(256) void pmf_backward_point_kernel(const float* __restrict__ verts, const long long* __restrict__ faces,
                                                                 const float* __restrict__ pts, const int* __restrict__ arg_p,
                                                                 const int* __restrict__ arg_f, float gp, float gf,
                                                                 float* __restrict__ gpts, int V, int F, int P)
{
    const int n = blockIdx.y, p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= P) return;
    const float* vn = verts + (size_t)n * V * 3;
    const float* q = pts + ((size_t)n * P + p) * 3;
    float acc[3] = {0.f, 0.f, 0.f};
    auto add = [&](int f, float wgt) {
        float a[3], b[3], c[3];
        load_tri(vn, faces + (size_t)f * 3, a, b, c);
        const Closest cl = point_triangle(q, a, b, c);
#pragma unroll
        for (int d = 0; d < 3; d++) acc[d] += 2.f * wgt * (q[d] - (cl.w[0] * a[d] + cl.w[1] * b[d] + cl.w[2] * c[d]));
    };
    if (F > 0) add(arg_p[(size_t)n * P + p], gp);
    // (four rounds of the table per step, their loads in flight together: one round at a time is a chain of F / 64 L2 round trips)
    for (int f0 = 0; f0 < F; f0 += 256) {
        int av[4];
#pragma unroll
        for (int u = 0; u < 4; u++) av[u] = arg_f[(size_t)n * F + min(f0 + 64 * u + lane, F - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int f = f0 + 64 * u + lane;
            unsigned long long hits = __ballot(f < F && av[u] == p);
            while (hits) {
                const int j = __ffsll((long long)hits) - 1;
                hits &= hits - 1;
                add(f0 + 64 * u + j, gf);
            }
        }
    }
    if (lane == 0) {
        float* o = gpts + ((size_t)n * P + p) * 3;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
    }
}

// ===========================================================================
// Perceptual-distance reduction, PerceptualSimilarity/util/util.py:71-83 + models/networks_basic.py:51-52:
//   d[n] = 1 - mean_{pixels} sum_c a_hat[c] b_hat[c],   x_hat = x / (sqrt(sum_c x_c^2) + 1e-10)
// fa [Na, C, P] (observed-image features; image n of fb pairs with fa[n / rep]), fb [N, C, P] (rendered-image features).
//
// Shape: a block owns COS_TP = 32 consecutive pixels of one image and ALL channels; its 256 threads are 8 channel groups x
// 32 pixels, thread (g, p) walks channels g, g+8, ...  A wave-load therefore touches two 128-B pixel rows, every thread has
// C/8 independent (a, b) load pairs in flight instead of one serial chain over C, and the grid has N * ceil(P/32) blocks
// (AlexNet layers 3-5, P = 225: 8 blocks per image instead of 1).  Channel partials meet in LDS and are folded in group
// order (deterministic).  The backward keeps its C/8 (a, b) pairs in registers (CPT <= 48, i.e. C <= 384: every AlexNet
// layer) so the features are read once; other channel counts take the generic two-pass instantiation.
// Algorithmic bytes: forward 2*N*C*P*4 (the observed side is re-read per hypothesis, from L2), backward 3*N*C*P*4.
// ===========================================================================
constexpr float COS_EPS = 1e-10f;
constexpr int COS_TP = 32;     // pixels per block
constexpr int COS_CG = 8;      // channel groups per block

__device__ __forceinline__ void cos_exchange(float (*red)[COS_CG][COS_TP], int cg, int pxl, float& dot, float& na, float& nb)
{
    red[0][cg][pxl] = dot; red[1][cg][pxl] = na; red[2][cg][pxl] = nb;
    __syncthreads();
    dot = na = nb = 0.f;
#pragma unroll
    for (int g = 0; g < COS_CG; g++) { dot += red[0][g][pxl]; na += red[1][g][pxl]; nb += red[2][g][pxl]; }
}

__global__ __launch_bounds__(256) void cosdist_forward_kernel(const float* __restrict__ fa, const float* __restrict__ fb,
                                                              float* __restrict__ part, int C, int P, int rep, int nch)
{
    __shared__ float red[3][COS_CG][COS_TP];
    const int n = blockIdx.x, pxl = threadIdx.x & 31, cg = threadIdx.x >> 5;
    const int p = blockIdx.y * COS_TP + pxl;
    float dot = 0.f, na = 0.f, nb = 0.f;
    if (p < P) {
        const float* a = fa + (size_t)(n / rep) * C * P + p;
        const float* b = fb + (size_t)n * C * P + p;
#pragma unroll 8
        for (int c = cg; c < C; c += COS_CG) {
            const float x = a[(size_t)c * P], y = b[(size_t)c * P];
            dot += x * y; na += x * x; nb += y * y;
        }
    }
    cos_exchange(red, cg, pxl, dot, na, nb);
    if (cg != 0) return;                                     // lanes 0..31 of wave 0 finish the tile
    float cosv = p < P ? dot / ((sqrtf(na) + COS_EPS) * (sqrtf(nb) + COS_EPS)) : 0.f;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) cosv += __shfl_xor(cosv, o, 32);
    if (pxl == 0) part[(size_t)n * nch + blockIdx.y] = cosv;
}

// one wave per image: lanes stride over the tile partials, fixed-order wave reduction
__global__ __launch_bounds__(64) void cosdist_fold_kernel(const float* __restrict__ part, float* __restrict__ d, int N, int nch, int P)
{
    const int n = blockIdx.x;
    float s = 0.f;
    for (int k = threadIdx.x; k < nch; k += 64) s += part[(size_t)n * nch + k];
    s = wave_sum_to_lane63(s);
    if (threadIdx.x == 63) d[n] = 1.f - s / (float)P;
}

// gradient w.r.t. fb only (the observed side is data): d cos / d b_c = a_c / (A B) - dot * b_c / (A * nb * B^2),
// A = |a| + eps, B = |b| + eps, second term 0 where |b| = 0.  CPT = channels per thread held in registers (0: re-read).
template <int CPT>
__global__ __launch_bounds__(256) void cosdist_backward_kernel(const float* __restrict__ fa, const float* __restrict__ fb,
                                                               const float* __restrict__ gd, float* __restrict__ gfb,
                                                               int C, int P, int rep)
{
    __shared__ float red[3][COS_CG][COS_TP];
    const int n = blockIdx.x, pxl = threadIdx.x & 31, cg = threadIdx.x >> 5;
    const int p = blockIdx.y * COS_TP + pxl;
    const bool in = p < P;
    const float* a = fa + (size_t)(n / rep) * C * P + (in ? p : 0);
    const float* b = fb + (size_t)n * C * P + (in ? p : 0);
    float* g = gfb + (size_t)n * C * P + p;
    float xa[CPT > 0 ? CPT : 1], xb[CPT > 0 ? CPT : 1];
    float dot = 0.f, na = 0.f, nb = 0.f;
    if (CPT > 0) {
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int c = cg + COS_CG * i;
            xa[i] = in ? a[(size_t)c * P] : 0.f;
            xb[i] = in ? b[(size_t)c * P] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < CPT; i++) { dot += xa[i] * xb[i]; na += xa[i] * xa[i]; nb += xb[i] * xb[i]; }
    } else if (in) {
#pragma unroll 8
        for (int c = cg; c < C; c += COS_CG) {
            const float x = a[(size_t)c * P], y = b[(size_t)c * P];
            dot += x * y; na += x * x; nb += y * y;
        }
    }
    cos_exchange(red, cg, pxl, dot, na, nb);
    if (!in) return;
    const float A = sqrtf(na) + COS_EPS, nbr = sqrtf(nb), B = nbr + COS_EPS;
    const float k = -gd[n] / (float)P;                       // d = 1 - mean cos
    const float ka = k / (A * B);
    const float kb = nbr > 0.f ? k * dot / (A * nbr * B * B) : 0.f;
    if (CPT > 0) {
#pragma unroll
        for (int i = 0; i < CPT; i++) g[(size_t)(cg + COS_CG * i) * P] = ka * xa[i] - kb * xb[i];
    } else {
#pragma unroll 8
        for (int c = cg; c < C; c += COS_CG) g[(size_t)c * P] = ka * a[(size_t)c * P] - kb * b[(size_t)c * P];
    }
}

// ---- all feature layers of the perceptual term in one launch each way ---------------------------------------------------------
// LASR's perceptual distance sums the cosine distance over the five AlexNet feature maps (networks_basic.py:51-64): with one
// operator call per layer that is 5 x (reduce + fold) + 5 backward launches per step, each 5-16 us for a few MB, plus the
// eager adds between them.  Here a block still owns one 32-pixel tile of one image of ONE layer (same arithmetic, same tile
// partials as cosdist_forward_kernel / cosdist_backward_kernel<CPT>), but the grid's second dimension runs over the tiles of
// every layer, and ONE fold launch (a wave per image) folds the tile partials exactly like cosdist_fold_kernel does (lane-
// strided, DPP tree) and adds the layers in list order: dist[n] = sum_l (1 - mean_l) is bit-identical to the per-layer calls
// added up by the caller.  (Folding by the reduce launch's last block -- ticket + __threadfence per block -- was measured at
// 224 us for the 5760 blocks of the spot3 sizes: every fence writes the XCD's L2 back.)
struct CosLayers {
    const float* fa[LASR_COSDIST_MAX_LAYERS];
    const float* fb[LASR_COSDIST_MAX_LAYERS];
    float* gfb[LASR_COSDIST_MAX_LAYERS];
    int C[LASR_COSDIST_MAX_LAYERS], P[LASR_COSDIST_MAX_LAYERS];
    int tile0[LASR_COSDIST_MAX_LAYERS + 1];          // first global tile of each layer; tile0[n_layers] = tiles per image
    int n_layers, rep;
};

__device__ __forceinline__ int cos_layer_of(const CosLayers& L, int tile)
{
    int l = 0;
    while (l + 1 < L.n_layers && tile >= L.tile0[l + 1]) l++;
    return l;
}

__global__ __launch_bounds__(256) void cosdist_multi_forward_kernel(CosLayers L, float* __restrict__ part)
{
    __shared__ float red[3][COS_CG][COS_TP];
    const int n = blockIdx.x, pxl = threadIdx.x & 31, cg = threadIdx.x >> 5;
    const int l = cos_layer_of(L, blockIdx.y), C = L.C[l], P = L.P[l], ntile = L.tile0[L.n_layers];
    const int p = (blockIdx.y - L.tile0[l]) * COS_TP + pxl;
    float dot = 0.f, na = 0.f, nb = 0.f;
    if (p < P) {
        const float* a = L.fa[l] + (size_t)(n / L.rep) * C * P + p;
        const float* b = L.fb[l] + (size_t)n * C * P + p;
#pragma unroll 8
        for (int c = cg; c < C; c += COS_CG) {
            const float x = a[(size_t)c * P], y = b[(size_t)c * P];
            dot += x * y; na += x * x; nb += y * y;
        }
    }
    cos_exchange(red, cg, pxl, dot, na, nb);
    if (cg == 0) {                                             // lanes 0..31 of wave 0 finish the tile
        float cosv = p < P ? dot / ((sqrtf(na) + COS_EPS) * (sqrtf(nb) + COS_EPS)) : 0.f;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) cosv += __shfl_xor(cosv, o, 32);
        if (pxl == 0) part[(size_t)n * ntile + blockIdx.y] = cosv;
    }
}

// a wave per image: cosdist_fold_kernel's order inside a layer, the layers added in list order
__global__ __launch_bounds__(64) void cosdist_multi_fold_kernel(CosLayers L, const float* __restrict__ part, float* __restrict__ d)
{
    const int m = blockIdx.x, lane = threadIdx.x, ntile = L.tile0[L.n_layers];
    float acc = 0.f;
    for (int k = 0; k < L.n_layers; k++) {
        const int t0 = L.tile0[k], nt = L.tile0[k + 1] - t0;
        float sm = 0.f;
        for (int j = lane; j < nt; j += 64) sm += part[(size_t)m * ntile + t0 + j];
        sm = wave_sum_to_lane63(sm);
        acc = acc + (1.f - sm / (float)L.P[k]);
    }
    if (lane == 63) d[m] = acc;
}

template <int CPT>
__device__ __forceinline__ void cosdist_backward_tile(const float* __restrict__ fa, const float* __restrict__ fb, float gdn,
                                                      float* __restrict__ gfb, int n, int tile, int C, int P, int rep,
                                                      float (*red)[COS_CG][COS_TP])
{
    const int pxl = threadIdx.x & 31, cg = threadIdx.x >> 5;
    const int p = tile * COS_TP + pxl;
    const bool in = p < P;
    const float* a = fa + (size_t)(n / rep) * C * P + (in ? p : 0);
    const float* b = fb + (size_t)n * C * P + (in ? p : 0);
    float* g = gfb + (size_t)n * C * P + p;
    float xa[CPT > 0 ? CPT : 1], xb[CPT > 0 ? CPT : 1];
    float dot = 0.f, na = 0.f, nb = 0.f;
    if (CPT > 0) {
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int c = cg + COS_CG * i;
            xa[i] = in ? a[(size_t)c * P] : 0.f;
            xb[i] = in ? b[(size_t)c * P] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < CPT; i++) { dot += xa[i] * xb[i]; na += xa[i] * xa[i]; nb += xb[i] * xb[i]; }
    } else if (in) {
#pragma unroll 8
        for (int c = cg; c < C; c += COS_CG) {
            const float x = a[(size_t)c * P], y = b[(size_t)c * P];
            dot += x * y; na += x * x; nb += y * y;
        }
    }
    cos_exchange(red, cg, pxl, dot, na, nb);
    if (!in) return;
    const float A = sqrtf(na) + COS_EPS, nbr = sqrtf(nb), B = nbr + COS_EPS;
    const float k = -gdn / (float)P;                         // d = 1 - mean cos
    const float ka = k / (A * B);
    const float kb = nbr > 0.f ? k * dot / (A * nbr * B * B) : 0.f;
    if (CPT > 0) {
#pragma unroll
        for (int i = 0; i < CPT; i++) g[(size_t)(cg + COS_CG * i) * P] = ka * xa[i] - kb * xb[i];
    } else {
#pragma unroll 8
        for (int c = cg; c < C; c += COS_CG) g[(size_t)c * P] = ka * a[(size_t)c * P] - kb * b[(size_t)c * P];
    }
}

__global__ __launch_bounds__(256) void cosdist_multi_backward_kernel(CosLayers L, const float* __restrict__ gd)
{
    __shared__ float red[3][COS_CG][COS_TP];
    const int n = blockIdx.x;
    const int l = cos_layer_of(L, blockIdx.y), C = L.C[l], P = L.P[l], tile = blockIdx.y - L.tile0[l];
    const float g = gd[n];                                   // every layer's distance enters the sum with weight 1
    switch (C % COS_CG == 0 ? C / COS_CG : 0) {              // block-uniform: AlexNet's 64 / 192 / 384 / 256 channels in registers
        case 8:  cosdist_backward_tile<8>(L.fa[l], L.fb[l], g, L.gfb[l], n, tile, C, P, L.rep, red); break;
        case 24: cosdist_backward_tile<24>(L.fa[l], L.fb[l], g, L.gfb[l], n, tile, C, P, L.rep, red); break;
        case 32: cosdist_backward_tile<32>(L.fa[l], L.fb[l], g, L.gfb[l], n, tile, C, P, L.rep, red); break;
        case 48: cosdist_backward_tile<48>(L.fa[l], L.fb[l], g, L.gfb[l], n, tile, C, P, L.rep, red); break;
        default: cosdist_backward_tile<0>(L.fa[l], L.fb[l], g, L.gfb[l], n, tile, C, P, L.rep, red);
    }
}

// ===========================================================================
// Texture atlas -> per-face surface texels, third_party/softras/soft_renderer/cuda/load_textures_cuda_kernel.cu:8-66:
// texel (w_x, w_y) of an R x R face texture sits at barycentric ((w_x + 1/3)/R, (w_y + 1/3)/R) of the lower triangle of
// the texel grid (w_x + w_y < R), else at the mirrored position of the upper one; the atlas is sampled bilinearly at the
// uv of that point, uv * (size - 1).  One thread per texel; 3 channels.  The reference reads row/column (int)(pos + 1)
// unchecked (one past the image when uv == 1, with weight 0); the index is clamped here.
// ===========================================================================
__global__ __launch_bounds__(256) void load_textures_kernel(const float* __restrict__ image, const float* __restrict__ faces_uv,
                                                            const int* __restrict__ is_update, float* __restrict__ textures,
                                                            int F, int R, int H, int W)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= F * R * R) return;
    const int fn = i / (R * R), w_y = (i % (R * R)) / R, w_x = i % R;
    if (is_update && is_update[fn] == 0) return;
    float w0, w1;
    if (w_x + w_y < R) {
        w0 = (float)((w_x + 1. / 3.) / R);
        w1 = (float)((w_y + 1. / 3.) / R);
    } else {
        w0 = (float)(((R - 1. - w_x) + 2. / 3.) / R);
        w1 = (float)(((R - 1. - w_y) + 2. / 3.) / R);
    }
    const float w2 = (float)(1. - (double)w0 - (double)w1);
    const float* uv = faces_uv + (size_t)fn * 6;
    const float pos_x = (uv[0] * w0 + uv[2] * w1 + uv[4] * w2) * (float)(W - 1);
    const float pos_y = (uv[1] * w0 + uv[3] * w1 + uv[5] * w2) * (float)(H - 1);
    const int x0 = (int)pos_x, y0 = (int)pos_y;
    const float wx1 = pos_x - (float)x0, wx0 = 1 - wx1, wy1 = pos_y - (float)y0, wy0 = 1 - wy1;
    const int xa = min(max(x0, 0), W - 1), xb = min(max(x0 + 1, 0), W - 1);
    const int ya = min(max(y0, 0), H - 1), yb = min(max((int)(pos_y + 1), 0), H - 1);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float c = 0;
        c += image[((size_t)ya * W + xa) * 3 + k] * (wx0 * wy0);
        c += image[((size_t)yb * W + xa) * 3 + k] * (wx0 * wy1);
        c += image[((size_t)ya * W + xb) * 3 + k] * (wx1 * wy0);
        c += image[((size_t)yb * W + xb) * 3 + k] * (wx1 * wy1);
        textures[(size_t)i * 3 + k] = c;
    }
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
                (float2*)flow, bgmask, P, nch, (long long)7 * P);
    return launch_ok();
}

extern "C" int lasr_flow_reproject_planes_forward(const float* pos6, long long batch_stride, const float* pp0, const float* pp1,
                                                  const float* fl0, const float* fl1, float* flow, unsigned char* bgmask, int N,
                                                  int P, void* hip_stream)
{
    if (N < 0 || P < 0 || batch_stride < (long long)6 * P) return LASR_E_BADARG;
    if (N == 0 || P == 0) return LASR_OK;
    if (!pos6 || !pp0 || !pp1 || !fl0 || !fl1 || !flow || !bgmask) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int nch = px_chunks(P);
    LASR_LAUNCH(K_FLOW_REPROJECT_FORWARD, flow_reproject_forward_kernel, dim3(N, nch), dim3(256), 0, pos6, pp0, pp1, fl0, fl1,
                (float2*)flow, bgmask, P, nch, batch_stride);
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
                (const float2*)grad_flow, grad_px, scratch, P, nch, (long long)7 * P, 7);
    int rc = launch_ok();
    if (rc) return rc;
    LASR_LAUNCH(K_FLOW_REPROJECT_BACKWARD, flow_reproject_fold_kernel, dim3((N + 255) / 256), dim3(256), 0, scratch, grad_pp1,
                grad_fl1, N, nch);
    return launch_ok();
}

extern "C" int lasr_flow_reproject_planes_backward(const float* pos6, long long batch_stride, const float* fl1,
                                                   const float* grad_flow, float* grad_pos6, float* grad_pp1, float* grad_fl1,
                                                   float* scratch, int N, int P, void* hip_stream)
{
    if (N < 0 || P < 0 || batch_stride < (long long)6 * P) return LASR_E_BADARG;
    if (N == 0) return LASR_OK;
    if (!pos6 || !fl1 || !grad_flow || !grad_pos6 || !grad_pp1 || !grad_fl1 || !scratch) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int nch = px_chunks(P);
    LASR_LAUNCH(K_FLOW_REPROJECT_BACKWARD, flow_reproject_backward_kernel, dim3(N, nch), dim3(256), 0, pos6, fl1,
                (const float2*)grad_flow, grad_pos6, scratch, P, nch, batch_stride, 6);
    int rc = launch_ok();
    if (rc) return rc;
    LASR_LAUNCH(K_FLOW_REPROJECT_BACKWARD, flow_reproject_fold_kernel, dim3((N + 255) / 256), dim3(256), 0, scratch, grad_pp1,
                grad_fl1, N, nch);
    return launch_ok();
}

extern "C" int lasr_quat_to_rotmat_forward(const float* quat, float* rotmat, int M, void* hip_stream)
{
    if (M < 0) return LASR_E_BADARG;
    if (M == 0) return LASR_OK;
    if (!quat || !rotmat) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_QUAT_FORWARD, quat_forward_kernel, dim3((M + 255) / 256), dim3(256), 0, quat, rotmat, M);
    return launch_ok();
}

extern "C" int lasr_quat_to_rotmat_backward(const float* quat, const float* grad_rotmat, float* grad_quat, int M,
                                            void* hip_stream)
{
    if (M < 0) return LASR_E_BADARG;
    if (M == 0) return LASR_OK;
    if (!quat || !grad_rotmat || !grad_quat) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_QUAT_BACKWARD, quat_backward_kernel, dim3((M + 255) / 256), dim3(256), 0, quat, grad_rotmat, grad_quat, M);
    return launch_ok();
}

extern "C" int lasr_skin_weights_forward(const float* ctl_ts, const float* ctl_rs, const float* log_ctl,
                                         const float* verts, float* skin, int H, int J, int V, void* hip_stream)
{
    if (H < 0 || J < 0 || V < 0 || J > SKIN_MAX_BONES) return LASR_E_BADARG;
    if (H == 0 || J == 0 || V == 0) return LASR_OK;
    if (!ctl_ts || !ctl_rs || !log_ctl || !verts || !skin) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const dim3 grid((V + 63) / 64, H);
    if (J <= 24)      LASR_LAUNCH(K_SKIN_FORWARD, skin_forward_kernel<24>, grid, dim3(64), 0, ctl_ts, ctl_rs, log_ctl, verts, skin, V, J);
    else if (J <= 40) LASR_LAUNCH(K_SKIN_FORWARD, skin_forward_kernel<40>, grid, dim3(64), 0, ctl_ts, ctl_rs, log_ctl, verts, skin, V, J);
    else              LASR_LAUNCH(K_SKIN_FORWARD, skin_forward_kernel<0>, grid, dim3(64), 0, ctl_ts, ctl_rs, log_ctl, verts, skin, V, J);
    return launch_ok();
}

extern "C" int lasr_skin_weights_backward(const float* ctl_ts, const float* ctl_rs, const float* log_ctl,
                                          const float* verts, const float* skin, const float* grad_skin,
                                          float* grad_ts, float* grad_rs, float* grad_log_ctl, float* scratch,
                                          int H, int J, int V, void* hip_stream)
{
    if (H < 0 || J < 0 || V < 0 || J > SKIN_MAX_BONES) return LASR_E_BADARG;
    if (H == 0 || J == 0) return LASR_OK;
    if (!ctl_ts || !ctl_rs || !log_ctl || !verts || !skin || !grad_skin || !grad_ts || !grad_rs || !grad_log_ctl)
        return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    (void)scratch;                                       // rounds 1-4: the dot[h,v] table of a separate first launch
    LASR_LAUNCH(K_SKIN_BACKWARD, skin_backward_kernel, dim3(H * J), dim3(256), 0, ctl_ts, ctl_rs, log_ctl, verts, skin,
                grad_skin, grad_ts, grad_rs, grad_log_ctl, V, J);
    return launch_ok();
}

extern "C" int lasr_flatten_forward(const float* x, const int* quads, float* loss, int N, int V, int E, void* hip_stream)
{
    if (N < 0 || V < 0 || E < 0) return LASR_E_BADARG;
    if (N == 0) return LASR_OK;
    if (!x || !loss || (E > 0 && !quads)) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_FLATTEN_FORWARD, flatten_forward_kernel, dim3(N), dim3(256), 0, x, quads, loss, V, E);
    return launch_ok();
}

extern "C" int lasr_flatten_backward(const float* x, const int* quads, const int* inc_ptr, const int* inc,
                                     const float* grad_loss, float* grad_x, float* scratch, int N, int V, int E,
                                     void* hip_stream)
{
    if (N < 0 || V < 0 || E < 0) return LASR_E_BADARG;
    if (N == 0 || V == 0) return LASR_OK;
    if (!x || !inc_ptr || !grad_loss || !grad_x || (E > 0 && (!quads || !inc || !scratch))) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    if (E > 0) {
        LASR_LAUNCH(K_FLATTEN_BACKWARD, flatten_backward_edge_kernel, dim3((E + 255) / 256, N), dim3(256), 0, x, quads,
                    grad_loss, scratch, V, E);
        int rc = launch_ok();
        if (rc) return rc;
    }
    LASR_LAUNCH(K_FLATTEN_BACKWARD, flatten_backward_vertex_kernel, dim3((V + 255) / 256, N), dim3(256), 0, scratch, inc_ptr,
                inc, grad_x, V, E);
    return launch_ok();
}

extern "C" int lasr_face_gather_forward(const float* attr, const long long* faces, float* out, int N, int V, int F, int C,
                                        void* hip_stream)
{
    if (N < 0 || V < 0 || F < 0 || C < 1) return LASR_E_BADARG;
    if (N == 0 || F == 0) return LASR_OK;
    if (!attr || !faces || !out) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_FACE_GATHER_FORWARD, face_gather_forward_kernel, dim3((3 * F + 255) / 256, N), dim3(256), 0, attr, faces, out,
                V, 3 * F, C);
    return launch_ok();
}

extern "C" int lasr_face_gather_backward(const float* grad_out, const long long* faces, float* grad_attr, int N, int V, int F,
                                         int C, void* hip_stream)
{
    if (N < 0 || V < 0 || F < 0 || C < 1) return LASR_E_BADARG;
    if (N == 0 || V == 0) return LASR_OK;
    if (!grad_attr || (F > 0 && (!grad_out || !faces))) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    if ((long long)N * ((V + 63) / 64) >= 2048) {
        LASR_LAUNCH(K_FACE_GATHER_BACKWARD, face_gather_backward_kernel<256>, dim3((V + 255) / 256, N), dim3(256), 0,
                    grad_out, faces, grad_attr, V, 3 * F, C);
    } else {
        // few blocks on a mostly empty chip: the scan of the mesh's 3 F corner ids is the critical path, 1024 threads walk it in a
        // quarter of the rounds (one frame of the bench mesh: 11 -> see profiles/r05_bench.json sweep)
        LASR_LAUNCH(K_FACE_GATHER_BACKWARD, (face_gather_backward_kernel<64, 1024>), dim3((V + 63) / 64, N), dim3(1024), 0,
                    grad_out, faces, grad_attr, V, 3 * F, C);
    }
    return launch_ok();
}

extern "C" int lasr_face_gather_backward_csr(const float* grad_out, const int* inc_ptr, const int* inc, int inc_shared,
                                             float* grad_attr, int N, int V, int F, int C, void* hip_stream)
{
    if (N < 0 || V < 0 || F < 0 || C < 1) return LASR_E_BADARG;
    if (N == 0 || V == 0) return LASR_OK;
    if (!grad_attr || !inc_ptr || (F > 0 && (!grad_out || !inc))) return LASR_E_BADARG;
    if ((long long)V * C > 0x7fffffffLL || N > 65535) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_FACE_GATHER_BACKWARD, face_gather_backward_csr_kernel, dim3((V * C + 255) / 256, N), dim3(256), 0, grad_out, inc_ptr,
                inc, grad_attr, V, 3 * F, C, inc_shared ? 1 : 0);
    return launch_ok();
}

extern "C" int lasr_nearest_point(const float* a, const float* b, float* d2, int* idx, int N, int P, int Q, void* hip_stream)
{
    if (N < 0 || P < 0 || Q < 1) return LASR_E_BADARG;
    if (N == 0 || P == 0) return LASR_OK;
    if (!a || !b || !d2 || !idx) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_NEAREST_POINT, nearest_point_kernel, dim3((P + 15) / 16, N), dim3(256), 0, a, b, d2, idx, P, Q);
    return launch_ok();
}

static void pmf_layout(int F, int P, int& FT, int& PT, int& FC, int& PC)
{
    FT = pmf_tile(F); PT = 256;                                      // face chunks of the tile size; point "chunks" = blocks of 256 points
    FC = (F + FT - 1) / FT; PC = (P + PT - 1) / PT;
}

extern "C" size_t lasr_point_mesh_scratch_floats(int N, int F, int P)
{
    if (N < 0 || F < 1 || P < 1) return 0;
    int FT, PT, FC, PC;
    pmf_layout(F, P, FT, PT, FC, PC);
    return 2 * (size_t)N * ((size_t)FC * P + (size_t)PC * F) + 4;        // per-chunk minima (float) and their indices (int)
}

extern "C" int lasr_point_mesh_forward(const float* verts, const long long* faces, const float* points, float* dmin_point,
                                       int* arg_point, float* dmin_face, int* arg_face, float* scratch, int N, int V, int F, int P,
                                       void* hip_stream)
{
    if (N < 0 || V < 0 || F < 1 || P < 1) return LASR_E_BADARG;
    if (N == 0) return LASR_OK;
    if (!verts || !faces || !points || !dmin_point || !arg_point || !dmin_face || !arg_face || !scratch) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    PmfArgs A;
    A.verts = verts; A.faces = faces; A.pts = points; A.V = V; A.F = F; A.P = P;
    pmf_layout(F, P, A.FT, A.PT, A.FC, A.PC);
    const size_t np = (size_t)N * A.FC * P, nf = (size_t)N * A.PC * F;
    A.part_pd = scratch; A.part_fd = scratch + np;
    A.part_pa = reinterpret_cast<int*>(scratch + np + nf); A.part_fa = A.part_pa + np;
    const int gx = ((P > F ? P : F) + 255) / 256;
    if (A.FC > 65535 || N > 65535) return LASR_E_BADARG;
    LASR_LAUNCH(K_POINT_MESH_FORWARD, pmf_partial_kernel, dim3(A.PC, A.FC, N), dim3(256), 0, A);
    int rc = launch_ok();
    if (rc) return rc;
    LASR_LAUNCH(K_POINT_MESH_FORWARD, pmf_fold_kernel, dim3(gx, 2, N), dim3(256), 0, A, dmin_point, arg_point, dmin_face, arg_face);
    return launch_ok();
}

extern "C" int lasr_point_mesh_backward(const float* verts, const long long* faces, const float* points, const int* arg_point,
                                        const int* arg_face, float grad_point_term, float grad_face_term, float* grad_tri,
                                        float* grad_points, int N, int V, int F, int P, void* hip_stream)
{
    if (N < 0 || V < 0 || F < 1 || P < 1) return LASR_E_BADARG;
    if (N == 0) return LASR_OK;
    if (!verts || !faces || !points || !arg_point || !arg_face || !grad_tri || !grad_points) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_POINT_MESH_BACKWARD, pmf_backward_face_kernel, dim3((F + 3) / 4, N), dim3(256), 0, verts, faces, points,
                arg_point, arg_face, grad_point_term, grad_face_term, grad_tri, V, F, P);
    int rc = launch_ok();
    if (rc) return rc;
    LASR_LAUNCH(K_POINT_MESH_BACKWARD, pmf_backward_point_kernel, dim3((P + 3) / 4, N), dim3(256), 0, verts, faces, points,
                arg_point, arg_face, grad_point_term, grad_face_term, grad_points, V, F, P);
    return launch_ok();
}

extern "C" size_t lasr_cosdist_scratch_floats(int N, int P)
{
    if (N < 0 || P < 0) return 0;
    return (size_t)N * ((P + COS_TP - 1) / COS_TP) + 4;
}

extern "C" int lasr_cosdist_forward(const float* feat_obs, const float* feat_rnd, float* dist, float* scratch, int N, int C,
                                    int P, int rep, void* hip_stream)
{
    if (N < 0 || C < 1 || P < 1 || rep < 1) return LASR_E_BADARG;
    if (N == 0) return LASR_OK;
    if (!feat_obs || !feat_rnd || !dist || !scratch) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int nch = (P + COS_TP - 1) / COS_TP;
    LASR_LAUNCH(K_COSDIST_FORWARD, cosdist_forward_kernel, dim3(N, nch), dim3(256), 0, feat_obs, feat_rnd, scratch, C, P, rep, nch);
    int rc = launch_ok();
    if (rc) return rc;
    LASR_LAUNCH(K_COSDIST_FORWARD, cosdist_fold_kernel, dim3(N), dim3(64), 0, scratch, dist, N, nch, P);
    return launch_ok();
}

extern "C" int lasr_cosdist_backward(const float* feat_obs, const float* feat_rnd, const float* grad_dist, float* grad_rnd,
                                     int N, int C, int P, int rep, void* hip_stream)
{
    if (N < 0 || C < 1 || P < 1 || rep < 1) return LASR_E_BADARG;
    if (N == 0) return LASR_OK;
    if (!feat_obs || !feat_rnd || !grad_dist || !grad_rnd) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const dim3 grid(N, (P + COS_TP - 1) / COS_TP);
#define LASR_COS_BWD(CPT)                                                                                          \
    LASR_LAUNCH(K_COSDIST_BACKWARD, (cosdist_backward_kernel<CPT>), grid, dim3(256), 0, feat_obs, feat_rnd, grad_dist, \
                grad_rnd, C, P, rep)
    switch (C % COS_CG == 0 ? C / COS_CG : 0) {         // AlexNet: 64, 192, 384, 256, 256 channels
        case 8:  LASR_COS_BWD(8); break;
        case 24: LASR_COS_BWD(24); break;
        case 32: LASR_COS_BWD(32); break;
        case 48: LASR_COS_BWD(48); break;
        default: LASR_COS_BWD(0);
    }
#undef LASR_COS_BWD
    return launch_ok();
}

static int cos_layers(CosLayers& L, const float* const* feat_obs, const float* const* feat_rnd, float* const* grad_rnd, const int* C,
                      const int* P, int n_layers, int rep)
{
    if (n_layers < 1 || n_layers > LASR_COSDIST_MAX_LAYERS || rep < 1 || !feat_obs || !feat_rnd || !C || !P) return LASR_E_BADARG;
    L.n_layers = n_layers; L.rep = rep;
    L.tile0[0] = 0;
    for (int l = 0; l < n_layers; l++) {
        if (C[l] < 1 || P[l] < 1 || !feat_obs[l] || !feat_rnd[l] || (grad_rnd && !grad_rnd[l])) return LASR_E_BADARG;
        L.fa[l] = feat_obs[l]; L.fb[l] = feat_rnd[l]; L.gfb[l] = grad_rnd ? grad_rnd[l] : nullptr;
        L.C[l] = C[l]; L.P[l] = P[l];
        L.tile0[l + 1] = L.tile0[l] + (P[l] + COS_TP - 1) / COS_TP;
        if (L.tile0[l + 1] > 65535) return LASR_E_BADARG;             // grid.y
    }
    return LASR_OK;
}

extern "C" size_t lasr_cosdist_multi_scratch_floats(const int* P, int n_layers, int N)
{
    if (!P || n_layers < 1 || n_layers > LASR_COSDIST_MAX_LAYERS || N < 0) return 0;
    size_t tiles = 0;
    for (int l = 0; l < n_layers; l++) tiles += (size_t)((P[l] > 0 ? P[l] : 0) + COS_TP - 1) / COS_TP;
    return (size_t)N * tiles + 4;                                       // tile partials
}

extern "C" int lasr_cosdist_multi_forward(const float* const* feat_obs, const float* const* feat_rnd, const int* C, const int* P,
                                          int n_layers, float* dist, float* scratch, int N, int rep, void* hip_stream)
{
    if (N < 0) return LASR_E_BADARG;
    CosLayers L;
    int rc = cos_layers(L, feat_obs, feat_rnd, nullptr, C, P, n_layers, rep);
    if (rc) return rc;
    if (N == 0) return LASR_OK;
    if (!dist || !scratch) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int ntile = L.tile0[n_layers];
    LASR_LAUNCH(K_COSDIST_FORWARD, cosdist_multi_forward_kernel, dim3(N, ntile), dim3(256), 0, L, scratch);
    if ((rc = launch_ok())) return rc;
    LASR_LAUNCH(K_COSDIST_FORWARD, cosdist_multi_fold_kernel, dim3(N), dim3(64), 0, L, (const float*)scratch, dist);
    return launch_ok();
}

extern "C" int lasr_cosdist_multi_backward(const float* const* feat_obs, const float* const* feat_rnd, const int* C, const int* P,
                                           int n_layers, const float* grad_dist, float* const* grad_rnd, int N, int rep,
                                           void* hip_stream)
{
    if (N < 0 || !grad_rnd) return LASR_E_BADARG;
    CosLayers L;
    int rc = cos_layers(L, feat_obs, feat_rnd, grad_rnd, C, P, n_layers, rep);
    if (rc) return rc;
    if (N == 0) return LASR_OK;
    if (!grad_dist) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_COSDIST_BACKWARD, cosdist_multi_backward_kernel, dim3(N, L.tile0[n_layers]), dim3(256), 0, L, grad_dist);
    return launch_ok();
}

extern "C" int lasr_load_textures(const float* image, const float* faces_uv, const int* is_update, float* textures, int F,
                                  int R, int H, int W, void* hip_stream)
{
    if (F < 0 || R < 1 || H < 1 || W < 1) return LASR_E_BADARG;
    if (F == 0) return LASR_OK;
    if (!image || !faces_uv || !textures) return LASR_E_BADARG;
    if ((long long)F * R * R > 0x7fffffffLL) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_LOAD_TEXTURES, load_textures_kernel, dim3((F * R * R + 255) / 256), dim3(256), 0, image, faces_uv, is_update,
                textures, F, R, H, W);
    return launch_ok();
}
