// mesh_losses.h -- device bodies of the three shape regularisers, shared by their own kernels (ops.hip: ARAP, Laplacian;
// fused.hip: flatten) and by the one-launch combination (mesh_reg.hip), so that every path runs the same arithmetic in the same
// order.  References: nnutils/loss_utils.py:29-64 (ARAP), third_party/ext_nnutils/loss_utils.py:34-65 (Laplacian), :110-152
// (flatten) under /root/reference/.
#pragma once
#include "ops_common.h"

namespace lasr {

__device__ __forceinline__ float mesh_sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// ---- ARAP: sum over directed edges of | |x_u - x_v|^2 - |dx_u - dx_v|^2 |; X / D = one mesh of x / dx ------------------------------
__device__ __forceinline__ float arap_forward_block(const float* __restrict__ X, const float* __restrict__ D, const int* __restrict__ row_ptr,
                                                    const int* __restrict__ col, int V, float* red)
{
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float a0 = X[3 * v], a1 = X[3 * v + 1], a2 = X[3 * v + 2];
        const float b0 = D[3 * v], b1 = D[3 * v + 1], b2 = D[3 * v + 2];
        for (int e = row_ptr[v]; e < row_ptr[v + 1]; e++) {
            const int u = col[e];
            const float p0 = X[3 * u] - a0, p1 = X[3 * u + 1] - a1, p2 = X[3 * u + 2] - a2;
            const float q0 = D[3 * u] - b0, q1 = D[3 * u + 1] - b1, q2 = D[3 * u + 2] - b2;
            s += fabsf((p0 * p0 + p1 * p1 + p2 * p2) - (q0 * q0 + q1 * q1 + q2 * q2));
        }
    }
    return block_sum(s, red);
}

// each undirected edge appears as (v,u) and (u,v) with the same value: 2 * d|e|/dx_v = 4 sign(e) (x_v - x_u); k = 4 g / edges
__device__ __forceinline__ void arap_backward_vertex(const float* __restrict__ X, const float* __restrict__ D, const int* __restrict__ row_ptr,
                                                     const int* __restrict__ col, float k, int v, float* __restrict__ gx, float* __restrict__ gdx)
{
    const float a0 = X[3 * v], a1 = X[3 * v + 1], a2 = X[3 * v + 2];
    const float b0 = D[3 * v], b1 = D[3 * v + 1], b2 = D[3 * v + 2];
    float gx0 = 0, gx1 = 0, gx2 = 0, gd0 = 0, gd1 = 0, gd2 = 0;
    for (int e = row_ptr[v]; e < row_ptr[v + 1]; e++) {
        const int u = col[e];
        const float p0 = a0 - X[3 * u], p1 = a1 - X[3 * u + 1], p2 = a2 - X[3 * u + 2];
        const float q0 = b0 - D[3 * u], q1 = b1 - D[3 * u + 1], q2 = b2 - D[3 * u + 2];
        const float sg = mesh_sgn((p0 * p0 + p1 * p1 + p2 * p2) - (q0 * q0 + q1 * q1 + q2 * q2));
        gx0 += sg * p0; gx1 += sg * p1; gx2 += sg * p2;
        gd0 -= sg * q0; gd1 -= sg * q1; gd2 -= sg * q2;
    }
    if (gx) { gx[0] = k * gx0; gx[1] = k * gx1; gx[2] = k * gx2; }
    if (gdx) { gdx[0] = k * gd0; gdx[1] = k * gd1; gdx[2] = k * gd2; }
}

// ---- Laplacian: lx[v] = x_v - mean_{u in nbr(v)} x_u (0 for isolated vertices); loss = sum |lx|^2 --------------------------------
__device__ __forceinline__ float laplacian_forward_block(const float* __restrict__ X, const int* __restrict__ row_ptr, const int* __restrict__ col,
                                                         float* __restrict__ lx_out, int V, float* red)
{
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) {
        const int e0 = row_ptr[v], e1 = row_ptr[v + 1];
        float l0 = 0.f, l1 = 0.f, l2 = 0.f;
        if (e1 > e0) {
            float m0 = 0.f, m1 = 0.f, m2 = 0.f;
            for (int e = e0; e < e1; e++) { const int u = col[e]; m0 += X[3 * u]; m1 += X[3 * u + 1]; m2 += X[3 * u + 2]; }
            const float inv = 1.f / (float)(e1 - e0);
            l0 = X[3 * v] - m0 * inv; l1 = X[3 * v + 1] - m1 * inv; l2 = X[3 * v + 2] - m2 * inv;
        }
        if (lx_out) { lx_out[3 * v] = l0; lx_out[3 * v + 1] = l1; lx_out[3 * v + 2] = l2; }
        s += l0 * l0 + l1 * l1 + l2 * l2;
    }
    return block_sum(s, red);
}

// g_x = 2 g L^T (L x):  out = k (lx[v] - sum_{u in nbr(v)} lx[u] / deg(u)), k = 2 g   (symmetric adjacency)
__device__ __forceinline__ void laplacian_backward_vertex(const float* __restrict__ L, const int* __restrict__ row_ptr, const int* __restrict__ col,
                                                          float k, int v, float* out)
{
    float a0 = L[3 * v], a1 = L[3 * v + 1], a2 = L[3 * v + 2];
    for (int e = row_ptr[v]; e < row_ptr[v + 1]; e++) {
        const int u = col[e];
        const float inv = 1.f / (float)(row_ptr[u + 1] - row_ptr[u]);
        a0 -= L[3 * u] * inv; a1 -= L[3 * u + 1] * inv; a2 -= L[3 * u + 2] * inv;
    }
    out[0] = k * a0; out[1] = k * a1; out[2] = k * a2;
}

// ---- flatten: for every listed interior edge (v0,v1) with opposite vertices v2, v3: (cos_e + 1)^2, cos_e = the cosine between the
// components of (v2-v0), (v3-v0) orthogonal to (v1-v0) -----------------------------------------------------------------------------
constexpr float FLAT_EPS = 1e-6f;

struct FlatSide { float b[3], cb[3], bl1, ab, den, cosb, sinb, t, nb; };

__device__ __forceinline__ void flat_side(const float* a, float al2, float sq_al2, const float* v0, const float* vb, FlatSide& s)
{
    float bl2 = 0.f; s.ab = 0.f;
#pragma unroll
    for (int d = 0; d < 3; d++) { s.b[d] = vb[d] - v0[d]; bl2 += s.b[d] * s.b[d]; s.ab += a[d] * s.b[d]; }
    s.bl1 = sqrtf(bl2 + FLAT_EPS);
    s.den = sq_al2 * s.bl1 + FLAT_EPS;
    s.cosb = s.ab / s.den;
    s.sinb = sqrtf(1.f - s.cosb * s.cosb + FLAT_EPS);
    s.t = s.ab / (al2 + FLAT_EPS);
#pragma unroll
    for (int d = 0; d < 3; d++) s.cb[d] = s.b[d] - a[d] * s.t;
    s.nb = s.bl1 * s.sinb;
}

struct FlatEdge { float a[3], al2, sq_al2; FlatSide s1, s2; float S, D, cos; };

__device__ __forceinline__ void flat_edge(const float* x, const int* q, FlatEdge& e)
{
    const float* v0 = x + 3 * (size_t)q[0];
    const float* v1 = x + 3 * (size_t)q[1];
    e.al2 = 0.f;
#pragma unroll
    for (int d = 0; d < 3; d++) { e.a[d] = v1[d] - v0[d]; e.al2 += e.a[d] * e.a[d]; }
    e.sq_al2 = sqrtf(e.al2 + FLAT_EPS);
    flat_side(e.a, e.al2, e.sq_al2, v0, x + 3 * (size_t)q[2], e.s1);
    flat_side(e.a, e.al2, e.sq_al2, v0, x + 3 * (size_t)q[3], e.s2);
    e.S = e.s1.cb[0] * e.s2.cb[0] + e.s1.cb[1] * e.s2.cb[1] + e.s1.cb[2] * e.s2.cb[2];
    e.D = e.s1.nb * e.s2.nb + FLAT_EPS;
    e.cos = e.S / e.D;
}

__device__ __forceinline__ float flatten_forward_block(const float* __restrict__ xn, const int* __restrict__ quads, int E, float* red)
{
    float acc = 0.f;
    for (int e = threadIdx.x; e < E; e += 256) {
        FlatEdge fe;
        flat_edge(xn, quads + 4 * (size_t)e, fe);
        acc += (fe.cos + 1.f) * (fe.cos + 1.f);
    }
    return block_sum(acc, red);
}

__device__ __forceinline__ void flat_side_backward(const FlatEdge& e, const FlatSide& s, const float* g_cb, float g_n,
                                                   float* g_a, float* g_b)
{
    float g_bl1 = g_n * s.sinb;
    const float g_cosb = (g_n * s.bl1) * (-s.cosb / s.sinb);
    float a_gcb = 0.f;
#pragma unroll
    for (int d = 0; d < 3; d++) a_gcb += e.a[d] * g_cb[d];
    const float g_t = -a_gcb;
    const float al2e = e.al2 + FLAT_EPS;
    const float g_ab = g_t / al2e + g_cosb / s.den;
    const float g_den = -g_cosb * s.ab / (s.den * s.den);
    float g_al2 = -g_t * s.ab / (al2e * al2e) + (g_den * s.bl1) / (2.f * e.sq_al2);
    g_bl1 += g_den * e.sq_al2;
    const float g_bl2 = g_bl1 / (2.f * s.bl1);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        g_b[d] = g_cb[d] + 2.f * s.b[d] * g_bl2 + e.a[d] * g_ab;
        g_a[d] += -s.t * g_cb[d] + s.b[d] * g_ab + 2.f * e.a[d] * g_al2;
    }
}

// gradient of one edge's term w.r.t. its four vertices: o[slot * 3 + d], slots (v0, v1, v2, v3); gl = d loss / d loss[n]
__device__ __forceinline__ void flatten_edge_gradient(const float* __restrict__ xn, const int* __restrict__ q, float gl, float* o)
{
    FlatEdge fe;
    flat_edge(xn, q, fe);
    const float gcos = 2.f * (fe.cos + 1.f) * gl;
    const float gD = -gcos * fe.S / (fe.D * fe.D);
    float g_cb1[3], g_cb2[3], g_a[3] = {0.f, 0.f, 0.f}, g_b1[3], g_b2[3];
#pragma unroll
    for (int d = 0; d < 3; d++) { g_cb1[d] = gcos / fe.D * fe.s2.cb[d]; g_cb2[d] = gcos / fe.D * fe.s1.cb[d]; }
    flat_side_backward(fe, fe.s1, g_cb1, gD * fe.s2.nb, g_a, g_b1);
    flat_side_backward(fe, fe.s2, g_cb2, gD * fe.s1.nb, g_a, g_b2);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        o[d] = -(g_a[d] + g_b1[d] + g_b2[d]); o[3 + d] = g_a[d]; o[6 + d] = g_b1[d]; o[9 + d] = g_b2[d];
    }
}

// vertex-centric: the vertex' incident (edge, slot) pairs in ascending order (inc [nnz] = edge * 4 + slot), each edge's gradient
// recomputed on the spot -- the values and the order of the two-stage form (edge table, then this gather)
__device__ __forceinline__ void flatten_backward_vertex(const float* __restrict__ xn, const int* __restrict__ quads, const int* __restrict__ inc_ptr,
                                                        const int* __restrict__ inc, float gl, int v, float* a)
{
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int i = inc_ptr[v]; i < inc_ptr[v + 1]; i++) {
        const int code = inc[i], e = code >> 2, slot = code & 3;
        float o[12];
        flatten_edge_gradient(xn, quads + 4 * (size_t)e, gl, o);
        a0 += slot == 0 ? o[0] : (slot == 1 ? o[3] : (slot == 2 ? o[6] : o[9]));
        a1 += slot == 0 ? o[1] : (slot == 1 ? o[4] : (slot == 2 ? o[7] : o[10]));
        a2 += slot == 0 ? o[2] : (slot == 1 ? o[5] : (slot == 2 ? o[8] : o[11]));
    }
    a[0] = a0; a[1] = a1; a[2] = a2;
}

// ---- symmetric squared Chamfer distance of one pair of small point sets (pytorch3d.loss.chamfer_distance()[0] as used on the bones'
// control points, nnutils/mesh_net.py:500-503): bodies shared by glue.hip's chamfer_* kernels and mesh_reg.hip's step launch.
// A [P,3], Bp [Q,3]; ab [P] / ba [Q] receive the nearest-neighbour indices (first minimum, like min(dim)).  256 threads.
__device__ __forceinline__ float chamfer_forward_item(const float* __restrict__ A, const float* __restrict__ Bp, int* __restrict__ ab,
                                                      int* __restrict__ ba, int P, int Q, float* red)
{
    float sa = 0.f, sb = 0.f;
    for (int i = threadIdx.x; i < P; i += 256) {
        const float x = A[3 * i], y = A[3 * i + 1], z = A[3 * i + 2];
        float best = 3.4e38f; int arg = 0;
        for (int j = 0; j < Q; j++) {
            const float dx = x - Bp[3 * j], dy = y - Bp[3 * j + 1], dz = z - Bp[3 * j + 2];
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < best) { best = d; arg = j; }                    // first minimum, like min(dim)
        }
        ab[i] = arg;
        sa += best;
    }
    for (int j = threadIdx.x; j < Q; j += 256) {
        const float x = Bp[3 * j], y = Bp[3 * j + 1], z = Bp[3 * j + 2];
        float best = 3.4e38f; int arg = 0;
        for (int i = 0; i < P; i++) {
            const float dx = x - A[3 * i], dy = y - A[3 * i + 1], dz = z - A[3 * i + 2];
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < best) { best = d; arg = i; }
        }
        ba[j] = arg;
        sb += best;
    }
    sa = block_sum(sa, red);
    sb = block_sum(sb, red);
    return sa / (float)P + sb / (float)Q;
}

__device__ __forceinline__ void chamfer_backward_item(const float* __restrict__ A, const float* __restrict__ Bp, const int* __restrict__ ab,
                                                      const int* __restrict__ ba, float g, float* __restrict__ ga,
                                                      float* __restrict__ gb, int P, int Q)
{
    const float wp = g * 2.f / (float)P, wq = g * 2.f / (float)Q;
    for (int i = threadIdx.x; i < P; i += 256) {
        const float x = A[3 * i], y = A[3 * i + 1], z = A[3 * i + 2];
        const int j0 = ab[i];
        float gx = wp * (x - Bp[3 * j0]), gy = wp * (y - Bp[3 * j0 + 1]), gz = wp * (z - Bp[3 * j0 + 2]);
        for (int j = 0; j < Q; j++)
            if (ba[j] == i) { gx += wq * (x - Bp[3 * j]); gy += wq * (y - Bp[3 * j + 1]); gz += wq * (z - Bp[3 * j + 2]); }
        ga[3 * i] = gx; ga[3 * i + 1] = gy; ga[3 * i + 2] = gz;
    }
    for (int j = threadIdx.x; j < Q; j += 256) {
        const float x = Bp[3 * j], y = Bp[3 * j + 1], z = Bp[3 * j + 2];
        const int i0 = ba[j];
        float gx = wq * (x - A[3 * i0]), gy = wq * (y - A[3 * i0 + 1]), gz = wq * (z - A[3 * i0 + 2]);
        for (int i = 0; i < P; i++)
            if (ab[i] == j) { gx += wp * (x - A[3 * i]); gy += wp * (y - A[3 * i + 1]); gz += wp * (z - A[3 * i + 2]); }
        gb[3 * j] = gx; gb[3 * j + 1] = gy; gb[3 * j + 2] = gz;
    }
}

}  // namespace lasr
