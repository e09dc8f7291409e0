// mesh_reg.hip -- LASR's three shape regularisers in one launch each way (include/lasr_ops.h: lasr_mesh_regularisers_*).
//
// The step evaluates, on fixed connectivity,
//   Laplacian smoothness of the mean shape   third_party/ext_nnutils/loss_utils.py:34-65     (nnutils/mesh_net.py:449-453)
//   dihedral flatness of the mean shape      third_party/ext_nnutils/loss_utils.py:110-152   (nnutils/mesh_net.py:454-459)
//   ARAP between the two frames' shapes      nnutils/loss_utils.py:29-64                      (nnutils/mesh_net.py:494-497)
// as three forward launches (one workgroup per mesh each) and five backward ones (ARAP, flatten edge + vertex stages, the
// Laplacian recomputed, its transpose) plus autograd's add -- 9 launches of 5-10 us for a few hundred KB.  Here:
//   forward : grid (meshes, 3 criteria); the Laplacian coordinates are kept for the backward;
//   backward: a block owns 16 vertices of one mesh, 16 lanes each: the flatten part straight from the vertex' incident
//             (edge, slot) pairs -- each pair's edge gradient is recomputed by the lane that holds it (4 x the arithmetic of the
//             two-stage form, no edge table through memory) -- then the Laplacian part, and the sum is stored once; ARAP's two
//             gradients come from the blocks of grid rows N .. N + NA - 1.
// The per-criterion arithmetic and every summation order are those of the separate kernels (ops.hip, fused.hip): the losses and
// the gradients are bit-identical to calling them one by one.
#include <hip/hip_runtime.h>

#include "../../include/lasr_ops.h"
#include "mesh_losses.h"
#include "ops_common.h"

namespace lasr {

// max over the workgroup (every thread gets it): the number of 16-element rounds the block's longest list needs, so that all
// threads reach the same barriers
__device__ __forceinline__ int __reduce_max_block16(int x)
{
    __shared__ int s_max;
    __syncthreads();
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    atomicMax(&s_max, x);
    __syncthreads();
    return s_max;
}

struct MeshRegArgs {
    const float* x;            // [N,V,3]   mean shape instances
    const float* dx;           // [NA,V,3]  ARAP: first argument  (the deformed shape of frame t)
    const float* ax;           // [NA,V,3]  ARAP: second argument (frame t')
    const int* lap_ptr; const int* lap_col;
    const int* arap_ptr; const int* arap_col;
    const int* quads; const int* inc_ptr; const int* inc;
    int N, NA, V, E;
    // optional fourth criterion of the step's launch (lasr_step_regularisers_*): the symmetric Chamfer distance of NC pairs of small
    // point sets (the bones' control points against their mirror images, nnutils/mesh_net.py:500-503)
    const float* ca; const float* cb;    // [NC,P,3], [NC,Q,3]
    int* nn_ab; int* nn_ba;              // [NC,P], [NC,Q] nearest-neighbour indices: written by the forward, read by the backward
    int NC, CP, CQ;
};

__global__ __launch_bounds__(256) void mesh_reg_forward_kernel(MeshRegArgs A, float* __restrict__ lap_loss, float* __restrict__ lx,
                                                               float* __restrict__ flat_loss, float* __restrict__ arap_loss,
                                                               float* __restrict__ cham_loss)
{
    __shared__ float red[4];
    const int n = blockIdx.x, V = A.V;
    if (blockIdx.y == 3) {
        if (n >= A.NC) return;
        const float s = chamfer_forward_item(A.ca + (size_t)n * A.CP * 3, A.cb + (size_t)n * A.CQ * 3, A.nn_ab + (size_t)n * A.CP,
                                             A.nn_ba + (size_t)n * A.CQ, A.CP, A.CQ, red);
        if (threadIdx.x == 0) cham_loss[n] = s;
        return;
    }
    if (blockIdx.y == 0) {
        if (n >= A.N) return;
        const float s = laplacian_forward_block(A.x + (size_t)n * V * 3, A.lap_ptr, A.lap_col, lx + (size_t)n * V * 3, V, red);
        if (threadIdx.x == 0) lap_loss[n] = s;
    } else if (blockIdx.y == 1) {
        if (n >= A.N) return;
        const float s = flatten_forward_block(A.x + (size_t)n * V * 3, A.quads, A.E, red);
        if (threadIdx.x == 0) flat_loss[n] = s;
    } else {
        if (n >= A.NA) return;
        const float s = arap_forward_block(A.ax + (size_t)n * V * 3, A.dx + (size_t)n * V * 3, A.arap_ptr, A.arap_col, V, red);
        if (threadIdx.x == 0) arap_loss[n] = s / (float)A.arap_ptr[V];
    }
}

// Backward: 16 lanes per vertex.  The per-vertex loops of the separate kernels (incident (edge, slot) pairs of the flatten term,
// neighbours of the Laplacian / ARAP terms) are chains of dependent gathers -- index, then indices of the edge's vertices, then
// coordinates -- walked one element after the other: 12 x 3 latency levels per vertex for the flatten part.  Here lane j of a
// vertex' group evaluates element j (its products / its edge gradient) and parks the three numbers in LDS; lane 0 then adds them
// up IN ELEMENT ORDER with the same expressions, so the sums are bit-identical to the sequential loops while the gathers of all
// elements are in flight together.
constexpr int MR_VPB = 16;         // vertices per block of the backward (x 16 lanes)

__global__ __launch_bounds__(256) void mesh_reg_backward_kernel(MeshRegArgs A, const float* __restrict__ lx, const float* __restrict__ g_lap,
                                                                const float* __restrict__ g_flat, const float* __restrict__ g_arap,
                                                                float* __restrict__ gx, float* __restrict__ gdx, float* __restrict__ gax,
                                                                const float* __restrict__ g_cham, float* __restrict__ gca,
                                                                float* __restrict__ gcb)
{
    __shared__ float term[MR_VPB][16][6];
    if ((int)blockIdx.y >= A.N + A.NA) {                       // ---- Chamfer pairs: grid rows N + NA .. N + NA + NC - 1, one block each
        if (blockIdx.x != 0) return;
        const int n = blockIdx.y - A.N - A.NA;
        chamfer_backward_item(A.ca + (size_t)n * A.CP * 3, A.cb + (size_t)n * A.CQ * 3, A.nn_ab + (size_t)n * A.CP,
                              A.nn_ba + (size_t)n * A.CQ, g_cham[n], gca + (size_t)n * A.CP * 3, gcb + (size_t)n * A.CQ * 3, A.CP, A.CQ);
        return;
    }
    const int V = A.V, tid = threadIdx.x, vl = tid >> 4, lane = tid & 15;
    const int v = blockIdx.x * MR_VPB + vl;
    const bool live = v < V;
    if ((int)blockIdx.y >= A.N) {                              // ---- ARAP rows: each neighbour's (sg * p, sg * q)
        const int n = blockIdx.y - A.N;
        const float* X = A.ax + (size_t)n * V * 3;
        const float* D = A.dx + (size_t)n * V * 3;
        const int e0 = live ? A.arap_ptr[v] : 0, e1 = live ? A.arap_ptr[v + 1] : 0;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
        if (live) { a0 = X[3 * v]; a1 = X[3 * v + 1]; a2 = X[3 * v + 2]; b0 = D[3 * v]; b1 = D[3 * v + 1]; b2 = D[3 * v + 2]; }
        float gx0 = 0, gx1 = 0, gx2 = 0, gd0 = 0, gd1 = 0, gd2 = 0;
        int rounds = (e1 - e0 + 15) >> 4;
        rounds = __reduce_max_block16(rounds);
        for (int r = 0; r < rounds; r++) {
            const int e = e0 + 16 * r + lane;
            if (e < e1) {
                const int u = A.arap_col[e];
                const float p0 = a0 - X[3 * u], p1 = a1 - X[3 * u + 1], p2 = a2 - X[3 * u + 2];
                const float q0 = b0 - D[3 * u], q1 = b1 - D[3 * u + 1], q2 = b2 - D[3 * u + 2];
                const float sg = mesh_sgn((p0 * p0 + p1 * p1 + p2 * p2) - (q0 * q0 + q1 * q1 + q2 * q2));
                float* t = term[vl][lane];
                t[0] = sg * p0; t[1] = sg * p1; t[2] = sg * p2; t[3] = sg * q0; t[4] = sg * q1; t[5] = sg * q2;
            }
            __syncthreads();
            if (lane == 0) {
                const int m = min(16, e1 - e0 - 16 * r);
                for (int j = 0; j < m; j++) {
                    const float* t = term[vl][j];
                    gx0 += t[0]; gx1 += t[1]; gx2 += t[2]; gd0 -= t[3]; gd1 -= t[4]; gd2 -= t[5];
                }
            }
            __syncthreads();
        }
        if (lane == 0 && live) {
            const float k = 4.f * g_arap[n] / (float)A.arap_ptr[V];
            const size_t o = ((size_t)n * V + v) * 3;
            if (gax) { gax[o] = k * gx0; gax[o + 1] = k * gx1; gax[o + 2] = k * gx2; }
            if (gdx) { gdx[o] = k * gd0; gdx[o + 1] = k * gd1; gdx[o + 2] = k * gd2; }
        }
        return;
    }
    const int n = blockIdx.y;
    const float* xn = A.x + (size_t)n * V * 3;
    const float* L = lx + (size_t)n * V * 3;
    // ---- flatten part: lane j = the vertex' j-th incident (edge, slot) pair
    float f0 = 0.f, f1 = 0.f, f2 = 0.f;
    {
        const int i0 = live ? A.inc_ptr[v] : 0, i1 = live ? A.inc_ptr[v + 1] : 0;
        const float gl = g_flat[n];
        int rounds = (i1 - i0 + 15) >> 4;
        rounds = __reduce_max_block16(rounds);
        for (int r = 0; r < rounds; r++) {
            const int i = i0 + 16 * r + lane;
            if (i < i1) {
                const int code = A.inc[i], e = code >> 2, slot = code & 3;
                float o[12];
                flatten_edge_gradient(xn, A.quads + 4 * (size_t)e, gl, o);
                float* t = term[vl][lane];
                t[0] = slot == 0 ? o[0] : (slot == 1 ? o[3] : (slot == 2 ? o[6] : o[9]));
                t[1] = slot == 0 ? o[1] : (slot == 1 ? o[4] : (slot == 2 ? o[7] : o[10]));
                t[2] = slot == 0 ? o[2] : (slot == 1 ? o[5] : (slot == 2 ? o[8] : o[11]));
            }
            __syncthreads();
            if (lane == 0) {
                const int m = min(16, i1 - i0 - 16 * r);
                for (int j = 0; j < m; j++) { f0 += term[vl][j][0]; f1 += term[vl][j][1]; f2 += term[vl][j][2]; }
            }
            __syncthreads();
        }
    }
    // ---- Laplacian part: lane j = the j-th neighbour's lx[u] / deg(u)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    {
        const int e0 = live ? A.lap_ptr[v] : 0, e1 = live ? A.lap_ptr[v + 1] : 0;
        if (live && lane == 0) { a0 = L[3 * v]; a1 = L[3 * v + 1]; a2 = L[3 * v + 2]; }
        int rounds = (e1 - e0 + 15) >> 4;
        rounds = __reduce_max_block16(rounds);
        for (int r = 0; r < rounds; r++) {
            const int e = e0 + 16 * r + lane;
            if (e < e1) {
                const int u = A.lap_col[e];
                const float inv = 1.f / (float)(A.lap_ptr[u + 1] - A.lap_ptr[u]);
                float* t = term[vl][lane];
                t[0] = L[3 * u] * inv; t[1] = L[3 * u + 1] * inv; t[2] = L[3 * u + 2] * inv;
            }
            __syncthreads();
            if (lane == 0) {
                const int m = min(16, e1 - e0 - 16 * r);
                for (int j = 0; j < m; j++) { a0 -= term[vl][j][0]; a1 -= term[vl][j][1]; a2 -= term[vl][j][2]; }
            }
            __syncthreads();
        }
    }
    if (lane == 0 && live) {
        const float k = 2.f * g_lap[n];
        float* o = gx + ((size_t)n * V + v) * 3;
        o[0] = k * a0 + f0; o[1] = k * a1 + f1; o[2] = k * a2 + f2;
    }
}

}  // namespace lasr

using namespace lasr;

static int step_reg_forward(MeshRegArgs A, float* lap_loss, float* lap_coords, float* flat_loss, float* arap_loss, float* cham_loss,
                            hipStream_t st)
{
    const int gx = (A.N > A.NA ? A.N : A.NA) > A.NC ? (A.N > A.NA ? A.N : A.NA) : A.NC;
    LASR_LAUNCH(K_MESH_REG, mesh_reg_forward_kernel, dim3(gx, A.NC > 0 ? 4 : 3), dim3(256), 0, A, lap_loss, lap_coords, flat_loss,
                arap_loss, cham_loss);
    return launch_ok();
}

extern "C" int lasr_mesh_regularisers_forward(const float* x, const float* arap_dx, const float* arap_x, const int* lap_row_ptr,
                                              const int* lap_col, const int* arap_row_ptr, const int* arap_col, const int* quads,
                                              float* lap_loss, float* lap_coords, float* flat_loss, float* arap_loss, int N, int NA,
                                              int V, int E, void* hip_stream)
{
    return lasr_step_regularisers_forward(x, arap_dx, arap_x, lap_row_ptr, lap_col, arap_row_ptr, arap_col, quads, lap_loss, lap_coords,
                                          flat_loss, arap_loss, N, NA, V, E, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, hip_stream);
}

extern "C" int lasr_step_regularisers_forward(const float* x, const float* arap_dx, const float* arap_x, const int* lap_row_ptr,
                                              const int* lap_col, const int* arap_row_ptr, const int* arap_col, const int* quads,
                                              float* lap_loss, float* lap_coords, float* flat_loss, float* arap_loss, int N, int NA,
                                              int V, int E, const float* cham_a, const float* cham_b, float* cham_loss, int* nn_ab,
                                              int* nn_ba, int NC, int P, int Q, void* hip_stream)
{
    if (N < 0 || NA < 0 || V < 0 || E < 0 || NC < 0 || (NC > 0 && (P < 1 || Q < 1))) return LASR_E_BADARG;
    if (V == 0) { N = 0; NA = 0; }
    if (N == 0 && NA == 0 && NC == 0) return LASR_OK;
    if (N > 0 && (!x || !lap_row_ptr || !lap_col || !lap_loss || !lap_coords || !flat_loss || (E > 0 && !quads))) return LASR_E_BADARG;
    if (NA > 0 && (!arap_dx || !arap_x || !arap_row_ptr || !arap_col || !arap_loss)) return LASR_E_BADARG;
    if (NC > 0 && (!cham_a || !cham_b || !cham_loss || !nn_ab || !nn_ba)) return LASR_E_BADARG;
    MeshRegArgs A{x, arap_dx, arap_x, lap_row_ptr, lap_col, arap_row_ptr, arap_col, quads, nullptr, nullptr, N, NA, V, E,
                  cham_a, cham_b, nn_ab, nn_ba, NC, P, Q};
    return step_reg_forward(A, lap_loss, lap_coords, flat_loss, arap_loss, cham_loss, (hipStream_t)hip_stream);
}

extern "C" int lasr_mesh_regularisers_backward(const float* x, const float* arap_dx, const float* arap_x, const int* lap_row_ptr,
                                               const int* lap_col, const int* arap_row_ptr, const int* arap_col, const int* quads,
                                               const int* inc_ptr, const int* inc, const float* lap_coords, const float* grad_lap,
                                               const float* grad_flat, const float* grad_arap, float* grad_x, float* grad_arap_dx,
                                               float* grad_arap_x, int N, int NA, int V, int E, void* hip_stream)
{
    return lasr_step_regularisers_backward(x, arap_dx, arap_x, lap_row_ptr, lap_col, arap_row_ptr, arap_col, quads, inc_ptr, inc, lap_coords,
                                           grad_lap, grad_flat, grad_arap, grad_x, grad_arap_dx, grad_arap_x, N, NA, V, E, nullptr, nullptr,
                                           nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, hip_stream);
}

extern "C" int lasr_step_regularisers_backward(const float* x, const float* arap_dx, const float* arap_x, const int* lap_row_ptr,
                                               const int* lap_col, const int* arap_row_ptr, const int* arap_col, const int* quads,
                                               const int* inc_ptr, const int* inc, const float* lap_coords, const float* grad_lap,
                                               const float* grad_flat, const float* grad_arap, float* grad_x, float* grad_arap_dx,
                                               float* grad_arap_x, int N, int NA, int V, int E, const float* cham_a, const float* cham_b,
                                               const int* nn_ab, const int* nn_ba, const float* grad_cham, float* grad_cham_a,
                                               float* grad_cham_b, int NC, int P, int Q, void* hip_stream)
{
    if (N < 0 || NA < 0 || V < 0 || E < 0 || NC < 0 || (NC > 0 && (P < 1 || Q < 1))) return LASR_E_BADARG;
    if (V == 0) { N = 0; NA = 0; }
    if (N == 0 && NA == 0 && NC == 0) return LASR_OK;
    if (N > 0 && (!x || !lap_row_ptr || !lap_col || !lap_coords || !grad_lap || !grad_flat || !grad_x || !inc_ptr ||
                  (E > 0 && (!quads || !inc)))) return LASR_E_BADARG;
    if (NA > 0 && (!arap_dx || !arap_x || !arap_row_ptr || !arap_col || !grad_arap)) return LASR_E_BADARG;
    if (NC > 0 && (!cham_a || !cham_b || !nn_ab || !nn_ba || !grad_cham || !grad_cham_a || !grad_cham_b)) return LASR_E_BADARG;
    if (N + NA + NC > 65535) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    MeshRegArgs A{x, arap_dx, arap_x, lap_row_ptr, lap_col, arap_row_ptr, arap_col, quads, inc_ptr, inc, N, NA, V, E,
                  cham_a, cham_b, const_cast<int*>(nn_ab), const_cast<int*>(nn_ba), NC, P, Q};
    const int gx = V > 0 && N + NA > 0 ? (V + MR_VPB - 1) / MR_VPB : 1;
    LASR_LAUNCH(K_MESH_REG, mesh_reg_backward_kernel, dim3(gx, N + NA + NC), dim3(256), 0, A, lap_coords, grad_lap,
                grad_flat, grad_arap, grad_x, grad_arap_dx, grad_arap_x, grad_cham, grad_cham_a, grad_cham_b);
    return launch_ok();
}
