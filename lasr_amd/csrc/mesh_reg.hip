// mesh_reg.hip -- LASR's three shape regularisers in one launch each way (include/lasr_ops.h: lasr_mesh_regularisers_*).
//
// The step evaluates, on fixed connectivity,
//   Laplacian smoothness of the mean shape   third_party/ext_nnutils/loss_utils.py:34-65     (nnutils/mesh_net.py:449-453)
//   dihedral flatness of the mean shape      third_party/ext_nnutils/loss_utils.py:110-152   (nnutils/mesh_net.py:454-459)
//   ARAP between the two frames' shapes      nnutils/loss_utils.py:29-64                      (nnutils/mesh_net.py:494-497)
// as three forward launches (one workgroup per mesh each) and five backward ones (ARAP, flatten edge + vertex stages, the
// Laplacian recomputed, its transpose) plus autograd's add -- 9 launches of 5-10 us for a few hundred KB.  Here:
//   forward : grid (meshes, 3 criteria); the Laplacian coordinates are kept for the backward;
//   backward: a block owns 128 vertices of one mesh; one half of its threads forms the Laplacian part of d loss / d x, the other
//             half the flatten part straight from the vertex' incident (edge, slot) pairs -- each pair's edge gradient is
//             recomputed by the thread that needs it (4 x the arithmetic of the two-stage form, no edge table through memory) --
//             and the sum is stored once; ARAP's two gradients come from the blocks of grid rows N .. N + NA - 1.
// The per-criterion arithmetic and every summation order are those of the separate kernels (ops.hip, fused.hip): the losses and
// the gradients are bit-identical to calling them one by one.
#include <hip/hip_runtime.h>

#include "../../include/lasr_ops.h"
#include "mesh_losses.h"
#include "ops_common.h"

namespace lasr {

struct MeshRegArgs {
    const float* x;            // [N,V,3]   mean shape instances
    const float* dx;           // [NA,V,3]  ARAP: first argument  (the deformed shape of frame t)
    const float* ax;           // [NA,V,3]  ARAP: second argument (frame t')
    const int* lap_ptr; const int* lap_col;
    const int* arap_ptr; const int* arap_col;
    const int* quads; const int* inc_ptr; const int* inc;
    int N, NA, V, E;
};

__global__ __launch_bounds__(256) void mesh_reg_forward_kernel(MeshRegArgs A, float* __restrict__ lap_loss, float* __restrict__ lx,
                                                               float* __restrict__ flat_loss, float* __restrict__ arap_loss)
{
    __shared__ float red[4];
    const int n = blockIdx.x, V = A.V;
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

constexpr int MR_VPB = 128;        // vertices per block of the backward
__global__ __launch_bounds__(256) void mesh_reg_backward_kernel(MeshRegArgs A, const float* __restrict__ lx, const float* __restrict__ g_lap,
                                                                const float* __restrict__ g_flat, const float* __restrict__ g_arap,
                                                                float* __restrict__ gx, float* __restrict__ gdx, float* __restrict__ gax)
{
    __shared__ float part[MR_VPB][3];
    const int V = A.V, tid = threadIdx.x, half = tid >> 7, t = tid & (MR_VPB - 1);
    const int v = blockIdx.x * MR_VPB + t;
    if ((int)blockIdx.y >= A.N) {                              // ARAP rows
        const int n = blockIdx.y - A.N;
        if (half == 0 && v < V) {
            const size_t o = ((size_t)n * V + v) * 3;
            arap_backward_vertex(A.ax + (size_t)n * V * 3, A.dx + (size_t)n * V * 3, A.arap_ptr, A.arap_col,
                                 4.f * g_arap[n] / (float)A.arap_ptr[V], v, gax ? gax + o : nullptr, gdx ? gdx + o : nullptr);
        }
        return;
    }
    const int n = blockIdx.y;
    float a[3] = {0.f, 0.f, 0.f};
    if (v < V) {
        if (half == 0) laplacian_backward_vertex(lx + (size_t)n * V * 3, A.lap_ptr, A.lap_col, 2.f * g_lap[n], v, a);
        else flatten_backward_vertex(A.x + (size_t)n * V * 3, A.quads, A.inc_ptr, A.inc, g_flat[n], v, a);
    }
    if (half == 1) { part[t][0] = a[0]; part[t][1] = a[1]; part[t][2] = a[2]; }
    __syncthreads();
    if (half == 0 && v < V) {
        float* o = gx + ((size_t)n * V + v) * 3;
        o[0] = a[0] + part[t][0]; o[1] = a[1] + part[t][1]; o[2] = a[2] + part[t][2];
    }
}

}  // namespace lasr

using namespace lasr;

extern "C" int lasr_mesh_regularisers_forward(const float* x, const float* arap_dx, const float* arap_x, const int* lap_row_ptr,
                                              const int* lap_col, const int* arap_row_ptr, const int* arap_col, const int* quads,
                                              float* lap_loss, float* lap_coords, float* flat_loss, float* arap_loss, int N, int NA,
                                              int V, int E, void* hip_stream)
{
    if (N < 0 || NA < 0 || V < 0 || E < 0) return LASR_E_BADARG;
    if ((N == 0 && NA == 0) || V == 0) return LASR_OK;
    if (N > 0 && (!x || !lap_row_ptr || !lap_col || !lap_loss || !lap_coords || !flat_loss || (E > 0 && !quads))) return LASR_E_BADARG;
    if (NA > 0 && (!arap_dx || !arap_x || !arap_row_ptr || !arap_col || !arap_loss)) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    MeshRegArgs A{x, arap_dx, arap_x, lap_row_ptr, lap_col, arap_row_ptr, arap_col, quads, nullptr, nullptr, N, NA, V, E};
    LASR_LAUNCH(K_MESH_REG, mesh_reg_forward_kernel, dim3(N > NA ? N : NA, 3), dim3(256), 0, A, lap_loss, lap_coords, flat_loss, arap_loss);
    return launch_ok();
}

extern "C" int lasr_mesh_regularisers_backward(const float* x, const float* arap_dx, const float* arap_x, const int* lap_row_ptr,
                                               const int* lap_col, const int* arap_row_ptr, const int* arap_col, const int* quads,
                                               const int* inc_ptr, const int* inc, const float* lap_coords, const float* grad_lap,
                                               const float* grad_flat, const float* grad_arap, float* grad_x, float* grad_arap_dx,
                                               float* grad_arap_x, int N, int NA, int V, int E, void* hip_stream)
{
    if (N < 0 || NA < 0 || V < 0 || E < 0) return LASR_E_BADARG;
    if ((N == 0 && NA == 0) || V == 0) return LASR_OK;
    if (N > 0 && (!x || !lap_row_ptr || !lap_col || !lap_coords || !grad_lap || !grad_flat || !grad_x || !inc_ptr ||
                  (E > 0 && (!quads || !inc)))) return LASR_E_BADARG;
    if (NA > 0 && (!arap_dx || !arap_x || !arap_row_ptr || !arap_col || !grad_arap)) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    MeshRegArgs A{x, arap_dx, arap_x, lap_row_ptr, lap_col, arap_row_ptr, arap_col, quads, inc_ptr, inc, N, NA, V, E};
    LASR_LAUNCH(K_MESH_REG, mesh_reg_backward_kernel, dim3((V + MR_VPB - 1) / MR_VPB, N + NA), dim3(256), 0, A, lap_coords, grad_lap,
                grad_flat, grad_arap, grad_x, grad_arap_dx, grad_arap_x);
    return launch_ok();
}
