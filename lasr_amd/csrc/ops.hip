// ops.hip -- geometry and loss kernels around the rasteriser (include/lasr_ops.h), gfx950 / wave64.
//
// All of these replace chains of eager PyTorch ops in the reference (paths relative to /root/reference/):
//   LBS        nnutils/geom_utils.py:45-71   K-1 separate bmm launches + a [N,K-1,V,3] temporary
//   pinhole    nnutils/geom_utils.py:27-34   6 elementwise kernels + clones
//   losses     nnutils/mesh_net.py:374-447   python loops over (image, hypothesis) with boolean-mask indexing
//   ARAP       nnutils/loss_utils.py:46-64   six dense [N,V,V] tensors (O(N V^3) flops)
//   Laplacian  third_party/ext_nnutils/loss_utils.py:57-65   dense [V,V] matmul
// Every reduction here is deterministic (fixed tree inside a block, no float atomics).
#include <hip/hip_runtime.h>

#include "../../include/lasr_ops.h"
#include "host_common.h"
#include "sr_device.h"
#include "ops_common.h"
#include "mesh_losses.h"

namespace lasr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float dpp_f(float v, const int ctrl)
{
    // compile-time ctrl required by the builtin: dispatch on the few patterns used here
    switch (ctrl) {
        case 0x111: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));
        case 0x112: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, false));
        case 0x113: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x113, 0xf, 0xf, false));
        case 0x116: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x116, 0xf, 0xf, false));
        case 0x55: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x55, 0xf, 0xf, false));
        case 0xAA: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xAA, 0xf, 0xf, false));
        default: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xFF, 0xf, 0xf, false));
    }
}

// ===========================================================================
// Linear-blend skinning
// ===========================================================================
// One wave = 16 vertices.  C[16 vertices, 16 cols] = A[16, K-1] (skin^T) x B[K-1, 16] where the 12 live
// columns of B are the bone's row-major R (9) and T (3): v_mfma_f32_16x16x4_f32, 4 bones per instruction.
// Fragment maps (cdna_hip_programming.md section 3): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// C[row = (lane >> 4) * 4 + r][col = lane & 15].
__global__ __launch_bounds__(256) void lbs_forward_kernel(const float* __restrict__ verts, const float* __restrict__ Rmat,
                                                          const float* __restrict__ Tmat, const float* __restrict__ skin,
                                                          float* __restrict__ out, int N, int V, int K, int tocam,
                                                          float* __restrict__ out_blend)
{
    // out_blend (optional, with tocam): the blended vertices BEFORE the body transform as a second output -- LASR.forward needs
    // both (deform_v, nnutils/mesh_net.py:291, and the camera-space vertices, :298) and the reference calls obj_to_cam twice
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v0 = (blockIdx.x * 4 + wave) * 16;
    if (v0 >= V) return;
    const int col = lane & 15, kc = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nb = K - 1;
    for (int kk = 0; kk < nb; kk += 4) {
        const int k = kk + kc;
        const int vi = v0 + col;
        float a = 0.f, b = 0.f;
        if (k < nb) {
            if (vi < V) a = skin[((size_t)n * nb + k) * V + vi];
            const size_t bone = (size_t)n * K + k + 1;
            if (col < 9) b = Rmat[bone * 9 + col];
            else if (col < 12) b = Tmat[bone * 3 + (col - 9)];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    const float* R0 = Rmat + (size_t)n * K * 9;
    const float* T0 = Tmat + (size_t)n * K * 3;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int vert = v0 + kc * 4 + r;
        const bool live = vert < V;
        const float* p = verts + ((size_t)n * V + (live ? vert : 0)) * 3;
        float x;
        if (nb > 0) {
            // lane `col` holds M[i][j] (col = 3i+j < 9) or t[j] (col = 9+j): form v_i*M[i][j] | t[j] and fold
            // the four lanes {j, 3+j, 6+j, 9+j} of the 16-lane row into lane 9+j
            x = col < 9 ? p[col / 3] * acc[r] : (col < 12 ? acc[r] : 0.f);
            x += dpp_f(x, 0x113);          // row_shr:3
            x += dpp_f(x, 0x116);          // row_shr:6
        } else {
            x = (col >= 9 && col < 12) ? p[col - 9] : 0.f;
        }
        if (out_blend && live && col >= 9 && col < 12) out_blend[((size_t)n * V + vert) * 3 + (col - 9)] = x;
        if (tocam) {
            // lanes 8..11 form a quad: broadcast vs_0..2 from its lanes 1..3
            const float s0 = dpp_f(x, 0x55), s1 = dpp_f(x, 0xAA), s2 = dpp_f(x, 0xFF);
            const int j = (col >= 9 && col < 12) ? col - 9 : 0;
            x = s0 * R0[j] + s1 * R0[3 + j] + s2 * R0[6 + j] + T0[j];
        }
        if (live && col >= 9 && col < 12) out[((size_t)n * V + vert) * 3 + (col - 9)] = x;
    }
}

// Backward.  Pass 1, one block per (mesh, 256-vertex chunk): thread per vertex -> g_verts, g_skin and the block's
// partial sums of the transform gradients: the body transform through block_sum, the part bones as
// partial[k][c] = sum_{v in chunk} skin[k][v] * G[v][c] with G (= d out / d blended transform, 12 columns) staged
// in LDS and one thread per (bone, column).  Pass 2 folds the chunk partials in a fixed order.  No atomics.
constexpr int LBS_CHUNK = 256;

__global__ __launch_bounds__(256) void lbs_backward_kernel(const float* __restrict__ verts, const float* __restrict__ Rmat,
                                                           const float* __restrict__ Tmat, const float* __restrict__ skin,
                                                           const float* __restrict__ gout, float* __restrict__ gverts,
                                                           float* __restrict__ gskin, float* __restrict__ partial,
                                                           int N, int V, int K, int tocam, const float* __restrict__ gout_blend)
{
    extern __shared__ float lds[];          // RT[K*12] | G[256*13] | red[4]
    const int n = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x, tid = threadIdx.x, nb = K - 1;
    float* RT = lds;                        // RT[k*12 + c], k = 0 body
    float* Gs = lds + K * 12;               // G[v_local*13 + c] (13: odd stride, conflict-free column reads)
    float* red = Gs + LBS_CHUNK * 13;
    for (int i = tid; i < K * 12; i += 256) {
        const int k = i / 12, c = i - k * 12;
        RT[i] = c < 9 ? Rmat[((size_t)n * K + k) * 9 + c] : Tmat[((size_t)n * K + k) * 3 + (c - 9)];
    }
    __syncthreads();
    const int v = chunk * LBS_CHUNK + tid;
    float accR0[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, accT0[3] = {0, 0, 0};
    float G[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (v < V) {
        const size_t o = ((size_t)n * V + v) * 3;
        const float px = verts[o], py = verts[o + 1], pz = verts[o + 2];
        const float g0 = gout[o], g1 = gout[o + 1], g2 = gout[o + 2];
        float h0 = g0, h1 = g1, h2 = g2;    // g_vs
        if (tocam) {                         // g_vs = g_out @ R0^T
            h0 = g0 * RT[0] + g1 * RT[1] + g2 * RT[2];
            h1 = g0 * RT[3] + g1 * RT[4] + g2 * RT[5];
            h2 = g0 * RT[6] + g1 * RT[7] + g2 * RT[8];
        }
        if (gout_blend) { h0 += gout_blend[o]; h1 += gout_blend[o + 1]; h2 += gout_blend[o + 2]; }   // second output's gradient
        float M[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
        if (nb > 0) {
#pragma unroll
            for (int c = 0; c < 12; c++) M[c] = 0.f;
            // G[c] = d out / d RT_blend[c] contracted with g_vs: (v_i * h_j | h_j)
            G[0] = px * h0; G[1] = px * h1; G[2] = px * h2; G[3] = py * h0; G[4] = py * h1; G[5] = py * h2;
            G[6] = pz * h0; G[7] = pz * h1; G[8] = pz * h2; G[9] = h0; G[10] = h1; G[11] = h2;
            for (int k = 0; k < nb; k++) {
                const float sk = skin[((size_t)n * nb + k) * V + v];
                const float* b = RT + (k + 1) * 12;
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < 12; c++) { M[c] += sk * b[c]; d += G[c] * b[c]; }
                if (gskin) gskin[((size_t)n * nb + k) * V + v] = d;
            }
        }
        if (gverts) {                        // g_v = g_vs @ M^T
            gverts[o] = h0 * M[0] + h1 * M[1] + h2 * M[2];
            gverts[o + 1] = h0 * M[3] + h1 * M[4] + h2 * M[5];
            gverts[o + 2] = h0 * M[6] + h1 * M[7] + h2 * M[8];
        }
        if (tocam) {                         // body transform: g_R0 += vs^T g_out, g_T0 += g_out
            const float s0 = px * M[0] + py * M[3] + pz * M[6] + M[9];
            const float s1 = px * M[1] + py * M[4] + pz * M[7] + M[10];
            const float s2 = px * M[2] + py * M[5] + pz * M[8] + M[11];
            accR0[0] = s0 * g0; accR0[1] = s0 * g1; accR0[2] = s0 * g2;
            accR0[3] = s1 * g0; accR0[4] = s1 * g1; accR0[5] = s1 * g2;
            accR0[6] = s2 * g0; accR0[7] = s2 * g1; accR0[8] = s2 * g2;
            accT0[0] = g0; accT0[1] = g1; accT0[2] = g2;
        }
    }
#pragma unroll
    for (int c = 0; c < 12; c++) Gs[tid * 13 + c] = G[c];      // zeros for v >= V
    float* P = partial + ((size_t)n * nchunks + chunk) * K * 12;
#pragma unroll
    for (int c = 0; c < 9; c++) { const float t = block_sum(accR0[c], red); if (tid == 0) P[c] = tocam ? t : 0.f; }
#pragma unroll
    for (int c = 0; c < 3; c++) { const float t = block_sum(accT0[c], red); if (tid == 0) P[9 + c] = tocam ? t : 0.f; }
    __syncthreads();
    const int nv = min(LBS_CHUNK, V - chunk * LBS_CHUNK);
    for (int i = tid; i < nb * 12; i += 256) {               // thread per (part bone, column)
        const int k = i / 12, c = i - k * 12;
        const float* sk = skin + ((size_t)n * nb + k) * V + (size_t)chunk * LBS_CHUNK;
        float a = 0.f;
        for (int u = 0; u < nv; u++) a += sk[u] * Gs[u * 13 + c];
        P[(k + 1) * 12 + c] = a;
    }
}

__global__ __launch_bounds__(256) void lbs_backward_fold_kernel(const float* __restrict__ partial, float* __restrict__ gR,
                                                                float* __restrict__ gT, int K, int nchunks)
{
    const int n = blockIdx.x;
    for (int i = threadIdx.x; i < K * 12; i += 256) {
        float a = 0.f;
#pragma unroll 8
        for (int ch = 0; ch < nchunks; ch++) a += partial[((size_t)n * nchunks + ch) * K * 12 + i];
        const int k = i / 12, c = i - k * 12;
        if (c < 9) { if (gR) gR[((size_t)n * K + k) * 9 + c] = a; }
        else if (gT) gT[((size_t)n * K + k) * 3 + (c - 9)] = a;
    }
}

// ---- backward on the matrix cores, one launch ----------------------------------------------------------------------------------
// Rounds 1-4 ran the kernel above: a per-thread loop over the bones for M and g_skin, and the transposed contraction
// g_RT[k][c] = sum_v skin[k][v] G[v][c] as one thread per (bone, column) walking 256 vertices serially -- 48 workgroups and
// 18 us per call at LASR's 16 x 642 x 21, plus the fold launch.  Here a block owns 64 vertices of one mesh (4 waves x 16: 176
// blocks at that size) and all three contractions run on v_mfma_f32_16x16x4_f32 from LDS-staged operands:
//   M      [16 v x 12]   = skin^T [16 v x nb]   x RT [nb x 12]        (the forward's product; needed for g_verts and g_R0)
//   g_skin [16 v x nb]   = G      [16 v x 12]   x RT^T [12 x nb]
//   g_RT   [nb x 12]    += skin   [nb x 16 v]   x G  [16 v x 12]      (per wave; the four waves' tiles meet in LDS, wave order)
// with G[v] = (v_i * h_j | h_j), h = d loss / d blended vertex.  The per-vertex pieces (h, G, g_verts, the body transform's
// outer products) are thread-per-vertex code around them.  Chunk partials of the transform gradients go to scratch and
// lbs_backward_fold_kernel folds them in chunk order; a caller that needs no transform / skin gradients (LASR's joint and
// control-point call: only the points receive gradient) passes NULL and the contractions and the fold are skipped.
constexpr int LBSB_VERTS = 64;
constexpr int LBSB_MAX_K = 65;          // nb <= 64 part bones: Ss / Cs tiles of at most 64 rows

__global__ __launch_bounds__(256) void lbs_backward_mfma_kernel(const float* __restrict__ verts, const float* __restrict__ Rmat,
                                                                const float* __restrict__ Tmat, const float* __restrict__ skin,
                                                                const float* __restrict__ gout, float* __restrict__ gverts,
                                                                float* __restrict__ gskin, float* __restrict__ partial,
                                                                int N, int V, int K, int tocam, const float* __restrict__ gout_blend)
{
    extern __shared__ float lds[];
    const int n = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x, tid = threadIdx.x, nb = K - 1;
    const int nbp = (nb + 15) & ~15;                        // bones padded to whole 16-row MFMA tiles
    float* RT = lds;                                         // [K][12], k = 0 body
    float* Gs = RT + ((K * 12 + 3) & ~3);                    // [64][16]  G rows (12 live columns), later the body-transform products
    float* Ms = Gs + LBSB_VERTS * 16;                        // [64][16]  blended transform per vertex
    float* Ss = Ms + LBSB_VERTS * 16;                        // [nbp][64] skin tile, zero padded
    float* Cs = Ss + nbp * LBSB_VERTS;                       // [4][nbp][16] per-wave g_RT tiles
    const int lane = tid & 63, wave = tid >> 6, col = lane & 15, kc = lane >> 4;
    const int v0 = chunk * LBSB_VERTS;
    for (int i = tid; i < K * 12; i += 256) {
        const int k = i / 12, c = i - k * 12;
        RT[i] = c < 9 ? Rmat[((size_t)n * K + k) * 9 + c] : Tmat[((size_t)n * K + k) * 3 + (c - 9)];
    }
    for (int i = tid; i < nbp * LBSB_VERTS; i += 256) {       // coalesced rows of 64 vertices
        const int k = i >> 6, u = i & 63;
        Ss[i] = (k < nb && v0 + u < V) ? skin[((size_t)n * nb + k) * V + v0 + u] : 0.f;
    }
    __syncthreads();
    // ---- per vertex, part 1 (wave 0, thread = vertex): h and G
    const int v = v0 + tid;
    const bool mine = tid < LBSB_VERTS && v < V;
    float px = 0.f, py = 0.f, pz = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, h0 = 0.f, h1 = 0.f, h2 = 0.f;
    if (tid < LBSB_VERTS) {
        if (mine) {
            const size_t o = ((size_t)n * V + v) * 3;
            px = verts[o]; py = verts[o + 1]; pz = verts[o + 2];
            g0 = gout[o]; g1 = gout[o + 1]; g2 = gout[o + 2];
            h0 = g0; h1 = g1; h2 = g2;
            if (tocam) {                         // g_vs = g_out @ R0^T
                h0 = g0 * RT[0] + g1 * RT[1] + g2 * RT[2];
                h1 = g0 * RT[3] + g1 * RT[4] + g2 * RT[5];
                h2 = g0 * RT[6] + g1 * RT[7] + g2 * RT[8];
            }
            if (gout_blend) { h0 += gout_blend[o]; h1 += gout_blend[o + 1]; h2 += gout_blend[o + 2]; }
        }
        float* G = Gs + tid * 16;
        G[0] = px * h0; G[1] = px * h1; G[2] = px * h2; G[3] = py * h0; G[4] = py * h1; G[5] = py * h2;
        G[6] = pz * h0; G[7] = pz * h1; G[8] = pz * h2; G[9] = h0; G[10] = h1; G[11] = h2;
        G[12] = G[13] = G[14] = G[15] = 0.f;
    }
    __syncthreads();
    // ---- the three contractions; wave w owns vertices vw .. vw + 15 of the block
    const int vw = wave * 16;
    if (nb > 0) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int kk = 0; kk < nb; kk += 4) {                  // M = skin^T x RT
            const int k = kk + kc;
            const float a = k < nb ? Ss[k * LBSB_VERTS + vw + col] : 0.f;
            const float b = (k < nb && col < 12) ? RT[(k + 1) * 12 + col] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) Ms[(vw + kc * 4 + r) * 16 + col] = acc[r];
        for (int bb = 0; bb < nbp; bb += 16) {
            const int bone = bb + col;
            if (gskin) {                                      // g_skin = G x RT^T
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s4 = 0; s4 < 12; s4 += 4) {
                    const float a = Gs[(vw + col) * 16 + s4 + kc];
                    const float b = bone < nb ? RT[(bone + 1) * 12 + s4 + kc] : 0.f;
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d, 0, 0, 0);
                }
                if (bone < nb) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int vv = v0 + vw + kc * 4 + r;
                        if (vv < V) gskin[((size_t)n * nb + bone) * V + vv] = d[r];
                    }
                }
            }
            if (partial) {
                f32x4 c4 = {0.f, 0.f, 0.f, 0.f};              // g_RT tile of this wave = skin x G
#pragma unroll
                for (int s4 = 0; s4 < 16; s4 += 4) {
                    const float a = Ss[(bb + col) * LBSB_VERTS + vw + s4 + kc];
                    const float b = Gs[(vw + s4 + kc) * 16 + col];
                    c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; r++) Cs[((size_t)wave * nbp + bb + kc * 4 + r) * 16 + col] = c4[r];
            }
        }
    }
    __syncthreads();
    float* P = partial ? partial + ((size_t)n * nchunks + chunk) * K * 12 : nullptr;
    for (int i = tid; P && i < nb * 12; i += 256) {          // part bones: the four waves' tiles in wave order
        const int k = i / 12, c = i - k * 12;
        const float* t = Cs + (size_t)k * 16 + c;
        P[(k + 1) * 12 + c] = ((t[0] + t[(size_t)nbp * 16]) + t[(size_t)2 * nbp * 16]) + t[(size_t)3 * nbp * 16];
    }
    // ---- per vertex, part 2: g_verts and the body transform's products (into Gs, free now)
    float q[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (mine) {
        float M[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
        if (nb > 0) {
#pragma unroll
            for (int c = 0; c < 12; c++) M[c] = Ms[tid * 16 + c];
        }
        const size_t o = ((size_t)n * V + v) * 3;
        if (gverts) {                        // g_v = g_vs @ M^T
            gverts[o] = h0 * M[0] + h1 * M[1] + h2 * M[2];
            gverts[o + 1] = h0 * M[3] + h1 * M[4] + h2 * M[5];
            gverts[o + 2] = h0 * M[6] + h1 * M[7] + h2 * M[8];
        }
        if (tocam && P) {                    // body transform: g_R0 += vs^T g_out, g_T0 += g_out
            const float s0 = px * M[0] + py * M[3] + pz * M[6] + M[9];
            const float s1 = px * M[1] + py * M[4] + pz * M[7] + M[10];
            const float s2 = px * M[2] + py * M[5] + pz * M[8] + M[11];
            q[0] = s0 * g0; q[1] = s0 * g1; q[2] = s0 * g2;
            q[3] = s1 * g0; q[4] = s1 * g1; q[5] = s1 * g2;
            q[6] = s2 * g0; q[7] = s2 * g1; q[8] = s2 * g2;
            q[9] = g0; q[10] = g1; q[11] = g2;
        }
    }
    if (!P) return;                                          // block-uniform
    __syncthreads();                                         // every wave is done reading Gs
    if (tid < LBSB_VERTS) {
#pragma unroll
        for (int c = 0; c < 12; c++) Gs[tid * 16 + c] = q[c];
    }
    __syncthreads();
    if (tid < 12) {                                          // vertex order
        float t = 0.f;
        for (int u = 0; u < LBSB_VERTS; u++) t += Gs[u * 16 + tid];
        P[tid] = tocam ? t : 0.f;
    }
}

// ===========================================================================
// Pinhole projection
// ===========================================================================
__global__ __launch_bounds__(256) void pinhole_forward_kernel(const float4* __restrict__ verts, const float* __restrict__ pp,
                                                              const float* __restrict__ fl, float4* __restrict__ out,
                                                              int N, int V)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * V) return;
    const int n = i / V;
    const float4 p = verts[i];
    const float f = fl[n];
    // same association as geom_utils.py:32-33: pp + (x * fl) / z
    out[i] = make_float4(pp[2 * n] + p.x * f / p.z, pp[2 * n + 1] + p.y * f / p.z, p.z, p.w);
}

__global__ __launch_bounds__(256) void pinhole_backward_kernel(const float4* __restrict__ verts, const float* __restrict__ fl,
                                                               const float4* __restrict__ gout, float4* __restrict__ gverts,
                                                               float* __restrict__ gpp, float* __restrict__ gfl, int N, int V)
{
    __shared__ float red[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    const float f = fl[n];
    float sx = 0.f, sy = 0.f, sf = 0.f;
    for (int v = tid; v < V; v += 256) {
        const size_t i = (size_t)n * V + v;
        const float4 p = verts[i], g = gout[i];
        const float iz = 1.f / p.z;
        const float xz = p.x * iz, yz = p.y * iz;
        if (gverts) gverts[i] = make_float4(g.x * f * iz, g.y * f * iz, g.z - (g.x * xz + g.y * yz) * f * iz, g.w);
        sx += g.x; sy += g.y; sf += g.x * xz + g.y * yz;
    }
    sx = block_sum(sx, red); sy = block_sum(sy, red); sf = block_sum(sf, red);
    if (tid == 0) {
        if (gpp) { gpp[2 * n] = sx; gpp[2 * n + 1] = sy; }
        if (gfl) gfl[n] = sf;
    }
}

// ===========================================================================
// Image-loss tables.  Each (image i, hypothesis j) row of P pixels is split into NCH chunks so that even
// a 2 x 1 table fills the chip; chunk partials go to scratch and a tiny second kernel folds them in a fixed
// order (deterministic).  Scratch layout (floats): part[IH][NCH][4] | tot[IH][4] | img[I][4] | ipart[I][H*NCH][2].
// Backward kernels are plain elementwise launches over (IH, NCH) that read tot / img.
// ===========================================================================
constexpr int LOSS_PX_PER_BLOCK = 2048;     // 256 threads x 8 pixels

__host__ __device__ inline int loss_nch(int P) { int n = (P + LOSS_PX_PER_BLOCK - 1) / LOSS_PX_PER_BLOCK; return n < 1 ? 1 : (n > 64 ? 64 : n); }

struct LossScratch {
    float* part; float* tot; float* img; float* ipart;
};
__host__ __device__ inline LossScratch loss_scratch(float* base, int I, int H, int P)
{
    const int nch = loss_nch(P);
    LossScratch s;
    s.part = base;
    s.tot = s.part + (size_t)I * H * nch * 4;
    s.img = s.tot + (size_t)I * H * 4;
    s.ipart = s.img + (size_t)I * 4;
    return s;
}

__device__ __forceinline__ void chunk_range(int P, int nch, int ch, int& p0, int& p1)
{
    const int per = (P + nch - 1) / nch;
    p0 = ch * per;
    p1 = min(P, p0 + per);
}

// fold part[ij][0..nch) -> tot[ij]; MODE selects how the loss is formed from the totals
template <int MODE>
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ part, float* __restrict__ tot,
                                                            float* __restrict__ loss, int IH, int nch, float scale)
{
    const int ij = blockIdx.x * 256 + threadIdx.x;
    if (ij >= IH) return;
    float a = 0.f, b = 0.f, c = 0.f;
    for (int k = 0; k < nch; k++) {
        const float* q = part + ((size_t)ij * nch + k) * 4;
        a += q[0]; b += q[1]; c += q[2];
    }
    tot[4 * ij] = a; tot[4 * ij + 1] = b; tot[4 * ij + 2] = c;
    if (MODE == 0) loss[ij] = 0.5f * (a / c);                         // mask: 0.5 * mean (NaN when empty, like torch)
    else if (MODE == 1) loss[ij] = c > 0.f ? 0.5f * (a / c) : 0.f;    // flow: 0 when nothing selected (mesh_net.py:412)
    else loss[ij] = (a / c + b / c) * scale;                          // texture: 2*wt*(mean1 + mean2)
}

__global__ __launch_bounds__(256) void mask_loss_forward_kernel(const float* __restrict__ pred, const float* __restrict__ masks,
                                                                const float* __restrict__ occ, float* __restrict__ part,
                                                                int H, int P, int nch)
{
    __shared__ float red[4];
    const int ij = blockIdx.x, i = ij / H, ch = blockIdx.y;
    const float* a = pred + (size_t)ij * P;
    const float* m = masks + (size_t)i * P;
    const float* oc = occ + (size_t)i * P;
    int p0, p1;
    chunk_range(P, nch, ch, p0, p1);
    float s = 0.f, c = 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += 256)
        if (oc[p] != 0.f) { const float d = a[p] - m[p]; s += d * d; c += 1.f; }
    s = block_sum(s, red); c = block_sum(c, red);
    if (threadIdx.x == 0) { float* q = part + ((size_t)ij * nch + ch) * 4; q[0] = s; q[1] = 0.f; q[2] = c; q[3] = 0.f; }
}

__global__ __launch_bounds__(256) void mask_loss_backward_kernel(const float* __restrict__ pred, const float* __restrict__ masks,
                                                                 const float* __restrict__ occ, const float* __restrict__ gloss,
                                                                 const float* __restrict__ tot, float* __restrict__ gpred,
                                                                 int H, int P, int nch)
{
    const int ij = blockIdx.x, i = ij / H;
    const float* a = pred + (size_t)ij * P;
    const float* m = masks + (size_t)i * P;
    const float* oc = occ + (size_t)i * P;
    const float k = gloss[ij] / tot[4 * ij + 2];            // 0.5 * 2 * g / count
    int p0, p1;
    chunk_range(P, nch, blockIdx.y, p0, p1);
    for (int p = p0 + threadIdx.x; p < p1; p += 256) gpred[(size_t)ij * P + p] = oc[p] != 0.f ? k * (a[p] - m[p]) : 0.f;
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// per image: sum and count of sigmoid(-occ) over sel[i] (all hypotheses): partials per (j, chunk)
__global__ __launch_bounds__(256) void flow_loss_stats_kernel(const unsigned char* __restrict__ bg, const float* __restrict__ occ,
                                                              const float* __restrict__ masks, float* __restrict__ ipart,
                                                              int H, int P, int nch)
{
    __shared__ float red[4];
    const int i = blockIdx.x, j = blockIdx.y / nch, ch = blockIdx.y - j * nch;
    const float* oc = occ + (size_t)i * P;
    const float* m = masks + (size_t)i * P;
    int p0, p1;
    chunk_range(P, nch, ch, p0, p1);
    float s = 0.f, c = 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += 256)
        if (!bg[((size_t)i * H + j) * P + p] && oc[p] != 0.f && m[p] > 0.f) { s += sigmoid_f(-oc[p]); c += 1.f; }
    s = block_sum(s, red); c = block_sum(c, red);
    if (threadIdx.x == 0) { float* q = ipart + ((size_t)i * H * nch + blockIdx.y) * 2; q[0] = s; q[1] = c; }
}

__global__ __launch_bounds__(256) void flow_loss_stats_fold_kernel(const float* __restrict__ ipart, float* __restrict__ img,
                                                                   int I, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= I) return;
    float s = 0.f, c = 0.f;
    for (int k = 0; k < n; k++) { s += ipart[((size_t)i * n + k) * 2]; c += ipart[((size_t)i * n + k) * 2 + 1]; }
    img[4 * i] = s; img[4 * i + 1] = c;
}

__global__ __launch_bounds__(256) void flow_loss_forward_kernel(const float2* __restrict__ flow_rd, const float* __restrict__ obs,
                                                                const unsigned char* __restrict__ bg, const float* __restrict__ occ,
                                                                const float* __restrict__ masks, const float* __restrict__ img,
                                                                float* __restrict__ part, float* __restrict__ fmap,
                                                                int H, int P, int obs_stride, int nch,
                                                                unsigned char* __restrict__ vis)
{
    __shared__ float red[4];
    const int ij = blockIdx.x, i = ij / H, ch = blockIdx.y;
    const float* oc = occ + (size_t)i * P;
    const float* m = masks + (size_t)i * P;
    const float* ox = obs + (size_t)i * obs_stride;
    const float* oy = ox + P;
    const float wmean = img[4 * i] / img[4 * i + 1];
    int p0, p1;
    chunk_range(P, nch, ch, p0, p1);
    float s = 0.f, c = 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
        const float2 f = flow_rd[(size_t)ij * P + p];
        const float dx = f.x - ox[p], dy = f.y - oy[p];
        const float e = sqrtf(dx * dx + dy * dy) * (sigmoid_f(-oc[p]) / wmean);
        fmap[(size_t)ij * P + p] = e;
        const bool sel = !bg[(size_t)ij * P + p] && oc[p] != 0.f && m[p] > 0.f;
        if (vis) vis[(size_t)ij * P + p] = sel ? 1 : 0;             // the vis_mask LASR logs (mesh_net.py:405), for free
        if (sel) { s += e; c += 1.f; }
    }
    s = block_sum(s, red); c = block_sum(c, red);
    if (threadIdx.x == 0) { float* q = part + ((size_t)ij * nch + ch) * 4; q[0] = s; q[1] = 0.f; q[2] = c; q[3] = 0.f; }
}

__global__ __launch_bounds__(256) void flow_loss_backward_kernel(const float2* __restrict__ flow_rd, const float* __restrict__ obs,
                                                                 const unsigned char* __restrict__ bg, const float* __restrict__ occ,
                                                                 const float* __restrict__ masks, const float* __restrict__ img,
                                                                 const float* __restrict__ tot, const float* __restrict__ gloss,
                                                                 float2* __restrict__ gflow, int H, int P, int obs_stride, int nch)
{
    const int ij = blockIdx.x, i = ij / H;
    const float* oc = occ + (size_t)i * P;
    const float* m = masks + (size_t)i * P;
    const float* ox = obs + (size_t)i * obs_stride;
    const float* oy = ox + P;
    const float wmean = img[4 * i] / img[4 * i + 1];
    const float c = tot[4 * ij + 2];
    const float k = c > 0.f ? 0.5f * gloss[ij] / c : 0.f;
    int p0, p1;
    chunk_range(P, nch, blockIdx.y, p0, p1);
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
        // d loss / d map = k on selected pixels, 0 elsewhere; d map / d norm = w.  When image i has no selected
        // pixel at all, w is NaN and 0 * NaN = NaN reaches every pixel -- exactly what autograd does with the
        // reference code (its trainer then drops the step, nnutils/train_utils.py:289-290).  Kept on purpose.
        const bool sel = !bg[(size_t)ij * P + p] && oc[p] != 0.f && m[p] > 0.f;
        const float gn = (sel ? k : 0.f) * (sigmoid_f(-oc[p]) / wmean);
        const float2 f = flow_rd[(size_t)ij * P + p];
        const float dx = f.x - ox[p], dy = f.y - oy[p];
        const float nrm = sqrtf(dx * dx + dy * dy);
        float2 g = make_float2(0.f, 0.f);
        if (nrm > 0.f) g = make_float2(gn / nrm * dx, gn / nrm * dy);   // torch.norm's subgradient at 0 is 0
        gflow[(size_t)ij * P + p] = g;
    }
}

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void tex_loss_forward_kernel(const float* __restrict__ img_obs, const float* __restrict__ img_white,
                                                               const float* __restrict__ rnd, const float* __restrict__ fg,
                                                               const float* __restrict__ occ, float* __restrict__ part,
                                                               int H, int P, int nch)
{
    __shared__ float red[4];
    const int ij = blockIdx.x, i = ij / H, ch = blockIdx.y;
    const float* oc = occ + (size_t)i * P;
    int p0, p1;
    chunk_range(P, nch, ch, p0, p1);
    float s1 = 0.f, s2 = 0.f, c = 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
        if (oc[p] == 0.f) continue;
        const float a = fg[(size_t)ij * P + p];
        float e1 = 0.f, e2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float r = rnd[((size_t)ij * 3 + k) * P + p];
            e1 += fabsf(img_obs[((size_t)i * 3 + k) * P + p] - r * a);
            e2 += fabsf(img_white[((size_t)i * 3 + k) * P + p] - r);
        }
        s1 += e1 / 3.f; s2 += e2 / 3.f; c += 1.f;
    }
    s1 = block_sum(s1, red); s2 = block_sum(s2, red); c = block_sum(c, red);
    if (threadIdx.x == 0) { float* q = part + ((size_t)ij * nch + ch) * 4; q[0] = s1; q[1] = s2; q[2] = c; q[3] = 0.f; }
}

__global__ __launch_bounds__(256) void tex_loss_backward_kernel(const float* __restrict__ img_obs, const float* __restrict__ img_white,
                                                                const float* __restrict__ rnd, const float* __restrict__ fg,
                                                                const float* __restrict__ occ, const float* __restrict__ tot,
                                                                const float* __restrict__ gloss, float* __restrict__ grnd,
                                                                float* __restrict__ gfg, float wt, int H, int P, int nch)
{
    const int ij = blockIdx.x, i = ij / H;
    const float* oc = occ + (size_t)i * P;
    const float k = gloss[ij] * (2.f * wt) / (3.f * tot[4 * ij + 2]);
    int p0, p1;
    chunk_range(P, nch, blockIdx.y, p0, p1);
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
        const bool on = oc[p] != 0.f;
        const float a = fg[(size_t)ij * P + p];
        float ga = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const size_t q = ((size_t)ij * 3 + ch) * P + p;
            float g = 0.f;
            if (on) {
                const float r = rnd[q];
                const float s1 = sgn(img_obs[((size_t)i * 3 + ch) * P + p] - r * a);
                const float s2 = sgn(img_white[((size_t)i * 3 + ch) * P + p] - r);
                g = -k * (s1 * a + s2);
                ga += -k * s1 * r;
            }
            grnd[q] = g;
        }
        gfg[(size_t)ij * P + p] = ga;
    }
}

// ===========================================================================
// Mesh regularisers on CSR adjacency: one block per mesh instance
// ===========================================================================
__global__ __launch_bounds__(256) void arap_forward_kernel(const float* __restrict__ dx, const float* __restrict__ x,
                                                           const int* __restrict__ row_ptr, const int* __restrict__ col,
                                                           float* __restrict__ loss, int V)
{
    __shared__ float red[4];
    const int n = blockIdx.x;
    const float s = arap_forward_block(x + (size_t)n * V * 3, dx + (size_t)n * V * 3, row_ptr, col, V, red);
    if (threadIdx.x == 0) loss[n] = s / (float)row_ptr[V];
}

__global__ __launch_bounds__(256) void arap_backward_kernel(const float* __restrict__ dx, const float* __restrict__ x,
                                                            const int* __restrict__ row_ptr, const int* __restrict__ col,
                                                            const float* __restrict__ gloss, float* __restrict__ gdx,
                                                            float* __restrict__ gx, int V)
{
    const int n = blockIdx.x;
    const float k = 4.f * gloss[n] / (float)row_ptr[V];
    for (int v = threadIdx.x; v < V; v += 256) {
        const size_t o = ((size_t)n * V + v) * 3;
        arap_backward_vertex(x + (size_t)n * V * 3, dx + (size_t)n * V * 3, row_ptr, col, k, v, gx ? gx + o : nullptr,
                             gdx ? gdx + o : nullptr);
    }
}

// lx[v] = x_v - mean_{u in nbr(v)} x_u  (0 for isolated vertices); loss = sum |lx|^2
__global__ __launch_bounds__(256) void laplacian_forward_kernel(const float* __restrict__ x, const int* __restrict__ row_ptr,
                                                                const int* __restrict__ col, float* __restrict__ loss,
                                                                float* __restrict__ lx_out, int V)
{
    __shared__ float red[4];
    const int n = blockIdx.x;
    const float s = laplacian_forward_block(x + (size_t)n * V * 3, row_ptr, col, lx_out ? lx_out + (size_t)n * V * 3 : nullptr, V, red);
    if (threadIdx.x == 0 && loss) loss[n] = s;
}

// g_x = 2 g L^T (L x):  g_x[v] = 2 g (lx[v] - sum_{u in nbr(v)} lx[u] / deg(u))   (symmetric adjacency)
__global__ __launch_bounds__(256) void laplacian_backward_kernel(const float* __restrict__ lx, const int* __restrict__ row_ptr,
                                                                 const int* __restrict__ col, const float* __restrict__ gloss,
                                                                 float* __restrict__ gx, int V)
{
    const int n = blockIdx.x;
    const float k = 2.f * gloss[n];
    for (int v = threadIdx.x; v < V; v += 256) {
        float a[3];
        laplacian_backward_vertex(lx + (size_t)n * V * 3, row_ptr, col, k, v, a);
        const size_t o = ((size_t)n * V + v) * 3;
        gx[o] = a[0]; gx[o + 1] = a[1]; gx[o + 2] = a[2];
    }
}

}  // namespace lasr

// ===========================================================================
// C ABI
// ===========================================================================
using namespace lasr;


extern "C" int lasr_lbs_forward(const float* verts, const float* Rmat, const float* Tmat, const float* skin, float* out,
                                int N, int V, int K, int tocam, void* hip_stream)
{
    if (N < 0 || V < 0 || K < 1) return LASR_E_BADARG;
    if (N == 0 || V == 0) return LASR_OK;
    if (!verts || !Rmat || !Tmat || !out || (K > 1 && !skin)) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_LBS_FORWARD, lbs_forward_kernel, dim3((V + 63) / 64, N), dim3(256), 0, verts, Rmat, Tmat, skin, out,
                N, V, K, tocam, (float*)nullptr);
    return launch_ok();
}

extern "C" int lasr_lbs_forward_both(const float* verts, const float* Rmat, const float* Tmat, const float* skin, float* out_cam,
                                     float* out_blend, int N, int V, int K, void* hip_stream)
{
    if (N < 0 || V < 0 || K < 1) return LASR_E_BADARG;
    if (N == 0 || V == 0) return LASR_OK;
    if (!verts || !Rmat || !Tmat || !out_cam || !out_blend || (K > 1 && !skin)) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_LBS_FORWARD, lbs_forward_kernel, dim3((V + 63) / 64, N), dim3(256), 0, verts, Rmat, Tmat, skin, out_cam,
                N, V, K, 1, out_blend);
    return launch_ok();
}

extern "C" size_t lasr_lbs_backward_scratch_floats(int N, int V, int K)
{
    if (N < 0 || V < 0 || K < 1) return 0;
    const size_t chunks = (size_t)(V + LBSB_VERTS - 1) / LBSB_VERTS > 0 ? (size_t)(V + LBSB_VERTS - 1) / LBSB_VERTS : 1;   // >= the 256-vertex chunks
    return (size_t)N * chunks * K * 12;
}

static int lbs_backward_impl(const float* verts, const float* Rmat, const float* Tmat, const float* skin,
                             const float* grad_out, const float* grad_out_blend, float* grad_verts, float* grad_Rmat,
                             float* grad_Tmat, float* grad_skin, float* scratch, int N, int V, int K, int tocam, void* hip_stream);

extern "C" int lasr_lbs_backward(const float* verts, const float* Rmat, const float* Tmat, const float* skin,
                                 const float* grad_out, float* grad_verts, float* grad_Rmat, float* grad_Tmat,
                                 float* grad_skin, float* scratch, int N, int V, int K, int tocam, void* hip_stream)
{
    return lbs_backward_impl(verts, Rmat, Tmat, skin, grad_out, nullptr, grad_verts, grad_Rmat, grad_Tmat, grad_skin, scratch,
                             N, V, K, tocam, hip_stream);
}

extern "C" int lasr_lbs_backward_both(const float* verts, const float* Rmat, const float* Tmat, const float* skin,
                                      const float* grad_out_cam, const float* grad_out_blend, float* grad_verts,
                                      float* grad_Rmat, float* grad_Tmat, float* grad_skin, float* scratch, int N, int V, int K,
                                      void* hip_stream)
{
    if (!grad_out_blend) return LASR_E_BADARG;
    return lbs_backward_impl(verts, Rmat, Tmat, skin, grad_out_cam, grad_out_blend, grad_verts, grad_Rmat, grad_Tmat, grad_skin,
                             scratch, N, V, K, 1, hip_stream);
}

static int lbs_backward_impl(const float* verts, const float* Rmat, const float* Tmat, const float* skin,
                             const float* grad_out, const float* grad_out_blend, float* grad_verts, float* grad_Rmat,
                             float* grad_Tmat, float* grad_skin, float* scratch, int N, int V, int K, int tocam, void* hip_stream)
{
    if (N < 0 || V < 0 || K < 1 || K > 1024) return LASR_E_BADARG;
    if (N == 0) return LASR_OK;
    if (!verts || !Rmat || !Tmat || !grad_out || (K > 1 && !skin)) return LASR_E_BADARG;
    const bool want_rt = grad_Rmat || grad_Tmat;
    if (want_rt && !scratch) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    int nchunks;
    if (K <= LBSB_MAX_K) {
        // the three contractions on the matrix cores; the transposed one (and the fold launch) only when a transform gradient is wanted
        nchunks = (V + LBSB_VERTS - 1) / LBSB_VERTS > 0 ? (V + LBSB_VERTS - 1) / LBSB_VERTS : 1;
        const int nbp = (K - 1 + 15) & ~15;
        const size_t lds = (size_t)(((K * 12 + 3) & ~3) + 2 * LBSB_VERTS * 16 + nbp * LBSB_VERTS + 4 * nbp * 16) * sizeof(float);
        LASR_LAUNCH(K_LBS_BACKWARD, lbs_backward_mfma_kernel, dim3(nchunks, N), dim3(256), lds, verts, Rmat, Tmat, skin, grad_out,
                    grad_verts, grad_skin, want_rt ? scratch : nullptr, N, V, K, tocam, grad_out_blend);
    } else {
        // more than 64 part bones: the VALU kernel of rounds 1-4 (256-vertex chunks, per-thread bone loop)
        if (!scratch) return LASR_E_BADARG;
        nchunks = (V + LBS_CHUNK - 1) / LBS_CHUNK > 0 ? (V + LBS_CHUNK - 1) / LBS_CHUNK : 1;
        const size_t lds = (size_t)(K * 12 + LBS_CHUNK * 13 + 4) * sizeof(float);
        LASR_LAUNCH(K_LBS_BACKWARD, lbs_backward_kernel, dim3(nchunks, N), dim3(256), lds, verts, Rmat, Tmat, skin, grad_out,
                    grad_verts, grad_skin, scratch, N, V, K, tocam, grad_out_blend);
    }
    int rc = launch_ok();
    if (rc || !want_rt) return rc;
    LASR_LAUNCH(K_LBS_BACKWARD_FOLD, lbs_backward_fold_kernel, dim3(N), dim3(256), 0, scratch, grad_Rmat, grad_Tmat, K, nchunks);
    return launch_ok();
}

extern "C" int lasr_pinhole_forward(const float* verts, const float* pp, const float* fl, float* out, int N, int V,
                                    void* hip_stream)
{
    if (N < 0 || V < 0) return LASR_E_BADARG;
    if (N == 0 || V == 0) return LASR_OK;
    if (!verts || !pp || !fl || !out) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_PINHOLE_FORWARD, pinhole_forward_kernel, dim3((N * V + 255) / 256), dim3(256), 0, (const float4*)verts,
                pp, fl, (float4*)out, N, V);
    return launch_ok();
}

extern "C" int lasr_pinhole_backward(const float* verts, const float* pp, const float* fl, const float* grad_out,
                                     float* grad_verts, float* grad_pp, float* grad_fl, int N, int V, void* hip_stream)
{
    (void)pp;
    if (N < 0 || V < 0) return LASR_E_BADARG;
    if (N == 0) return LASR_OK;
    if (!verts || !fl || !grad_out) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_PINHOLE_BACKWARD, pinhole_backward_kernel, dim3(N), dim3(256), 0, (const float4*)verts, fl,
                (const float4*)grad_out, (float4*)grad_verts, grad_pp, grad_fl, N, V);
    return launch_ok();
}

static int check_ihp(int I, int H, int P) { return (I < 0 || H < 0 || P < 0) ? LASR_E_BADARG : LASR_OK; }

extern "C" size_t lasr_loss_scratch_floats(int I, int H, int P)
{
    if (I < 0 || H < 0 || P < 0) return 0;
    const size_t nch = (size_t)loss_nch(P);
    return (size_t)I * H * nch * 4 + (size_t)I * H * 4 + (size_t)I * 4 + (size_t)I * H * nch * 2 + 16;
}

extern "C" int lasr_mask_loss_forward(const float* mask_pred, const float* masks, const float* occ, float* loss,
                                      float* scratch, int I, int H, int P, void* hip_stream)
{
    if (check_ihp(I, H, P)) return LASR_E_BADARG;
    if (I * H == 0) return LASR_OK;
    if (!mask_pred || !masks || !occ || !loss || !scratch) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int nch = loss_nch(P);
    const LossScratch sc = loss_scratch(scratch, I, H, P);
    LASR_LAUNCH(K_MASK_LOSS_FORWARD, mask_loss_forward_kernel, dim3(I * H, nch), dim3(256), 0, mask_pred, masks, occ,
                sc.part, H, P, nch);
    int rc = launch_ok();
    if (rc) return rc;
    LASR_LAUNCH(K_LOSS_FINALIZE, loss_finalize_kernel<0>, dim3((I * H + 255) / 256), dim3(256), 0, sc.part, sc.tot, loss,
                I * H, nch, 1.f);
    return launch_ok();
}

extern "C" int lasr_mask_loss_backward(const float* mask_pred, const float* masks, const float* occ,
                                       const float* grad_loss, const float* scratch, float* grad_pred, int I, int H, int P,
                                       void* hip_stream)
{
    if (check_ihp(I, H, P)) return LASR_E_BADARG;
    if (I * H == 0) return LASR_OK;
    if (!mask_pred || !masks || !occ || !grad_loss || !grad_pred || !scratch) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int nch = loss_nch(P);
    const LossScratch sc = loss_scratch(const_cast<float*>(scratch), I, H, P);
    LASR_LAUNCH(K_MASK_LOSS_BACKWARD, mask_loss_backward_kernel, dim3(I * H, nch), dim3(256), 0, mask_pred, masks, occ,
                grad_loss, sc.tot, grad_pred, H, P, nch);
    return launch_ok();
}

extern "C" int lasr_flow_loss_forward(const float* flow_rd, const float* flow_obs, const unsigned char* bg,
                                      const float* occ, const float* masks, float* loss, float* flow_rd_map,
                                      float* scratch, int I, int H, int P, int obs_image_stride, void* hip_stream)
{
    return lasr_flow_loss_forward_vis(flow_rd, flow_obs, bg, occ, masks, loss, flow_rd_map, nullptr, scratch, I, H, P,
                                      obs_image_stride, hip_stream);
}

extern "C" int lasr_flow_loss_forward_vis(const float* flow_rd, const float* flow_obs, const unsigned char* bg,
                                          const float* occ, const float* masks, float* loss, float* flow_rd_map,
                                          unsigned char* vis_mask, float* scratch, int I, int H, int P, int obs_image_stride,
                                          void* hip_stream)
{
    if (check_ihp(I, H, P)) return LASR_E_BADARG;
    if (I * H == 0) return LASR_OK;
    if (!flow_rd || !flow_obs || !bg || !occ || !masks || !loss || !flow_rd_map || !scratch) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int nch = loss_nch(P);
    const LossScratch sc = loss_scratch(scratch, I, H, P);
    LASR_LAUNCH(K_FLOW_LOSS_STATS, flow_loss_stats_kernel, dim3(I, H * nch), dim3(256), 0, bg, occ, masks, sc.ipart, H, P, nch);
    int rc = launch_ok();
    if (rc) return rc;
    LASR_LAUNCH(K_LOSS_FINALIZE, flow_loss_stats_fold_kernel, dim3((I + 255) / 256), dim3(256), 0, sc.ipart, sc.img, I, H * nch);
    if ((rc = launch_ok())) return rc;
    LASR_LAUNCH(K_FLOW_LOSS_FORWARD, flow_loss_forward_kernel, dim3(I * H, nch), dim3(256), 0, (const float2*)flow_rd,
                flow_obs, bg, occ, masks, sc.img, sc.part, flow_rd_map, H, P, obs_image_stride, nch, vis_mask);
    if ((rc = launch_ok())) return rc;
    LASR_LAUNCH(K_LOSS_FINALIZE, loss_finalize_kernel<1>, dim3((I * H + 255) / 256), dim3(256), 0, sc.part, sc.tot, loss,
                I * H, nch, 1.f);
    return launch_ok();
}

extern "C" int lasr_flow_loss_backward(const float* flow_rd, const float* flow_obs, const unsigned char* bg,
                                       const float* occ, const float* masks, const float* scratch,
                                       const float* grad_loss, float* grad_flow_rd, int I, int H, int P,
                                       int obs_image_stride, void* hip_stream)
{
    if (check_ihp(I, H, P)) return LASR_E_BADARG;
    if (I * H == 0) return LASR_OK;
    if (!flow_rd || !flow_obs || !bg || !occ || !masks || !scratch || !grad_loss || !grad_flow_rd) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int nch = loss_nch(P);
    const LossScratch sc = loss_scratch(const_cast<float*>(scratch), I, H, P);
    LASR_LAUNCH(K_FLOW_LOSS_BACKWARD, flow_loss_backward_kernel, dim3(I * H, nch), dim3(256), 0, (const float2*)flow_rd,
                flow_obs, bg, occ, masks, sc.img, sc.tot, grad_loss, (float2*)grad_flow_rd, H, P, obs_image_stride, nch);
    return launch_ok();
}

extern "C" int lasr_tex_loss_forward(const float* img_obs, const float* img_white, const float* rnd, const float* fg,
                                     const float* occ, float* loss, float* scratch, float wt, int I, int H, int P,
                                     void* hip_stream)
{
    if (check_ihp(I, H, P)) return LASR_E_BADARG;
    if (I * H == 0) return LASR_OK;
    if (!img_obs || !img_white || !rnd || !fg || !occ || !loss || !scratch) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int nch = loss_nch(P);
    const LossScratch sc = loss_scratch(scratch, I, H, P);
    LASR_LAUNCH(K_TEX_LOSS_FORWARD, tex_loss_forward_kernel, dim3(I * H, nch), dim3(256), 0, img_obs, img_white, rnd, fg,
                occ, sc.part, H, P, nch);
    int rc = launch_ok();
    if (rc) return rc;
    LASR_LAUNCH(K_LOSS_FINALIZE, loss_finalize_kernel<2>, dim3((I * H + 255) / 256), dim3(256), 0, sc.part, sc.tot, loss,
                I * H, nch, 2.f * wt);
    return launch_ok();
}

extern "C" int lasr_tex_loss_backward(const float* img_obs, const float* img_white, const float* rnd, const float* fg,
                                      const float* occ, const float* grad_loss, const float* scratch, float* grad_rnd,
                                      float* grad_fg, float wt, int I, int H, int P, void* hip_stream)
{
    if (check_ihp(I, H, P)) return LASR_E_BADARG;
    if (I * H == 0) return LASR_OK;
    if (!img_obs || !img_white || !rnd || !fg || !occ || !grad_loss || !grad_rnd || !grad_fg || !scratch) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const int nch = loss_nch(P);
    const LossScratch sc = loss_scratch(const_cast<float*>(scratch), I, H, P);
    LASR_LAUNCH(K_TEX_LOSS_BACKWARD, tex_loss_backward_kernel, dim3(I * H, nch), dim3(256), 0, img_obs, img_white, rnd, fg,
                occ, sc.tot, grad_loss, grad_rnd, grad_fg, wt, H, P, nch);
    return launch_ok();
}

extern "C" int lasr_arap_forward(const float* dx, const float* x, const int* row_ptr, const int* col, float* loss,
                                 int N, int V, void* hip_stream)
{
    if (N < 0 || V < 0) return LASR_E_BADARG;
    if (N == 0) return LASR_OK;
    if (!dx || !x || !row_ptr || !col || !loss) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_ARAP_FORWARD, arap_forward_kernel, dim3(N), dim3(256), 0, dx, x, row_ptr, col, loss, V);
    return launch_ok();
}

extern "C" int lasr_arap_backward(const float* dx, const float* x, const int* row_ptr, const int* col,
                                  const float* grad_loss, float* grad_dx, float* grad_x, int N, int V, void* hip_stream)
{
    if (N < 0 || V < 0) return LASR_E_BADARG;
    if (N == 0) return LASR_OK;
    if (!dx || !x || !row_ptr || !col || !grad_loss) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_ARAP_BACKWARD, arap_backward_kernel, dim3(N), dim3(256), 0, dx, x, row_ptr, col, grad_loss, grad_dx,
                grad_x, V);
    return launch_ok();
}

extern "C" int lasr_laplacian_forward(const float* x, const int* row_ptr, const int* col, float* loss, int N, int V,
                                      void* hip_stream)
{
    if (N < 0 || V < 0) return LASR_E_BADARG;
    if (N == 0) return LASR_OK;
    if (!x || !row_ptr || !col || !loss) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_LAP_FORWARD, laplacian_forward_kernel, dim3(N), dim3(256), 0, x, row_ptr, col, loss, (float*)nullptr, V);
    return launch_ok();
}

extern "C" int lasr_laplacian_backward(const float* x, const int* row_ptr, const int* col, const float* grad_loss,
                                       float* grad_x, float* scratch_lx, int N, int V, void* hip_stream)
{
    if (N < 0 || V < 0) return LASR_E_BADARG;
    if (N == 0) return LASR_OK;
    if (!x || !row_ptr || !col || !grad_loss || !grad_x || !scratch_lx) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    LASR_LAUNCH(K_LAP_FORWARD, laplacian_forward_kernel, dim3(N), dim3(256), 0, x, row_ptr, col, (float*)nullptr,
                scratch_lx, V);
    int rc = launch_ok();
    if (rc) return rc;
    LASR_LAUNCH(K_LAP_BACKWARD, laplacian_backward_kernel, dim3(N), dim3(256), 0, scratch_lx, row_ptr, col, grad_loss,
                grad_x, V);
    return launch_ok();
}
