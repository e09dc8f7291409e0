// post_raster.hip -- everything LASR.forward does with the render between the rasteriser and the loss sum, as ONE pass over
// the [N,10,IS,IS] output of the nine-attribute render (planes 0-2 texture colours, 3-5 own camera-space position, 6-8 the
// other frame's position, 9 alpha) and ONE pass back:
//   flow reprojection + background mask            nnutils/mesh_net.py:87-104 (render_flow_soft_2's tail)
//   silhouette loss table                          :374-390
//   flow loss table (+ weighted error map, vis)    :393-416
//   texture L1 loss table                          :419-441
//   the perceptual network's input pair            :436-441 (render * alpha | render), written here instead of mul + cat
// The separate operators (ops.hip / fused.hip: lasr_mask_loss_*, lasr_flow_loss_*, lasr_tex_loss_*, lasr_flow_reproject_*)
// stay for callers that render the pieces separately; this file reads the planes IN PLACE (no contiguous copies of channel
// slices) and its backward writes every plane of grad_px itself (no split / cat / accumulate kernels of autograd).
// Arithmetic, chunking and fold order are those of the separate kernels, so the tables are bit-identical to theirs.
//
// Scratch (floats): part[N][nch][8] | tot[N][8] | img[I][4] | part2[N][nch][4] | part3[N][nch][4]
//
// Launches: forward = 3 (the pass over the render; the flow pass, whose blocks fold the first pass' chunk partials themselves in
// the fixed chunk order -- rounds 1-4 ran that fold as a one-wave launch of 20 us; a 256-thread fold of the flow partials),
// backward = 2 (the pass back + the fold of the intrinsics' partials).  Folding inside the producing launch by its LAST block
// (device-scope ticket + __threadfence) was built and measured in round 5: with 512 blocks the L2 write-back of every block's
// fence costs 20-26 us per launch on this part (8 XCD-private L2s) -- a 5 us fold launch is cheaper (profiles/experiments/).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lasr_ops.h"
#include "ops_common.h"

namespace lasr {

constexpr int PR_PX_PER_BLOCK = 2048;
__host__ __device__ inline int pr_nch(int P) { int n = (P + PR_PX_PER_BLOCK - 1) / PR_PX_PER_BLOCK; return n < 1 ? 1 : (n > 64 ? 64 : n); }

struct PrScratch { float *part, *tot, *img, *part2, *part3; };
__host__ __device__ inline PrScratch pr_scratch(float* base, int I, int H, int P)
{
    const size_t N = (size_t)I * H, nch = (size_t)pr_nch(P);
    PrScratch s;
    s.part = base;
    s.tot = s.part + N * nch * 8;
    s.img = s.tot + N * 8;
    s.part2 = s.img + (size_t)I * 4;
    s.part3 = s.part2 + N * nch * 4;
    return s;
}

__device__ __forceinline__ float pr_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float pr_sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ void pr_range(int P, int nch, int ch, int& p0, int& p1)
{
    const int per = (P + nch - 1) / nch;
    p0 = ch * per;
    p1 = min(P, p0 + per);
}

struct PrArgs {
    const float* px;        // [N,10,P]
    const float* masks;     // [I,P]
    const float* occ;       // [I,P]
    const float* obs;       // [I,C>=2,P] observed flow, image stride obs_stride
    const float* img_obs;   // [I,3,P]
    const float* img_white; // [I,3,P]
    const float* pp;        // [N,2] principal points; image n reprojects the other frame's position with pp[(n + half) % N]
    const float* fl;        // [N]   focal lengths, same indexing
    int I, H, P, nch, half;
    long long obs_stride;
};

// ---- forward pass 1: everything that needs one look at the render ------------------------------------------------------------
// FROM_IMGS: A.img_obs holds the observed images themselves; the object on black / on white (nnutils/mesh_net.py:364-366,
// obs_pair_kernel's expressions) is formed here and written to obs_out [2I,3,P] by the blocks of hypothesis 0 -- the launch of
// lasr_obs_pair and three of the pass' input planes are saved.
template <bool FROM_IMGS>
__global__ __launch_bounds__(256) void render_tables_forward_kernel(PrArgs A, float2* __restrict__ flow, unsigned char* __restrict__ bg,
                                                                    float* __restrict__ rndpair, float* __restrict__ part,
                                                                    float* __restrict__ obs_out)
{
    __shared__ float red[4];
    const int ij = blockIdx.x, i = ij / A.H, ch = blockIdx.y, P = A.P, N = A.I * A.H;
    const float* q = A.px + (size_t)ij * 10 * P;
    const float* m = A.masks + (size_t)i * P;
    const float* oc = A.occ + (size_t)i * P;
    const float* io = A.img_obs + (size_t)i * 3 * P;
    const float* iw = A.img_white + (size_t)i * 3 * P;
    const int other = (ij + A.half) % N;
    const float c0x = A.pp[2 * ij], c0y = A.pp[2 * ij + 1], c1x = A.pp[2 * other], c1y = A.pp[2 * other + 1];
    const float f0 = A.fl[ij], f1 = A.fl[other];
    int p0, p1;
    pr_range(P, A.nch, ch, p0, p1);
    float s_mask = 0.f, c_occ = 0.f, s1 = 0.f, s2 = 0.f, s_sig = 0.f, c_sel = 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
        const float a = q[9 * (size_t)P + p];
        const float r0 = q[p], r1 = q[P + p], r2 = q[2 * (size_t)P + p];
        float x0 = q[3 * (size_t)P + p], y0 = q[4 * (size_t)P + p], z0 = q[5 * (size_t)P + p];
        float x1 = q[6 * (size_t)P + p], y1 = q[7 * (size_t)P + p], z1 = q[8 * (size_t)P + p];
        // reprojection (mesh_net.py:93-104 == flow_reproject_forward_kernel)
        const bool b = (z0 < 1e-9f) | (z1 < 1e-9f);
        if (b) x0 = y0 = z0 = x1 = y1 = z1 = 10.f;
        const float u0 = c0x + (x0 * f0) / z0, v0 = c0y + (y0 * f0) / z0;
        const float u1 = c1x + (x1 * f1) / z1, v1 = c1y + (y1 * f1) / z1;
        flow[(size_t)ij * P + p] = make_float2(u1 - u0, v1 - v0);
        bg[(size_t)ij * P + p] = b ? 1 : 0;
        const float o = oc[p];
        float b0, b1, b2, w0, w1, w2;                                   // observed object on black | on white
        if (FROM_IMGS) {
            const float fg = m[p] > 0.f ? 1.f : 0.f;
            b0 = io[p] * fg; b1 = io[P + p] * fg; b2 = io[2 * (size_t)P + p] * fg;
            w0 = 1.f - fg + b0; w1 = 1.f - fg + b1; w2 = 1.f - fg + b2;
            if (ij == i * A.H) {
                float* ob = obs_out + (size_t)i * 3 * P + p;
                float* ow = obs_out + ((size_t)A.I + i) * 3 * P + p;
                ob[0] = b0; ob[P] = b1; ob[2 * (size_t)P] = b2;
                ow[0] = w0; ow[P] = w1; ow[2 * (size_t)P] = w2;
            }
        } else if (o != 0.f) {
            b0 = io[p]; b1 = io[P + p]; b2 = io[2 * (size_t)P + p];
            w0 = iw[p]; w1 = iw[P + p]; w2 = iw[2 * (size_t)P + p];
        }
        if (o != 0.f) {
            const float d = a - m[p];                                   // silhouette (== mask_loss_forward_kernel)
            s_mask += d * d; c_occ += 1.f;
            const float e1 = fabsf(b0 - r0 * a) + fabsf(b1 - r1 * a) + fabsf(b2 - r2 * a);
            const float e2 = fabsf(w0 - r0) + fabsf(w1 - r1) + fabsf(w2 - r2);
            s1 += e1 / 3.f; s2 += e2 / 3.f;                             // texture L1 (== tex_loss_forward_kernel)
            if (!b && m[p] > 0.f) { s_sig += pr_sigmoid(-o); c_sel += 1.f; }   // (== flow_loss_stats_kernel)
        }
        if (rndpair) {
            float* o1 = rndpair + (size_t)ij * 3 * P + p;
            float* o2 = rndpair + ((size_t)N + ij) * 3 * P + p;
            o1[0] = r0 * a; o1[P] = r1 * a; o1[2 * (size_t)P] = r2 * a;
            o2[0] = r0; o2[P] = r1; o2[2 * (size_t)P] = r2;
        }
    }
    s_mask = block_sum(s_mask, red); c_occ = block_sum(c_occ, red); s1 = block_sum(s1, red); s2 = block_sum(s2, red);
    s_sig = block_sum(s_sig, red); c_sel = block_sum(c_sel, red);
    if (threadIdx.x == 0) {
        float* o = part + ((size_t)ij * A.nch + ch) * 8;
        o[0] = s_mask; o[1] = c_occ; o[2] = s1; o[3] = s2; o[4] = s_sig; o[5] = c_sel; o[6] = 0.f; o[7] = 0.f;
    }
}

// ---- forward pass 2: the flow loss needs the image's mean weight first (== flow_loss_forward_kernel) ----------------------------
// Prologue (the former render_tables_fold_kernel, 20 us as its own one-wave launch): every block folds the (sum, count) of
// sigmoid(-occ) over its image's (hypothesis, chunk) partials IN THAT ORDER -- staged through LDS by all threads, added up by
// one -- so the mean weight it uses is bit-identical to the value the fold kernel used to leave in img[]; the chunk-0 block of
// each (image, hypothesis) also folds that row's silhouette / texture partials (chunk order) into tot[] and the two tables,
// and the chunk-0 block of hypothesis 0 records img[].
constexpr int PR_FOLD_TILE = 1024;
__global__ __launch_bounds__(256) void render_tables_flow_kernel(PrArgs A, const float2* __restrict__ flow, const unsigned char* __restrict__ bg,
                                                                 const float* __restrict__ part, float* __restrict__ tot,
                                                                 float* __restrict__ img, float* __restrict__ mask_tab,
                                                                 float* __restrict__ tex_tab, float tex_scale,
                                                                 float* __restrict__ part2, float* __restrict__ fmap,
                                                                 unsigned char* __restrict__ vis)
{
    __shared__ float red[4];
    __shared__ float2 stage[PR_FOLD_TILE];
    __shared__ float s_img[2];
    const int ij = blockIdx.x, i = ij / A.H, ch = blockIdx.y, P = A.P, H = A.H, nch = A.nch, tid = threadIdx.x;
    {
        const float* pi = part + (size_t)i * H * nch * 8;            // the image's H * nch partial rows, (j, chunk) order
        const int rows = H * nch;
        float s = 0.f, c = 0.f;
        for (int r0 = 0; r0 < rows; r0 += PR_FOLD_TILE) {
            const int m = min(PR_FOLD_TILE, rows - r0);
            __syncthreads();
            for (int r = tid; r < m; r += 256) stage[r] = make_float2(pi[(size_t)(r0 + r) * 8 + 4], pi[(size_t)(r0 + r) * 8 + 5]);
            __syncthreads();
            if (tid == 0)
                for (int r = 0; r < m; r++) { s += stage[r].x; c += stage[r].y; }
        }
        if (tid == 0) {
            s_img[0] = s; s_img[1] = c;
            if (ch == 0 && ij == i * H) { img[4 * i] = s; img[4 * i + 1] = c; }
        }
        if (ch == 0 && tid == 64) {                                   // another wave: this row's silhouette / texture totals
            float a = 0.f, cc = 0.f, s1 = 0.f, s2 = 0.f;
            for (int k = 0; k < nch; k++) {
                const float* o = part + ((size_t)ij * nch + k) * 8;
                a += o[0]; cc += o[1]; s1 += o[2]; s2 += o[3];
            }
            float* t = tot + (size_t)ij * 8;
            t[0] = a; t[1] = cc; t[2] = s1; t[3] = s2;
            mask_tab[ij] = 0.5f * (a / cc);                     // NaN when empty, like torch (mesh_net.py:388)
            tex_tab[ij] = (s1 / cc + s2 / cc) * tex_scale;      // 2 * wt * (mean1 + mean2)
        }
        __syncthreads();
    }
    const float* oc = A.occ + (size_t)i * P;
    const float* m = A.masks + (size_t)i * P;
    const float* ox = A.obs + (size_t)i * A.obs_stride;
    const float* oy = ox + P;
    const float wmean = s_img[0] / s_img[1];
    int p0, p1;
    pr_range(P, nch, ch, p0, p1);
    float s = 0.f, c = 0.f;
    for (int p = p0 + tid; p < p1; p += 256) {
        const float2 f = flow[(size_t)ij * P + p];
        const float dx = f.x - ox[p], dy = f.y - oy[p];
        const float e = sqrtf(dx * dx + dy * dy) * (pr_sigmoid(-oc[p]) / wmean);
        fmap[(size_t)ij * P + p] = e;
        const bool sel = !bg[(size_t)ij * P + p] && oc[p] != 0.f && m[p] > 0.f;
        vis[(size_t)ij * P + p] = sel ? 1 : 0;
        if (sel) { s += e; c += 1.f; }
    }
    s = block_sum(s, red); c = block_sum(c, red);
    if (tid == 0) { float* o = part2 + ((size_t)ij * nch + ch) * 4; o[0] = s; o[1] = c; o[2] = 0.f; o[3] = 0.f; }
}

__global__ __launch_bounds__(256) void render_tables_flow_fold_kernel(const float* __restrict__ part2, float* __restrict__ tot,
                                                                      float* __restrict__ flow_tab, int N, int nch)
{
    const int ij = blockIdx.x * 256 + threadIdx.x;
    if (ij >= N) return;
    float s = 0.f, c = 0.f;
    for (int k = 0; k < nch; k++) { s += part2[((size_t)ij * nch + k) * 4]; c += part2[((size_t)ij * nch + k) * 4 + 1]; }
    tot[(size_t)ij * 8 + 4] = s; tot[(size_t)ij * 8 + 5] = c;
    flow_tab[ij] = c > 0.f ? 0.5f * (s / c) : 0.f;          // 0 when nothing is selected (mesh_net.py:412)
}

// ---- backward: every plane of grad_px in one pass --------------------------------------------------------------------------------
// g_rndpair may be null (perceptual term off).  Planes 3-5 (the rendering frame's own position) get zeros: its projection is
// detached (mesh_net.py:101-102).  part3[n][chunk] = (d pp.x, d pp.y, d fl, -) of the OTHER frame's intrinsics.
// per-image / per-(image, hypothesis) constants of the backward pass
struct PrBwdK { float c0x, c0y, c1x, c1y, f0, f1, k_mask, k_tex, k_flow, wmean; };

// one pixel: inputs in registers, the ten gradient values out, the three intrinsics sums accumulated
struct PrBwdIn { float q[10], m, oc, io[3], iw[3], ox, oy, g1[3], g2[3]; };
__device__ __forceinline__ void pr_backward_pixel(const PrBwdK& K, const PrBwdIn& I, bool pair, float* o, float& sx, float& sy, float& sf)
{
    const float a = I.q[9];
    const float r[3] = {I.q[0], I.q[1], I.q[2]};
    const bool on = I.oc != 0.f;
    float ga = on ? K.k_mask * (a - I.m) : 0.f;
    float gr[3] = {0.f, 0.f, 0.f};
    if (on) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float s1 = pr_sgn(I.io[c] - r[c] * a);
            const float s2 = pr_sgn(I.iw[c] - r[c]);
            gr[c] = -K.k_tex * (s1 * a + s2);
            ga += -K.k_tex * s1 * r[c];
        }
    }
    if (pair) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            gr[c] += I.g1[c] * a + I.g2[c];
            ga += I.g1[c] * r[c];
        }
    }
    // flow: loss -> rendered flow -> the other frame's position (planes 6-8) and intrinsics
    float x0 = I.q[3], y0 = I.q[4], z0 = I.q[5], x1 = I.q[6], y1 = I.q[7], z1 = I.q[8];
    const bool b = (z0 < 1e-9f) | (z1 < 1e-9f);
    float gx1 = 0.f, gy1 = 0.f, gz1 = 0.f;
    if (b) x0 = y0 = z0 = x1 = y1 = z1 = 10.f;
    const float u0 = K.c0x + (x0 * K.f0) / z0, v0 = K.c0y + (y0 * K.f0) / z0;
    const float u1 = K.c1x + (x1 * K.f1) / z1, v1 = K.c1y + (y1 * K.f1) / z1;
    const float dx = (u1 - u0) - I.ox, dy = (v1 - v0) - I.oy;
    const bool sel = !b && on && I.m > 0.f;
    // when image i has no selected pixel at all wmean is NaN and 0 * NaN = NaN reaches every pixel, exactly what
    // autograd does with the reference code (flow_loss_backward_kernel keeps the same behaviour)
    const float gn = (sel ? K.k_flow : 0.f) * (pr_sigmoid(-I.oc) / K.wmean);
    const float nrm = sqrtf(dx * dx + dy * dy);
    float2 gf = make_float2(0.f, 0.f);
    if (nrm > 0.f) gf = make_float2(gn / nrm * dx, gn / nrm * dy);
    if (!b) {
        const float ax = gf.x / z1, ay = gf.y / z1;
        gx1 = ax * K.f1; gy1 = ay * K.f1;
        gz1 = -(ax * ((x1 * K.f1) / z1) + ay * ((y1 * K.f1) / z1));
        sx += gf.x; sy += gf.y; sf += ax * x1 + ay * y1;
    }
    o[0] = gr[0]; o[1] = gr[1]; o[2] = gr[2]; o[3] = 0.f; o[4] = 0.f; o[5] = 0.f; o[6] = gx1; o[7] = gy1; o[8] = gz1; o[9] = ga;
}

// V4: four consecutive pixels per thread and round through 16-byte loads / stores (every plane, chunk start and chunk length a
// multiple of four pixels: any power-of-two image); the launch streams ~7 MB per image at LASR's sizes and the 4-byte form left
// it at 2.7 TB/s.  Same per-pixel arithmetic; the intrinsics' sums associate differently (gradient bar 2e-4 relative).
template <bool V4>
__global__ __launch_bounds__(256) void render_tables_backward_kernel(PrArgs A, const float* __restrict__ tot, const float* __restrict__ img,
                                                                     const float* __restrict__ g_mask, const float* __restrict__ g_flow,
                                                                     const float* __restrict__ g_tex, const float* __restrict__ g_rndpair,
                                                                     float wt, float* __restrict__ gpx, float* __restrict__ part3)
{
    __shared__ float red[4];
    const int ij = blockIdx.x, i = ij / A.H, ch = blockIdx.y, P = A.P, N = A.I * A.H;
    const float* q = A.px + (size_t)ij * 10 * P;
    float* g = gpx + (size_t)ij * 10 * P;
    const float* m = A.masks + (size_t)i * P;
    const float* oc = A.occ + (size_t)i * P;
    const float* io = A.img_obs + (size_t)i * 3 * P;
    const float* iw = A.img_white + (size_t)i * 3 * P;
    const float* ox = A.obs + (size_t)i * A.obs_stride;
    const float* oy = ox + P;
    const float* gp1 = g_rndpair ? g_rndpair + (size_t)ij * 3 * P : nullptr;
    const float* gp2 = g_rndpair ? g_rndpair + ((size_t)N + ij) * 3 * P : nullptr;
    const int other = (ij + A.half) % N;
    const float* t = tot + (size_t)ij * 8;
    PrBwdK K;
    K.c0x = A.pp[2 * ij]; K.c0y = A.pp[2 * ij + 1]; K.c1x = A.pp[2 * other]; K.c1y = A.pp[2 * other + 1];
    K.f0 = A.fl[ij]; K.f1 = A.fl[other];
    K.k_mask = g_mask[ij] / t[1];                               // 0.5 * 2 * g / count (== mask_loss_backward_kernel)
    K.k_tex = g_tex[ij] * (2.f * wt) / (3.f * t[1]);            // (== tex_loss_backward_kernel)
    K.k_flow = t[5] > 0.f ? 0.5f * g_flow[ij] / t[5] : 0.f;     // (== flow_loss_backward_kernel)
    K.wmean = img[4 * i] / img[4 * i + 1];
    int p0, p1;
    pr_range(P, A.nch, ch, p0, p1);
    float sx = 0.f, sy = 0.f, sf = 0.f;
    const bool pair = g_rndpair != nullptr;
    if (V4) {
        for (int p = p0 + 4 * threadIdx.x; p < p1; p += 1024) {
            float4 Q[10], M, OC, IO[3], IW[3], OX, OY, G1[3], G2[3];
#pragma unroll
            for (int c = 0; c < 10; c++) Q[c] = *reinterpret_cast<const float4*>(q + (size_t)c * P + p);
            M = *reinterpret_cast<const float4*>(m + p); OC = *reinterpret_cast<const float4*>(oc + p);
            OX = *reinterpret_cast<const float4*>(ox + p); OY = *reinterpret_cast<const float4*>(oy + p);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                IO[c] = *reinterpret_cast<const float4*>(io + (size_t)c * P + p);
                IW[c] = *reinterpret_cast<const float4*>(iw + (size_t)c * P + p);
                G1[c] = pair ? *reinterpret_cast<const float4*>(gp1 + (size_t)c * P + p) : make_float4(0.f, 0.f, 0.f, 0.f);
                G2[c] = pair ? *reinterpret_cast<const float4*>(gp2 + (size_t)c * P + p) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float out[10][4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                PrBwdIn In;
#define LASR_EL(v) (e == 0 ? (v).x : e == 1 ? (v).y : e == 2 ? (v).z : (v).w)
#pragma unroll
                for (int c = 0; c < 10; c++) In.q[c] = LASR_EL(Q[c]);
                In.m = LASR_EL(M); In.oc = LASR_EL(OC); In.ox = LASR_EL(OX); In.oy = LASR_EL(OY);
#pragma unroll
                for (int c = 0; c < 3; c++) { In.io[c] = LASR_EL(IO[c]); In.iw[c] = LASR_EL(IW[c]); In.g1[c] = LASR_EL(G1[c]); In.g2[c] = LASR_EL(G2[c]); }
#undef LASR_EL
                float o[10];
                pr_backward_pixel(K, In, pair, o, sx, sy, sf);
#pragma unroll
                for (int c = 0; c < 10; c++) out[c][e] = o[c];
            }
#pragma unroll
            for (int c = 0; c < 10; c++)
                *reinterpret_cast<float4*>(g + (size_t)c * P + p) = make_float4(out[c][0], out[c][1], out[c][2], out[c][3]);
        }
    } else {
        for (int p = p0 + threadIdx.x; p < p1; p += 256) {
            PrBwdIn In;
#pragma unroll
            for (int c = 0; c < 10; c++) In.q[c] = q[(size_t)c * P + p];
            In.m = m[p]; In.oc = oc[p]; In.ox = ox[p]; In.oy = oy[p];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                In.io[c] = io[(size_t)c * P + p]; In.iw[c] = iw[(size_t)c * P + p];
                In.g1[c] = pair ? gp1[(size_t)c * P + p] : 0.f; In.g2[c] = pair ? gp2[(size_t)c * P + p] : 0.f;
            }
            float o[10];
            pr_backward_pixel(K, In, pair, o, sx, sy, sf);
#pragma unroll
            for (int c = 0; c < 10; c++) g[(size_t)c * P + p] = o[c];
        }
    }
    sx = block_sum(sx, red); sy = block_sum(sy, red); sf = block_sum(sf, red);
    if (threadIdx.x == 0) {
        float* o = part3 + ((size_t)ij * A.nch + ch) * 4;
        o[0] = sx; o[1] = sy; o[2] = sf; o[3] = 0.f;
    }
}

// image n's sums belong to the intrinsics of image (n + half) % N
__global__ __launch_bounds__(256) void render_tables_intrinsics_fold_kernel(const float* __restrict__ part3, float* __restrict__ gpp,
                                                                            float* __restrict__ gfl, int N, int nch, int half)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float a = 0.f, b = 0.f, c = 0.f;
    for (int k = 0; k < nch; k++) {
        const float* o = part3 + ((size_t)n * nch + k) * 4;
        a += o[0]; b += o[1]; c += o[2];
    }
    const int other = (n + half) % N;
    gpp[2 * other] = a; gpp[2 * other + 1] = b; gfl[other] = c;
}

// =====================================================================================================================
// The other side of the render: what LASR.forward builds from the camera-space vertices before it calls the rasteriser
// (nnutils/mesh_net.py:298-311, :350-356; geom_utils.py:27-34), in one launch:
//   pinhole projection of [x y z 1] with the frame's principal point and the hypothesis' focal length,
//   raster-space vertices verts_pre = (proj + eye) * (1,-1,1)   (the renderer's look_at subtracts eye again, :81-82, :354-355),
//   the nine vertex attributes (colour | own camera-space position | the other frame's position),
//   and the min / max of the projected depth for the near / far planes (:304-311), finished by a one-block second kernel.
// verts_cam [N,V,3], tex [N,V,3], pp [N/H... expanded by the caller to N,2], fl [N]; image n's "other frame" is (n + N/2) % N.
struct RiArgs { const float* verts_cam; const float* tex; const float* pp; const float* fl; int N, V, half; float ex, ey, ez; };

__global__ __launch_bounds__(256) void raster_inputs_forward_kernel(RiArgs A, float* __restrict__ verts_pre, float* __restrict__ attrs,
                                                                    float* __restrict__ zpart)
{
    __shared__ float red[8];
    const int n = blockIdx.x, tid = threadIdx.x, V = A.V;
    const int other = (n + A.half) % A.N;
    const float f = A.fl[n], cx = A.pp[2 * n], cy = A.pp[2 * n + 1];
    float zmin = 3.4e38f, zmax = -3.4e38f;
    for (int v = tid; v < V; v += 256) {
        const size_t i = (size_t)n * V + v, o = (size_t)other * V + v;
        const float x = A.verts_cam[3 * i], y = A.verts_cam[3 * i + 1], z = A.verts_cam[3 * i + 2];
        // geom_utils.py:32-33: pp + (x * fl) / z; then (+ eye) * (1, -1, 1)
        const float px = cx + x * f / z, py = cy + y * f / z;
        verts_pre[3 * i] = (px + A.ex) * 1.f;
        verts_pre[3 * i + 1] = (py + A.ey) * -1.f;
        verts_pre[3 * i + 2] = (z + A.ez) * 1.f;
        float* a = attrs + 9 * i;
        a[0] = A.tex[3 * i]; a[1] = A.tex[3 * i + 1]; a[2] = A.tex[3 * i + 2];
        a[3] = x; a[4] = y; a[5] = z;
        a[6] = A.verts_cam[3 * o]; a[7] = A.verts_cam[3 * o + 1]; a[8] = A.verts_cam[3 * o + 2];
        zmin = fminf(zmin, z); zmax = fmaxf(zmax, z);
    }
    // block min / max (NaN depth: fminf / fmaxf drop it, like torch.aminmax does not -- a NaN vertex already poisons the step)
    for (int d = 32; d >= 1; d >>= 1) { zmin = fminf(zmin, __shfl_xor(zmin, d)); zmax = fmaxf(zmax, __shfl_xor(zmax, d)); }
    if ((tid & 63) == 0) { red[tid >> 6] = zmin; red[4 + (tid >> 6)] = zmax; }
    __syncthreads();
    if (tid == 0) {
        zpart[2 * n] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
        zpart[2 * n + 1] = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    }
}

__global__ __launch_bounds__(64) void raster_inputs_nearfar_kernel(const float* __restrict__ zpart, float* __restrict__ near_far, int N)
{
    float zmin = 3.4e38f, zmax = -3.4e38f;
    for (int n = threadIdx.x; n < N; n += 64) { zmin = fminf(zmin, zpart[2 * n]); zmax = fmaxf(zmax, zpart[2 * n + 1]); }
    for (int d = 32; d >= 1; d >>= 1) { zmin = fminf(zmin, __shfl_xor(zmin, d)); zmax = fmaxf(zmax, __shfl_xor(zmax, d)); }
    if (threadIdx.x == 0) {
        const float half_range = (zmax - zmin) / 2.f;              // mesh_net.py:306-311
        near_far[0] = zmin - half_range;
        near_far[1] = zmax + half_range;
    }
}

// one block per image: d verts_cam (own projection + own position attribute + the position attribute it lends to the other
// frame), d tex, d pp, d fl
__global__ __launch_bounds__(256) void raster_inputs_backward_kernel(RiArgs A, const float* __restrict__ g_pre, const float* __restrict__ g_attrs,
                                                                     float* __restrict__ g_cam, float* __restrict__ g_tex,
                                                                     float* __restrict__ g_pp, float* __restrict__ g_fl)
{
    __shared__ float red[4];
    const int n = blockIdx.x, tid = threadIdx.x, V = A.V;
    const int other = (n + A.half) % A.N;              // the image whose attributes 6..8 are THIS image's positions
    const float f = A.fl[n];
    float sx = 0.f, sy = 0.f, sf = 0.f;
    for (int v = tid; v < V; v += 256) {
        const size_t i = (size_t)n * V + v, o = (size_t)other * V + v;
        const float x = A.verts_cam[3 * i], y = A.verts_cam[3 * i + 1], z = A.verts_cam[3 * i + 2];
        const float gx = g_pre[3 * i], gy = -g_pre[3 * i + 1], gz = g_pre[3 * i + 2];
        const float iz = 1.f / z, xz = x * iz, yz = y * iz;                   // == pinhole_backward_kernel
        const float* ga = g_attrs + 9 * i;
        const float* go = g_attrs + 9 * o;
        g_cam[3 * i] = gx * f * iz + ga[3] + go[6];
        g_cam[3 * i + 1] = gy * f * iz + ga[4] + go[7];
        g_cam[3 * i + 2] = gz - (gx * xz + gy * yz) * f * iz + ga[5] + go[8];
        g_tex[3 * i] = ga[0]; g_tex[3 * i + 1] = ga[1]; g_tex[3 * i + 2] = ga[2];
        sx += gx; sy += gy; sf += gx * xz + gy * yz;
    }
    sx = block_sum(sx, red); sy = block_sum(sy, red); sf = block_sum(sf, red);
    if (tid == 0) { g_pp[2 * n] = sx; g_pp[2 * n + 1] = sy; g_fl[n] = sf; }
}


// =====================================================================================================================
// The same stage written PER FACE CORNER: what the rasteriser actually consumes is face_vertices [N,F,3,3] and the per-face
// attribute stack [N,F,3,9]; via raster_inputs_forward_kernel they take five launches (that kernel, its near / far fold, the
// camera stage's `vertices - eye`, two face gathers) and three on the way back (two vertex-centric gathers, the kernel above).
// Here one launch each way:
//   forward : corner c of face f of mesh n reads its vertex once and writes the projected, eye-shifted position
//             ((pinhole + eye) * (1,-1,1)) - eye  -- the expression of raster_inputs_forward_kernel followed by look_at's
//             subtraction (soft_renderer/functional/look_at.py:6-62 with LASR's constant eye on the -z axis: R = I), bit for bit --
//             and the nine attributes; blocks also take min / max depth of a 256-vertex slice; raster_inputs_nearfar_kernel folds
//             them into near / far (mesh_net.py:304-311).
//   backward: 16 lanes per vertex walk the vertex' incident corners (CSR built once per connectivity, corners ascending: the
//             summation order of face_gather_backward_kernel) -- 3 position sums, 6 own-attribute sums, 3 sums of the position
//             attribute this vertex lends to the other frame's mesh -- then one lane per vertex applies the projection's
//             Jacobian; d pp / d fl: block partials in LDS order, folded in block order by a second small launch.
// faces: [Nf,F,3] int64 with Nf = N, or Nf = 1 when all meshes share the connectivity (LASR: always); same for the CSR.
struct RfArgs {
    const float* verts_cam; const float* tex; const float* pp; const float* fl;
    const long long* faces; int faces_shared;
    int N, V, F3, half; float ex, ey, ez;
};

__global__ __launch_bounds__(256) void raster_faces_forward_kernel(RfArgs A, float* __restrict__ fv, float* __restrict__ fa,
                                                                   float* __restrict__ zpart)
{
    __shared__ float red[8];
    const int n = blockIdx.y, tid = threadIdx.x, V = A.V;
    const int other = (n + A.half) % A.N;
    const int c = blockIdx.x * 256 + tid;
    if (c < A.F3) {
        const long long vi = A.faces[(A.faces_shared ? (size_t)0 : (size_t)n * A.F3) + c];
        const size_t i = (size_t)n * V + (size_t)vi, o = (size_t)other * V + (size_t)vi;
        const float f = A.fl[n], cx = A.pp[2 * n], cy = A.pp[2 * n + 1];
        const float x = A.verts_cam[3 * i], y = A.verts_cam[3 * i + 1], z = A.verts_cam[3 * i + 2];
        const float px = cx + x * f / z, py = cy + y * f / z;           // geom_utils.py:32-33
        float* q = fv + ((size_t)n * A.F3 + c) * 3;
        q[0] = (px + A.ex) * 1.f - A.ex;                                  // (+ eye) * (1,-1,1), then look_at's - eye
        q[1] = (py + A.ey) * -1.f - A.ey;
        q[2] = (z + A.ez) * 1.f - A.ez;
        float* a = fa + ((size_t)n * A.F3 + c) * 9;
        a[0] = A.tex[3 * i]; a[1] = A.tex[3 * i + 1]; a[2] = A.tex[3 * i + 2];
        a[3] = x; a[4] = y; a[5] = z;
        a[6] = A.verts_cam[3 * o]; a[7] = A.verts_cam[3 * o + 1]; a[8] = A.verts_cam[3 * o + 2];
    }
    // depth range over the VERTICES (referenced by a face or not, like the reference's min / max over the projected batch)
    float zmin = 3.4e38f, zmax = -3.4e38f;
    if (c < V) { const float z = A.verts_cam[3 * ((size_t)n * V + c) + 2]; zmin = z; zmax = z; }
    for (int d = 32; d >= 1; d >>= 1) { zmin = fminf(zmin, __shfl_xor(zmin, d)); zmax = fmaxf(zmax, __shfl_xor(zmax, d)); }
    if ((tid & 63) == 0) { red[tid >> 6] = zmin; red[4 + (tid >> 6)] = zmax; }
    __syncthreads();
    const int nblk = gridDim.x;
    if (tid == 0) {
        float* zp = zpart + 2 * ((size_t)n * nblk + blockIdx.x);
        zp[0] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
        zp[1] = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    }
}

constexpr int RF_VPB = 16;            // vertices per block of the backward (16 lanes each)
__global__ __launch_bounds__(256) void raster_faces_backward_kernel(RfArgs A, const int* __restrict__ inc_ptr, const int* __restrict__ inc,
                                                                    const float* __restrict__ g_fv, const float* __restrict__ g_fa,
                                                                    float* __restrict__ g_cam, float* __restrict__ g_tex,
                                                                    float* __restrict__ part)
{
    __shared__ float sums[RF_VPB][16];
    __shared__ float bsum[RF_VPB][3];
    const int n = blockIdx.y, tid = threadIdx.x, V = A.V, F3 = A.F3;
    const int other = (n + A.half) % A.N;              // the mesh whose attributes 6..8 are THIS mesh's positions
    const int vl = tid >> 4, lane = tid & 15, v = blockIdx.x * RF_VPB + vl;
    // lane 0-2: d verts_pre; 3-8: own attributes 0..5 (colour, own position); 9-11: attributes 6..8 of mesh `other`
    float acc = 0.f;
    if (v < V && lane < 12) {
        const size_t inc_base = A.faces_shared ? 0 : (size_t)n;
        const int* ptr = inc_ptr + inc_base * (V + 1);
        const int* lst = inc + inc_base * F3;
        const float* src; int stride, off;
        if (lane < 3) { src = g_fv + (size_t)n * F3 * 3; stride = 3; off = lane; }
        else if (lane < 9) { src = g_fa + (size_t)n * F3 * 9; stride = 9; off = lane - 3; }
        else { src = g_fa + (size_t)other * F3 * 9; stride = 9; off = lane - 3; }
        if (lane >= 9 && !A.faces_shared) {                // the other mesh's own incidence lists
            ptr = inc_ptr + (size_t)other * (V + 1);
            lst = inc + (size_t)other * F3;
        }
        const int e0 = ptr[v], e1 = ptr[v + 1];
        for (int e = e0; e < e1; e += 4) {                 // corner ids first, then the four gathers: two latency levels per round
            int cid[4]; float g[4];
#pragma unroll
            for (int j = 0; j < 4; j++) cid[j] = e + j < e1 ? lst[e + j] : -1;
#pragma unroll
            for (int j = 0; j < 4; j++) g[j] = cid[j] >= 0 ? src[(size_t)cid[j] * stride + off] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) if (cid[j] >= 0) acc += g[j];          // ascending corner order
        }
    }
    sums[vl][lane] = acc;
    __syncthreads();
    if (lane == 0) {
        float sx = 0.f, sy = 0.f, sf = 0.f;
        if (v < V) {
            const size_t i = (size_t)n * V + v;
            const float f = A.fl[n];
            const float x = A.verts_cam[3 * i], y = A.verts_cam[3 * i + 1], z = A.verts_cam[3 * i + 2];
            const float* S = sums[vl];
            const float gx = S[0], gy = -S[1], gz = S[2];
            const float iz = 1.f / z, xz = x * iz, yz = y * iz;                   // == raster_inputs_backward_kernel
            g_cam[3 * i] = gx * f * iz + S[6] + S[9];
            g_cam[3 * i + 1] = gy * f * iz + S[7] + S[10];
            g_cam[3 * i + 2] = gz - (gx * xz + gy * yz) * f * iz + S[8] + S[11];
            g_tex[3 * i] = S[3]; g_tex[3 * i + 1] = S[4]; g_tex[3 * i + 2] = S[5];
            sx = gx; sy = gy; sf = gx * xz + gy * yz;
        }
        bsum[vl][0] = sx; bsum[vl][1] = sy; bsum[vl][2] = sf;
    }
    __syncthreads();
    const int nblk = gridDim.x;
    if (tid < 3) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < RF_VPB; k++) t += bsum[k][tid];                          // vertex order
        part[((size_t)n * nblk + blockIdx.x) * 4 + tid] = t;
    }
}

// one block per mesh: the block partials are staged into LDS by all threads (coalesced, all loads in flight), 1024 blocks at a
// time -- any number of vertices -- then three threads add them up in block order
constexpr int RF_FOLD_CHUNK = 1024;
__global__ __launch_bounds__(256) void raster_faces_fold_kernel(const float* __restrict__ part, float* __restrict__ g_pp,
                                                                float* __restrict__ g_fl, int N, int nblk)
{
    __shared__ float stage[RF_FOLD_CHUNK * 4];
    const int m = blockIdx.x, tid = threadIdx.x;
    float t = 0.f;
    for (int b0 = 0; b0 < nblk; b0 += RF_FOLD_CHUNK) {
        const int nb = min(RF_FOLD_CHUNK, nblk - b0);
        if (b0) __syncthreads();
        for (int i = tid; i < nb * 4; i += 256) stage[i] = part[((size_t)m * nblk + b0) * 4 + i];
        __syncthreads();
        if (tid < 3)
            for (int b = 0; b < nb; b++) t += stage[b * 4 + tid];                         // block order
    }
    if (tid < 2) g_pp[2 * m + tid] = t;
    else if (tid == 2) g_fl[m] = t;
}

}  // namespace lasr

using namespace lasr;

extern "C" size_t lasr_raster_faces_scratch_floats(int N, int V, int F)
{
    if (N < 0 || V < 0 || F < 0) return 0;
    const size_t cov = (size_t)(3 * F > V ? 3 * F : V);
    const size_t fwd = 2 * (size_t)N * ((cov + 255) / 256);
    const size_t bwd = 4 * (size_t)N * (((size_t)V + RF_VPB - 1) / RF_VPB);
    return (fwd > bwd ? fwd : bwd) + 4;
}

extern "C" int lasr_raster_faces_forward(const float* verts_cam, const float* tex, const float* pp, const float* fl, const float* eye,
                                         const long long* faces, int faces_shared, float* face_vertices, float* face_attrs,
                                         float* near_far, float* scratch, int N, int V, int F, void* hip_stream)
{
    if (N < 0 || V < 0 || F < 0 || (N % 2)) return LASR_E_BADARG;
    if (N == 0 || V == 0) return LASR_OK;
    if (!verts_cam || !tex || !pp || !fl || !eye || !near_far || !scratch) return LASR_E_BADARG;
    if (F > 0 && (!faces || !face_vertices || !face_attrs)) return LASR_E_BADARG;
    if ((long long)3 * F > 0x7fffffffLL) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    RfArgs A{verts_cam, tex, pp, fl, faces, faces_shared, N, V, 3 * F, N / 2, eye[0], eye[1], eye[2]};
    const int cov = 3 * F > V ? 3 * F : V;
    const int nblk = (cov + 255) / 256;
    LASR_LAUNCH(K_RASTER_FACES, raster_faces_forward_kernel, dim3(nblk, N), dim3(256), 0, A, face_vertices, face_attrs, scratch);
    int rc = launch_ok();
    if (rc) return rc;
    LASR_LAUNCH(K_RASTER_FACES, raster_inputs_nearfar_kernel, dim3(1), dim3(64), 0, scratch, near_far, N * nblk);
    return launch_ok();
}

extern "C" int lasr_raster_faces_backward(const float* verts_cam, const float* fl, const int* inc_ptr, const int* inc,
                                          int faces_shared, const float* grad_face_vertices, const float* grad_face_attrs,
                                          float* grad_verts_cam, float* grad_tex, float* grad_pp, float* grad_fl, float* scratch,
                                          int N, int V, int F, void* hip_stream)
{
    if (N < 0 || V < 0 || F < 0 || (N % 2)) return LASR_E_BADARG;
    if (N == 0 || V == 0) return LASR_OK;
    if (!verts_cam || !fl || !inc_ptr || !grad_verts_cam || !grad_tex || !grad_pp || !grad_fl || !scratch) return LASR_E_BADARG;
    if (F > 0 && (!inc || !grad_face_vertices || !grad_face_attrs)) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    RfArgs A{verts_cam, nullptr, nullptr, fl, nullptr, faces_shared, N, V, 3 * F, N / 2, 0.f, 0.f, 0.f};
    const int nblk = (V + RF_VPB - 1) / RF_VPB;
    LASR_LAUNCH(K_RASTER_FACES, raster_faces_backward_kernel, dim3(nblk, N), dim3(256), 0, A, inc_ptr, inc, grad_face_vertices,
                grad_face_attrs, grad_verts_cam, grad_tex, scratch);
    int rc = launch_ok();
    if (rc) return rc;
    LASR_LAUNCH(K_RASTER_FACES, raster_faces_fold_kernel, dim3(N), dim3(256), 0, scratch, grad_pp, grad_fl, N, nblk);
    return launch_ok();
}


extern "C" int lasr_raster_inputs_forward(const float* verts_cam, const float* tex, const float* pp, const float* fl, const float* eye,
                                          float* verts_pre, float* attrs, float* near_far, float* scratch, int N, int V,
                                          void* hip_stream)
{
    if (N < 0 || V < 0 || (N % 2)) return LASR_E_BADARG;
    if (N == 0 || V == 0) return LASR_OK;
    if (!verts_cam || !tex || !pp || !fl || !eye || !verts_pre || !attrs || !near_far || !scratch) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    RiArgs A{verts_cam, tex, pp, fl, N, V, N / 2, eye[0], eye[1], eye[2]};
    LASR_LAUNCH(K_RASTER_INPUTS, raster_inputs_forward_kernel, dim3(N), dim3(256), 0, A, verts_pre, attrs, scratch);
    int rc = launch_ok();
    if (rc) return rc;
    LASR_LAUNCH(K_RASTER_INPUTS, raster_inputs_nearfar_kernel, dim3(1), dim3(64), 0, scratch, near_far, N);
    return launch_ok();
}

extern "C" int lasr_raster_inputs_backward(const float* verts_cam, const float* fl, const float* grad_verts_pre,
                                           const float* grad_attrs, float* grad_verts_cam, float* grad_tex, float* grad_pp,
                                           float* grad_fl, int N, int V, void* hip_stream)
{
    if (N < 0 || V < 0 || (N % 2)) return LASR_E_BADARG;
    if (N == 0 || V == 0) return LASR_OK;
    if (!verts_cam || !fl || !grad_verts_pre || !grad_attrs || !grad_verts_cam || !grad_tex || !grad_pp || !grad_fl) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    RiArgs A{verts_cam, nullptr, nullptr, fl, N, V, N / 2, 0.f, 0.f, 0.f};
    LASR_LAUNCH(K_RASTER_INPUTS, raster_inputs_backward_kernel, dim3(N), dim3(256), 0, A, grad_verts_pre, grad_attrs, grad_verts_cam,
                grad_tex, grad_pp, grad_fl);
    return launch_ok();
}

extern "C" size_t lasr_render_tables_scratch_floats(int I, int H, int P)
{
    if (I < 0 || H < 0 || P < 0) return 0;
    const size_t N = (size_t)I * H, nch = (size_t)pr_nch(P);
    return N * nch * 8 + N * 8 + (size_t)I * 4 + N * nch * 4 + N * nch * 4 + 16;
}

static int pr_args(PrArgs& A, const float* px, const float* masks, const float* occ, const float* flow_obs, long long obs_stride,
                   const float* img_obs, const float* img_white, const float* pp, const float* fl, int I, int H, int P)
{
    if (I < 0 || H < 0 || P < 0) return LASR_E_BADARG;
    if (I == 0 || H == 0 || P == 0) return LASR_OK;
    if (!px || !masks || !occ || !flow_obs || !img_obs || !img_white || !pp || !fl || obs_stride < 2LL * P) return LASR_E_BADARG;
    if (((long long)I * H) % 2) return LASR_E_BADARG;                 // [frame t block ; frame t' block]
    A.px = px; A.masks = masks; A.occ = occ; A.obs = flow_obs; A.img_obs = img_obs; A.img_white = img_white; A.pp = pp; A.fl = fl;
    A.I = I; A.H = H; A.P = P; A.nch = pr_nch(P); A.half = I * H / 2; A.obs_stride = obs_stride;
    return LASR_OK;
}

static int render_tables_forward_impl(const float* px, const float* masks, const float* occ, const float* flow_obs,
                                      long long flow_obs_image_stride, const float* img_obs, const float* img_white,
                                      const float* pp, const float* fl, float l1tex_wt, float* mask_tab, float* flow_tab,
                                      float* tex_tab, float* flow_rd, unsigned char* bgmask, float* flow_map, unsigned char* vis_mask,
                                      float* rndpair, float* scratch, int I, int H, int P, void* hip_stream, float* obs_pair_out)
{
    PrArgs A;
    int rc = pr_args(A, px, masks, occ, flow_obs, flow_obs_image_stride, img_obs, img_white, pp, fl, I, H, P);
    if (rc || I == 0 || H == 0 || P == 0) return rc;
    if (!mask_tab || !flow_tab || !tex_tab || !flow_rd || !bgmask || !flow_map || !vis_mask || !scratch) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const PrScratch S = pr_scratch(scratch, I, H, P);
    const int N = I * H;
    if (obs_pair_out)
        LASR_LAUNCH(K_RENDER_TABLES_FORWARD, render_tables_forward_kernel<true>, dim3(N, A.nch), dim3(256), 0, A, (float2*)flow_rd, bgmask,
                    rndpair, S.part, obs_pair_out);
    else
        LASR_LAUNCH(K_RENDER_TABLES_FORWARD, render_tables_forward_kernel<false>, dim3(N, A.nch), dim3(256), 0, A, (float2*)flow_rd, bgmask,
                    rndpair, S.part, (float*)nullptr);
    if ((rc = launch_ok())) return rc;
    LASR_LAUNCH(K_RENDER_TABLES_FLOW, render_tables_flow_kernel, dim3(N, A.nch), dim3(256), 0, A, (const float2*)flow_rd, bgmask,
                S.part, S.tot, S.img, mask_tab, tex_tab, 2.f * l1tex_wt, S.part2, flow_map, vis_mask);
    if ((rc = launch_ok())) return rc;
    LASR_LAUNCH(K_RENDER_TABLES_FOLD, render_tables_flow_fold_kernel, dim3((N + 255) / 256), dim3(256), 0, S.part2, S.tot, flow_tab,
                N, A.nch);
    return launch_ok();
}

extern "C" int lasr_render_tables_forward(const float* px, const float* masks, const float* occ, const float* flow_obs,
                                          long long flow_obs_image_stride, const float* img_obs, const float* img_white,
                                          const float* pp, const float* fl, float l1tex_wt, float* mask_tab, float* flow_tab,
                                          float* tex_tab, float* flow_rd, unsigned char* bgmask, float* flow_map, unsigned char* vis_mask,
                                          float* rndpair, float* scratch, int I, int H, int P, void* hip_stream)
{
    return render_tables_forward_impl(px, masks, occ, flow_obs, flow_obs_image_stride, img_obs, img_white, pp, fl, l1tex_wt, mask_tab,
                                      flow_tab, tex_tab, flow_rd, bgmask, flow_map, vis_mask, rndpair, scratch, I, H, P, hip_stream, nullptr);
}

extern "C" int lasr_render_tables_forward_imgs(const float* px, const float* masks, const float* occ, const float* flow_obs,
                                               long long flow_obs_image_stride, const float* imgs, float* obs_pair_out,
                                               const float* pp, const float* fl, float l1tex_wt, float* mask_tab, float* flow_tab,
                                               float* tex_tab, float* flow_rd, unsigned char* bgmask, float* flow_map,
                                               unsigned char* vis_mask, float* rndpair, float* scratch, int I, int H, int P,
                                               void* hip_stream)
{
    if (I > 0 && H > 0 && P > 0 && !obs_pair_out) return LASR_E_BADARG;
    return render_tables_forward_impl(px, masks, occ, flow_obs, flow_obs_image_stride, imgs, imgs, pp, fl, l1tex_wt, mask_tab, flow_tab,
                                      tex_tab, flow_rd, bgmask, flow_map, vis_mask, rndpair, scratch, I, H, P, hip_stream, obs_pair_out);
}

extern "C" int lasr_render_tables_backward(const float* px, const float* masks, const float* occ, const float* flow_obs,
                                           long long flow_obs_image_stride, const float* img_obs, const float* img_white,
                                           const float* pp, const float* fl, float l1tex_wt, const float* grad_mask_tab,
                                           const float* grad_flow_tab, const float* grad_tex_tab, const float* grad_rndpair,
                                           const float* scratch, float* grad_px, float* grad_pp, float* grad_fl, int I, int H, int P,
                                           void* hip_stream)
{
    PrArgs A;
    int rc = pr_args(A, px, masks, occ, flow_obs, flow_obs_image_stride, img_obs, img_white, pp, fl, I, H, P);
    if (rc || I == 0 || H == 0 || P == 0) return rc;
    if (!grad_mask_tab || !grad_flow_tab || !grad_tex_tab || !scratch || !grad_px || !grad_pp || !grad_fl) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const PrScratch S = pr_scratch(const_cast<float*>(scratch), I, H, P);
    const int N = I * H;
    // 16-byte form: every plane 16-byte aligned and the chunks whole quads of pixels
    const int per = (P + A.nch - 1) / A.nch;
    const uintptr_t bits = (uintptr_t)px | (uintptr_t)masks | (uintptr_t)occ | (uintptr_t)flow_obs | (uintptr_t)img_obs | (uintptr_t)img_white |
                           (uintptr_t)grad_px | (uintptr_t)grad_rndpair;
    const bool v4 = (P % 4) == 0 && (per % 4) == 0 && (flow_obs_image_stride % 4) == 0 && (bits & 15) == 0;
    if (v4)
        LASR_LAUNCH(K_RENDER_TABLES_BACKWARD, render_tables_backward_kernel<true>, dim3(N, A.nch), dim3(256), 0, A, S.tot, S.img,
                    grad_mask_tab, grad_flow_tab, grad_tex_tab, grad_rndpair, l1tex_wt, grad_px, S.part3);
    else
        LASR_LAUNCH(K_RENDER_TABLES_BACKWARD, render_tables_backward_kernel<false>, dim3(N, A.nch), dim3(256), 0, A, S.tot, S.img,
                    grad_mask_tab, grad_flow_tab, grad_tex_tab, grad_rndpair, l1tex_wt, grad_px, S.part3);
    if ((rc = launch_ok())) return rc;
    LASR_LAUNCH(K_RENDER_TABLES_FOLD, render_tables_intrinsics_fold_kernel, dim3((N + 255) / 256), dim3(256), 0, S.part3, grad_pp,
                grad_fl, N, A.nch, A.half);
    return launch_ok();
}
