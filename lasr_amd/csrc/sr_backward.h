// sr_backward.h -- the face-major backward raster kernel (template).  Two translation units instantiate it:
//   sr_raster.hip        <false, 3>  every mode combination, compiled like the forward pass with -ffp-contract=off (the
//                                    surface-texture path picks texels from (int)(w * res) and must see the forward's w);
//   sr_backward_fast.hip <true, 3|6|9> LASR's mode combination (vertex attributes), compiled with -ffp-contract=fast: the
//                                    reference's own backward is only defined up to float-atomic ordering (bar: 1e-3 of the
//                                    largest gradient), so multiply-add pairs may fuse -- the kernel is VALU-issue bound and
//                                    without contraction 672 M of its 1131 M VALU instructions per 256 frames are lone
//                                    v_mul_f32 / v_add_f32 (profiles/r02c_v0_pmc.txt).
#pragma once
#include "sr_common.h"

namespace lasr {

// ---------------------------------------------------------------------------
// Backward: one wave per (image, face); K.cu:486-668 evaluated face-major, in two stages.
//   stage 1 (cheap, 64 bbox pixels per round): exact bbox test + barycentrics + the conservative
//            line-distance reject; survivors are compacted (ballot + prefix) into a per-wave LDS ring.
//   stage 2 (heavy, runs whenever 64 survivors are queued): full fragment + gradient math on dense lanes.
// Roughly half of the bbox pixels of a face are farther than sqrt(threshold) from it; without the
// compaction they would idle through the heavy code.  The heavy code uses v_rcp/v_exp based math
// (FM = true): the reference backward is itself only defined up to float-atomic ordering.
constexpr bool BWD_FM = true;
#ifndef LASR_BWD_LDSREC
#define LASR_BWD_LDSREC 1      // LASR's modes: stage 2 reads the face's record and attributes from LDS (see the kernel)
#endif
#ifndef LASR_BWD_WPE8
#define LASR_BWD_WPE8 0
#endif
#ifndef LASR_BWD_DIVREC
#define LASR_BWD_DIVREC 1
#endif
#ifndef LASR_BWD_ONE
#define LASR_BWD_ONE 1      // one edge projection per pixel for well-conditioned faces (sr_device.h: euclid_one)
#endif
constexpr int QCAP = 128;   // ring entries per wave (power of two, >= 2 * 64)
constexpr int BWD_THREADS = 64;   // one wave per workgroup (see backward_impl)

// (Register budget of the six- / nine-channel instantiations: 90 / 106 VGPRs = 5 / 4 waves per SIMD.  Asking for one wave more,
// amdgpu_waves_per_eu(6 / 5), gives 78 / 88 VGPRs without a spill and a 28-50 % SLOWER kernel -- the schedule that fits re-derives
// instead of keeping -- profiles/r05_backward_ab.txt.  Not requested.)
template <bool LASR_FAST, int NCH>
__global__ __launch_bounds__(BWD_THREADS)
#if LASR_BWD_WPE8
__attribute__((amdgpu_waves_per_eu(LASR_FAST && NCH == 3 ? 8 : 1, LASR_FAST && NCH == 3 ? 8 : 8)))      // measurement build
#endif
void sr_backward_kernel(RasterArgs A, const float* __restrict__ colors,
                                                          const float* __restrict__ aggrs,
                                                          const float* __restrict__ gcolors,
                                                          float* __restrict__ gfaces, float* __restrict__ gtex)
{
    __shared__ unsigned int s_ring[BWD_THREADS / 64][QCAP];
#if LASR_BWD_LDSREC
    // LASR's modes: stage 2 reads the face's record and vertex attributes from LDS -- broadcast ds_reads whose results are VGPR
    // operands -- instead of scalar registers: a VALU instruction with an SGPR operand issues in ~4 cycles, with VGPR operands in
    // ~2.5 (profiles/r02_valu_issue.txt), and stage 2 has ~130 record operands per batch.  (Holding the record in VGPRs for the
    // whole face costs 33 registers = half the resident waves: slower, profiles/experiments/README.md.)  Measured: backward
    // 1.139 -> 1.092 ms at 256 frames, 65 VGPRs (the distance code only: with the vertex attributes and 1 / z from LDS as well the
    // kernel needs 70 VGPRs and takes 1.105 ms).
    __shared__ __attribute__((aligned(16))) float s_rec[BWD_THREADS / 64][REC];
#endif
    constexpr bool FM = BWD_FM;
    const Modes m = LASR_FAST ? Modes{2, 1, 2, 1, 1} : A.m;
    if (A.near_far_dev) { A.near = A.near_far_dev[0]; A.far = A.near_far_dev[1]; }
    const int lane = threadIdx.x & 63;
    unsigned int* ring = s_ring[threadIdx.x >> 6];
    // blocks of one image stay on one XCD (block b runs on XCD b % 8): its 10 pixel planes (2.6 MB at 256x256) then
    // live in a single 4 MB L2 instead of being fetched by all eight.  Every XCD walks its faces LAST TO FIRST: the backward
    // pass follows the forward pass, whose most recently written images are the ones still in L2 / Infinity Cache (measured,
    // same box, first-to-last -> last-to-first: backward 1.292 -> 1.262 ms at 256 frames, 0.318 -> 0.310 at 64, 0.099 -> 0.092
    // at 16, 0.998 -> 0.961 at 64 frames of 512x512; profiles/r04_tile_order_ab.txt).  A face's gradients are written by its
    // own wave: the walk order does not enter the result.
#if defined(LASR_BWD_ORDER) && LASR_BWD_ORDER == 1      // measurement build: the XCD's images interleaved face by face
    int blk = xcd_remap(blockIdx.x, gridDim.x);
    {
        const int total = gridDim.x, per = total >> 3;
        if ((total & 7) == 0 && per % A.F == 0) {
            const int m = per / A.F, i = blockIdx.x >> 3, rank = i / m;
            blk = ((blockIdx.x & 7) * m + (i - rank * m)) * A.F + rank;
        }
    }
#elif defined(LASR_BWD_ORDER) && LASR_BWD_ORDER == 3    // measurement build: first to last (rounds 2-4 until this change)
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
#else
    int blk;
    {
        const int total = gridDim.x, per = total >> 3;
        blk = (total & 7) == 0 ? (blockIdx.x & 7) * per + (per - 1 - (blockIdx.x >> 3)) : total - 1 - blockIdx.x;
    }
#endif
    const int gw = __builtin_amdgcn_readfirstlane((int)((blk * blockDim.x + threadIdx.x) >> 6));
    if (gw >= A.N * A.F) return;
    const int bn = gw / A.F, fn = gw - bn * A.F;
    const int IS = A.IS, P = IS * IS;
    const cptr_t rec = as_const(A.recs + (size_t)gw * REC);
    const cptr_t tex = as_const(A.textures + (size_t)gw * A.T * NCH);
    const int flags = __float_as_int(rec[R_FLAGS]);
#if LASR_BWD_LDSREC
    typedef const float __attribute__((address_space(3)))* lptr_t;
    if (LASR_FAST && (flags & 16)) {
        float* dst = s_rec[threadIdx.x >> 6];
        if (lane < REC) dst[lane] = A.recs[(size_t)gw * REC + lane];
        __builtin_amdgcn_wave_barrier();
    }
    const lptr_t lrec = (lptr_t)s_rec[threadIdx.x >> 6];
#endif

    // exact pixel rectangle of the bbox test (columns x0..x1, rows r0..r1 from the top); empty when x0 > x1.  Read from the
    // record's first cache line (the flags sit there too) rather than from the separate rect array the forward's binning scans
    const unsigned rlo = (unsigned)__float_as_int(rec[R_BB + 0]), rext = (unsigned)__float_as_int(rec[R_BB + 1]);
    const int x0 = (int)(rlo & 0xffff), r0 = (int)(rlo >> 16);
    const int bw = (int)(rext & 0xffff) + 1, bh = (int)(rext >> 16) + 1;
    const bool empty = rlo == 0xffffffffu;
    const int npx = empty ? 0 : bw * bh;

    float gv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // d/d(x0 y0 z0 x1 y1 z1 x2 y2 z2)
    constexpr int NGT = 3 * NCH < 18 ? 18 : 3 * NCH;
    constexpr int THIRD = NCH == 9 ? 18 : 0;     // start of the third group of nine attribute components (NCH = 9 only)
    float gt[NGT];                               // vertex attributes: [vertex j][channel k] at NCH*j + k (3*NCH used)
#pragma unroll
    for (int k = 0; k < NGT; k++) gt[k] = 0.f;
    const bool front = (flags & 8) != 0;
    const bool vertex_tex = (m.tex == 1);
    const int lim = (A.N * A.F - gw) * A.T;      // texels from this face to the end of the tensor
    // cheap reject only for the soft distance modes and well-conditioned faces; 2 % slack on thr
    const bool use_far = (m.dist == 2) && (flags & 16);
    const float thr_pad = A.thr * 1.05f;
    const float inv_is = A.inv_is;               // 1.f / (float)IS, from the host (sr_common.h)
    const bool pow2 = (IS & (IS - 1)) == 0;      // then n * (1/IS) == n / IS exactly: skip the division per pixel
    // pixel centres as ONE fma of the converted index: (2 i + 1 - IS) / IS = i * (2 / IS) + (1 - IS) / IS (K.cu:343-346; rows count
    // from the top: yi = IS - 1 - row).  For a power-of-two image every term and the result are exact in fp32; other sizes keep
    // the division in stage 2 (the distance code is ill-conditioned in the pixel position on edge-on faces, and the surface
    // texel choice must see the forward's barycentrics) and use the fma -- within an ulp -- for stage 1's conservative reject only
    const float cx_a = A.cx_a, cx_b = A.cx_b, cy_b = A.cy_b;      // 2 / IS, (1 - IS) / IS, (IS - 1) / IS
    auto centre_x = [&](int xi) { return OPT_BWD_S1 && pow2 ? __builtin_fmaf((float)xi, cx_a, cx_b) : pix_center_p2(xi, IS, inv_is, pow2); };
    auto centre_y = [&](int row) { return OPT_BWD_S1 && pow2 ? __builtin_fmaf((float)row, -cx_a, cy_b) : pix_center_p2(IS - 1 - row, IS, inv_is, pow2); };
    // stage 1's reject as signed line distances: d_k = w_k * h_k (h_k = height of vertex k over its opposite edge) is linear in
    // the pixel, so the coefficients are scaled once per face and a pixel costs six fmas, one v_min3 and one compare:
    // "some d_k < -sqrt(thr_pad)" == certainly_far (conservative either way: 5 % slack on thr, fused vs unfused ~1e-7)
    float ld[9];
    const float far_t = A.far_t;                 // -sqrtf(thr_pad)
    if (OPT_BWD_S1 && use_far) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            // (v_sqrt_f32, 1 ulp, instead of the 14-instruction IEEE expansion per height: the reject carries 5 % slack on thr,
            // and flag 16 guarantees a well-scaled argument)
            const float h = __builtin_amdgcn_sqrtf(rec[R_HK2 + k]);
            ld[3 * k] = rec[R_INV + 3 * k] * h; ld[3 * k + 1] = rec[R_INV + 3 * k + 1] * h; ld[3 * k + 2] = rec[R_INV + 3 * k + 2] * h;
        }
    }
    const float tie_scale = (flags & 16) ? near_tie_scale(rec[R_HK2], rec[R_HK2 + 1], rec[R_HK2 + 2]) : 0.f;     // (sr_device.h: near_tie)
    // K.cu:599: a fragment the forward pass depth-culled gets no gradient.  A well-conditioned face whose vertex depths lie
    // strictly inside (near, far) cannot be culled anywhere (its clipped barycentrics are >= 0 and sum to 1 within 1e-4, so the
    // interpolated depth stays within the vertex range up to rounding): one test per face instead of one per fragment
    bool depth_safe = false;
    if (OPT_BWD_MATH && (flags & 16)) {
        const float z0 = rec[R_FACE + 2], z1 = rec[R_FACE + 5], z2 = rec[R_FACE + 8];
        const float zlo = fminf(fminf(z0, z1), z2), zhi = fmaxf(fmaxf(z0, z1), z2);
        depth_safe = zlo > 0.f && zlo * (1.f - 1e-4f) > A.near && zhi * (1.f + 1e-4f) < A.far;
    }
    // (the ten pixel planes a face reads go through plain 64-bit addressed loads: buffer descriptors with the plane offset as a
    // scalar -- one 32-bit byte offset per pixel instead of an address computation per load -- were measured 1.5 % SLOWER,
    // profiles/r04_opt_ab.txt)
    auto ld_plane = [&](const float* base, int nplanes, int plane, int pn_) -> float {
#if defined(LASR_BWD_ABL) && LASR_BWD_ABL == 3       // measurement build: everything but the pixel-plane loads
        return 0.25f + 1e-3f * (float)(pn_ & 255) + 0.01f * (float)plane;
#endif
        return base[((size_t)bn * nplanes + plane) * P + pn_];
    };

    // lane -> (row, col) inside the bbox, advanced incrementally (one division per face)
    int r = 0, c = 0;
#if LASR_BWD_DIVREC
    // (the setup kernel left the divisions in the record: rec[15] = m | dr << 17 with lane / bw == (lane * m) >> 16, sr_device.h)
    const int walk = __float_as_int(rec[15]);
    const int dr = empty ? 0 : walk >> 17, dc = empty ? 0 : 64 - dr * bw;
    if (!empty) { r = (lane * (walk & 0x1ffff)) >> 16; c = lane - r * bw; }
#else
    const int dr = empty ? 0 : 64 / bw, dc = empty ? 0 : 64 - dr * bw;
    if (!empty) { r = lane / bw; c = lane - r * bw; }
#endif

    int head = 0, tail = 0, base = 0;            // wave-uniform ring state
    while (true) {
        if (base < npx) {
            // ---------------- stage 1
            const int xi = x0 + c, row = r0 + r;
            const bool in_range = base + lane < npx;
            c += dc; r += dr;
            if (c >= bw) { c -= bw; r += 1; }
            base += 64;
            bool keep = in_range;                              // every pixel of the rect passes the bbox test
            if (OPT_BWD_S1) {
                if (use_far) {
                    const float xp = __builtin_fmaf((float)xi, cx_a, cx_b), yp = __builtin_fmaf((float)row, -cx_a, cy_b);
                    const float d0 = __builtin_fmaf(ld[0], xp, __builtin_fmaf(ld[1], yp, ld[2]));
                    const float d1 = __builtin_fmaf(ld[3], xp, __builtin_fmaf(ld[4], yp, ld[5]));
                    const float d2 = __builtin_fmaf(ld[6], xp, __builtin_fmaf(ld[7], yp, ld[8]));
                    keep = in_range && !(fminf(fminf(d0, d1), d2) < far_t);
                }
            } else if (in_range && use_far) {
                float w0, w1, w2;
                barycentric(rec, pix_center_p2(xi, IS, inv_is, pow2), pix_center_p2(IS - 1 - row, IS, inv_is, pow2), w0, w1, w2);
                keep = !certainly_far(rec, w0, w1, w2, thr_pad);
            }
            const unsigned long long mask = wave_mask(keep);
            if (keep) ring[(tail + bits_below_lane(mask)) & (QCAP - 1)] = (unsigned)xi | ((unsigned)row << 16);
            tail += __popcll(mask);
        }
        const int avail = tail - head;
        if (!(avail >= 64 || (base >= npx && avail > 0))) {
            if (base >= npx) break;
            continue;
        }
        // ---------------- stage 2 on up to 64 queued pixels
        __builtin_amdgcn_wave_barrier();
        const bool active = lane < avail;
        const unsigned int packed = ring[(head + lane) & (QCAP - 1)];
        head += min(avail, 64);
#if defined(LASR_BWD_ABL) && LASR_BWD_ABL == 1       // measurement build: prologue + stage 1 + ring + epilogue only
        gv[0] += (float)packed;
        continue;
#endif
        if (!active) continue;
        const int xi = packed & 0xffff, row = packed >> 16;
        const int pn = row * IS + xi;
        const float xp = centre_x(xi), yp = centre_y(row);

        float w0, w1, w2;
        Frag fr;
#if LASR_BWD_ONE
        // (wave-uniform choice: the face's flags)
        if (LASR_FAST && (flags & 16)) {
#if LASR_BWD_LDSREC
            if (!fragment_one(lrec, A.thr, A.sigma, xp, yp, w0, w1, w2, fr, tie_scale)) continue;
#else
            if (!fragment_one(rec, A.thr, A.sigma, xp, yp, w0, w1, w2, fr, tie_scale)) continue;
#endif
        } else
#endif
        if (!fragment<FM, cptr_t, (OPT_BWD_MATH && LASR_FAST)>(rec, m.dist, A.thr, A.sigma, xp, yp, w0, w1, w2, fr)) continue;
        const float D = fr.D;
#if defined(LASR_BWD_ABL) && LASR_BWD_ABL == 2       // measurement build: + the distance code, nothing after it (no plane loads)
        gv[0] += D + fr.dx + fr.dy + fr.t0 + fr.t1 + fr.t2 + fr.sign;
        continue;
#endif

        // alpha path (K.cu:583-593); hard alpha: the reference still adds g_alpha into C
        float Ca = ld_plane(gcolors, NCH + 1, NCH, pn);
        if (m.alpha == 1) Ca = div_<FM>(Ca, (float)A.F);
        else if (m.alpha == 2) {
            const float a_out = ld_plane(colors, NCH + 1, NCH, pn);
            Ca *= div_<FM>(1 - a_out, fmaxf(1 - D, 1e-6f));
        }
        float C = Ca;

        const float u0 = w0, u1 = w1, u2 = w2;       // unclipped barycentrics (w0 of K.cu:596)
        // surface sampling picks a texel from (int)(w * res): keep the exact division there
        if (vertex_tex) clip_normalise<FM, (OPT_BWD_MATH && LASR_FAST)>(w0, w1, w2); else clip_normalise<false>(w0, w1, w2);
        // (the record holds the correctly rounded 1 / z_k: no v_rcp per fragment)
        const float iz0r = rec[R_IZ + 0], iz1r = rec[R_IZ + 1], iz2r = rec[R_IZ + 2];
        const float q0 = w0 * iz0r, q1 = w1 * iz1r, q2 = w2 * iz2r;
        const float zp = OPT_BWD_MATH && LASR_FAST ? __builtin_amdgcn_rcpf(q0 + q1 + q2) : depth_at<FM>(rec, w0, w1, w2);
        if (!depth_safe) {
            // K.cu:599: no gradient at all for a fragment the forward pass depth-culled.  The fast-math depth decides unless it
            // lies within 1e-4 relative of a plane; then the forward's own arithmetic is re-run so both passes agree.
            float zc = zp;
            const float tol = 1e-4f * fabsf(zp);
            if (fabsf(zp - A.near) <= tol || fabsf(zp - A.far) <= tol) zc = depth_forward_exact(rec, xp, yp);
            if (zc < A.near || zc > A.far) continue;
        }

        float gz0 = 0, gz1 = 0, gz2 = 0;
        if (m.rgb == 0) {
            if ((float)fn == ld_plane(aggrs, 2, 1, pn)) {       // K.cu:603
                float g[NCH];
#pragma unroll
                for (int k = 0; k < NCH; k++) g[k] = ld_plane(gcolors, NCH + 1, k, pn);
                if (vertex_tex) {
#pragma unroll
                    for (int k = 0; k < NCH; k++) {
                        gt[k] += w0 * g[k]; gt[NCH + k] += w1 * g[k]; gt[2 * NCH + k] += w2 * g[k];
                    }
                } else {
                    const int j = surface_texel(w0, w1, A.res);
                    if (j >= 0 && j < A.T) {   // the reference only credits texels j < T (K.cu:605)
                        float* gtp = gtex + (size_t)gw * A.T * 3 + 3 * j;
                        atomicAdd(gtp + 0, g[0]); atomicAdd(gtp + 1, g[1]); atomicAdd(gtp + 2, g[2]);
                    }
                }
            }
        } else if (front || m.double_side) {                                 // K.cu:611-640
            const float ssum = ld_plane(aggrs, 2, 0, pn);
            const float smax = ld_plane(aggrs, 2, 1, pn);
            const float zn = div_<FM>(A.far - zp, A.far - A.near);
            const float sm = div_<FM>(D * exp_<FM>(div_<FM>(zn - smax, A.gamma)), ssum);
            float g[NCH];
#pragma unroll
            for (int k = 0; k < NCH; k++) g[k] = ld_plane(gcolors, NCH + 1, k, pn);
            if (vertex_tex) {
#pragma unroll
                for (int k = 0; k < NCH; k++) {
                    gt[k] += sm * (w0 * g[k]); gt[NCH + k] += sm * (w1 * g[k]); gt[2 * NCH + k] += sm * (w2 * g[k]);
                }
            } else {
                const int j = surface_texel(w0, w1, A.res);
                if (j >= 0 && j < A.T) {       // K.cu:620
                    float* gtp = gtex + (size_t)gw * A.T * 3 + 3 * j;
                    atomicAdd(gtp + 0, sm * g[0]); atomicAdd(gtp + 1, sm * g[1]); atomicAdd(gtp + 2, sm * g[2]);
                }
            }
            float Crgb = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; k++)
                Crgb += g[k] * (sample_colour(tex, w0, w1, w2, A.res, k, m.tex, lim, NCH) - ld_plane(colors, NCH + 1, k, pn));
            Crgb *= sm;
            C += div_<FM>(Crgb, D);
            const float Cz = div_<FM>(div_<FM>(Crgb, A.gamma), A.near - A.far) * zp * zp;
            if (OPT_BWD_MATH && LASR_FAST) {
                gz0 = Cz * q0 * iz0r; gz1 = Cz * q1 * iz1r; gz2 = Cz * q2 * iz2r;       // q_k = w_k / z_k
            } else {
                const float iz0 = __builtin_amdgcn_rcpf(rec[R_FACE + 2]), iz1 = __builtin_amdgcn_rcpf(rec[R_FACE + 5]), iz2 = __builtin_amdgcn_rcpf(rec[R_FACE + 8]);
                gz0 = Cz * w0 * iz0 * iz0;
                gz1 = Cz * w1 * iz1 * iz1;
                gz2 = Cz * w2 * iz2 * iz2;
            }
        }

        C *= div_<FM>(D * (1 - D), A.sigma);                                  // K.cu:644
        float gx0 = 0, gy0 = 0, gx1 = 0, gy1 = 0, gx2 = 0, gy2 = 0;
        if (m.dist == 1) {                                                    // K.cu:161-175
            const float t0 = fr.t0, t1 = fr.t1, t2 = fr.t2;
            const int p = t0 > t1 ? (t1 > t2 ? 2 : 1) : (t0 > t2 ? 2 : 0);
            const float ipx = rec[R_INV + 3 * p + 0], ipy = rec[R_INV + 3 * p + 1];   // divergent gather: cold path
            const float dis = fr.dis;
            const float sc = 2.f * sqrtf(fabsf(dis));
            float gxy[3][2];
#pragma unroll
            for (int l = 0; l < 2; l++) {
                const float ipl = l == 0 ? ipx : ipy;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    float acc = 0.f;
                    acc += -ipl * rec[R_INV + 3 * k + 0] * xp;
                    acc += -ipl * rec[R_INV + 3 * k + 1] * yp;
                    acc += -ipl * rec[R_INV + 3 * k + 2] * 1.f;
                    gxy[k][l] = acc * C * sc;
                }
            }
            gx0 = gxy[0][0]; gy0 = gxy[0][1]; gx1 = gxy[1][0]; gy1 = gxy[1][1]; gx2 = gxy[2][0]; gy2 = gxy[2][1];
        } else if (m.dist == 2) {                                             // K.cu:649-655
            const float k2 = 2 * fr.sign * C;
            gx0 = k2 * (fr.t0 + u0) * fr.dx; gy0 = k2 * (fr.t0 + u0) * fr.dy;
            gx1 = k2 * (fr.t1 + u1) * fr.dx; gy1 = k2 * (fr.t1 + u1) * fr.dy;
            gx2 = k2 * (fr.t2 + u2) * fr.dx; gy2 = k2 * (fr.t2 + u2) * fr.dy;
        }
        gv[0] += gx0; gv[1] += gy0; gv[2] += gz0;
        gv[3] += gx1; gv[4] += gy1; gv[5] += gz1;
        gv[6] += gx2; gv[7] += gy2; gv[8] += gz2;
    }

    // one wave reduction per face (sr_device.h: wave_reduce18), then a plain, non-atomic accumulate: this wave
    // owns the face.  Lanes 15/31/47/63 each end up with the totals of up to five components.
    float v18[18], red[5];
    const int lane_row = lane >> 4;
    const int sub = lane_row == 0 ? 0 : lane_row == 1 ? 2 : lane_row == 2 ? 1 : 3;   // component offset inside a register
    float* gf = gfaces + (size_t)gw * 9;
    float* gtp = gtex + (size_t)gw * 3 * NCH;
    // pass 0: 9 face components + the first 9 attribute components; pass 1 (NCH = 6, 9): attribute components 9..17 and,
    // for NCH = 9, 18..26 in the second half of the 18-wide reduction
#pragma unroll
    for (int pass = 0; pass < (NCH > 3 ? 2 : 1); pass++) {
#pragma unroll
        for (int k = 0; k < 9; k++) {
            v18[k] = pass == 0 ? gv[k] : (vertex_tex ? gt[9 + k] : 0.f);
            v18[9 + k] = pass == 0 ? (vertex_tex ? gt[k] : 0.f) : (NCH == 9 && vertex_tex ? gt[THIRD + k] : 0.f);
        }
        wave_reduce18(v18, red);
        if ((lane & 15) == 15) {
#pragma unroll
            for (int i = 0; i < 5; i++) {
                if (i == 4 && (lane_row & 1)) continue;                          // register 4 only carries v[16], v[17]
                const int comp = i == 4 ? 16 + (lane_row >> 1) : 4 * i + sub;
                // this wave owns the face: a plain read-modify-write, or a plain store when the caller did not zero the buffers
                // (LASR_SR_GRADS_OVERWRITE; vertex attributes only -- surface texels are credited with atomics above)
                const bool ow = A.overwrite_grads && vertex_tex;
                if (pass == 0) {
                    if (comp < 9) gf[comp] = ow ? red[i] : gf[comp] + red[i];
                    else if (vertex_tex) gtp[comp - 9] = ow ? red[i] : gtp[comp - 9] + red[i];
                } else if ((comp < 9 || NCH == 9) && vertex_tex) gtp[9 + comp] = ow ? red[i] : gtp[9 + comp] + red[i];
            }
        }
    }
}

}  // namespace lasr
