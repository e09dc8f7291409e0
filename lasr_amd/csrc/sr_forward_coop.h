// sr_forward_coop.h -- the forward raster kernel for SMALL launches: four or eight waves share one 8x8-pixel tile.
//
// With few frames per launch the forward pass is a latency problem, not a throughput problem: a wave that is alone on its SIMD
// issues one instruction every ~5 cycles whatever the instruction is, a list entry costs ~380 VALU + SALU instructions, and the
// busiest 8x8 quadrant of a LASR frame walks ~140 entries one after the other -- 0.2 ms for ONE frame while most of the chip
// idles (profiles/r03_pmc_sq_n4.txt: 13 cycles per VALU instruction per wave, 40 % of the wave time in s_waitcnt).  The pixels'
// face lists cannot be split between waves (the alpha product and the online depth-softmax see the faces in index order, and
// neither is associative in floating point), but most of an entry's work does not touch that state:
//
//   part I  (producers, ~250 instructions per entry): rect test, barycentrics, distance, sigmoid, clip / normalise, depth,
//           normalised depth, the interpolated attributes -- functions of (pixel, face) only;
//   part S  (consumer, ~40 instructions per entry):   alpha product, running maximum, the one exponential, the rescaled sums.
//
// Per step of seven list entries waves 1..3 evaluate part I of two entries each and wave 0 of one (eight-wave form: waves 1..7
// one entry each, wave 0 none; nine channels: waves 1..3 one entry each -- forward_impl has the measurements), into an LDS buffer [entry][field][pixel]; wave 0 then applies part S to the previous step's
// seven entries IN LIST ORDER while the others are already on the next step (two buffers, one workgroup barrier per step).  Every (pixel, face) pair goes through exactly the
// arithmetic of sr_forward_kernel's forward_face, in the same order per pixel: the output is bit-identical.  The host picks
// this kernel by launch size (forward_impl); large launches keep the one-wave-per-tile kernel, which spends fewer instructions
// per entry in total.
#pragma once

namespace lasr {

constexpr int COOP_CAP = 1024;        // list entries per round (u16 ids relative to the round's first face)

// NW waves per tile; per step waves 1..NW-1 evaluate part I of EPW entries each (slots (w-1)*EPW ..) and wave 0 of E0 (the last slots)
// Latency or throughput?  What decides is how many 8x8 tiles have work: below ~8 busy tiles per SIMD the chip is not full and
// the serial walks set the time (cooperative kernel), above it the total instruction count does (one wave per tile).  The
// host only knows frames x tiles; LASR frames are cropped around the object (dataloader/vidbase.py:105-135), so most tiles are
// busy there, while a small object leaves three quarters of them empty.  For launches in the range where that matters one
// wave estimates the busy tiles from the group rects the setup kernel has just written (lane = image: the union of its group
// rects is the mesh's pixel bounding box) and leaves its choice in the workspace; both candidates are launched and the one
// not chosen returns at its first instruction.
__global__ __launch_bounds__(64) void sr_choose_kernel(const short4* __restrict__ grects, int N, int G, int IS,
                                                       long long coop_max_tiles, int* __restrict__ choice)
{
    long long tiles = 0;
    for (int i = threadIdx.x; i < N; i += 64) {
        int x0 = 32767, x1 = -1, y0 = 32767, y1 = -1;
#pragma unroll 8
        for (int g = 0; g < G; g++) {                 // (an empty group rect is (32767, -1, 32767, -1): it changes nothing)
            const short4 q = grects[(size_t)i * G + g];
            x0 = min(x0, (int)q.x); x1 = max(x1, (int)q.y); y0 = min(y0, (int)q.z); y1 = max(y1, (int)q.w);
        }
        x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, IS - 1); y1 = min(y1, IS - 1);
        if (x0 <= x1 && y0 <= y1) tiles += (long long)((x1 >> 3) - (x0 >> 3) + 1) * ((y1 >> 3) - (y0 >> 3) + 1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tiles += __shfl_xor(tiles, o);
    if (threadIdx.x == 0) *choice = tiles <= coop_max_tiles ? CHOICE_COOP : CHOICE_ONE_WAVE;
}

// LDS of one cooperative tile: the ordered face list, the per-wave counts of the list building, the two hand-over buffers
template <int NCH, int NW, int EPW, int E0>
struct CoopLds {
    static constexpr int STEP = (NW - 1) * EPW + E0;
    static constexpr int FIELDS = 3 + NCH;              // flags, D, zn, NCH interpolated attributes
    float buf[2][STEP][FIELDS][64];
    int wcnt[2][NW];
    unsigned short list[COOP_CAP];
};

// The tile body as a device function: sr_forward_coop_kernel is a thin wrapper, sr_forward_mixed_kernel (sr_raster.hip) calls it
// for the crowded head of the launch's ordered tile table.
template <int NCH, int NW, int EPW, int E0>
__device__ __forceinline__ void coop_tile_body(RasterArgs A, float* __restrict__ aggrs, float* __restrict__ colors, int bn, int tx, int ty,
                                               CoopLds<NCH, NW, EPW, E0>& L)
{
    constexpr int COOP_STEP = (NW - 1) * EPW + E0;
    auto& s_list = L.list;
    auto& s_wcnt = L.wcnt;
    auto& s_buf = L.buf;
    const Modes m = Modes{2, 1, 2, 1, 1};               // LASR's configuration: euclidean, softmax, prod, vertex, double-sided
    if (A.near_far_dev) { A.near = A.near_far_dev[0]; A.far = A.near_far_dev[1]; }
    const int IS = A.IS, P = IS * IS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int qx0 = tx * 8, qy0 = ty * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool valid = px < IS && py < IS;
    const int pn = py * IS + px;
    const int pxy = valid ? px | (py << 16) : (int)0xfffefffeu;    // (lies in no rect, the empty one included: rect_has)

    PixState<NCH> s;                                    // lives in wave 0 only
    s.a = 1.f;
    s.fbest = -1;
    s.ssum = expf(A.eps / A.gamma); s.smax = A.eps;
#pragma unroll
    for (int k = 0; k < NCH; k++) {
        const float bg = A.use_bg ? A.bg[k] : (valid ? colors[((size_t)bn * (NCH + 1) + k) * P + pn] : 1.f);
        s.c[k] = bg * s.ssum;
    }

    // level 0: the groups of 64 consecutive faces whose union rect meets the tile (every wave evaluates the same test)
    const int G = groups_of(A.F);
    const short4* __restrict__ grects = A.grects + (size_t)bn * G;
    const int tX1 = qx0 + 7, tY1 = qy0 + 7;
    unsigned long long gmask;
    {
        bool t = false;
        if (lane < G) {
            const short4 q = grects[lane];
            t = !(q.x > tX1 || q.y < qx0 || q.z > tY1 || q.w < qy0);
        }
        gmask = wave_mask(t);
    }
    if (gmask != 0 || G > 64) {
    const float xp = pix_center(px, IS);
    const float yp = pix_center(IS - 1 - py, IS);
    const short4* __restrict__ rects = A.rects + (size_t)bn * A.F;
    const float* __restrict__ recs = A.recs + (size_t)bn * A.F * REC;
    const float* __restrict__ texs = A.textures + (size_t)bn * A.F * A.T * NCH;
    const int texstride = A.T * NCH;
    const UniRecip U = uni_recip(A);
    const int ok_bit = U.ok ? 32 : 0;
    const float fmn = A.far - A.near;
    const float thr_pad2 = A.thr * 1.10f;
    const float q_xlo = pix_center(qx0, IS), q_xhi = pix_center(min(qx0 + 7, IS - 1), IS);
    const float q_yhi = pix_center(IS - 1 - qy0, IS), q_ylo = pix_center(IS - 1 - min(qy0 + 7, IS - 1), IS);

    // rect overlap + the conservative corner cull of sr_forward_kernel (same faces contribute, fewer entries to walk)
    auto touches_tile = [&](int f) -> bool {
        const short4 q = rects[f];
        bool hit = !(q.x > tX1 || q.y < qx0 || q.z > tY1 || q.w < qy0);
        if (hit) {
            const float* R = recs + (size_t)f * REC;
            if (__float_as_int(R[R_FLAGS]) & 16) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float a = R[R_INV + 3 * k], b = R[R_INV + 3 * k + 1], c = R[R_INV + 3 * k + 2];
                    const float w00 = a * q_xlo + b * q_ylo + c, w01 = a * q_xhi + b * q_ylo + c;
                    const float w10 = a * q_xlo + b * q_yhi + c, w11 = a * q_xhi + b * q_yhi + c;
                    const float wmax = fmaxf(fmaxf(w00, w01), fmaxf(w10, w11));
                    if (wmax < 0.f && wmax * wmax * R[R_HK2 + k] > thr_pad2) hit = false;
                }
            }
        }
        return hit;
    };

    // part I of list entry e into slot `slot` of buffer `b` (the calling wave's 64 pixels)
    auto produce = [&](int b, int slot, int e, int base) {
        const int fn = __builtin_amdgcn_readfirstlane(base + (int)s_list[e]);
        const cptr_t rec = as_const(recs + (size_t)fn * REC);
        const cptr_t tex = as_const(texs + (size_t)fn * texstride);
#if LASR_PREFETCH       // the entry's other two record lines and its attribute line in the same round trip as the rect (sr_raster.hip)
        int pf1, pf2, pf3;
        {
            const float* rn = recs + (size_t)fn * REC;
            const float* tn = texs + (size_t)fn * texstride;
            asm volatile("s_load_dword %0, %3, 0x40\n\ts_load_dword %1, %3, 0x80\n\ts_load_dword %2, %4, 0x0"
                         : "=&s"(pf1), "=&s"(pf2), "=&s"(pf3) : "s"(rn), "s"(tn) : "memory");
        }
#endif
        const bool cand = rect_has(__float_as_int(rec[R_BB + 0]), __float_as_int(rec[R_BB + 1]), pxy);
#if LASR_PREFETCH
        asm volatile("s_waitcnt lgkmcnt(0)" : : "s"(pf1), "s"(pf2), "s"(pf3));
#endif
        float w0, w1, w2;
        barycentric(rec, xp, yp, w0, w1, w2);
        const bool mk = (__float_as_int(rec[R_FLAGS]) & ok_bit) != 0;     // wave-uniform; ok_bit = U.ok ? 32 : 0 (no branch on U.ok per entry)
        int fl = 0;
        if (cand) {
            Frag fr;
            const bool ok = mk ? fragment_w<false, true>(rec, m.dist, A.thr, A.sigma, xp, yp, w0, w1, w2, fr, U.inv_sigma)
                               : fragment_w<false, false>(rec, m.dist, A.thr, A.sigma, xp, yp, w0, w1, w2, fr, U.inv_sigma);
            if (ok) {
                fl = mk ? 5 : 1;
                s_buf[b][slot][1][lane] = fr.D;
                float c0 = w0, c1 = w1, c2 = w2;
                float zp;
                if (mk) { clip_normalise<false, OPT_MED3>(c0, c1, c2); zp = depth_at<false, true>(rec, c0, c1, c2); }
                else { clip_normalise<false>(c0, c1, c2); zp = depth_at<false, false>(rec, c0, c1, c2); }
                // (a tame record's NaN depth stands for the reference's 1 / 0 = inf: cut, see forward_face)
                if (mk && OPT_NOSCALE ? (zp >= A.near && zp <= A.far) : !(zp < A.near || zp > A.far)) {
                    fl |= 2;
                    s_buf[b][slot][2][lane] = mk ? div_by_recip(A.far - zp, fmn, U.inv_fmn) : (A.far - zp) / fmn;
#pragma unroll
                    for (int k = 0; k < NCH; k++)
                        s_buf[b][slot][3 + k][lane] = sample_colour(tex, c0, c1, c2, A.res, k, m.tex, 0, NCH);
                }
            }
        }
        s_buf[b][slot][0][lane] = __int_as_float(fl);
    };

    // part S of slot `slot` of buffer `b`: forward_face's state update, K.cu:409-446 (wave 0)
    auto consume = [&](int b, int slot) {
        const int fl = __float_as_int(s_buf[b][slot][0][lane]);
        if (fl & 1) {
            const float D = s_buf[b][slot][1][lane];
            s.a = (float)((double)s.a * (1. - (double)D));
            if (fl & 2) {
                const float zn = s_buf[b][slot][2][lane];
                const bool up = zn > s.smax;
                const float d = OPT_SOFTMAX ? -fabsf(zn - s.smax) : (up ? s.smax - zn : zn - s.smax);      // see forward_face
                const float E = exp_1ulp((fl & 4) ? div_by_recip(d, A.gamma, U.inv_gamma) : d / A.gamma);
                const float hist = up ? E : 1.f, wgt = up ? D : E * D;
                s.smax = OPT_SOFTMAX && (fl & 4) ? max_finite(zn, s.smax) : (up ? zn : s.smax);
                s.ssum = hist * s.ssum + wgt;
#pragma unroll
                for (int k = 0; k < NCH; k++) s.c[k] = hist * s.c[k] + wgt * s_buf[b][slot][3 + k][lane];
            }
        }
    };

    int g_next = 64, g_mask0 = 0;
    bool more = true;
    while (more) {                                    // one round unless the tile meets more than COOP_CAP - 64 NW faces
        // ---- ordered list of the faces that reach the tile: four touched groups per step, one per wave
        int count = 0, flip = 0, base = -1;
        for (;;) {
            if (gmask == 0) {
                if (g_next >= G) { more = false; break; }
                g_mask0 = g_next;
                bool t = false;
                if (g_next + lane < G) {
                    const short4 q = grects[g_next + lane];
                    t = !(q.x > tX1 || q.y < qx0 || q.z > tY1 || q.w < qy0);
                }
                gmask = wave_mask(t);
                g_next += 64;
                continue;
            }
            unsigned long long mm = gmask;
            int mine_g = -1, last_g = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                if (mm) {
                    const int bit = __builtin_ctzll(mm);
                    mm &= mm - 1;
                    if (k == wave) mine_g = g_mask0 + bit;
                    last_g = g_mask0 + bit;
                }
            }
            const int first_g = g_mask0 + __builtin_ctzll(gmask);
            if (base < 0) base = first_g * GROUP;
            if (count + NW * 64 > COOP_CAP || (last_g + 1) * GROUP - base > 65536) break;  // walk what we have, then continue
            gmask = mm;
            const int f = mine_g * GROUP + lane;
            const bool hit = mine_g >= 0 && f < A.F && touches_tile(f);
            const unsigned long long mask = wave_mask(hit);
            if (lane == 0) s_wcnt[flip][wave] = __popcll(mask);
            __syncthreads();
            int before = 0, all = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const int c = s_wcnt[flip][k];
                if (k < wave) before += c;
                all += c;
            }
            if (hit) s_list[count + before + bits_below_lane(mask)] = (unsigned short)(f - base);
            count += all;
            flip ^= 1;
        }
        if (base < 0) base = 0;
        __syncthreads();
        // ---- the walk, software-pipelined over steps of COOP_STEP entries: step k is produced while step k - 1 is consumed
#if defined(LASR_ABL) && LASR_ABL == 2              // measurement build: tile order + list building + stores, no walk
        count = 0;
#endif
        const int steps = (count + COOP_STEP - 1) / COOP_STEP;
        for (int k = 0; k <= steps; k++) {
            if (k < steps) {
                const int e0 = k * COOP_STEP, b = k & 1;
                if (wave == 0) {
#pragma unroll
                    for (int j = (NW - 1) * EPW; j < COOP_STEP; j++)
                        if (e0 + j < count) produce(b, j, e0 + j, base);
                } else {
#pragma unroll
                    for (int i = 0; i < EPW; i++) {
                        const int j = (wave - 1) * EPW + i;
                        if (e0 + j < count) produce(b, j, e0 + j, base);
                    }
                }
            }
            if (k > 0 && wave == 0) {
                const int e0 = (k - 1) * COOP_STEP, b = (k - 1) & 1;
                const int n = min(COOP_STEP, count - e0);
                for (int j = 0; j < n; j++) consume(b, j);
            }
            __syncthreads();
        }
    }
    }   // tile meets at least one group

    if (!valid || wave != 0) return;
#if defined(LASR_ABL) && LASR_ABL == 4              // measurement build: everything but the output stores
    if (s.a != 12345.678f || s.ssum != 3.25f) return;
#endif
    // ---- finalise (K.cu:458-482)
    colors[((size_t)bn * (NCH + 1) + NCH) * P + pn] = (float)(1. - (double)s.a);
#pragma unroll
    for (int k = 0; k < NCH; k++) colors[((size_t)bn * (NCH + 1) + k) * P + pn] = s.c[k] / s.ssum;
    aggrs[((size_t)bn * 2 + 0) * P + pn] = s.ssum;
    aggrs[((size_t)bn * 2 + 1) * P + pn] = s.smax;
}

template <int NCH, int NW = 4, int EPW = 2, int E0 = 1>
__global__ __launch_bounds__(NW * 64) void sr_forward_coop_kernel(RasterArgs A, float* __restrict__ aggrs,
                                                                  float* __restrict__ colors)
{
    __shared__ CoopLds<NCH, NW, EPW, E0> L;
    if (A.choice && chosen_kernel(A) != CHOICE_COOP) return;   // the launch was left to the device, which took the other kernel
    int bn, tx, ty;
    tile_of_block(blockIdx.x, gridDim.x, (A.IS + 7) / 8, bn, tx, ty, A.order);
    coop_tile_body<NCH, NW, EPW, E0>(A, aggrs, colors, bn, tx, ty, L);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Opt-in (LASR_SR_SEGMENTED, include/lasr_sr.h): the SAME small-launch problem attacked from the other side.  The cooperative
// kernel above keeps the reference's accumulation order, so its ordered part (wave 0: ~50 instructions per entry on a lone wave)
// stays a serial chain.  Alpha product and depth-softmax are, in exact arithmetic, symmetric in the fragments: a list split into
// NW index-ordered segments can be folded by NW waves independently -- each with the full per-entry arithmetic of
// sr_forward_kernel's forward_face, records through the scalar cache, no per-entry hand-over -- and the NW partial states
// (alpha product, running maximum, rescaled sums) merged in segment order at the end:
//     a = a_0 a_1 ...;   m = max_w m_w;   sum = SUM_w sum_w exp((m_w - m) / gamma);   c likewise.
// Only the rounding sequence changes (measured <= 1e-6 on the image against the default path, tests/test_forward_segmented_gpu.py;
// the north-star bar is 1e-4); the default path stays bit-faithful to the reference's order.  Segment 0 carries the background
// term (K.cu:354-368); the others start from an empty state (sum 0, maximum -1e30: the first fragment's rescale is exp(-inf) = 0).
// Measured (profiles/experiments/README.md, round 4): forward 0.0535 -> 0.0445 ms at one frame, 0.0896 -> 0.0838 at four, slower
// from sixteen on -- a lone wave needs ~4500 cycles per list entry whichever way the entry is evaluated, so the host applies the
// flag only to launches of at most twice the eight-wave range.
template <int NCH, int NW>
__global__ __launch_bounds__(NW * 64) void sr_forward_seg_kernel(RasterArgs A, float* __restrict__ aggrs, float* __restrict__ colors)
{
    constexpr int FIELDS = 3 + NCH;                     // a, smax, ssum, NCH accumulators
    __shared__ unsigned short s_list[COOP_CAP];
    __shared__ int s_wcnt[2][NW];
    __shared__ float s_part[NW][FIELDS][64];

    if (A.choice && chosen_kernel(A) != CHOICE_COOP) return;
    const Modes m = Modes{2, 1, 2, 1, 1};
    if (A.near_far_dev) { A.near = A.near_far_dev[0]; A.far = A.near_far_dev[1]; }
    const int IS = A.IS, P = IS * IS;
    const int tiles_x = (IS + 7) / 8;
    int bn, tx, ty;
    tile_of_block(blockIdx.x, gridDim.x, tiles_x, bn, tx, ty, A.order);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int qx0 = tx * 8, qy0 = ty * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool valid = px < IS && py < IS;
    const int pn = py * IS + px;
    const int pxy = valid ? px | (py << 16) : (int)0xfffefffeu;

    PixState<NCH> s;
    s.a = 1.f;
    s.fbest = -1;
    if (wave == 0) {
        s.ssum = expf(A.eps / A.gamma); s.smax = A.eps;
#pragma unroll
        for (int k = 0; k < NCH; k++) {
            const float bg = A.use_bg ? A.bg[k] : (valid ? colors[((size_t)bn * (NCH + 1) + k) * P + pn] : 1.f);
            s.c[k] = bg * s.ssum;
        }
    } else {
        s.ssum = 0.f; s.smax = -1e30f;
#pragma unroll
        for (int k = 0; k < NCH; k++) s.c[k] = 0.f;
    }

    const int G = groups_of(A.F);
    const short4* __restrict__ grects = A.grects + (size_t)bn * G;
    const int tX1 = qx0 + 7, tY1 = qy0 + 7;
    unsigned long long gmask;
    {
        bool t = false;
        if (lane < G) {
            const short4 q = grects[lane];
            t = !(q.x > tX1 || q.y < qx0 || q.z > tY1 || q.w < qy0);
        }
        gmask = wave_mask(t);
    }
    if (gmask != 0 || G > 64) {
    const float xp = pix_center(px, IS);
    const float yp = pix_center(IS - 1 - py, IS);
    const short4* __restrict__ rects = A.rects + (size_t)bn * A.F;
    const float* __restrict__ recs = A.recs + (size_t)bn * A.F * REC;
    const float* __restrict__ texs = A.textures + (size_t)bn * A.F * A.T * NCH;
    const int texstride = A.T * NCH;
    const UniRecip U = uni_recip(A);
    const int ok_bit = U.ok ? 32 : 0;
    const float thr_pad2 = A.thr * 1.10f;
    const float q_xlo = pix_center(qx0, IS), q_xhi = pix_center(min(qx0 + 7, IS - 1), IS);
    const float q_yhi = pix_center(IS - 1 - qy0, IS), q_ylo = pix_center(IS - 1 - min(qy0 + 7, IS - 1), IS);

    auto touches_tile = [&](int f) -> bool {          // as in sr_forward_coop_kernel
        const short4 q = rects[f];
        bool hit = !(q.x > tX1 || q.y < qx0 || q.z > tY1 || q.w < qy0);
        if (hit) {
            const float* R = recs + (size_t)f * REC;
            if (__float_as_int(R[R_FLAGS]) & 16) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float a = R[R_INV + 3 * k], b = R[R_INV + 3 * k + 1], c = R[R_INV + 3 * k + 2];
                    const float w00 = a * q_xlo + b * q_ylo + c, w01 = a * q_xhi + b * q_ylo + c;
                    const float w10 = a * q_xlo + b * q_yhi + c, w11 = a * q_xhi + b * q_yhi + c;
                    const float wmax = fmaxf(fmaxf(w00, w01), fmaxf(w10, w11));
                    if (wmax < 0.f && wmax * wmax * R[R_HK2 + k] > thr_pad2) hit = false;
                }
            }
        }
        return hit;
    };

    int g_next = 64, g_mask0 = 0;
    bool more = true;
    while (more) {
        int count = 0, flip = 0, base = -1;
        for (;;) {                                      // the ordered list of the faces that reach the tile (as above)
            if (gmask == 0) {
                if (g_next >= G) { more = false; break; }
                g_mask0 = g_next;
                bool t = false;
                if (g_next + lane < G) {
                    const short4 q = grects[g_next + lane];
                    t = !(q.x > tX1 || q.y < qx0 || q.z > tY1 || q.w < qy0);
                }
                gmask = wave_mask(t);
                g_next += 64;
                continue;
            }
            unsigned long long mm = gmask;
            int mine_g = -1, last_g = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                if (mm) {
                    const int bit = __builtin_ctzll(mm);
                    mm &= mm - 1;
                    if (k == wave) mine_g = g_mask0 + bit;
                    last_g = g_mask0 + bit;
                }
            }
            const int first_g = g_mask0 + __builtin_ctzll(gmask);
            if (base < 0) base = first_g * GROUP;
            if (count + NW * 64 > COOP_CAP || (last_g + 1) * GROUP - base > 65536) break;
            gmask = mm;
            const int f = mine_g * GROUP + lane;
            const bool hit = mine_g >= 0 && f < A.F && touches_tile(f);
            const unsigned long long mask = wave_mask(hit);
            if (lane == 0) s_wcnt[flip][wave] = __popcll(mask);
            __syncthreads();
            int before = 0, all = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const int c = s_wcnt[flip][k];
                if (k < wave) before += c;
                all += c;
            }
            if (hit) s_list[count + before + bits_below_lane(mask)] = (unsigned short)(f - base);
            count += all;
            flip ^= 1;
        }
        if (base < 0) base = 0;
        __syncthreads();
        // ---- this wave's segment of the round: entries [e0, e1), in list order, straight into its own partial state
        const int per = (count + NW - 1) / NW;
        const int e0 = min(wave * per, count), e1 = min(e0 + per, count);
        for (int e = e0; e < e1; e++) {
            const int fn = __builtin_amdgcn_readfirstlane(base + (int)s_list[e]);
            const cptr_t rec = as_const(recs + (size_t)fn * REC);
            const bool cand = rect_has(__float_as_int(rec[R_BB + 0]), __float_as_int(rec[R_BB + 1]), pxy);
            float w0, w1, w2;
            barycentric(rec, xp, yp, w0, w1, w2);
            const cptr_t tex = as_const(texs + (size_t)fn * texstride);
            const bool mk = (__float_as_int(rec[R_FLAGS]) & ok_bit) != 0;
            if (cand) {
                if (mk) forward_face<true, true, NCH, false>(A, m, rec, tex, fn, 0, xp, yp, w0, w1, w2, s, U);
                else forward_face<true, false, NCH>(A, m, rec, tex, fn, 0, xp, yp, w0, w1, w2, s, U);
            }
        }
        __syncthreads();                                // the list is rebuilt by the next round
    }
    }   // tile meets at least one group

    // ---- merge the partial states in segment order (wave 0), then finalise as K.cu:458-482
    if (wave != 0) {
        s_part[wave][0][lane] = s.a; s_part[wave][1][lane] = s.smax; s_part[wave][2][lane] = s.ssum;
#pragma unroll
        for (int k = 0; k < NCH; k++) s_part[wave][3 + k][lane] = s.c[k];
    }
    __syncthreads();
    if (!valid || wave != 0) return;
    const float inv_gamma = 1.f / A.gamma;
#pragma unroll
    for (int w = 1; w < NW; w++) {
        const float sw = s_part[w][2][lane];
        s.a *= s_part[w][0][lane];
        if (sw > 0.f) {
            const float mw = s_part[w][1][lane];
            const float mx = fmaxf(s.smax, mw);
            const float e_a = __expf((s.smax - mx) * inv_gamma), e_b = __expf((mw - mx) * inv_gamma);
            s.ssum = s.ssum * e_a + sw * e_b;
#pragma unroll
            for (int k = 0; k < NCH; k++) s.c[k] = s.c[k] * e_a + s_part[w][3 + k][lane] * e_b;
            s.smax = mx;
        }
    }
    colors[((size_t)bn * (NCH + 1) + NCH) * P + pn] = (float)(1. - (double)s.a);
#pragma unroll
    for (int k = 0; k < NCH; k++) colors[((size_t)bn * (NCH + 1) + k) * P + pn] = s.c[k] / s.ssum;
    aggrs[((size_t)bn * 2 + 0) * P + pn] = s.ssum;
    aggrs[((size_t)bn * 2 + 1) * P + pn] = s.smax;
}

}  // namespace lasr
