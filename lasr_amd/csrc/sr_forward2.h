// sr_forward2.h -- two-phase forward raster kernel for LASR's mode combination (euclidean / softmax / prod / vertex
// attributes / double sided), included by sr_raster.hip.
//
// Why: in the one-phase kernel (sr_forward_kernel) a wave walks its face list one wave-uniform face at a time and every
// lane = one pixel of an 8x8 quadrant runs the whole fragment code; a face's survivors in a quadrant are a blob of ~26
// pixels, so ~60 % of the issued lane slots idle through ~230 VALU instructions (PMC: SQ_THREAD_CYCLES_VALU /
// (64 * SQ_ACTIVE_INST_VALU) = 0.44, profiles/r02a_pmc.txt), and every iteration starts with a dependent scalar load of the
// 176-B record (SQ_WAIT_ANY = 48 % of the wave cycles).  Here the work of a wave is split by cost:
//
//   cheap  (lanes = the 64 pixels of the quadrant, face wave-uniform, 15 record dwords through the scalar cache):
//          exact pixel-rect test, barycentrics, inside/outside class, conservative line-distance reject.  A surviving
//          (pixel, face) pair is appended to the wave's INSIDE or OUTSIDE entry list in LDS, and the pixel's lane notes the
//          entry index under its own running rank: refs[rank][pixel].  Faces are consumed in index order, so a pixel's
//          ranks are in face-index order.
//   heavy  (lanes = 64 consecutive ENTRIES of one class: dense, and branch-uniform on the inside/outside split that
//          dominates the distance code): record gathered per lane with vector loads (neighbouring entries share a face, so
//          the 64 addresses of a load fall into 2-3 cache lines), the reference's distance / sigmoid / clip / depth /
//          attribute arithmetic unchanged, result {D, zn, attributes} written back to the entry's slot.
//   fold   (lanes = pixels again): every pixel folds ITS results in rank order = face-index order: alpha product and the
//          online depth-softmax see exactly the reference's sequence, so the image is bit-identical to the one-phase kernel.
//
// A chunk ends after FOLD_RANKS faces (a pixel's rank never exceeds the faces consumed) or when an entry list could
// overflow; all LDS state is private to a wave, so after the workgroup-level binning there is no barrier.
#pragma once

namespace lasr {

constexpr int F2_CAP_OUT = 192;      // outside-class entries per chunk and wave
constexpr int F2_CAP_IN = 64;        // inside-class entries per chunk and wave
constexpr int F2_CAP = F2_CAP_OUT + F2_CAP_IN;
static_assert(F2_CAP <= 256, "entry indices are stored as bytes");
constexpr int F2_RANKS = 16;         // faces per chunk == max rank of a pixel inside a chunk

// one record field gathered per lane (plain global pointer: vector loads through L1)
typedef const float* __restrict__ lptr_t;

// K.cu:132-148 with a per-lane edge index a (the one-phase kernel dispatches to three compile-time variants and runs all
// of them when the lanes of a wave disagree).  Same operations, same order.
template <bool MKT>
__device__ __forceinline__ void edge_project_outside(lptr_t rec, int a, bool mk, float w0, float w1, float w2,
                                                     float& u0, float& u1, float& u2)
{
    const float e0 = rec[R_E + 3 * a + 0], e1 = rec[R_E + 3 * a + 1], e2 = rec[R_E + 3 * a + 2];
    const float eb = a == 0 ? e1 : (a == 1 ? e2 : e0);                 // e[a][(a + 1) % 3]
    const float den = rec[R_DEN + a];
    const float num = w0 * e0 + w1 * e1 + w2 * e2 - eb;
    float ta;
    if (MKT) ta = mk ? div_by_recip(num, den, rec[R_IDEN + a]) : num / den;
    else ta = num / den;
    float tb = 1 - ta;
    ta = fminf(fmaxf(ta, 0.f), 1.f);
    tb = fminf(fmaxf(tb, 0.f), 1.f);
    const float t0 = a == 0 ? ta : (a == 1 ? 0.f : tb);
    const float t1 = a == 0 ? tb : (a == 1 ? ta : 0.f);
    const float t2 = a == 0 ? 0.f : (a == 1 ? tb : ta);
    u0 = t0 - w0; u1 = t1 - w1; u2 = t2 - w2;
}

// K.cu:96-131: which edge an outside pixel projects to
__device__ __forceinline__ int outside_edge(int flags, float xp, float yp, float x0, float y0, float x1, float y1,
                                            float x2, float y2, float w0, float w1, float w2)
{
    int a = -1;
    if (w1 <= 0 && w2 <= 0) {
        a = 0;
        if ((flags & 1) && (xp - x0) * (x2 - x0) + (yp - y0) * (y2 - y0) > 0) a = 2;
    } else if (w2 <= 0 && w0 <= 0) {
        a = 1;
        if ((flags & 2) && (xp - x1) * (x0 - x1) + (yp - y1) * (y0 - y1) > 0) a = 0;
    } else if (w0 <= 0 && w1 <= 0) {
        a = 2;
        if ((flags & 4) && (xp - x2) * (x1 - x2) + (yp - y2) * (y1 - y2) > 0) a = 1;
    } else if (w0 <= 0) a = 1;
    else if (w1 <= 0) a = 2;
    else if (w2 <= 0) a = 0;
    return a < 0 ? 0 : a;           // the reference indexes [-1] here (UB); pinned to edge 0 like the oracle
}

// tail shared by both classes: fragment probability -> clip/normalise -> depth -> attributes; K.cu:397-447 minus the
// running-state updates, which the fold applies in face order.  Returns D with the sign bit set when the fragment takes
// no part in the colour blend (depth outside [near, far], K.cu:424) -- its alpha contribution stays.
template <int NCH, bool RX>
__device__ __forceinline__ void heavy_tail(const RasterArgs& A, const UniRecip& U, lptr_t rec, lptr_t tex, bool mk,
                                           float sdis /* -sign * dis */, float w0, float w1, float w2,
                                           float& D_out, float& zn_out, float (&col)[NCH])
{
    float D;
    if (RX) D = __builtin_amdgcn_rcpf(1.f + __expf(sdis * U.inv_sigma));
    else D = sigmoid_neg_<false>(mk ? div_by_recip(sdis, A.sigma, U.inv_sigma) : sdis / A.sigma);
    float c0 = w0, c1 = w1, c2 = w2;
    clip_normalise<RX>(c0, c1, c2);
    float zp;
    if (RX) zp = __builtin_amdgcn_rcpf(c0 * rec[R_IZ + 0] + c1 * rec[R_IZ + 1] + c2 * rec[R_IZ + 2]);
    else if (mk) zp = 1.f / (div_by_recip(c0, rec[2], rec[R_IZ + 0]) + div_by_recip(c1, rec[5], rec[R_IZ + 1]) +
                             div_by_recip(c2, rec[8], rec[R_IZ + 2]));
    else zp = 1.f / (c0 / rec[2] + c1 / rec[5] + c2 / rec[8]);
    const bool culled = zp < A.near || zp > A.far;
    const float fmn = A.far - A.near;
    zn_out = RX ? (A.far - zp) * U.inv_fmn : (mk ? div_by_recip(A.far - zp, fmn, U.inv_fmn) : (A.far - zp) / fmn);
#pragma unroll
    for (int k = 0; k < NCH; k++) col[k] = c0 * tex[k] + c1 * tex[NCH + k] + c2 * tex[2 * NCH + k];
    D_out = culled ? __uint_as_float(__float_as_uint(D) | 0x80000000u) : D;
}

template <int NCH, bool RX>
__global__ __launch_bounds__(256) void sr_forward2_kernel(RasterArgs A, float* __restrict__ aggrs,
                                                          float* __restrict__ colors)
{
    __shared__ unsigned short s_all[LIST_CAP];                 // faces whose pixel rect touches the 16x16 tile, index order
    __shared__ int s_wcnt[2][4];
    __shared__ unsigned short s_blk[4][64];                    // per wave: one level-2 block of face ids
    __shared__ unsigned char s_refs[4][F2_RANKS * 64];         // per wave: refs[rank][pixel] -> entry index (< F2_CAP = 256)
    __shared__ float s_res[4][(2 + NCH) * F2_CAP];             // per wave: D | zn | attributes, SoA over the entry index
                                                               // (an entry's packed (pixel, face) word lives in its D slot
                                                               //  until the heavy stage replaces it with the result)
    if (A.near_far_dev) { A.near = A.near_far_dev[0]; A.far = A.near_far_dev[1]; }
    const int IS = A.IS, P = IS * IS;
    const int tiles_x = (IS + TILE - 1) / TILE;
    const int tiles = tiles_x * tiles_x;
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int bn = blk / tiles;
    const int tl = blk - bn * tiles;
    const int ty = tl / tiles_x, tx = tl - ty * tiles_x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int qx0 = tx * TILE + (wave & 1) * 8, qy0 = ty * TILE + (wave >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool valid = px < IS && py < IS;
    const int pn = py * IS + px;
    const float xp = pix_center(px, IS);
    const float yp = pix_center(IS - 1 - py, IS);

    PixState<NCH> s;
    s.a = 1.f;
    s.fbest = -1;
    s.ssum = expf(A.eps / A.gamma);
    s.smax = A.eps;
#pragma unroll
    for (int k = 0; k < NCH; k++) {
        const float bg = valid ? colors[((size_t)bn * (NCH + 1) + k) * P + pn] : 1.f;
        s.c[k] = bg * s.ssum;
    }

    const short4* __restrict__ rects = A.rects + (size_t)bn * A.F;
    const float* __restrict__ recs = A.recs + (size_t)bn * A.F * REC;
    const float* __restrict__ texs = A.textures + (size_t)bn * A.F * 3 * NCH;
    UniRecip U;
    U.inv_sigma = 1.f / A.sigma; U.inv_gamma = 1.f / A.gamma; U.inv_fmn = 1.f / (A.far - A.near);
    U.ok = recip_safe(A.sigma) && recip_safe(A.gamma) && recip_safe(A.far - A.near);
    const int tX0 = tx * TILE, tX1 = tX0 + TILE - 1, tY0 = ty * TILE, tY1 = tY0 + TILE - 1;
    const float thr_pad = A.thr * 1.05f, thr_pad2 = A.thr * 1.10f;
    const float q_xlo = pix_center(qx0, IS), q_xhi = pix_center(min(qx0 + 7, IS - 1), IS);
    const float q_yhi = pix_center(IS - 1 - qy0, IS), q_ylo = pix_center(IS - 1 - min(qy0 + 7, IS - 1), IS);
    const float inv_is = 1.f / (float)IS;
    const bool pow2 = (IS & (IS - 1)) == 0;

    unsigned short* const blkbuf = s_blk[wave];
    unsigned char* const refs = s_refs[wave];
    float* const resD = s_res[wave];
    float* const resZ = resD + F2_CAP;
    float* const resC = resD + 2 * F2_CAP;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    for (int base = 0; base < A.F; base += LIST_CAP) {          // one round unless F > LIST_CAP
        const int end = min(base + LIST_CAP, A.F);
        if (base > 0) __syncthreads();
        // ---- level 1 (workgroup): ordered compaction of the faces whose rect touches the tile (as the one-phase kernel)
        int count = 0, flip = 0;
        for (int c = base; c < end; c += 256, flip ^= 1) {
            const int f = c + tid;
            bool hit = false;
            if (f < end) {
                const short4 q = rects[f];
                hit = !(q.x > tX1 || q.y < tX0 || q.z > tY1 || q.w < tY0);
            }
            const unsigned long long mask = __ballot(hit);
            if (lane == 0) s_wcnt[flip][wave] = __popcll(mask);
            __syncthreads();
            const int c0 = s_wcnt[flip][0], c1 = s_wcnt[flip][1], c2 = s_wcnt[flip][2], c3 = s_wcnt[flip][3];
            const int before = wave == 0 ? 0 : wave == 1 ? c0 : wave == 2 ? c0 + c1 : c0 + c1 + c2;
            if (hit) s_all[count + before + __popcll(mask & lt_mask)] = (unsigned short)(f - base);
            count += c0 + c1 + c2 + c3;
        }
        __syncthreads();
        count = __builtin_amdgcn_readfirstlane(count);

        // ---- per wave from here on: stream of faces -> chunks of (cheap*, heavy, fold)
        int i0 = 0;               // next level-2 block of s_all
        int n_blk = 0, j = 0;     // current block: faces in `myfaces` (lane i = i-th face), next to consume
        int myfaces = 0;
        bool more = true;
        while (more) {
            int n_out = 0, n_in = 0, nfaces = 0, cnt = 0;
            // ================= cheap: classify (pixel, face) pairs until the chunk is full or the stream ends
            while (true) {
                if (j == n_blk) {
                    if (i0 >= count) { more = false; break; }
                    // level 2: 64 list entries, keep those that can touch this quadrant (rect overlap + corner cull)
                    bool hit = false;
                    int e = 0;
                    if (i0 + lane < count) {
                        e = s_all[i0 + lane];
                        const short4 q = rects[base + e];
                        hit = !(q.x > qx0 + 7 || q.y < qx0 || q.z > qy0 + 7 || q.w < qy0);
                        if (hit) {
                            const float* R = recs + (size_t)(base + e) * REC;
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
                    }
                    const unsigned long long mask = __ballot(hit);
                    __builtin_amdgcn_wave_barrier();
                    if (hit) blkbuf[__popcll(mask & lt_mask)] = (unsigned short)e;
                    __builtin_amdgcn_wave_barrier();
                    n_blk = __popcll(mask);
                    myfaces = lane < n_blk ? (int)blkbuf[lane] : 0;
                    j = 0;
                    i0 += 64;
                    continue;
                }
                if (nfaces == F2_RANKS || n_out + 64 > F2_CAP_OUT || n_in + 64 > F2_CAP_IN) break;
                const int fn = base + __builtin_amdgcn_readlane(myfaces, j);
                j++;
                nfaces++;
                const cptr_t rec = as_const(recs + (size_t)fn * REC);
                const int rx = __float_as_int(rec[R_BB + 0]), ry = __float_as_int(rec[R_BB + 1]);
                const bool cand = valid && px >= (int)(short)(rx & 0xffff) && px <= (rx >> 16) &&
                                  py >= (int)(short)(ry & 0xffff) && py <= (ry >> 16);
                float w0, w1, w2;
                barycentric(rec, xp, yp, w0, w1, w2);
                const bool inside = w0 > 0 && w1 > 0 && w2 > 0 && w0 < 1 && w1 < 1 && w2 < 1;      // K.cu:66
                bool far = false;
                if (__float_as_int(rec[R_FLAGS]) & 16) far = certainly_far(rec, w0, w1, w2, thr_pad);   // never true inside
                const bool surv = cand && !far;
                const unsigned long long b_in = __ballot(surv && inside), b_out = __ballot(surv && !inside);
                if (surv) {
                    const int idx = inside ? F2_CAP_OUT + n_in + __popcll(b_in & lt_mask) : n_out + __popcll(b_out & lt_mask);
                    resD[idx] = __uint_as_float((unsigned)lane | ((unsigned)fn << 6));
                    refs[cnt * 64 + lane] = (unsigned char)idx;
                    cnt++;
                }
                n_in += __popcll(b_in);
                n_out += __popcll(b_out);
            }
            if (n_out + n_in == 0) continue;
            __builtin_amdgcn_wave_barrier();

            // ================= heavy: dense batches, one class at a time (cls 0: outside entries, 1: inside entries)
#pragma unroll 1
            for (int cls = 0; cls < 2; cls++) {
                const int n = cls ? n_in : n_out;
                const int off = cls ? F2_CAP_OUT : 0;
#pragma unroll 1
                for (int b = 0; b < n; b += 64) {
                    if (b + lane >= n) continue;
                    const int idx = off + b + lane;
                    const unsigned packed = __float_as_uint(resD[idx]);
                    const int lp = packed & 63, fn = (int)(packed >> 6);
                    const float exp_ = pix_center_p2(qx0 + (lp & 7), IS, inv_is, pow2);
                    const float eyp = pix_center_p2(IS - 1 - (qy0 + (lp >> 3)), IS, inv_is, pow2);
                    lptr_t rec = recs + (size_t)fn * REC;
                    lptr_t tex = texs + (size_t)fn * 3 * NCH;
                    const int flags = __float_as_int(rec[R_FLAGS]);
                    const bool mk = U.ok && (flags & 32);
                    float w0, w1, w2;
                    barycentric(rec, exp_, eyp, w0, w1, w2);
                    float sdis;                                                          // -sign * dis (K.cu:397,403)
                    bool skip = false;
                    if (cls == 0) {
                        const float x0 = rec[0], y0 = rec[1], x1 = rec[3], y1 = rec[4], x2 = rec[6], y2 = rec[7];
                        const int a = outside_edge(flags, exp_, eyp, x0, y0, x1, y1, x2, y2, w0, w1, w2);
                        float u0, u1, u2;
                        edge_project_outside<true>(rec, a, mk, w0, w1, w2, u0, u1, u2);
                        const float dx = u0 * x0 + u1 * x1 + u2 * x2;
                        const float dy = u0 * y0 + u1 * y1 + u2 * y2;
                        sdis = dx * dx + dy * dy;                                        // sign = -1
                        skip = sdis >= A.thr;                                            // K.cu:402
                    } else {
                        Frag fr;
                        if (mk) euclid<false, true, lptr_t>(rec, exp_, eyp, w0, w1, w2, fr);
                        else euclid<false, false, lptr_t>(rec, exp_, eyp, w0, w1, w2, fr);
                        sdis = -fr.sign * (fr.dx * fr.dx + fr.dy * fr.dy);
                    }
                    float D, zn, col[NCH];
                    heavy_tail<NCH, RX>(A, U, rec, tex, mk, sdis, w0, w1, w2, D, zn, col);
                    if (skip) { D = -0.f; zn = 0.f; }                                    // a no-op for the fold
                    resD[idx] = D;
                    resZ[idx] = zn;
#pragma unroll
                    for (int k = 0; k < NCH; k++) resC[k * F2_CAP + idx] = col[k];
                }
            }
            __builtin_amdgcn_wave_barrier();

            // ================= fold: every pixel applies its results in rank (= face index) order; K.cu:409-447
#pragma unroll 1
            for (int r = 0; r < nfaces; r++) {
                const bool act = r < cnt;
                if (__ballot(act) == 0ull) break;
                if (!act) continue;
                const int idx = refs[r * 64 + lane];
                const float Dm = resD[idx];
                const float zn = resZ[idx];
                const float D = fabsf(Dm);
                if (RX) s.a *= 1.f - D;
                else s.a = (float)((double)s.a * (1. - (double)D));
                if (__float_as_uint(Dm) & 0x80000000u) continue;                         // depth-culled or beyond the threshold
                const bool up = zn > s.smax;
                const float d = up ? s.smax - zn : zn - s.smax;
                // exp((smax - zn)/gamma) rescales the history when zn is the new maximum and the fragment's own weight is
                // exp(0) = 1 exactly; otherwise the history keeps weight 1 and the fragment gets exp((zn - smax)/gamma):
                // one exponential instead of the reference's two, identical values (exp_1ulp(0) == 1, x * 1 == x)
                const float E = RX ? __expf(d * U.inv_gamma)
                                   : exp_1ulp(U.ok ? div_by_recip(d, A.gamma, U.inv_gamma) : d / A.gamma);
                const float ED = E * D;
                if (up) {
                    s.smax = zn;
                    s.ssum = E * s.ssum + D;
#pragma unroll
                    for (int k = 0; k < NCH; k++) s.c[k] = E * s.c[k] + D * resC[k * F2_CAP + idx];
                } else {
                    s.ssum = s.ssum + ED;
#pragma unroll
                    for (int k = 0; k < NCH; k++) s.c[k] = s.c[k] + ED * resC[k * F2_CAP + idx];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    if (!valid) return;
    // ---- finalise (K.cu:458-482)
    colors[((size_t)bn * (NCH + 1) + NCH) * P + pn] = (float)(1. - (double)s.a);
#pragma unroll
    for (int k = 0; k < NCH; k++) colors[((size_t)bn * (NCH + 1) + k) * P + pn] = s.c[k] / s.ssum;
    aggrs[((size_t)bn * 2 + 0) * P + pn] = s.ssum;
    aggrs[((size_t)bn * 2 + 1) * P + pn] = s.smax;
}

}  // namespace lasr
