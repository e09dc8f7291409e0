// sr_forward_pairs.h -- the forward raster kernel of LASR's mode combination for launches that fill the chip: every LANE walks
// the (pixel, face) pairs of its own pixel.
//
// The one-wave-per-tile kernel (sr_raster.hip: forward_tile_body) keeps the face index wave-uniform: every one of the ~290
// instructions of a list entry is issued for the 64 pixels of the tile while ~27 of them lie within the face's reach (41.5 % live
// lanes, profiles/r05_valu.json), and the record arrives as SGPR operands, which halve the issue rate of an fp32 instruction
// (profiles/r02_valu_issue.txt).  Here the face index is PER LANE:
//
//   list     four waves share a 16x16-pixel tile and build its ordered face list as before (group rects, pixel rects, corner cull);
//   stage    64 list entries at a time: their records (+ vertex attributes) are copied into LDS in a layout made for per-lane
//            gathers -- 16-byte quads grouped by use, each edge's constants in a block of its own (so the edge an outside pixel
//            projects to is ONE run-time offset, not three exec-masked code variants), the obtuse-corner test's operands
//            precomputed -- at an odd quad stride, so that 16 lanes reading the same quad of 16 different entries hit 16
//            different bank groups;
//   classify (wave-uniform entry, lanes = the wave's 8x8 pixels, record by LDS broadcast): exact integer rect test, barycentrics,
//            then one bit per (pixel, entry) in one of three per-lane 64-bit masks: INSIDE the face, OUTSIDE but not certainly
//            beyond the distance threshold (the backward's conservative line-distance reject, sr_device.h: certainly_far), or
//            SLOW (a record that is not tame: handled by the generic arithmetic, wave-uniform, at the end of the chunk);
//   balance  the lanes are ranked by their pair count (six ballots) and lane of rank r is partnered with rank 63 - r: the lighter
//            partner takes the upper half of the difference from the heavier one's outside mask and folds it into a private
//            partial state;
//   walk     every lane pops its own masks, inside pairs first, gathers the entry's record from LDS with 16-byte reads and applies
//            the pair with the arithmetic of forward_face.  Inside and outside pixels share ONE clamped edge projection: an inside
//            pixel's nearest edge line follows from the three products w_k^2 hk2_k (sr_device.h: euclid_one); only where two
//            lines are equidistant within 1.5 % or within the face's rounding scale (sr_device.h: near_tie) -- the reference's own
//            choice is then decided by its rounding -- the reference's
//            three projections run (a region most iterations skip).  Before: in 43 % of the walk's iterations some lane was
//            inside and the 90-instruction three-projection branch ran at 27 % live lanes (tools/pair_stats.py);
//   merge    partial states (alpha product, running maximum, rescaled sums) are folded into their pixel's state.
//
// What changes against the one-wave kernel is the ORDER in which a pixel's fragments meet its state (inside fragments first,
// then outside fragments in index order, a stolen run merged at the end of each chunk): alpha product and depth softmax are
// symmetric in the fragments, only the rounding sequence moves (image within ~1e-6 of the reference-order kernels; the running
// maximum is exact).  Every (pixel, face) pair itself goes through the same instruction sequence as in forward_face
// (K.cu:370-453).
#pragma once

namespace lasr {

constexpr int PW_CAP = 64;            // list entries per staged chunk = bits per lane mask
constexpr int PW_LIST = 1024;         // tile list entries per round (u16 ids relative to the round's first face)
constexpr int PW_TILE = 16;

typedef unsigned long long u64_t;

// ---- the staged record (floats; 14 quads + the attributes)
//   q0  inv0..3      q1  inv4..7      q2  inv8, flags, rect lo, edge table
//   q3  far[0..2], rect ext: far[k] = -sqrt(1.05 thr / hk2[k]), the barycentric w_k below which a pixel is certainly beyond the threshold
//       edge table: which edge an OUTSIDE pixel projects to (K.cu:113-125), two bits per case: bit pair n0 + 2 n1 + 4 n2
//       (n_k = "w_k <= 0") of the low half, or of the high half when the pixel lies beyond the face's obtuse corner (q7's test)
//   q4  x0 y0 x1 y1  q5  x2 y2 z0 z1  q6  z2, 1/z0, 1/z1, 1/z2
//   q7  obtuse corner c (flags bit0..2): x_c, y_c, x_o - x_c, y_o - y_c with o = (c + 2) % 3   (K.cu:113-125's override test)
//   q8 + 2k, q9 + 2k   edge k: e[k][0..2], e[k][(k+1)%3]  |  den[k], 1/den[k], -, -
//   q14 hq[0..2]: the squared height over edge k's line (hk2 of the vertex opposite edge k; 0 unless the face is well conditioned),
//       hq[3]: the face's absolute near-tie scale (sr_device.h: near_tie_scale)
//   60 ..  attributes [vertex][channel]
constexpr int PR_INV = 0, PR_FLAGS = 9, PR_BB = 10, PR_ETBL = 11, PR_HK2 = 12, PR_BBE = 15, PR_XY = 16, PR_Z = 22, PR_IZ = 25, PR_OBT = 28, PR_EDGE = 32, PR_HQ = 56, PR_TEX = 60;

// record field (sr_device.h index) -> staged slot; compile-time for the generic arithmetic, which indexes with constants
__host__ __device__ constexpr int pr_of(int i)
{
    return i == R_BB ? PR_BB : i == R_BB + 1 ? PR_BBE : i == R_FLAGS ? PR_FLAGS
         : (i >= R_INV && i < R_INV + 9) ? PR_INV + (i - R_INV)
         : (i >= R_HK2 && i < R_HK2 + 3) ? PR_HK2 + (i - R_HK2)
         : (i >= R_FACE && i < R_FACE + 9) ? ((i - R_FACE) % 3 == 2 ? PR_Z + (i - R_FACE) / 3 : PR_XY + 2 * ((i - R_FACE) / 3) + (i - R_FACE) % 3)
         : (i >= R_DEN && i < R_DEN + 3) ? PR_EDGE + 8 * (i - R_DEN) + 4
         : (i >= R_IDEN && i < R_IDEN + 3) ? PR_EDGE + 8 * (i - R_IDEN) + 5
         : (i >= R_E && i < R_E + 9) ? PR_EDGE + 8 * ((i - R_E) / 3) + (i - R_E) % 3
         : (i >= R_IZ && i < R_IZ + 3) ? PR_IZ + (i - R_IZ) : 15;
}
struct PairRecView {                                   // the staged record behind the record indices of sr_device.h
    const float* p;
    __device__ __forceinline__ float operator[](int i) const { return p[pr_of(i)]; }
};

// LDS: the chunk's staged records at an ODD number of quads per slot, the tile's list, the list builder's counts
// SPLIT teams of four waves share a tile (SPLIT = 2: launches whose time is the heaviest tile's chain of dependent phases, not the
// chip's throughput): they build the list together, team t walks the chunks t, t + SPLIT, ... in its own staging buffer, and the
// teams' partial states meet in `merge` at the end.
template <int NCH, int SPLIT = 1>
struct PairLds {
    static constexpr int Q0 = (PR_TEX + 3 * NCH + 3) / 4;
    static constexpr int RS = 4 * (Q0 | 1);
    float rec[SPLIT][PW_CAP * RS];
    unsigned short list[PW_LIST];
    int wcnt[2][4 * SPLIT];
    float sel[24];                    // edge k: (k == 0, k == 1, k == 2 | k + 1 == 0, k + 1 == 1, k + 1 == 2 mod 3) as 0 / 1, 8 floats apart
    unsigned long long ball[SPLIT][2][16];   // the chunk's rect ballots: [0][x] entries whose rect holds column x, [1][y] row y (of the tile)
    float merge[SPLIT > 1 ? (SPLIT - 1) * 256 * (3 + NCH) : 1];       // partial states of the teams >= 1, [team - 1][field][pixel lane]
};

__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }
// "these values are needed HERE": keeps the compiler from sinking an LDS read to its first use, where the wave would sit out the
// whole round trip alone -- reads requested together come back together (nothing is emitted)
#ifndef LASR_PW_FENCE
#define LASR_PW_FENCE 0
#endif
#ifndef LASR_PW_WAVES
#define LASR_PW_WAVES 0      // no occupancy request: the compiler's own budget (see the kernel below)
#endif
#define PW_LANDED(q) asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w))

// the conservative corner cull of the tile kernels on a block of pixels [xlo, xhi] x [ylo, yhi]: false when all four corners lie
// beyond one edge's line by more than sqrt(thr_cull)
template <typename RP>
__device__ __forceinline__ bool reaches_block(RP R, float xlo, float xhi, float ylo, float yhi, float thr_cull)
{
    bool hit = true;
    if (__float_as_int(R[R_FLAGS]) & 16) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float a = R[R_INV + 3 * k], b = R[R_INV + 3 * k + 1], c = R[R_INV + 3 * k + 2];
            const float w00 = a * xlo + b * ylo + c, w01 = a * xhi + b * ylo + c;
            const float w10 = a * xlo + b * yhi + c, w11 = a * xhi + b * yhi + c;
            const float wmax = fmaxf(fmaxf(w00, w01), fmaxf(w10, w11));
            if (wmax < 0.f && wmax * wmax * R[R_HK2 + k] > thr_cull) hit = false;
        }
    }
    return hit;
}

// staging: source quads [Q0, Q1) of a record (global layout, sr_device.h) into their staged slots
__host__ __device__ constexpr bool rec_used(int i) { return i < 15 || (i >= 16 && i < 31) || (i >= 32 && i < 44); }
template <int Q0, int Q1>
__device__ __forceinline__ void stage_quads(const float4* __restrict__ src, float* __restrict__ dst, float thr_far, bool well)
{
    float4 v[Q1 - Q0];
#pragma unroll
    for (int q = Q0; q < Q1; q++) v[q - Q0] = src[q];
#pragma unroll
    for (int q = Q0; q < Q1; q++) {
        const float w[4] = {v[q - Q0].x, v[q - Q0].y, v[q - Q0].z, v[q - Q0].w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = 4 * q + j;
            if (!rec_used(i)) continue;
            if (i >= R_HK2 && i < R_HK2 + 3) {                 // the far threshold; -inf (never far) unless the face is well conditioned
                dst[pr_of(i)] = well ? -sqrtf(thr_far / w[j]) : -__builtin_huge_valf();
                dst[PR_HQ + (i - R_HK2 + 1) % 3] = well ? w[j] : 0.f;     // vertex k lies opposite edge (k + 1) % 3
                if (i == R_HK2) dst[PR_HQ + 3] = near_tie_scale(w[0], w[1], w[2]);        // (R_HK2 .. + 2 are words 0..2 of this source quad)
                continue;
            }
            dst[pr_of(i)] = w[j];
            if (i == R_E + 1) dst[PR_EDGE + 3] = w[j];            // e[k][(k+1)%3] once more, as the fourth word of edge k's block
            if (i == R_E + 5) dst[PR_EDGE + 8 + 3] = w[j];
            if (i == R_E + 6) dst[PR_EDGE + 16 + 3] = w[j];
        }
    }
}

// K.cu:113-125 as a table: the edge (0..2) an outside pixel projects to, for the sign pattern idx = n0 + 2 n1 + 4 n2 of its
// barycentrics (n_k = "w_k <= 0") and `over` = "beyond the face's obtuse corner" (only ever true for the face's one obtuse
// corner, flags bit0..2).  Same truth table as euclid<>'s lane-mask algebra (sr_device.h); idx 0 ("none <= 0", which the
// reference leaves undefined) is pinned to edge 0 like there.  Two bits per case, `over` cases in the high half.
__host__ __device__ constexpr unsigned edge_table(int obtuse_bits)
{
    unsigned t = 0;
    for (int over = 0; over < 2; over++)
        for (int idx = 0; idx < 8; idx++) {
            const bool n0 = idx & 1, n1 = idx & 2, n2 = idx & 4;
            const bool o0 = over && (obtuse_bits & 1), o1 = over && (obtuse_bits & 2), o2 = over && (obtuse_bits & 4);
            const bool c12 = n1 && n2, c20 = n2 && n0 && !n1, c01 = n0 && n1 && !n2;
            const bool e1 = (c20 && !o1) || (c01 && o2) || (n0 && !n1 && !n2);
            const bool e2 = (c01 && !o2) || (c12 && o0) || (n1 && !n0 && !n2);
            t |= (unsigned)((e1 ? 1 : 0) + (e2 ? 2 : 0)) << (2 * idx + 16 * over);
        }
    return t;
}

// a / b from y = RN(1 / b) with ONE Newton-Markstein correction: the correctly rounded quotient unless a / b lies within 2^-48
// (relative) of a rounding boundary -- about one quotient in 2^24 then differs from div_by_recip's by an ulp.  For the
// well-conditioned quotients after the distance code (clip / normalise, depth, softmax exponents); the edge projection, whose
// result decides `dis >= threshold`, keeps the two-correction form.
__device__ __forceinline__ float div_by_recip1(float a, float b, float y)
{
    const float q = a * y;
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);
}

// (float)(1. / (1. + (double)e)) for 0 < e < 2^116 (K.cu:397,403 promote the sigmoid to double) as v_rcp_f64 and ONE Newton step:
// v_rcp_f64 is good to 4.6e-8, the step squares that -- far below what narrowing to float can see (tools/ubench/acc.hip: no
// mismatch against recip64_noscale's full expansion in 2^24 arguments)
__device__ __forceinline__ float sigmoid_from_exp(float e)
{
    const double x = 1. + (double)e;
    const double y = __builtin_amdgcn_rcp(x);
    return (float)__builtin_fma(__builtin_fma(-x, y, 1.), y, y);
}

// clamp to [0, 1] of a finite value as the output modifier of a multiplication by one (v_mul_f32 issues at the fma rate, v_med3 /
// v_max at 0.6 of it, profiles/r04_ubench_mix.txt); same value as v_med3(x, 0, 1) up to the sign of a zero
__device__ __forceinline__ float clamp01(float x)
{
    float r;
    asm("v_mul_f32_e64 %0, 1.0, %1 clamp" : "=v"(r) : "v"(x));
    return r;
}

// merge of a partial state P (fragments folded on their own, starting from "no fragment": a = 1, sum 0, reference level
// smax = eps) into S:  a = a_S a_P;  m = max(m_S, m_P);  sum = sum_S e^((m_S - m) / gamma) + sum_P e^((m_P - m) / gamma)
template <int NCH>
__device__ __forceinline__ void merge_partial(PixState<NCH>& S, float pa, float pm, float ps, const float (&pc)[NCH],
                                              float gamma, float inv_gamma)
{
    S.a = (float)((double)S.a * (double)pa);
    const float mx = fmaxf(S.smax, pm);
    const float ds = S.smax - mx, dp = pm - mx;            // one of them is exactly 0
    const float es = exp_1ulp(div_by_recip(ds, gamma, inv_gamma));
    const float ep = exp_1ulp(div_by_recip(dp, gamma, inv_gamma));
    S.ssum = S.ssum * es + ps * ep;
#pragma unroll
    for (int k = 0; k < NCH; k++) S.c[k] = S.c[k] * es + pc[k] * ep;
    S.smax = mx;
}

// One (pixel, face) pair of a TAME record against the pixel state: forward_face<LASR, MK = true> (sr_raster.hip) with the record
// gathered per lane from its staged slot R.  is_in: the classification's inside test (the same predicate on the same
// barycentrics as euclid<.., TAME>).  Same operations in the same order per pair: the fragment's D, depth and weights are the
// bits the one-wave kernel computes.
template <int NCH>
__device__ __forceinline__ void pair_apply(const RasterArgs& A, const UniRecip& U, const float* __restrict__ R, const float* __restrict__ SEL,
                                           bool is_in, bool has, float xp, float yp, PixState<NCH>& s)
{
#pragma clang fp contract(off)
    float4 q0 = ld4(R), q1 = ld4(R + 4), q2 = ld4(R + 8);
    float4 q4 = ld4(R + PR_XY), q5 = ld4(R + PR_XY + 4), ob = ld4(R + PR_OBT);      // one LDS round trip for the six quads
#if LASR_PW_FENCE & 1
    PW_LANDED(q0); PW_LANDED(q1); PW_LANDED(q2); PW_LANDED(q4); PW_LANDED(q5); PW_LANDED(ob);
#endif
    const float w0 = q0.x * xp + q0.y * yp + q0.z;                    // K.cu:24-29 (barycentric())
    const float w1 = q0.w * xp + q1.x * yp + q1.y;
    const float w2 = q1.z * xp + q1.w * yp + q2.x;
    const float x0 = q4.x, y0 = q4.y, x1 = q4.z, y1 = q4.w, x2 = q5.x, y2 = q5.y, z0 = q5.z, z1 = q5.w;
    float narg;
    // which edge: an inside pixel's nearest edge LINE from the three products w^2 hk2 (sr_device.h: euclid_one), an outside pixel's
    // from the sign pattern of its barycentrics (K.cu:113-125); inside pixels near an angle bisector -- or of a face that is not
    // well conditioned (hq = 0) -- take the reference's three projections
    bool tie = false;
    int k_in = 0;
    if (wave_mask(is_in) != 0) {
        const float4 hq = ld4(R + PR_HQ);
        const float g0 = w2 * w2 * hq.x, g1 = w0 * w0 * hq.y, g2 = w1 * w1 * hq.z;
        const float glo = fminf(fminf(g0, g1), g2), gmid = __builtin_amdgcn_fmed3f(g0, g1, g2);
        tie = (bool)((int)is_in & (int)near_tie(glo, gmid, hq.w));
        k_in = g1 == glo ? 1 : g2 == glo ? 2 : 0;
    }
    if (tie) {
        // K.cu:61-110: project on all three edges, keep the nearest (euclid<.., FWD>'s inside branch)
        float best = 100000000.f, bx = 0, by = 0;
#pragma unroll
        for (int K = 0; K < 3; K++) {
            const float4 ea = ld4(R + PR_EDGE + 8 * K);
            const float2 eb = *(const float2*)(R + PR_EDGE + 8 * K + 4);
            const float num = w0 * ea.x + w1 * ea.y + w2 * ea.z - ea.w;
            const float ta = div_by_recip(num, eb.x, eb.y);
            const float tb = 1 - ta;
            float t[3];
            t[K] = ta; t[(K + 1) % 3] = tb; t[(K + 2) % 3] = 0;
            const float u0 = t[0] - w0, u1 = t[1] - w1, u2 = t[2] - w2;
            const float px = u0 * x0 + u1 * x1 + u2 * x2;
            const float py = u0 * y0 + u1 * y1 + u2 * y2;
            const float d2 = px * px + py * py;
            if (d2 < best) { best = d2; bx = px; by = py; }
        }
        narg = (-bx) * bx - by * by;
    } else {
        // K.cu:113-150: which edge (sign pattern of the barycentrics, obtuse-corner override), ONE clamped projection (the clamp
        // leaves an inside pixel's projection on its nearest line as it is: the foot lies on the triangle's boundary)
        const bool over = (xp - ob.x) * ob.z + (yp - ob.y) * ob.w > 0;
        const int sh = (w0 <= 0 ? 2 : 0) | (w1 <= 0 ? 4 : 0) | (w2 <= 0 ? 8 : 0) | (over ? 16 : 0);
        int k = (int)__builtin_amdgcn_ubfe(__float_as_uint(q2.w), (unsigned)sh, 2u);
        k = is_in ? k_in : k;
        const float* E = R + PR_EDGE + 8 * k;
        const float4 ea = ld4(E);
        const float2 eb = *(const float2*)(E + 4);
        const float num = w0 * ea.x + w1 * ea.y + w2 * ea.z - ea.w;
        float ta = div_by_recip(num, eb.x, eb.y);
        float tb = 1 - ta;
        ta = clamp01(ta);
        tb = clamp01(tb);
        // t[k] = ta, t[(k + 1) % 3] = tb, the third 0, by exact 0 / 1 factors (ta, tb lie in [0, 1]: every product and sum is exact)
        const float4 fa = ld4(SEL + 8 * k);
        const float2 fb = *(const float2*)(SEL + 8 * k + 4);
        const float t0 = __builtin_fmaf(tb, fa.w, ta * fa.x), t1 = __builtin_fmaf(tb, fb.x, ta * fa.y), t2 = __builtin_fmaf(tb, fb.y, ta * fa.z);
        const float u0 = t0 - w0, u1 = t1 - w1, u2 = t2 - w2;
        const float dx = u0 * x0 + u1 * x1 + u2 * x2;
        const float dy = u0 * y0 + u1 * y1 + u2 * y2;
        const float d2 = dx * dx + dy * dy;
        narg = is_in ? -d2 : d2;                        // inside: (-dx) dx - dy dy, the same bits
        if ((bool)((int)!is_in & ((int)(d2 >= A.thr) | (int)!has))) return;          // K.cu:402 (has: false for a lane that rides along)
    }
    float4 q6 = ld4(R + PR_XY + 8);                                                  // z2, 1/z0, 1/z1, 1/z2
    float4 tq[(3 * NCH + 3) / 4];
#pragma unroll
    for (int q = 0; q < (3 * NCH + 3) / 4; q++) tq[q] = ld4(R + PR_TEX + 4 * q);
    const float D = sigmoid_from_exp(exp_1ulp(div_by_recip1(narg, A.sigma, U.inv_sigma)));
    // K.cu:409-417, (float)((double)a * (1. - (double)D)): a - a D in ONE rounding is that product rounded once -- the same float
    // except where the double rounding of the reference's form shows (about one product in 2^29)
    s.a = __builtin_fmaf(-s.a, D, s.a);
#if LASR_PW_FENCE & 2
    PW_LANDED(q6);
#pragma unroll
    for (int q = 0; q < (3 * NCH + 3) / 4; q++) PW_LANDED(tq[q]);
#endif
    // clip / normalise (K.cu:53-58) and depth (K.cu:423) with single-correction quotients (div_by_recip1)
    float c0 = clamp01(w0), c1 = clamp01(w1), c2 = clamp01(w2);
    {
        const float sm = fmaxf(c0 + c1 + c2, 1e-5f);
        float y = __builtin_amdgcn_rcpf(sm);
        y = __builtin_fmaf(__builtin_fmaf(-sm, y, 1.f), y, y);
        c0 = div_by_recip1(c0, sm, y); c1 = div_by_recip1(c1, sm, y); c2 = div_by_recip1(c2, sm, y);
    }
    const float zs = div_by_recip1(c0, z0, q6.y) + div_by_recip1(c1, z1, q6.z) + div_by_recip1(c2, q6.x, q6.w);
    float zp;
    {
        float y = __builtin_amdgcn_rcpf(zs);
        y = __builtin_fmaf(__builtin_fmaf(-zs, y, 1.f), y, y);
        zp = __builtin_fmaf(__builtin_fmaf(-zs, y, 1.f), y, y);
    }
    if (!(zp >= A.near && zp <= A.far)) return;
    const float zn = div_by_recip1(A.far - zp, A.far - A.near, U.inv_fmn);
    const bool up = zn > s.smax;
    const float d = -fabsf(zn - s.smax);
    const float Ex = exp_1ulp(div_by_recip1(d, A.gamma, U.inv_gamma));
    const float hist = up ? Ex : 1.f, wgt = up ? D : Ex * D;
    s.smax = max_finite(zn, s.smax);
    // (the sums as fused multiply-adds: the pair walk folds a pixel's fragments in its own order anyway, see the header)
    s.ssum = __builtin_fmaf(hist, s.ssum, wgt);
    float tex[3 * NCH];
#pragma unroll
    for (int q = 0; q < (3 * NCH + 3) / 4; q++) {
        const float4 t = tq[q];
        tex[4 * q] = t.x;
        if (4 * q + 1 < 3 * NCH) tex[4 * q + 1] = t.y;
        if (4 * q + 2 < 3 * NCH) tex[4 * q + 2] = t.z;
        if (4 * q + 3 < 3 * NCH) tex[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int k = 0; k < NCH; k++) {
        const float col = __builtin_fmaf(c2, tex[2 * NCH + k], __builtin_fmaf(c1, tex[NCH + k], c0 * tex[k]));
        s.c[k] = __builtin_fmaf(wgt, col, hist * s.c[k]);
    }
}

template <int NCH, int SPLIT>
__device__ __forceinline__ void pairs_tile_body(RasterArgs A, float* __restrict__ aggrs, float* __restrict__ colors, PairLds<NCH, SPLIT>& L)
{
    constexpr int RS = PairLds<NCH, SPLIT>::RS;
    constexpr int NW = 4 * SPLIT;                       // waves per tile

    const Modes m = Modes{2, 1, 2, 1, 1};               // LASR's configuration: euclidean, softmax, prod, vertex, double-sided
    if (A.near_far_dev) { A.near = A.near_far_dev[0]; A.far = A.near_far_dev[1]; }
    const int IS = A.IS, P = IS * IS;
    const int tiles_x = (IS + PW_TILE - 1) / PW_TILE;
    int bn, tx, ty;
    tile_of_block(blockIdx.x, gridDim.x, tiles_x, bn, tx, ty, A.order);      // this launch's own order when the host built one (16x16 tiles)
    const int tid = threadIdx.x, wave_all = tid >> 6, lane = tid & 63;
    const int team = SPLIT > 1 ? wave_all >> 2 : 0, wave = wave_all & 3;        // wave: within its team (the pixel rows it owns)
#ifndef LASR_PW_ROWS
#define LASR_PW_ROWS 1
#endif
#if LASR_PW_ROWS
    // wave w takes the rows w, w + 4, w + 8, w + 12 of the tile (16 pixels = 64 bytes each): the four waves of a tile see the same
    // load (they meet at two barriers per chunk), and a store instruction covers whole 64-byte segments
    const int px = tx * PW_TILE + (lane & 15), py = ty * PW_TILE + wave + 4 * (lane >> 4);
#else
    const int qx0 = tx * PW_TILE + (wave & 1) * 8, qy0 = ty * PW_TILE + (wave >> 1) * 8;       // this wave's 8x8 quadrant
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
#endif
    const bool valid = px < IS && py < IS;
    const int pn = py * IS + px;

    PixState<NCH> s;
    s.a = 1.f;
    s.fbest = -1;
    s.ssum = expf(A.eps / A.gamma); s.smax = A.eps;
#pragma unroll
    for (int k = 0; k < NCH; k++) {
        const float bg = A.use_bg ? A.bg[k] : (valid ? colors[((size_t)bn * (NCH + 1) + k) * P + pn] : 1.f);
        s.c[k] = bg * s.ssum;
    }
    if (SPLIT > 1 && team > 0) {                        // "no fragment yet": merged into team 0's state at the end (merge_partial)
        s.ssum = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; k++) s.c[k] = 0.f;
    }

    if (tid < 24) {
        const int k = tid >> 3, j = tid & 7;          // edge k, slot j: [0..2] = (vertex == k), [3] = (k + 1) % 3 == 0, [4..5] = ... == 1, == 2
        L.sel[tid] = j < 3 ? (j == k ? 1.f : 0.f) : (j >= 3 && j < 6 && (j - 3) == (k + 1) % 3) ? 1.f : 0.f;
    }
    // level 0: the groups of 64 consecutive faces whose union rect meets the 16x16 tile (every wave evaluates the same test)
    const int G = groups_of(A.F);
    const short4* __restrict__ grects = A.grects + (size_t)bn * G;
    const int tX0 = tx * PW_TILE, tX1 = tX0 + PW_TILE - 1, tY0 = ty * PW_TILE, tY1 = tY0 + PW_TILE - 1;
    u64_t gmask;
    {
        bool t = false;
        if (lane < G) {
            const short4 q = grects[lane];
            t = !(q.x > tX1 || q.y < tX0 || q.z > tY1 || q.w < tY0);
        }
        gmask = wave_mask(t);
    }
    if (gmask != 0 || G > 64) {
    const float xp = pix_center(px, IS);
    const float yp = pix_center(IS - 1 - py, IS);
    const short4* __restrict__ rects = A.rects + (size_t)bn * A.F;
    const float* __restrict__ recs = A.recs + (size_t)bn * A.F * REC;
    const int texstride = A.T * NCH;
    const float* __restrict__ texs = A.textures + (size_t)bn * A.F * texstride;
    const UniRecip U = uni_recip(A);
    const int ok_bit = U.ok ? 32 : 0;
    const float thr_cull = A.thr * 1.10f, thr_far = A.thr * 1.05f;
    // corners of the tile (list building) and of this wave's quadrant (which entries of a chunk the wave looks at); wave-uniform
    // values the VALU computed (divisions) go back to scalar registers
    auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    const float t_xlo = uni(pix_center(tX0, IS)), t_xhi = uni(pix_center(min(tX1, IS - 1), IS));
    const float t_yhi = uni(pix_center(IS - 1 - tY0, IS)), t_ylo = uni(pix_center(IS - 1 - min(tY1, IS - 1), IS));
#if !LASR_PW_ROWS
    const float q_xlo = uni(pix_center(qx0, IS)), q_xhi = uni(pix_center(min(qx0 + 7, IS - 1), IS));
    const float q_yhi = uni(pix_center(IS - 1 - qy0, IS)), q_ylo = uni(pix_center(IS - 1 - min(qy0 + 7, IS - 1), IS));
#endif
    int g_next = 64, g_mask0 = 0;
    bool more = true;
    while (more) {                                    // one round unless the tile meets more than PW_LIST - 256 faces
        // ---- ordered list of the faces that reach the tile: four touched groups per step, one per wave
        int count = 0, flip = 0, base = -1;
        for (;;) {
            if (gmask == 0) {
                if (g_next >= G) { more = false; break; }
                g_mask0 = g_next;
                bool t = false;
                if (g_next + lane < G) {
                    const short4 q = grects[g_next + lane];
                    t = !(q.x > tX1 || q.y < tX0 || q.z > tY1 || q.w < tY0);
                }
                gmask = wave_mask(t);
                g_next += 64;
                continue;
            }
            u64_t mm = gmask;
            int mine_g = -1, last_g = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                if (mm) {
                    const int bit = __builtin_ctzll(mm);
                    mm &= mm - 1;
                    if (k == wave_all) mine_g = g_mask0 + bit;
                    last_g = g_mask0 + bit;
                }
            }
            const int first_g = g_mask0 + __builtin_ctzll(gmask);
            if (base < 0) base = first_g * GROUP;
            if (count + 64 * NW > PW_LIST || (last_g + 1) * GROUP - base > 65536) break;  // walk what we have, then continue
            gmask = mm;
            const int f = mine_g * GROUP + lane;
            bool hit = false;
            if (mine_g >= 0 && f < A.F) {
                const short4 q = rects[f];
                hit = !(q.x > tX1 || q.y < tX0 || q.z > tY1 || q.w < tY0);
                if (hit) hit = reaches_block(recs + (size_t)f * REC, t_xlo, t_xhi, t_ylo, t_yhi, thr_cull);
            }
            const u64_t mask = wave_mask(hit);
            if (lane == 0) L.wcnt[flip][wave_all] = __popcll(mask);
            __syncthreads();
            int before = 0, step_total = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const int ck = L.wcnt[flip][k];
                before += k < wave_all ? ck : 0;
                step_total += ck;
            }
            if (hit) L.list[count + before + bits_below_lane(mask)] = (unsigned short)(f - base);
            count += step_total;
            flip ^= 1;
        }
        if (base < 0) base = 0;
        __syncthreads();

        float* const Lrec = L.rec[team];
        for (int cb = 0; cb < count; cb += SPLIT * PW_CAP) {       // team t takes the chunks t, t + SPLIT, ...: same trip count, same barriers
            const int c0 = cb + team * PW_CAP;
            const int n = max(0, min(PW_CAP, count - c0));
            // ---- stage: lane = entry, every wave a quarter of the record's source quads (all loads of a chunk in flight at once)
            int cm = 0, rm = 0;                       // the entry's rect as a 16-bit column mask and row mask of the tile
            if (lane < n) {
                const int fn = base + (int)L.list[c0 + lane];
                {
                    const short4 q = rects[fn];       // x0, x1, row0, row1 (empty: 32767, -1, 32767, -1)
                    const int a0 = max((int)q.x - tX0, 0), a1 = min((int)q.y - tX0, PW_TILE - 1);
                    const int b0 = max((int)q.z - tY0, 0), b1 = min((int)q.w - tY0, PW_TILE - 1);
                    if (a0 <= a1) cm = (2 << a1) - (1 << a0);
                    if (b0 <= b1) rm = (2 << b1) - (1 << b0);
                }
                const float4* __restrict__ src = (const float4*)(recs + (size_t)fn * REC);
                const float* __restrict__ ta = texs + (size_t)fn * texstride;
                float* dst = Lrec + lane * RS;
                float tv[(3 * NCH + 3) / 4];
#pragma unroll
                for (int i = 0; i < (3 * NCH + 3) / 4; i++) tv[i] = wave + 4 * i < 3 * NCH ? ta[wave + 4 * i] : 0.f;
                const bool well = (__float_as_int(((const float*)src)[R_FLAGS]) & 16) != 0;
                if (wave == 0) stage_quads<0, 3>(src, dst, thr_far, well);
                else if (wave == 1) stage_quads<3, 6>(src, dst, thr_far, well);
                else if (wave == 2) stage_quads<6, 9>(src, dst, thr_far, well);
                else {
                    stage_quads<9, 11>(src, dst, thr_far, well);
                    // the obtuse corner's operands (see the layout above)
                    const float* f = (const float*)src;
                    const int fl = __float_as_int(f[R_FLAGS]);
                    const int c = (fl & 1) ? 0 : (fl & 2) ? 1 : 2, o = c == 0 ? 2 : c - 1;
                    const float xc = f[R_FACE + 3 * c], yc = f[R_FACE + 3 * c + 1], xo = f[R_FACE + 3 * o], yo = f[R_FACE + 3 * o + 1];
                    *(float4*)(dst + PR_OBT) = make_float4(xc, yc, xo - xc, yo - yc);
                    const unsigned tb = (fl & 1) ? edge_table(1) : (fl & 2) ? edge_table(2) : (fl & 4) ? edge_table(4) : edge_table(0);
                    dst[PR_ETBL] = __uint_as_float(tb);
                }
#pragma unroll
                for (int i = 0; i < (3 * NCH + 3) / 4; i++)
                    if (wave + 4 * i < 3 * NCH) dst[PR_TEX + wave + 4 * i] = tv[i];
            }
            // the rect ballots of the chunk (the exact bbox test, K.cu:375): wave w forms those of columns 4w .. 4w + 3 and of rows
            // 4w .. 4w + 3 -- lanes = entries -- and leaves them in LDS for the four waves' pixel lanes
            {
                u64_t mine = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const u64_t Bc = wave_mask((cm >> (4 * wave + j)) & 1), Br = wave_mask((rm >> (4 * wave + j)) & 1);
                    if (lane == j) mine = Bc;
                    if (lane == 4 + j) mine = Br;
                }
                if (lane < 8) L.ball[team][lane >> 2][4 * wave + (lane & 3)] = mine;
            }
            __syncthreads();

            // ---- candidates: bit e of a lane's mask = "entry e's pixel rect holds my pixel": the ballots of its column and of its row
            u64_t cand, tame_mask;
            {
                bool tame_e = false;
                if (lane < n) tame_e = (__float_as_int(Lrec[lane * RS + PR_FLAGS]) & ok_bit) != 0;
                const u64_t cxv = L.ball[team][0][px - tX0], ryv = L.ball[team][1][py - tY0];
                cand = valid ? cxv & ryv : 0ull;
                tame_mask = wave_mask(tame_e);
            }
#if defined(LASR_PW_ABL) && LASR_PW_ABL == 2        // measurement build: list + stage only
            s.a += (float)__popcll(cand); cand = 0;
#endif
            // ---- classify: every lane pops its own candidates of TAME records (the others -- or every record of a launch whose
            // uniforms are not safe for the reciprocal arithmetic -- go to the slow mask as they are): inside the face, or outside
            // and not certainly beyond the threshold (the staged far[k]: w_k < far[k] puts the pixel farther than sqrt(1.05 thr)
            // beyond edge k's line; -inf for faces the conservative reject does not apply to)
            u64_t ms = cand & ~tame_mask, mi = 0, mo = 0;
            cand &= tame_mask;
            // (no divergent region: a lane without candidates rides along on slot 63 with an empty bit)
            if (wave_mask(cand != 0) != 0) do {
                const int e = __builtin_ctzll(cand | (1ull << 63));
                const u64_t rest = cand & (cand - 1), bit = cand ^ rest;
                cand = rest;
                const float* R = Lrec + e * RS;
                const float4 q0 = ld4(R), q1 = ld4(R + 4);
                const float inv8 = R[8];
                const float4 q3 = ld4(R + PR_HK2);
                // (unfused like the walk's: on an edge-on face the fused form moves the barycentrics by whole pixels, and which
                // distance formula a pixel gets must be the reference's choice)
                const float w0 = q0.x * xp + q0.y * yp + q0.z;                  // barycentric()
                const float w1 = q0.w * xp + q1.x * yp + q1.y;
                const float w2 = q1.z * xp + q1.w * yp + inv8;
                const bool inside = (bool)((int)(fminf(fminf(w0, w1), w2) > 0) & (int)(fmaxf(fmaxf(w0, w1), w2) < 1));
                const bool far = (bool)((int)(w0 < q3.x) | (int)(w1 < q3.y) | (int)(w2 < q3.z));
                mi |= inside ? bit : 0ull;
                mo |= (bool)((int)!inside & (int)!far) ? bit : 0ull;
            } while (wave_mask(cand != 0) != 0);

#if defined(LASR_PW_ABL) && LASR_PW_ABL == 1        // measurement build: no walk
            s.a += (float)(__popcll(mi) + 2 * __popcll(mo)); mi = mo = 0;
#endif
#if defined(LASR_PW_ABL) && LASR_PW_ABL == 3        // measurement build: no balance (every lane its own pairs)
            const bool no_steal = true;
#else
            const bool no_steal = false;
#endif
            // ---- balance: rank the lanes by their pair count (heaviest first), partner rank r with rank 63 - r
            const int kmine = min((int)__popcll(mi) + (int)__popcll(mo), 63);
            int rank;
            {
                u64_t Gm = ~0ull;
                int greater = 0;
#pragma unroll
                for (int b = 5; b >= 0; b--) {
                    const bool mine = (kmine >> b) & 1;
                    const u64_t B = wave_mask(mine);
                    const u64_t GB = Gm & B;
                    if (!mine) { greater += __popcll(GB); Gm ^= GB; }
                    else Gm = GB;
                }
                rank = greater + bits_below_lane(Gm);
            }
            const int id_at_rank = __builtin_amdgcn_ds_permute(rank << 2, lane);                 // lane r: the lane of rank r
            const int partner = __builtin_amdgcn_ds_bpermute((63 - rank) << 2, id_at_rank);
            const int kpart = __builtin_amdgcn_ds_bpermute(partner << 2, kmine);
            const unsigned pmo_lo = (unsigned)__builtin_amdgcn_ds_bpermute(partner << 2, (int)(unsigned)mo);
            const unsigned pmo_hi = (unsigned)__builtin_amdgcn_ds_bpermute(partner << 2, (int)(unsigned)(mo >> 32));
            const float xq = __int_as_float(__builtin_amdgcn_ds_bpermute(partner << 2, __float_as_int(xp)));
            const float yq = __int_as_float(__builtin_amdgcn_ds_bpermute(partner << 2, __float_as_int(yp)));
            const bool heavy = rank < 32;
            // the heavier partner's outside mask and the number of its pairs that change hands (both partners compute the same)
            const u64_t hm = heavy ? mo : ((u64_t)pmo_hi << 32 | pmo_lo);
            const int moved = min((heavy ? kmine - kpart : kpart - kmine) >> 1, (int)__popcll(hm));
            u64_t st = 0;                                                       // light lane: the pairs it takes over
            bool gave = false;
            if (moved > 0 && !no_steal) {
                // smallest position whose upper part holds at most `moved` bits
                int p = 0;
                u64_t top = hm;
                if (__popcll(hm) > moved) {
#pragma unroll
                    for (int sft = 32; sft >= 1; sft >>= 1)
                        if (p + sft <= 63 && __popcll(hm >> (p + sft)) > moved) p += sft;
                    top = p >= 63 ? 0ull : (hm >> (p + 1)) << (p + 1);
                }
                if (heavy) { mo ^= top; gave = top != 0; }
                else st = top;
            }

            // ---- walk: every lane pops its runs of pairs -- its inside pairs, its outside pairs, the partner's share -- one pair per
            // iteration; a lane moves on to its next run inside the (rarely entered) region below
            float cx = xp, cy = yp;
            bool in_stolen = false;
            PixState<NCH> saved = s;
            u64_t cur = mi, nxt = mo;
            bool is_in = mi != 0;
            u64_t more_m = ~0ull;
            // a lane whose current run is used up moves on to its next one (entered when some lane has to)
            auto advance = [&]() {
                if ((wave_mask(cur == 0) & more_m) != 0) {
                    if (cur == 0) {
                        if (nxt != 0) { cur = nxt; nxt = 0; }
                        else if (st != 0) {                                         // the partner's share, into a fresh partial state
                            saved = s;
                            s.a = 1.f; s.ssum = 0.f; s.smax = A.eps;
#pragma unroll
                            for (int k = 0; k < NCH; k++) s.c[k] = 0.f;
                            cx = xq; cy = yq; cur = st; st = 0; in_stolen = true;
                        }
                        is_in = false;
                    }
                    more_m = wave_mask((nxt | st) != 0);
                }
            };
            advance();
            if (wave_mask(cur != 0) != 0) do {
                // (no divergent region around the pair: a lane without work rides along the outside branch on entry 63's slot and
                // leaves at the threshold cut -- the instructions are issued for the wave either way)
                const bool has = cur != 0;
                const int e = __builtin_ctzll(cur | (1ull << 63));
                cur &= cur - 1;
                pair_apply<NCH>(A, U, Lrec + e * RS, L.sel, (bool)((int)is_in & (int)has), has, cx, cy, s);
                advance();
            } while (wave_mask(cur != 0) != 0);
            // ---- merge the partial states into their pixels
            float pa = s.a, pm = s.smax, ps = s.ssum, pc[NCH];
#pragma unroll
            for (int k = 0; k < NCH; k++) pc[k] = s.c[k];
            if (in_stolen) s = saved;
            if (wave_mask(gave) != 0) {
                pa = __int_as_float(__builtin_amdgcn_ds_bpermute(partner << 2, __float_as_int(pa)));
                pm = __int_as_float(__builtin_amdgcn_ds_bpermute(partner << 2, __float_as_int(pm)));
                ps = __int_as_float(__builtin_amdgcn_ds_bpermute(partner << 2, __float_as_int(ps)));
#pragma unroll
                for (int k = 0; k < NCH; k++) pc[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(partner << 2, __float_as_int(pc[k])));
                if (gave) merge_partial<NCH>(s, pa, pm, ps, pc, A.gamma, U.inv_gamma);
            }
            // ---- the chunk's slow pairs (records that are not tame): wave-uniform entry, generic arithmetic
            for (int e = 0; wave_mask(ms != 0) != 0 && e < n; e++) {
                if (wave_mask((ms >> e) & 1) == 0) continue;
                const PairRecView R{Lrec + e * RS};
                if ((ms >> e) & 1) {
                    float w0, w1, w2;
                    barycentric(R, xp, yp, w0, w1, w2);
                    forward_face<true, false, NCH>(A, m, R, Lrec + e * RS + PR_TEX, 0, 0, xp, yp, w0, w1, w2, s, U);
                }
            }
            __syncthreads();                            // the slots are rewritten by the next chunk
        }
    }
    }   // tile meets at least one group

    if (SPLIT > 1) {
        // ---- the teams' partial states meet: team t >= 1 leaves its state in LDS, team 0 folds them in (team order) and finalises
        const int pl = wave * 64 + lane;                    // the pixel's slot: same (wave, lane) -> pixel mapping in every team
        if (team > 0) {
            float* o = L.merge + (size_t)(team - 1) * (3 + NCH) * 256 + pl;
            o[0] = s.a; o[256] = s.smax; o[512] = s.ssum;
#pragma unroll
            for (int k = 0; k < NCH; k++) o[(3 + k) * 256] = s.c[k];
        }
        __syncthreads();
        if (team > 0) return;
        const float inv_gamma = 1.f / A.gamma;
#pragma unroll
        for (int t = 1; t < SPLIT; t++) {
            const float* o = L.merge + (size_t)(t - 1) * (3 + NCH) * 256 + pl;
            float pc[NCH];
#pragma unroll
            for (int k = 0; k < NCH; k++) pc[k] = o[(3 + k) * 256];
            merge_partial<NCH>(s, o[0], o[256], o[512], pc, A.gamma, inv_gamma);
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

// six / nine channels: the compiler's own register budget (95 / 112 VGPRs)
template <int NCH>
__global__ __launch_bounds__(256) void sr_forward_pairs_kernel(RasterArgs A, float* __restrict__ aggrs, float* __restrict__ colors)
{
    __shared__ __attribute__((aligned(16))) PairLds<NCH, 1> L;
    pairs_tile_body<NCH, 1>(A, aggrs, colors, L);
}
// three channels: no occupancy request (87 VGPRs = 5 waves per SIMD; six / seven forced were slower, profiles/experiments/README.md)
__global__ __launch_bounds__(256)
#if LASR_PW_WAVES
__attribute__((amdgpu_waves_per_eu(LASR_PW_WAVES, LASR_PW_WAVES)))
#endif
void sr_forward_pairs3_kernel(RasterArgs A, float* __restrict__ aggrs, float* __restrict__ colors)
{
    __shared__ __attribute__((aligned(16))) PairLds<3, 1> L;
    pairs_tile_body<3, 1>(A, aggrs, colors, L);
}
// two / four teams of four waves per tile (512 / 1024 threads): launches bound by the heaviest tile's chain of phases, not by throughput
template <int NCH, int SPLIT>
__global__ __launch_bounds__(256 * SPLIT) void sr_forward_pairs_teams_kernel(RasterArgs A, float* __restrict__ aggrs, float* __restrict__ colors)
{
    __shared__ __attribute__((aligned(16))) PairLds<NCH, SPLIT> L;
    pairs_tile_body<NCH, SPLIT>(A, aggrs, colors, L);
}

}  // namespace lasr
