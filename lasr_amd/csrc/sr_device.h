// sr_device.h -- gfx950 device-side arithmetic of the soft-rasteriser.
//
// Every expression keeps the operation order and the float/double promotions of
// the reference kernel (/root/reference/third_party/softras/soft_renderer/cuda/
// soft_rasterize_cuda_kernel.cu, "K.cu" below) so that, compiled with
// -ffp-contract=off, the geometry (barycentrics, edge projections, depth) is
// bit-identical to an un-contracted fp32 evaluation of the reference; only
// exp() may differ by an ulp from a host libm.  What is NOT taken from the
// reference is the execution shape: per-face constants live in a 48-float
// record that a wave reads through the scalar cache (the face index is
// wave-uniform), the bbox reject becomes an exact integer pixel rectangle, divisions by
// per-face constants go through correctly rounded reciprocals, and runtime-indexed arrays
// are replaced by compile-time edge indices so nothing spills to scratch.
#pragma once
#include <hip/hip_runtime.h>

namespace lasr {

// ---- per-face record -------------------------------------------------------
// 48 floats = 192 B on 64-B boundaries, fields grouped by the cache line the raster walks need them in (a record is
// fetched through the scalar cache with dependent loads: three aligned lines instead of the 3-4 a packed 176-B record
// straddles; measured -1.8 % on the backward kernel, which starts every wave with this fetch).
// line 0  [0..1]   rect : the pixels that pass the bbox test of K.cu:33-38, as EXACT integer bounds (own addition):
//                         [0] = x0 | r0 << 16 (first column, first row from the top), [1] = (x1 - x0) | (r1 - r0) << 16
//                         (extents), so that "pixel in rect" is one packed u16 subtract + min + compare;
//                         empty = [0] all ones, [1] zero
//         [2]      flags: bit0..2 first obtuse corner (K.cu:296-304), bit3 front-facing (K.cu:41-44),
//                         bit4 well-conditioned (the cheap line-distance reject below may be used), bit5 TAME record:
//                         reciprocals usable, every inv entry finite and below 1e30 (so the barycentrics of a pixel are
//                         finite: min3 / max3 / med3 then equal the reference's compare chains), vertex depths positive
//                         and within [1e-6, 1e6] (so 1 / sum(c_k / z_k) needs no scaling)
//         [3..11]  inv  : rows of adj([x y 1])/det                         (K.cu:274-286)
//         [12..14] hk2  : squared height of vertex k over its opposite edge (own addition: w_k * h_k is the
//                         signed distance of a pixel to that edge's line, a lower bound of the true distance)
// line 1  [16..24] x0 y0 z0 x1 y1 z1 x2 y2 z2
//         [25..27] den  : den[k]  = e[k][k] - e[k][(k+1)%3]                 (K.cu:85,136, hoisted)
//         [28..30] iden : RN(1/den[k])   } correctly rounded reciprocals for the exact-division-by-reciprocal below;
// line 2  [32..40] e    : e[k][j] = sym[3k+j] - sym[3((k+1)%3)+j]           (K.cu:81-83,132-134, hoisted)
//         [41..43] iz   : RN(1/z_k)      } flags bit5 says they are usable (finite, denominators in a safe range)
//         [15], [31], [44..47] padding
constexpr int REC = 48;

// Round-4 instruction-mix options (profiles/r04_opt_ab.txt has the A/B of each bit; every one leaves the forward's output
// bit-identical -- the kernel-choice, reference-vector and oracle suites run with all of them on).  Measurement builds switch
// bits off: make variant NAME=x DEFS=-DLASR_OPT=<mask>.
#ifndef LASR_OPT
#define LASR_OPT 0xffff
#endif
constexpr bool OPT_PKRECT = (LASR_OPT & 1) != 0;      // rect test as one packed u16 compare
constexpr bool OPT_MED3 = (LASR_OPT & 2) != 0;        // clamps / inside test as v_med3 / v_min3 / v_max3 on tame records
constexpr bool OPT_NOSCALE = (LASR_OPT & 4) != 0;     // 1/x and the f64 sigmoid division without v_div_scale / v_div_fixup
constexpr bool OPT_SIGNFOLD = (LASR_OPT & 8) != 0;    // threshold cut inside the outside branch, no float sign
constexpr bool OPT_SOFTMAX = (LASR_OPT & 16) != 0;    // depth-softmax update with -|zn - smax| and v_max
// (bit 32, the backward's pixel planes through buffer descriptors, was measured slower and removed)
constexpr bool OPT_BWD_S1 = (LASR_OPT & 64) != 0;     // backward stage 1: branch-free conservative reject, fused centres
constexpr bool OPT_BWD_MATH = (LASR_OPT & 128) != 0;
constexpr bool OPT_EDGESEL = (LASR_OPT & 256) != 0;   // which edge an outside pixel projects to: lane-mask algebra, not a nested if chain  // backward stage 2: record reciprocals, med3, per-face depth-range test
constexpr int R_BB = 0, R_FLAGS = 2, R_INV = 3, R_HK2 = 12, R_FACE = 16, R_DEN = 25, R_IDEN = 28, R_E = 32, R_IZ = 41;

// Read-only buffers written by an EARLIER kernel are viewed through the constant address
// space: with a wave-uniform index the compiler then emits s_load (scalar cache -> SGPRs)
// even after barriers, instead of 64 identical vector loads.
typedef const float __attribute__((address_space(4)))* cptr_t;
__device__ __forceinline__ cptr_t as_const(const float* p) { return (cptr_t)(unsigned long long)p; }

struct Modes {
    int dist, rgb, alpha, tex, double_side;
};

__device__ __forceinline__ float pix_center(int i, int is)
{
    // K.cu:345-346 evaluates (2.*i + 1. - is) / is in double and narrows.  The numerator is an exact
    // integer and narrowing a double quotient of two floats is the correctly rounded float quotient
    // (53 >= 2*24+2 bits), so one IEEE fp32 division gives the identical value.
    return (float)(2 * i + 1 - is) / (float)is;
}

// ---- exact division by a precomputed reciprocal ------------------------------------------------------
// q = RN(a / b) from y = RN(1 / b): two Newton-Markstein corrections with exact FMA residuals.  With y correctly
// rounded the second correction returns the correctly rounded quotient (Markstein 1990; the only exception, a
// divisor whose significand is all ones, has probability 2^-23 and costs at most one ulp).  5 VALU ops instead of
// the ~11 of the IEEE division expansion.  Only used when the divisor is in a range where neither y nor the
// quotient can overflow/underflow (recip_safe); otherwise the plain division runs.
__device__ __forceinline__ bool recip_safe(float b)
{
    const float m = fabsf(b);
    return m > 1e-18f && m < 1e18f;              // also false for NaN / inf / 0
}
__device__ __forceinline__ float div_by_recip(float a, float b, float y)
{
    float q = a * y;
    float e = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e, y, q);
    e = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e, y, q);
}

// 1 / x bit-identical to the IEEE division `1.f / x` for 2^-60 < |x| < 2^60: the instruction sequence the compiler emits for
// that division (v_rcp_f32, one Newton step, quotient, two residual corrections) minus v_div_scale x 2 / v_div_fixup, which
// are identities in that range, and with v_div_fmas as a plain fma (7 instead of 11 VALU ops; lasr_selftest_recip).
__device__ __forceinline__ float recip_noscale(float x)
{
    float y = __builtin_amdgcn_rcpf(x);
    y = __builtin_fmaf(__builtin_fmaf(-x, y, 1.f), y, y);
    float q = y;                                             // 1 * y
    q = __builtin_fmaf(__builtin_fmaf(-x, q, 1.f), y, q);
    return __builtin_fmaf(__builtin_fmaf(-x, q, 1.f), y, q);
}
// (float)(1. / x) for a double 1 <= x < 2^128: the compiler's f64 division expansion (v_rcp_f64, two Newton steps, quotient,
// one residual correction) without v_div_scale_f64 x 2 / v_div_fixup_f64 (identities here); same bits (lasr_selftest_recip).
__device__ __forceinline__ double recip64_noscale(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = __builtin_fma(__builtin_fma(-x, y, 1.), y, y);
    y = __builtin_fma(__builtin_fma(-x, y, 1.), y, y);
    const double q = y;                                      // 1 * y
    return __builtin_fma(__builtin_fma(-x, q, 1.), y, q);
}

// max(a, b) of two finite floats as ONE v_max_f32 (fmaxf makes the compiler quiet a possible signalling NaN first: two ops)
__device__ __forceinline__ float max_finite(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// "pixel (px | py << 16) lies in the record's rect" (record fields R_BB, R_BB + 1): per 16-bit half d = p - first must not
// exceed the extent; a pixel left of / above the rect wraps to d >= 32769 > any extent, and the empty rect (first = 0xffff,
// extent 0) holds no pixel of an image (coordinates <= 32766).  Lanes outside the image pass 0xfffe in both halves: against a
// real rect d >= 0x7fff > any extent, against the empty one d = 0xffff > 0 -- no rect holds them (0xffff would match the
// empty rect: d = 0).
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool rect_has(int lo, int ext, int pxy)
{
    if (OPT_PKRECT) {
        const u16x2_t d = __builtin_bit_cast(u16x2_t, pxy) - __builtin_bit_cast(u16x2_t, lo);
        const u16x2_t m = __builtin_elementwise_min(d, __builtin_bit_cast(u16x2_t, ext));
        return __builtin_bit_cast(int, m) == __builtin_bit_cast(int, d);
    }
    const int px = pxy & 0xffff, py = (unsigned)pxy >> 16, x0 = lo & 0xffff, r0 = (unsigned)lo >> 16;
    return px >= x0 && px <= x0 + (ext & 0xffff) && py >= r0 && py <= r0 + (int)((unsigned)ext >> 16);
}

// Pixel-index bounds equivalent to the float test `lo <= pix_center(i) <= hi` (pix_center is monotone in i):
// an estimate from the inverse map, then fixed up with the exact function so the integer test decides
// exactly like the reference's float comparisons.  NaN bounds never reject (K.cu:33-38 compares with > / <).
__device__ __forceinline__ int first_pixel_ge(float lo, int IS)
{
    if (lo != lo) return 0;
    const float e = ceilf((lo * (float)IS + (float)IS - 1.f) * 0.5f);
    int i = e < 0.f ? 0 : (e > (float)IS ? IS : (int)e);
    for (int k = 0; k < 4 && i > 0 && pix_center(i - 1, IS) >= lo; k++) i--;
    for (int k = 0; k < 4 && i < IS && pix_center(i, IS) < lo; k++) i++;
    return i;                                   // in [0, IS]; IS = no pixel
}
__device__ __forceinline__ int last_pixel_le(float hi, int IS)
{
    if (hi != hi) return IS - 1;
    const float e = floorf((hi * (float)IS + (float)IS - 1.f) * 0.5f);
    int i = e < -1.f ? -1 : (e > (float)(IS - 1) ? IS - 1 : (int)e);
    for (int k = 0; k < 4 && i < IS - 1 && pix_center(i + 1, IS) <= hi; k++) i++;
    for (int k = 0; k < 4 && i >= 0 && pix_center(i, IS) > hi; k++) i--;
    return i;                                   // in [-1, IS-1]; -1 = no pixel
}

// same value, cheaper when the image size is a power of two (1/IS and the product are then exact)
__device__ __forceinline__ float pix_center_p2(int i, int is, float inv_is, bool pow2)
{
    return pow2 ? (float)(2 * i + 1 - is) * inv_is : pix_center(i, is);
}

__device__ __forceinline__ void build_record(const float* __restrict__ f, float* __restrict__ rec,
                                             short4* __restrict__ rect, float margin, int IS,
                                             float* __restrict__ info27)
{
    const float x0 = f[0], y0 = f[1], x1 = f[3], y1 = f[4], x2 = f[6], y2 = f[7];
    float adj[9];
    adj[0] = y1 - y2; adj[1] = x2 - x1; adj[2] = x1 * y2 - x2 * y1;
    adj[3] = y2 - y0; adj[4] = x0 - x2; adj[5] = x2 * y0 - x0 * y2;
    adj[6] = y0 - y1; adj[7] = x1 - x0; adj[8] = x0 * y1 - x1 * y0;
    float det = x2 * (y0 - y1) + x0 * (y1 - y2) + x1 * (y2 - y0);
    if (det > 0) { double d = (double)det; det = (float)(d > 1e-10 ? d : 1e-10); }
    else         { double d = (double)det; det = (float)(d < -1e-10 ? d : -1e-10); }
    float inv[9], sym[9];
#pragma unroll
    for (int k = 0; k < 9; k++) inv[k] = adj[k] / det;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int k = 0; k < 3; k++) sym[3 * j + k] = f[3 * j] * f[3 * k] + f[3 * j + 1] * f[3 * k + 1] + 1;
    int flags = 0;
    {
        const float px[3] = {x0, x1, x2}, py[3] = {y0, y1, y2};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int b = (k + 1) % 3, c = (k + 2) % 3;
            const bool obt = (px[b] - px[k]) * (px[c] - px[k]) + (py[b] - py[k]) * (py[c] - py[k]) < 0;
            if (obt && (flags & 7) == 0) flags |= 1 << k;
        }
    }
    if ((y2 - y0) * (x1 - x0) < (y1 - y0) * (x2 - x0)) flags |= 8;
#pragma unroll
    for (int k = 0; k < 9; k++) rec[R_FACE + k] = f[k];
#pragma unroll
    for (int k = 0; k < 9; k++) rec[R_INV + k] = inv[k];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int b = (k + 1) % 3;
        float e[3];
#pragma unroll
        for (int j = 0; j < 3; j++) { e[j] = sym[3 * k + j] - sym[3 * b + j]; rec[R_E + 3 * k + j] = e[j]; }
        rec[R_DEN + k] = e[k] - e[b];
    }
    {
        // heights^2 = det^2 / |opposite edge|^2 (unclamped det); a face is "well conditioned" when all
        // three heights exceed 1e-2 NDC (about a pixel at 256x256) and no edge is longer than 4 NDC, so that
        // the fp32 distance the reference computes for it is accurate to a small fraction of a percent
        const float det0 = x2 * (y0 - y1) + x0 * (y1 - y2) + x1 * (y2 - y0);
        const float px[3] = {x0, x1, x2}, py[3] = {y0, y1, y2};
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int b = (k + 1) % 3, c = (k + 2) % 3;
            const float ex = px[c] - px[b], ey = py[c] - py[b];
            const float l2 = ex * ex + ey * ey;
            const float h2 = l2 > 0.f ? det0 * det0 / l2 : 0.f;
            rec[R_HK2 + k] = h2;
            ok = ok && (h2 > 1e-4f) && (h2 < 1e4f) && (l2 < 16.f);
        }
        if (ok) flags |= 16;
    }
    rec[R_FLAGS] = __int_as_float(flags);
    // K.cu:33-38 with the max/min +- margin hoisted (same float ops, done once)
    const float xmax = fmaxf(fmaxf(x0, x1), x2) + margin, xmin = fminf(fminf(x0, x1), x2) - margin;
    const float ymax = fmaxf(fmaxf(y0, y1), y2) + margin, ymin = fminf(fminf(y0, y1), y2) - margin;
    int px0 = first_pixel_ge(xmin, IS), px1 = last_pixel_le(xmax, IS);
    const int yi0 = first_pixel_ge(ymin, IS), yi1 = last_pixel_le(ymax, IS);    // yi counts from the bottom (K.cu:343)
    int r0 = IS - 1 - yi1, r1 = IS - 1 - yi0;                                    // rows from the top
    if (px0 > px1 || r0 > r1) { px0 = 32767; px1 = -1; r0 = 32767; r1 = -1; }    // empty
    *rect = make_short4((short)px0, (short)px1, (short)r0, (short)r1);          // caller-owned (register or global)
    if (px0 > px1) { rec[R_BB + 0] = __int_as_float(-1); rec[R_BB + 1] = __int_as_float(0); }
    else {
        rec[R_BB + 0] = __int_as_float((px0 & 0xffff) | (r0 << 16));
        rec[R_BB + 1] = __int_as_float(((px1 - px0) & 0xffff) | ((r1 - r0) << 16));
    }
    {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float d = rec[R_DEN + k], z = f[3 * k + 2];
            rec[R_IDEN + k] = 1.f / d;
            rec[R_IZ + k] = 1.f / z;
            ok = ok && recip_safe(d) && z >= 1e-6f && z <= 1e6f;
        }
#pragma unroll
        for (int k = 0; k < 9; k++) ok = ok && fabsf(inv[k]) < 1e30f;          // false for NaN / inf as well
        if (ok) rec[R_FLAGS] = __int_as_float(__float_as_int(rec[R_FLAGS]) | 32);
        // rec[15]: the backward's lane -> (row, column) walk over the pixel rect, prepared here (one division per face instead of three
        // in every backward wave): bits 0..16 m = 65536 / bw + 1 for bw < 64 (lane / bw == (lane * m) >> 16 for lane < 64; 0: one row
        // holds the whole wave), bits 17..23 dr = 64 / bw
        {
            const int bw = px0 > px1 ? 1 : px1 - px0 + 1;
            const int mdiv = bw < 64 ? 65536 / bw + 1 : 0, dr = 64 / bw;
            rec[15] = __int_as_float(mdiv | (dr << 17));
        }
        rec[31] = 0.f;                                              // padding: defined bytes in the workspace
#pragma unroll
        for (int k = 44; k < REC; k++) rec[k] = 0.f;
    }
    if (info27) {   // reference layout, for callers that still want the tensor
#pragma unroll
        for (int k = 0; k < 9; k++) { info27[k] = inv[k]; info27[9 + k] = sym[k]; }
        info27[18] = (flags & 1) ? 1.f : 0.f; info27[19] = (flags & 2) ? 1.f : 0.f; info27[20] = (flags & 4) ? 1.f : 0.f;
#pragma unroll
        for (int k = 21; k < 27; k++) info27[k] = 0.f;
    }
}

__device__ __forceinline__ bool inside_closed(float w0, float w1, float w2)
{
    return w0 <= 1 && w0 >= 0 && w1 <= 1 && w1 >= 0 && w2 <= 1 && w2 >= 0;   // K.cu:47-50
}

// Three IEEE-754 quotients a_k / b with one divisor.  This is the instruction sequence the compiler emits for a
// float division (v_rcp_f32, one Newton step on the reciprocal, quotient, two residual corrections) with the
// divisor-only part shared: 18 VALU ops instead of 3 x 11.  v_div_scale / v_div_fmas / v_div_fixup are identities
// here -- the caller guarantees 1e-5 <= b <= 3 and a_k = 0 or 2^-100 <= a_k <= 1, so nothing is scaled and no special
// value occurs -- hence the quotients are bit-identical to a_k / b (lasr_selftest_div covers this path as well).
// (0 < a_k < 2^-100 would differ in the last denormal bit at most: a clipped barycentric of 1e-30.)
__device__ __forceinline__ void div3_shared(float& a0, float& a1, float& a2, float b)
{
    float y = __builtin_amdgcn_rcpf(b);
    y = __builtin_fmaf(__builtin_fmaf(-b, y, 1.f), y, y);
#define LASR_QUOT(a)                                                     \
    {                                                                    \
        float q = a * y;                                                 \
        q = __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);              \
        a = __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);              \
    }
    LASR_QUOT(a0) LASR_QUOT(a1) LASR_QUOT(a2)
#undef LASR_QUOT
}

// K.cu:53-58
// TAME: the barycentrics are known finite (record flag bit5, or the backward pass where NaN propagation is not pinned): the clamp
// is one v_med3_f32 (for finite w, med3(w, 0, 1) == max(min(w, 1), 0) up to the sign of a zero, which no later result depends on)
template <bool FM = false, bool TAME = false>
__device__ __forceinline__ void clip_normalise(float& w0, float& w1, float& w2)
{
    if (TAME) {
        w0 = __builtin_amdgcn_fmed3f(w0, 0.f, 1.f);
        w1 = __builtin_amdgcn_fmed3f(w1, 0.f, 1.f);
        w2 = __builtin_amdgcn_fmed3f(w2, 0.f, 1.f);
    } else {
        w0 = fmaxf(fminf(w0, 1.f), 0.f);     // == the reference's double-literal clamp (the bounds are exact)
        w1 = fmaxf(fminf(w1, 1.f), 0.f);
        w2 = fmaxf(fminf(w2, 1.f), 0.f);
    }
    const float s = fmaxf(w0 + w1 + w2, 1e-5f);   // (float)max((double)s, 1e-5) == fmaxf(s, 1e-5f) for every float s
    if (FM) { const float r = __builtin_amdgcn_rcpf(s); w0 *= r; w1 *= r; w2 *= r; }
    else div3_shared(w0, w1, w2, s);
}

// ---- math flavours -----------------------------------------------------------------
// FM = false : the reference's rounding sequence (forward pass: parity to ~1 ulp of expf)
// FM = true  : v_rcp_f32 / v_exp_f32 based (backward pass only: the reference's own backward sums
//              its terms with unordered float atomics, the gradient bar is relative 1e-3)
template <bool FM> __device__ __forceinline__ float div_(float a, float b)
{
    return FM ? a * __builtin_amdgcn_rcpf(b) : a / b;
}
// exp(x) to ~1 ulp in 7 VALU ops: n = rint(x log2 e), f = x log2 e - n evaluated with a two-term log2 e (so the
// reduction error does not grow with |x|), 2^f by v_exp_f32 (|f| <= 0.5), scaled by 2^n with v_ldexp_f32.
// The library expf spends about twice that on special cases that cannot occur here (|x| < ~110, results that
// underflow simply flush towards 0 as exp() itself does).
__device__ __forceinline__ float exp_1ulp(float x)
{
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-8f;
    const float n = rintf(x * L2E_HI);
    const float f = __builtin_fmaf(x, L2E_LO, __builtin_fmaf(x, L2E_HI, -n));
    return ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}
template <bool FM> __device__ __forceinline__ float exp_(float x) { return FM ? __expf(x) : exp_1ulp(x); }
// NOSCALE: the caller guarantees neg_arg < 80, so 1 + e is a double in [1, 2^116]
template <bool FM, bool NOSCALE = false> __device__ __forceinline__ float sigmoid_neg_(float neg_arg)
{
    if (FM) return __builtin_amdgcn_rcpf(1.f + __expf(neg_arg));
    if (NOSCALE) return (float)recip64_noscale(1. + (double)exp_1ulp(neg_arg));
    return (float)(1. / (1. + (double)exp_1ulp(neg_arg)));   // K.cu:397,403 promote this to double
}

// Conservative reject for the compaction stages: true only if the pixel is certainly farther than
// sqrt(thr) from the face (so the reference's `dis >= threshold` test would skip it as well).
template <typename RP>
__device__ __forceinline__ bool certainly_far(RP rec, float w0, float w1, float w2, float thr_pad)
{
    return (w0 < 0.f && w0 * w0 * rec[R_HK2 + 0] > thr_pad) ||
           (w1 < 0.f && w1 * w1 * rec[R_HK2 + 1] > thr_pad) ||
           (w2 < 0.f && w2 * w2 * rec[R_HK2 + 2] > thr_pad);
}

// One edge projection with compile-time edge K (a=K, b=K+1, c=K+2 mod 3): K.cu:85-95 / 136-148.
// Returns u (already minus w) in (u0,u1,u2).  CLAMP selects the outside-branch variant.
template <int K, bool CLAMP, bool FM = false, bool MK = false, typename RP = cptr_t, bool IDEN = false>
__device__ __forceinline__ void edge_project(RP rec, float w0, float w1, float w2,
                                             float& u0, float& u1, float& u2)
{
    // The reference's point-to-face distance is ill-conditioned on edge-on faces (a fused multiply-add in `num` moves the
    // result by percents there), and the backward pass must re-derive the SAME distance the forward pass turned into D.  So
    // this arithmetic is pinned to separate multiplies and adds in every translation unit; sr_backward_fast.hip
    // (-ffp-contract=fast-honor-pragmas) fuses only the well-conditioned gradient arithmetic around it.
#pragma clang fp contract(off)
    constexpr int B = (K + 1) % 3;
    const float e0 = rec[R_E + 3 * K + 0], e1 = rec[R_E + 3 * K + 1], e2 = rec[R_E + 3 * K + 2];
    const float eb = rec[R_E + 3 * K + B];
    const float num = w0 * e0 + w1 * e1 + w2 * e2 - eb;
    float ta = MK ? div_by_recip(num, rec[R_DEN + K], rec[R_IDEN + K])
                  : (FM && IDEN) ? num * rec[R_IDEN + K] : div_<FM>(num, rec[R_DEN + K]);     // IDEN: the record's RN(1 / den)
    float tb = 1 - ta;
    float tc = 0;
    if (CLAMP) {
        ta = fminf(fmaxf(ta, 0.f), 1.f);
        tb = fminf(fmaxf(tb, 0.f), 1.f);
    }
    float t[3];
    t[K] = ta; t[B] = tb; t[(K + 2) % 3] = tc;
    u0 = t[0] - w0; u1 = t[1] - w1; u2 = t[2] - w2;
}

struct Frag {
    float D;            // fragment probability
    float sign, dx, dy; // euclidean: signed displacement
    float t0, t1, t2;   // euclidean/barycentric: saved for backward
    float dis;
    float narg;         // euclidean, forward-only form (euclid<.., FWD = true>): -sign * dis, the numerator of the sigmoid's exponent
};

// Euclidean point-to-face distance: K.cu:61-151
// FWD (forward pass, tame records): the caller only needs D, so the two branches leave -sign * dis in fr.narg themselves (the
// negation folds into the last subtraction: -(a + b) == (-a) - b bit for bit) and the `sign < 0 && dis >= thr` cut of
// K.cu:402 is taken inside the outside branch -- no float sign, one compare less.  Returns false when the face is cut.
// TAME: w0..w2 are finite, so the strict-inside test is min3 > 0 && max3 < 1.
template <bool FM = false, bool MK = false, typename RP = cptr_t, bool FWD = false, bool TAME = false, bool IDEN = false>
__device__ __forceinline__ bool euclid(RP rec, float xp, float yp,
                                       float w0, float w1, float w2, Frag& fr, float thr = 0.f)
{
#pragma clang fp contract(off)   // see edge_project
    const float x0 = rec[R_FACE + 0], y0 = rec[R_FACE + 1], x1 = rec[R_FACE + 3], y1 = rec[R_FACE + 4], x2 = rec[R_FACE + 6], y2 = rec[R_FACE + 7];
    const bool inside = TAME ? (bool)((int)(fminf(fminf(w0, w1), w2) > 0) & (int)(fmaxf(fmaxf(w0, w1), w2) < 1))     // (no short-circuit branch)
                             : (w0 > 0 && w1 > 0 && w2 > 0 && w0 < 1 && w1 < 1 && w2 < 1);
    if (inside) {
        float best = 100000000.f, bx = 0, by = 0, b0 = 0, b1 = 0, b2 = 0;
        float u0, u1, u2;
#define LASR_TRY_EDGE(K)                                                          \
        edge_project<K, false, FM, MK, RP, IDEN>(rec, w0, w1, w2, u0, u1, u2);            \
        {                                                                         \
            const float px = u0 * x0 + u1 * x1 + u2 * x2;                         \
            const float py = u0 * y0 + u1 * y1 + u2 * y2;                         \
            const float d2 = px * px + py * py;                                   \
            if (d2 < best) { best = d2; bx = px; by = py; b0 = u0; b1 = u1; b2 = u2; } \
        }
        LASR_TRY_EDGE(0) LASR_TRY_EDGE(1) LASR_TRY_EDGE(2)
#undef LASR_TRY_EDGE
        if (FWD) { fr.narg = (-bx) * bx - by * by; return true; }
        fr.dx = bx; fr.dy = by; fr.t0 = b0; fr.t1 = b1; fr.t2 = b2; fr.sign = 1.f;
    } else {
        const int flags = __float_as_int(rec[R_FLAGS]);
        float u0, u1, u2;
        if (OPT_EDGESEL) {
            // K.cu:113-125 as lane-mask algebra instead of a nested if chain (which the compiler turns into a dozen exec-mask
            // regions per entry): which edge a pixel outside the face projects to follows from the sign pattern of its
            // barycentrics, with the obtuse-corner override; every lane takes exactly one of the three projections
            const bool n0 = w0 <= 0, n1 = w1 <= 0, n2 = w2 <= 0;
            bool o0 = false, o1 = false, o2 = false;                 // at most one corner of a face is obtuse (wave-uniform flags)
            if (flags & 7) {
                if (flags & 1) o0 = (xp - x0) * (x2 - x0) + (yp - y0) * (y2 - y0) > 0;
                if (flags & 2) o1 = (xp - x1) * (x0 - x1) + (yp - y1) * (y0 - y1) > 0;
                if (flags & 4) o2 = (xp - x2) * (x1 - x2) + (yp - y2) * (y1 - y2) > 0;
            }
            const bool c12 = n1 & n2, c20 = n2 & n0 & !n1, c01 = n0 & n1 & !n2;        // two (or three) non-positive: a corner region
            const bool e1 = (c20 & !o1) | (c01 & o2) | (n0 & !n1 & !n2);
            const bool e2 = (c01 & !o2) | (c12 & o0) | (n1 & !n0 & !n2);
            const bool e0 = !(e1 | e2);             // incl. "none <= 0" (the reference indexes [-1] there: UB; pinned to edge 0 like the oracle)
            if (e0) edge_project<0, true, FM, MK, RP, IDEN>(rec, w0, w1, w2, u0, u1, u2);
            if (e1) edge_project<1, true, FM, MK, RP, IDEN>(rec, w0, w1, w2, u0, u1, u2);
            if (e2) edge_project<2, true, FM, MK, RP, IDEN>(rec, w0, w1, w2, u0, u1, u2);
        } else {
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
        if (a < 0) a = 0;   // reference indexes [-1] here (UB); pinned to edge 0 like the oracle
        if (a == 0) edge_project<0, true, FM, MK, RP, IDEN>(rec, w0, w1, w2, u0, u1, u2);
        else if (a == 1) edge_project<1, true, FM, MK, RP, IDEN>(rec, w0, w1, w2, u0, u1, u2);
        else edge_project<2, true, FM, MK, RP, IDEN>(rec, w0, w1, w2, u0, u1, u2);
        }
        fr.dx = u0 * x0 + u1 * x1 + u2 * x2;
        fr.dy = u0 * y0 + u1 * y1 + u2 * y2;
        if (FWD) { fr.narg = fr.dx * fr.dx + fr.dy * fr.dy; return !(fr.narg >= thr); }
        fr.t0 = u0; fr.t1 = u1; fr.t2 = u2; fr.sign = -1.f;
    }
    return true;
}

// Well-conditioned faces (flags bit 4), euclidean distance: ONE edge projection per pixel, inside or outside, wherever the choice of
// the edge is clear.  K.cu:61-110 projects an inside pixel on all three edge LINES (no clamp) and keeps the nearest: that is the
// line with the smallest perpendicular distance d_k = w_k h_k (h_k = height of vertex k over its opposite edge, hk2 = h_k^2 in
// the record).  The three products q = w_k^2 hk2_k cost six multiplications; when the smallest is below the second smallest by
// more than NEAR_TIE and by more than the face's rounding scale (near_tie below: the reference's own projections are only accurate
// to a fraction of a percent on small faces, so its choice near an angle bisector is decided by its rounding), only that projection
// is evaluated -- by the same instructions on the
// same barycentrics as edge_project<k>, hence with the bits the three-projection form gives for that edge.  The clamp of the
// outside branch is a no-op for an inside pixel (the foot of the perpendicular on the nearest line lies on the triangle's
// boundary), so both kinds of pixel share the three exec-masked projections.  Inside pixels near a bisector take the reference's
// three projections (a region most batches skip).  Why: in the bench launch 15 % of the surviving pixels of a face are inside and
// nearly every batch holds some -- the three-projection branch ran for every batch at ~10 live lanes.
constexpr float NEAR_TIE = 0.985f;
// ... and an ABSOLUTE margin on the distances: the reference's projection p - foot = sum u_k v_k carries the rounding error of its
// barycentrics (~2^-24 |inv| |coords| ~ 2e-7 / h_min each) times the vertex coordinates, ~6e-7 / h_min in the position
// (h_min = the face's smallest height); which of two lines it finds nearer is its rounding's choice while
// |d_mid - d_lo| < E = NEAR_TIE_ABS / h_min (four times that estimate).  As squared distances, without a root:
// (q_mid - q_lo)^2 < 4 E^2 q_mid  (since q_mid - q_lo = (d_mid - d_lo)(d_mid + d_lo) <= 2 d_mid (d_mid - d_lo)).
// near_tie_scale(hk2) = 4 E^2, per face.  Measured need: with the relative margin alone the image of LASR's own meshes (slivers at
// the silhouette) moved by up to 5e-5 against the reference build (profiles/r06_whole_forward_parity.jsonl, first version).
constexpr float NEAR_TIE_ABS = 2.4e-6f;
__host__ __device__ __forceinline__ float near_tie_scale(float h2a, float h2b, float h2c)
{
    return 4.f * NEAR_TIE_ABS * NEAR_TIE_ABS / fminf(fminf(h2a, h2b), h2c);
}
__device__ __forceinline__ bool near_tie(float qlo, float qmid, float scale)
{
    const float gap = qmid - qlo;
    return (bool)((int)!(qlo < NEAR_TIE * qmid) | (int)!(gap * gap > scale * qmid));
}
#ifndef LASR_BWD_FMA
#define LASR_BWD_FMA 0
#endif
// edge_project<K, CLAMP = true, FM = true, IDEN = true> for the backward's single projection; with LASR_BWD_FMA its multiply-add
// pairs may fuse (the translation unit's -ffp-contract=fast): a well-conditioned face's distance moves by ~1e-6 relative
template <int K, typename RP>
__device__ __forceinline__ void edge_project_one(RP rec, float w0, float w1, float w2, float& u0, float& u1, float& u2)
{
#if !LASR_BWD_FMA
#pragma clang fp contract(off)
#endif
    constexpr int B = (K + 1) % 3;
    const float num = w0 * rec[R_E + 3 * K + 0] + w1 * rec[R_E + 3 * K + 1] + w2 * rec[R_E + 3 * K + 2] - rec[R_E + 3 * K + B];
    float ta = num * rec[R_IDEN + K];
    float tb = 1 - ta;
    ta = fminf(fmaxf(ta, 0.f), 1.f);
    tb = fminf(fmaxf(tb, 0.f), 1.f);
    float t[3];
    t[K] = ta; t[B] = tb; t[(K + 2) % 3] = 0;
    u0 = t[0] - w0; u1 = t[1] - w1; u2 = t[2] - w2;
}
template <typename RP>
__device__ __forceinline__ void euclid_one(RP rec, float xp, float yp, float w0, float w1, float w2, Frag& fr, float tie_scale)
{
#if !LASR_BWD_FMA
#pragma clang fp contract(off)   // see edge_project
#endif
    const float x0 = rec[R_FACE + 0], y0 = rec[R_FACE + 1], x1 = rec[R_FACE + 3], y1 = rec[R_FACE + 4], x2 = rec[R_FACE + 6], y2 = rec[R_FACE + 7];
    const bool inside = (bool)((int)(fminf(fminf(w0, w1), w2) > 0) & (int)(fmaxf(fmaxf(w0, w1), w2) < 1));
    const int flags = __float_as_int(rec[R_FLAGS]);
    const bool n0 = w0 <= 0, n1 = w1 <= 0, n2 = w2 <= 0;
    bool o0 = false, o1 = false, o2 = false;
    if (flags & 7) {
        if (flags & 1) o0 = (xp - x0) * (x2 - x0) + (yp - y0) * (y2 - y0) > 0;
        if (flags & 2) o1 = (xp - x1) * (x0 - x1) + (yp - y1) * (y0 - y1) > 0;
        if (flags & 4) o2 = (xp - x2) * (x1 - x2) + (yp - y2) * (y1 - y2) > 0;
    }
    const bool c12 = n1 & n2, c20 = n2 & n0 & !n1, c01 = n0 & n1 & !n2;
    bool e1 = (c20 & !o1) | (c01 & o2) | (n0 & !n1 & !n2);
    bool e2 = (c01 & !o2) | (c12 & o0) | (n1 & !n0 & !n2);
    // inside: edge k (from vertex k to k + 1, edge_project<k>) is the one opposite vertex (k + 2) % 3, at distance w_{k+2} h_{k+2}
    const float q0 = w2 * w2 * rec[R_HK2 + 2], q1 = w0 * w0 * rec[R_HK2 + 0], q2 = w1 * w1 * rec[R_HK2 + 1];   // edge 0, 1, 2
    const float qlo = fminf(fminf(q0, q1), q2), qmid = __builtin_amdgcn_fmed3f(q0, q1, q2);
    const bool tie = (bool)((int)inside & (int)near_tie(qlo, qmid, tie_scale));
    float u0, u1, u2;
    if (tie) {
        float best = 100000000.f, bx = 0, by = 0, b0 = 0, b1 = 0, b2 = 0;
#define LASR_TRY_EDGE(K)                                                          \
        edge_project<K, false, true, false, RP, true>(rec, w0, w1, w2, u0, u1, u2);   \
        {                                                                         \
            const float px = u0 * x0 + u1 * x1 + u2 * x2;                         \
            const float py = u0 * y0 + u1 * y1 + u2 * y2;                         \
            const float d2 = px * px + py * py;                                   \
            if (d2 < best) { best = d2; bx = px; by = py; b0 = u0; b1 = u1; b2 = u2; } \
        }
        LASR_TRY_EDGE(0) LASR_TRY_EDGE(1) LASR_TRY_EDGE(2)
#undef LASR_TRY_EDGE
        fr.dx = bx; fr.dy = by; u0 = b0; u1 = b1; u2 = b2;
    } else {
        const bool i1 = q1 == qlo, i2 = q2 == qlo;      // (clear of a tie: exactly one)
        e1 = inside ? i1 : e1;
        e2 = inside ? i2 : e2;
        const bool e0 = !(e1 | e2);
        if (e0) edge_project_one<0>(rec, w0, w1, w2, u0, u1, u2);
        if (e1) edge_project_one<1>(rec, w0, w1, w2, u0, u1, u2);
        if (e2) edge_project_one<2>(rec, w0, w1, w2, u0, u1, u2);
        fr.dx = u0 * x0 + u1 * x1 + u2 * x2;
        fr.dy = u0 * y0 + u1 * y1 + u2 * y2;
    }
    fr.t0 = u0; fr.t1 = u1; fr.t2 = u2; fr.sign = inside ? 1.f : -1.f;
}

// Fragment probability of the face in `rec` at (xp,yp): K.cu:387-404.  false = face skipped.
template <typename RP>
__device__ __forceinline__ void barycentric(RP rec, float xp, float yp, float& w0, float& w1, float& w2)
{
#pragma clang fp contract(off)   // distance code: never fused, also in the contracted backward unit (see edge_project)
    w0 = rec[R_INV + 0] * xp + rec[R_INV + 1] * yp + rec[R_INV + 2];   // K.cu:24-29
    w1 = rec[R_INV + 3] * xp + rec[R_INV + 4] * yp + rec[R_INV + 5];
    w2 = rec[R_INV + 6] * xp + rec[R_INV + 7] * yp + rec[R_INV + 8];
}

template <bool FM = false, bool MK = false, typename RP = cptr_t, bool BT = false>
__device__ __forceinline__ bool fragment_w(RP rec, int dist, float thr, float sigma,
                                           float xp, float yp, float w0, float w1, float w2, Frag& fr,
                                           float inv_sigma = 0.f);

// BT (backward pass, LASR's modes): NaN propagation is not pinned there (gradient bar 1e-3 relative), so the inside test runs
// as min3 / max3 and the edge projection multiplies by the record's 1 / den instead of a v_rcp per fragment
template <bool FM = false, typename RP = cptr_t, bool BT = false>
__device__ __forceinline__ bool fragment(RP rec, int dist, float thr, float sigma,
                                         float xp, float yp, float& w0, float& w1, float& w2, Frag& fr)
{
    barycentric(rec, xp, yp, w0, w1, w2);
    return fragment_w<FM, false, RP, BT>(rec, dist, thr, sigma, xp, yp, w0, w1, w2, fr);
}

template <bool FM, bool MK, typename RP, bool BT>
__device__ __forceinline__ bool fragment_w(RP rec, int dist, float thr, float sigma,
                                           float xp, float yp, float w0, float w1, float w2, Frag& fr, float inv_sigma)
{
#pragma clang fp contract(off)   // see edge_project
    if (dist == 0) {
        if (!inside_closed(w0, w1, w2)) return false;
        fr.D = 1.f;
    } else if (dist == 1) {
        float d = w0 > w1 ? (w1 > w2 ? w2 : w1) : (w0 > w2 ? w2 : w0);      // K.cu:155-158
        d = d > 0 ? d * d : -(d * d);
        fr.dis = d; fr.t0 = w0; fr.t1 = w1; fr.t2 = w2;
        if (-d >= thr) return false;
        fr.D = sigmoid_neg_<FM>(MK ? div_by_recip(-d, sigma, inv_sigma) : div_<FM>(-d, sigma));
    } else if (!FM && MK && OPT_SIGNFOLD) {
        if (!euclid<FM, MK, RP, true, OPT_MED3>(rec, xp, yp, w0, w1, w2, fr, thr)) return false;
        fr.D = sigmoid_neg_<FM, OPT_NOSCALE>(div_by_recip(fr.narg, sigma, inv_sigma));
    } else {
        euclid<FM, MK, RP, false, (!FM && MK && OPT_MED3) || BT, BT>(rec, xp, yp, w0, w1, w2, fr);
        fr.dis = fr.dx * fr.dx + fr.dy * fr.dy;
        if (fr.sign < 0 && fr.dis >= thr) return false;
        fr.D = sigmoid_neg_<FM, (!FM && MK && OPT_NOSCALE)>(MK ? div_by_recip(-fr.sign * fr.dis, sigma, inv_sigma) : div_<FM>(-fr.sign * fr.dis, sigma));
    }
    return true;
}

// fragment<FM = true, .., BT = true> for a well-conditioned face in the euclidean mode, on euclid_one
template <typename RP>
__device__ __forceinline__ bool fragment_one(RP rec, float thr, float sigma, float xp, float yp, float& w0, float& w1, float& w2, Frag& fr,
                                             float tie_scale)
{
#if LASR_BWD_FMA
    w0 = rec[R_INV + 0] * xp + rec[R_INV + 1] * yp + rec[R_INV + 2];   // K.cu:24-29, contractable
    w1 = rec[R_INV + 3] * xp + rec[R_INV + 4] * yp + rec[R_INV + 5];
    w2 = rec[R_INV + 6] * xp + rec[R_INV + 7] * yp + rec[R_INV + 8];
#else
#pragma clang fp contract(off)   // see edge_project
    barycentric(rec, xp, yp, w0, w1, w2);
#endif
    euclid_one(rec, xp, yp, w0, w1, w2, fr, tie_scale);
    fr.dis = fr.dx * fr.dx + fr.dy * fr.dy;
    if (fr.sign < 0 && fr.dis >= thr) return false;
    fr.D = sigmoid_neg_<true>(div_<true>(-fr.sign * fr.dis, sigma));
    return true;
}

// K.cu:423 (1. / float-sum evaluated in double; narrowing a double quotient of
// floats is the correctly rounded float quotient, so a float division is identical)
template <bool FM = false, bool MK = false, typename RP = cptr_t>
__device__ __forceinline__ float depth_at(RP rec, float c0, float c1, float c2)
{
    if (MK) {
        // tame record: 0 <= c_k <= 1 with sum >= 1e-5-ish and 1e-6 <= z_k <= 1e6, so the sum lies in [1e-11, 3e6]: no scaling
        const float s = div_by_recip(c0, rec[R_FACE + 2], rec[R_IZ + 0]) + div_by_recip(c1, rec[R_FACE + 5], rec[R_IZ + 1]) +
                        div_by_recip(c2, rec[R_FACE + 8], rec[R_IZ + 2]);
        return OPT_NOSCALE ? recip_noscale(s) : 1.f / s;
    }
    return div_<FM>(1.f, div_<FM>(c0, rec[R_FACE + 2]) + div_<FM>(c1, rec[R_FACE + 5]) + div_<FM>(c2, rec[R_FACE + 8]));
}

// The forward pass's depth of a pixel on a face, bit for bit (barycentrics K.cu:24-29, clip/normalise :53-58, depth :423),
// whatever contraction setting the including translation unit uses.  The backward pass evaluates it only for fragments whose
// fast-math depth falls within rounding distance of the near / far planes, so that "this fragment was depth-culled in the
// forward pass" (K.cu:424 / :599: no gradient at all) is decided from the same number in both passes.
template <typename RP>
__device__ __forceinline__ float depth_forward_exact(RP rec, float xp, float yp)
{
#pragma clang fp contract(off)
    float w0 = rec[R_INV + 0] * xp + rec[R_INV + 1] * yp + rec[R_INV + 2];
    float w1 = rec[R_INV + 3] * xp + rec[R_INV + 4] * yp + rec[R_INV + 5];
    float w2 = rec[R_INV + 6] * xp + rec[R_INV + 7] * yp + rec[R_INV + 8];
    w0 = fmaxf(fminf(w0, 1.f), 0.f);
    w1 = fmaxf(fminf(w1, 1.f), 0.f);
    w2 = fmaxf(fminf(w2, 1.f), 0.f);
    const float s = fmaxf(w0 + w1 + w2, 1e-5f);
    w0 = w0 / s; w1 = w1 / s; w2 = w2 / s;                  // == div3_shared (bit-identical quotients, self-tested)
    return 1.f / (w0 / rec[R_FACE + 2] + w1 / rec[R_FACE + 5] + w2 / rec[R_FACE + 8]);  // == the reciprocal forms of depth_at (correctly rounded)
}

// texel index a surface sample lands in (K.cu:181-188 == 200-211).  A clipped barycentric
// of exactly 1 yields ix (or iy) == res, i.e. an index >= T that runs into the next
// face's texels: the reference reads it unchecked, and its backward (a loop over
// j < T) then finds no texel to credit.  Both behaviours are kept; `lim` (texels
// from this face to the end of the tensor) keeps the read inside the allocation.
__device__ __forceinline__ int surface_texel(float c0, float c1, int res)
{
    const int ix = (int)(c0 * res), iy = (int)(c1 * res);
    if ((c0 + c1) * res - ix - iy <= 1) return iy * res + ix;
    return (res - 1 - iy) * res + (res - 1 - ix);
}

// K.cu:178-194
template <typename TP>
__device__ __forceinline__ float sample_colour(TP tex, float c0, float c1, float c2,
                                               int res, int ch, int tex_type, int lim, int nch = 3)
{
    if (tex_type == 0) {
        const int j = max(min(surface_texel(c0, c1, res), lim - 1), 0);
        return tex[j * 3 + ch];
    }
    return c0 * tex[ch] + c1 * tex[nch + ch] + c2 * tex[2 * nch + ch];   // vertex attributes, [3 vertices][nch channels]
}

// ---- wave64 helpers ----------------------------------------------------------
// Lane mask of a per-lane condition, straight from the condition's own mask (HIP's __ballot(int) first turns the bool into a
// 0 / 1 register and compares it again), and the number of set bits below the calling lane as v_mbcnt_lo / v_mbcnt_hi (two
// instructions; the shift-and-popcount form needs a per-lane 64-bit mask and four).
__device__ __forceinline__ unsigned long long wave_mask(bool c) { return __builtin_amdgcn_ballot_w64(c); }
__device__ __forceinline__ int bits_below_lane(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Sum over the 64 lanes of a wave with DPP row operations (no LDS traffic);
// the total lands in lane 63.
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
#define LASR_DPP_ADD(ctrl, rmask, bmask)                                                              \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, bmask, false))
    LASR_DPP_ADD(0x111, 0xf, 0xf);   // row_shr:1
    LASR_DPP_ADD(0x112, 0xf, 0xf);   // row_shr:2
    LASR_DPP_ADD(0x114, 0xf, 0xe);   // row_shr:4
    LASR_DPP_ADD(0x118, 0xf, 0xc);   // row_shr:8
    LASR_DPP_ADD(0x142, 0xa, 0xf);   // row_bcast:15
    LASR_DPP_ADD(0x143, 0xc, 0xf);   // row_bcast:31
#undef LASR_DPP_ADD
    return v;
}

// Reduce 18 per-lane partials across the wave in 68 VALU ops instead of 18 x 12 (one DPP tree each):
//   v_permlane32_swap folds two values across the wave halves at once, v_permlane16_swap across row pairs;
//   the surviving 5 registers hold, per 16-lane row, the partial sums of different components and finish with
//   one DPP row tree each.  Component c ends in lane out_lane(c) of register out_reg(c):
//     reg = c / 4, row = {0,2,1,3}[c % 4]  (c = 16,17: reg 4, rows 0 and 2), lane = 16 * row + 15.
__device__ __forceinline__ float swap_add32(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);     // lanes 0-31: sum_halves(a), lanes 32-63: sum_halves(b)
}
__device__ __forceinline__ float swap_add16(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);     // rows 0,2: a's row pairs; rows 1,3: b's row pairs
}
__device__ __forceinline__ float row_sum_to_lane15(float v)
{
#define LASR_DPP_ADD(ctrl, bmask) \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, bmask, false))
    LASR_DPP_ADD(0x111, 0xf);   // row_shr:1
    LASR_DPP_ADD(0x112, 0xf);   // row_shr:2
    LASR_DPP_ADD(0x114, 0xe);   // row_shr:4
    LASR_DPP_ADD(0x118, 0xc);   // row_shr:8
#undef LASR_DPP_ADD
    return v;                   // lane 15 of every row holds that row's sum
}
// v[0..17] -> out[0..4]; see the layout above
__device__ __forceinline__ void wave_reduce18(const float (&v)[18], float (&out)[5])
{
    float s[9];
#pragma unroll
    for (int i = 0; i < 9; i++) s[i] = swap_add32(v[2 * i], v[2 * i + 1]);
    // after level 32: lower half of s[i] = partials of v[2i], upper half = partials of v[2i+1]
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = row_sum_to_lane15(swap_add16(s[2 * i], s[2 * i + 1]));
    out[4] = row_sum_to_lane15(swap_add16(s[8], 0.f));
    // out[i]: row0 = v[4i], row1 = v[4i+2], row2 = v[4i+1], row3 = v[4i+3]   (i = 4: row0 = v[16], row2 = v[17])
}

}  // namespace lasr
