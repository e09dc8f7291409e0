// sr_fp64.hip -- the soft-rasteriser for float64 tensors (every mode combination, 3 colour channels).
//
// The reference dispatches its kernels on the tensor type (AT_DISPATCH_FLOATING_TYPES, K.cu:701,716,780: scalar_t = float or
// double, the scalar arguments near / far / eps / sigma / gamma stay float).  LASR itself only ever renders fp32, so this path is
// about completeness of the operator, not speed: no tile binning, no records in the scalar cache, no reciprocal tricks -- the
// arithmetic of K.cu:245-668 in double, in the reference's operation order, with the shape that is natural on this machine for a
// brute-force pass:
//   setup    1 thread / face: adj([x y 1]) / det, the Gram matrix + 1, the first obtuse corner (the reference's 27-slot layout,
//            which doubles as the caller-visible `faces_info`)
//   forward  1 workgroup / 16x16-pixel tile, lane = pixel; the image's faces stream through LDS 64 at a time (coalesced, one
//            fetch per workgroup instead of one per pixel) and every pixel walks them IN INDEX ORDER -- same accumulation order
//            as the reference, so the hard-mode index map is identical and the soft modes agree to rounding
//   backward the same walk per pixel; the 9 + 3 T gradient components of a (pixel, face) pair go out as hardware fp64 atomics
//            (the reference does the same; its order of additions is as undefined as this one's)
// Entry points: lasr_sr_forward_f64 / lasr_sr_backward_f64 (include/lasr_sr.h).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lasr_sr.h"
#include "host_common.h"

namespace lasr64 {

constexpr int INFO = 27;        // per-face: inv[9] | sym[9] | obt[3] | 6 unused   (K.cu:245-305)
constexpr int CHUNK = 64;       // faces staged per LDS round

struct Args {
    int N, F, T, res, IS;
    float near, far, eps, sigma, gamma, thr;      // float scalars, as in the reference's kernel signature
    int dist, rgb, alpha, tex, double_side;
};

__global__ __launch_bounds__(256) void setup_kernel(const double* __restrict__ faces, double* __restrict__ info, int total)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const double* f = faces + (size_t)i * 9;
    double* o = info + (size_t)i * INFO;
    const double x0 = f[0], y0 = f[1], x1 = f[3], y1 = f[4], x2 = f[6], y2 = f[7];
    double adj[9] = {y1 - y2, x2 - x1, x1 * y2 - x2 * y1, y2 - y0, x0 - x2, x2 * y0 - x0 * y2, y0 - y1, x1 - x0, x0 * y1 - x1 * y0};
    double det = x2 * (y0 - y1) + x0 * (y1 - y2) + x1 * (y2 - y0);
    det = det > 0 ? fmax(det, 1e-10) : fmin(det, -1e-10);                                  // K.cu:282
    for (int k = 0; k < 9; k++) o[k] = adj[k] / det;
    for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++) o[9 + 3 * j + k] = f[3 * j] * f[3 * k] + f[3 * j + 1] * f[3 * k + 1] + 1;
    const double px[3] = {x0, x1, x2}, py[3] = {y0, y1, y2};
    o[18] = o[19] = o[20] = 0;
    for (int k = 0; k < 3; k++) {                                                          // K.cu:296-304: the first obtuse corner
        const int b = (k + 1) % 3, c = (k + 2) % 3;
        if ((px[b] - px[k]) * (px[c] - px[k]) + (py[b] - py[k]) * (py[c] - py[k]) < 0) { o[18 + k] = 1; break; }
    }
    for (int k = 21; k < INFO; k++) o[k] = 0;
}

struct FaceD {          // one staged face: vertices + info
    double v[9];
    double inv[9];
    double sym[9];
    int obt;            // bit k: corner k is the flagged obtuse one
};

// point-to-face displacement in the image plane, K.cu:61-151.  Returns the sign (+1 strictly inside, -1 otherwise);
// (dx, dy) the displacement, t[3] the barycentric offsets (t - w) of the closest point.
__device__ inline double euclid64(const FaceD& fc, double xp, double yp, const double w[3], double& dx, double& dy, double t[3])
{
    const double* f = fc.v;
    const double* sym = fc.sym;
    if (w[0] > 0 && w[1] > 0 && w[2] > 0 && w[0] < 1 && w[1] < 1 && w[2] < 1) {
        double best = 100000000., bx = 0, by = 0, bt[3] = {0, 0, 0};
        for (int ka = 0; ka < 3; ka++) {               // all three edges, unclamped (K.cu:81-95): the nearest one wins
            const int kb = (ka + 1) % 3, kc = (ka + 2) % 3;
            double ds[3], u[3];
            for (int j = 0; j < 3; j++) ds[j] = sym[3 * ka + j] - sym[3 * kb + j];
            const double along = (w[0] * ds[0] + w[1] * ds[1] + w[2] * ds[2] - ds[kb]) / (ds[ka] - ds[kb]);
            u[ka] = along - w[ka]; u[kb] = (1 - along) - w[kb]; u[kc] = 0 - w[kc];
            const double qx = u[0] * f[0] + u[1] * f[3] + u[2] * f[6];
            const double qy = u[0] * f[1] + u[1] * f[4] + u[2] * f[7];
            const double d2 = qx * qx + qy * qy;
            if (d2 < best) { best = d2; bx = qx; by = qy; bt[0] = u[0]; bt[1] = u[1]; bt[2] = u[2]; }
        }
        dx = bx; dy = by; t[0] = bt[0]; t[1] = bt[1]; t[2] = bt[2];
        return 1.;
    }
    // Which edge the pixel projects to (K.cu:113-125), in the form the fp32 kernels use: without an obtuse corner the sign pattern
    // of the barycentrics decides -- edge 1 iff w0 <= 0 < w1, edge 2 iff w1 <= 0 < w2, edge 0 otherwise (incl. "none <= 0", where
    // the reference indexes [-1]: undefined there, pinned to edge 0 like the fp32 path and the oracle) -- and beyond a flagged
    // obtuse corner the pixel goes to the corner's OTHER edge when it lies on that side.
    const bool n0 = w[0] <= 0, n1 = w[1] <= 0, n2 = w[2] <= 0;
    int ka = (n0 && !n1) ? 1 : ((n1 && !n2) ? 2 : 0);
    if (n1 && n2) { if ((fc.obt & 1) && (xp - f[0]) * (f[6] - f[0]) + (yp - f[1]) * (f[7] - f[1]) > 0) ka = 2; }
    else if (n2 && n0) { if ((fc.obt & 2) && (xp - f[3]) * (f[0] - f[3]) + (yp - f[4]) * (f[1] - f[4]) > 0) ka = 0; }
    else if (n0 && n1) { if ((fc.obt & 4) && (xp - f[6]) * (f[3] - f[6]) + (yp - f[7]) * (f[4] - f[7]) > 0) ka = 1; }
    const int kb = (ka + 1) % 3, kc = (ka + 2) % 3;
    double ds[3];                                   // row ka minus row kb of the Gram matrix: the edge's direction in barycentric space
    for (int j = 0; j < 3; j++) ds[j] = sym[3 * ka + j] - sym[3 * kb + j];
    const double along = (w[0] * ds[0] + w[1] * ds[1] + w[2] * ds[2] - ds[kb]) / (ds[ka] - ds[kb]);
    t[ka] = fmin(fmax(along, 0.), 1.);              // clamped to the segment (the outside branch only, K.cu:141-145)
    t[kb] = fmin(fmax(1 - along, 0.), 1.);
    t[kc] = 0;
    t[0] -= w[0]; t[1] -= w[1]; t[2] -= w[2];
    dx = t[0] * f[0] + t[1] * f[3] + t[2] * f[6];
    dy = t[0] * f[1] + t[1] * f[4] + t[2] * f[7];
    return -1.;
}

__device__ inline int surface_texel64(double c0, double c1, int res)
{
    const int ix = (int)(c0 * res), iy = (int)(c1 * res);                                  // K.cu:181-188
    if ((c0 + c1) * res - ix - iy <= 1) return iy * res + ix;
    return (res - 1 - iy) * res + (res - 1 - ix);
}

// Everything a (pixel, face) pair needs in both passes: false = the face does not touch the pixel (K.cu:375-404)
struct Pair {
    double w[3], wc[3];     // barycentrics, clipped + normalised barycentrics
    double D, sign, dx, dy, t[3], dis;
};

__device__ inline bool pair_prob(const Args& A, const FaceD& fc, double xp, double yp, Pair& p)
{
    const double* f = fc.v;
    const double margin = sqrt((double)A.thr);
    // K.cu:33-38
    if (xp > fmax(fmax(f[0], f[3]), f[6]) + margin || xp < fmin(fmin(f[0], f[3]), f[6]) - margin ||
        yp > fmax(fmax(f[1], f[4]), f[7]) + margin || yp < fmin(fmin(f[1], f[4]), f[7]) - margin) return false;
    for (int k = 0; k < 3; k++) p.w[k] = fc.inv[3 * k] * xp + fc.inv[3 * k + 1] * yp + fc.inv[3 * k + 2];
    if (A.dist == 0) {
        if (!(p.w[0] <= 1 && p.w[0] >= 0 && p.w[1] <= 1 && p.w[1] >= 0 && p.w[2] <= 1 && p.w[2] >= 0)) return false;
        p.D = 1;
    } else if (A.dist == 1) {
        double d = p.w[0] > p.w[1] ? (p.w[1] > p.w[2] ? p.w[2] : p.w[1]) : (p.w[0] > p.w[2] ? p.w[2] : p.w[0]);
        d = d > 0 ? d * d : -(d * d);
        p.dis = d; p.t[0] = p.w[0]; p.t[1] = p.w[1]; p.t[2] = p.w[2];
        if (-d >= A.thr) return false;
        p.D = 1. / (1. + exp(-d / A.sigma));
    } else {
        p.sign = euclid64(fc, xp, yp, p.w, p.dx, p.dy, p.t);
        p.dis = p.dx * p.dx + p.dy * p.dy;
        if (p.sign < 0 && p.dis >= A.thr) return false;
        p.D = 1. / (1. + exp(-p.sign * p.dis / A.sigma));
    }
    double s = 0;
    for (int k = 0; k < 3; k++) { p.wc[k] = fmax(fmin(p.w[k], 1.), 0.); s += p.wc[k]; }
    s = fmax(s, 1e-5);
    for (int k = 0; k < 3; k++) p.wc[k] /= s;                                               // K.cu:53-58
    return true;
}

__device__ inline void stage_faces(const double* __restrict__ faces, const double* __restrict__ info, int first, int n, FaceD* sf)
{
    for (int i = threadIdx.x; i < n * 27; i += blockDim.x) {
        const int e = i / 27, k = i - e * 27;
        const size_t fi = (size_t)(first + e);
        if (k < 9) sf[e].v[k] = faces[fi * 9 + k];
        else if (k < 18) sf[e].inv[k - 9] = info[fi * INFO + (k - 9)];
        else sf[e].sym[k - 18] = info[fi * INFO + 9 + (k - 18)];
    }
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const double* o = info + (size_t)(first + e) * INFO + 18;
        sf[e].obt = (o[0] != 0 ? 1 : 0) | (o[1] != 0 ? 2 : 0) | (o[2] != 0 ? 4 : 0);
    }
}

__global__ __launch_bounds__(256) void forward_kernel(Args A, const double* __restrict__ faces, const double* __restrict__ textures,
                                                      const double* __restrict__ info, double* __restrict__ aggrs,
                                                      double* __restrict__ colors)
{
    __shared__ FaceD sf[CHUNK];
    const int IS = A.IS, P = IS * IS;
    const int tiles_x = (IS + 15) / 16;
    const int bn = blockIdx.x / (tiles_x * tiles_x), tl = blockIdx.x - bn * tiles_x * tiles_x;
    const int px = (tl % tiles_x) * 16 + (threadIdx.x & 15), py = (tl / tiles_x) * 16 + (threadIdx.x >> 4);
    const bool valid = px < IS && py < IS;
    const int pn = py * IS + px;
    const double xp = (2. * px + 1. - IS) / IS, yp = (2. * (IS - 1 - py) + 1. - IS) / IS;   // K.cu:343-346 (row 0 is the top)

    double a_acc = A.alpha == 2 ? 1. : 0.;
    double ssum = 0, smax = 0, c[3] = {0, 0, 0}, zbest = 10000000.;
    int fbest = -1;
    if (valid) for (int k = 0; k < 3; k++) c[k] = colors[((size_t)bn * 4 + k) * P + pn];     // the caller's background (soft_rasterize.py:50-53)
    if (A.rgb == 1) {
        ssum = (double)expf(A.eps / A.gamma); smax = A.eps;           // K.cu:362-366: `exp(eps / gamma_val)` of two floats IS the float exp
        for (int k = 0; k < 3; k++) c[k] *= ssum;
    }
    for (int f0 = 0; f0 < A.F; f0 += CHUNK) {
        const int n = min(CHUNK, A.F - f0);
        __syncthreads();
        stage_faces(faces, info, bn * A.F + f0, n, sf);
        __syncthreads();
        if (!valid) continue;
        for (int e = 0; e < n; e++) {
            const FaceD& fc = sf[e];
            const int fn = f0 + e;
            Pair p;
            if (!pair_prob(A, fc, xp, yp, p)) continue;
            if (A.alpha == 0) { if (p.D > 0.5) a_acc = 1; }                                  // K.cu:409-417, before the depth test
            else if (A.alpha == 1) a_acc += p.D;
            else a_acc *= 1. - p.D;
            const double zp = 1. / (p.wc[0] / fc.v[2] + p.wc[1] / fc.v[5] + p.wc[2] / fc.v[8]);
            if (zp < A.near || zp > A.far) continue;                                          // K.cu:421-424
            const bool front = (fc.v[7] - fc.v[1]) * (fc.v[3] - fc.v[0]) < (fc.v[4] - fc.v[1]) * (fc.v[6] - fc.v[0]);
            const double* tx = textures + ((size_t)bn * A.F + fn) * A.T * 3;
            double col[3];
            if (A.tex == 0) {
                const size_t lim = ((size_t)A.N * A.F - ((size_t)bn * A.F + fn)) * A.T;     // texels to the end of the tensor (K.cu:181-188 reads unchecked)
                size_t j = (size_t)max(surface_texel64(p.wc[0], p.wc[1], A.res), 0);
                if (j >= lim) j = lim - 1;
                for (int k = 0; k < 3; k++) col[k] = tx[j * 3 + k];
            } else for (int k = 0; k < 3; k++) col[k] = p.wc[0] * tx[k] + p.wc[1] * tx[3 + k] + p.wc[2] * tx[6 + k];
            if (A.rgb == 0) {
                const bool inside = p.w[0] <= 1 && p.w[0] >= 0 && p.w[1] <= 1 && p.w[1] >= 0 && p.w[2] <= 1 && p.w[2] >= 0;
                if (zp < zbest && inside && (A.double_side || front)) {
                    zbest = zp; fbest = fn;
                    for (int k = 0; k < 3; k++) c[k] = col[k];
                }
            } else if (front || A.double_side) {
                const double zn = (A.far - zp) / (A.far - A.near);
                double ez = exp((zn - smax) / A.gamma);                                       // K.cu:437-452
                if (zn > smax) {
                    const double r = exp((smax - zn) / A.gamma);
                    ssum *= r; for (int k = 0; k < 3; k++) c[k] *= r;
                    smax = zn; ez = 1.;
                }
                ssum += ez * p.D;
                for (int k = 0; k < 3; k++) c[k] += ez * p.D * col[k];
            }
        }
    }
    if (!valid) return;
    double a_out = a_acc;                                                                     // K.cu:458-482
    if (A.alpha == 1) a_out = a_acc / A.F;
    else if (A.alpha == 2) a_out = 1. - a_acc;
    colors[((size_t)bn * 4 + 3) * P + pn] = a_out;
    if (A.rgb == 0) {
        if (fbest != -1) for (int k = 0; k < 3; k++) colors[((size_t)bn * 4 + k) * P + pn] = c[k];
        aggrs[((size_t)bn * 2 + 0) * P + pn] = zbest;
        aggrs[((size_t)bn * 2 + 1) * P + pn] = (double)fbest;
    } else {
        for (int k = 0; k < 3; k++) colors[((size_t)bn * 4 + k) * P + pn] = c[k] / ssum;
        aggrs[((size_t)bn * 2 + 0) * P + pn] = ssum;
        aggrs[((size_t)bn * 2 + 1) * P + pn] = smax;
    }
}

__global__ __launch_bounds__(256) void backward_kernel(Args A, const double* __restrict__ faces, const double* __restrict__ textures,
                                                       const double* __restrict__ info, const double* __restrict__ colors,
                                                       const double* __restrict__ aggrs, const double* __restrict__ gcolors,
                                                       double* __restrict__ gfaces, double* __restrict__ gtex)
{
    __shared__ FaceD sf[CHUNK];
    const int IS = A.IS, P = IS * IS;
    const int tiles_x = (IS + 15) / 16;
    const int bn = blockIdx.x / (tiles_x * tiles_x), tl = blockIdx.x - bn * tiles_x * tiles_x;
    const int px = (tl % tiles_x) * 16 + (threadIdx.x & 15), py = (tl / tiles_x) * 16 + (threadIdx.x >> 4);
    const bool valid = px < IS && py < IS;
    const int pn = py * IS + px;
    const double xp = (2. * px + 1. - IS) / IS, yp = (2. * (IS - 1 - py) + 1. - IS) / IS;
    double g[4] = {0, 0, 0, 0}, out[4] = {0, 0, 0, 0}, ag0 = 0, ag1 = 0;
    if (valid) {
        for (int k = 0; k < 4; k++) { g[k] = gcolors[((size_t)bn * 4 + k) * P + pn]; out[k] = colors[((size_t)bn * 4 + k) * P + pn]; }
        ag0 = aggrs[((size_t)bn * 2 + 0) * P + pn]; ag1 = aggrs[((size_t)bn * 2 + 1) * P + pn];
    }
    for (int f0 = 0; f0 < A.F; f0 += CHUNK) {
        const int n = min(CHUNK, A.F - f0);
        __syncthreads();
        stage_faces(faces, info, bn * A.F + f0, n, sf);
        __syncthreads();
        if (!valid) continue;
        for (int e = 0; e < n; e++) {
            const FaceD& fc = sf[e];
            const int fn = f0 + e;
            Pair p;
            if (!pair_prob(A, fc, xp, yp, p)) continue;
            const double D = p.D;
            double C = g[3];                                                                  // K.cu:583-593 (hard alpha: the reference still adds g_alpha)
            if (A.alpha == 1) C = g[3] / A.F;
            else if (A.alpha == 2) C = g[3] * ((1. - out[3]) / fmax(1. - D, 1e-6));
            const double zp = 1. / (p.wc[0] / fc.v[2] + p.wc[1] / fc.v[5] + p.wc[2] / fc.v[8]);
            if (zp < A.near || zp > A.far) continue;                                          // K.cu:596-599: no gradient at all
            const bool front = (fc.v[7] - fc.v[1]) * (fc.v[3] - fc.v[0]) < (fc.v[4] - fc.v[1]) * (fc.v[6] - fc.v[0]);
            const size_t face = (size_t)bn * A.F + fn;
            const double* tx = textures + face * A.T * 3;
            double* gt = gtex + face * A.T * 3;
            double gz[3] = {0, 0, 0};
            if (A.rgb == 0) {
                if ((double)fn == ag1) {                                                      // K.cu:602-609
                    if (A.tex == 1) { for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) atomicAdd(gt + 3 * j + k, p.wc[j] * g[k]); }
                    else {
                        const int j = surface_texel64(p.wc[0], p.wc[1], A.res);
                        if (j >= 0 && j < A.T) for (int k = 0; k < 3; k++) atomicAdd(gt + 3 * j + k, g[k]);
                    }
                }
            } else if (front || A.double_side) {                                              // K.cu:611-640
                const double zn = (A.far - zp) / (A.far - A.near);
                const double sm = D * exp((zn - ag1) / A.gamma) / ag0;
                double col[3];
                if (A.tex == 1) {
                    for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) atomicAdd(gt + 3 * j + k, sm * p.wc[j] * g[k]);
                    for (int k = 0; k < 3; k++) col[k] = p.wc[0] * tx[k] + p.wc[1] * tx[3 + k] + p.wc[2] * tx[6 + k];
                } else {
                    const int j = surface_texel64(p.wc[0], p.wc[1], A.res);
                    if (j >= 0 && j < A.T) for (int k = 0; k < 3; k++) atomicAdd(gt + 3 * j + k, sm * g[k]);
                    const size_t lim = ((size_t)A.N * A.F - face) * A.T;
                    size_t jj = (size_t)max(j, 0);
                    if (jj >= lim) jj = lim - 1;
                    for (int k = 0; k < 3; k++) col[k] = tx[jj * 3 + k];
                }
                double Crgb = 0;
                for (int k = 0; k < 3; k++) Crgb += g[k] * (col[k] - out[k]);
                Crgb *= sm;
                C += Crgb / D;
                const double Cz = Crgb / A.gamma / (A.near - A.far) * zp * zp;
                for (int k = 0; k < 3; k++) gz[k] = Cz * p.wc[k] / fc.v[3 * k + 2] / fc.v[3 * k + 2];
            }
            C *= D * (1. - D) / A.sigma;                                                      // K.cu:644
            double gxy[3][2] = {{0, 0}, {0, 0}, {0, 0}};
            if (A.dist == 1) {                                                                // K.cu:161-175
                const int q = p.t[0] > p.t[1] ? (p.t[1] > p.t[2] ? 2 : 1) : (p.t[0] > p.t[2] ? 2 : 0);
                const double sc = 2. * sqrt(fabs(p.dis));
                for (int l = 0; l < 2; l++)
                    for (int k = 0; k < 3; k++) {
                        const double ipl = fc.inv[3 * q + l];
                        gxy[k][l] = (-ipl * fc.inv[3 * k] * xp + -ipl * fc.inv[3 * k + 1] * yp + -ipl * fc.inv[3 * k + 2]) * C * sc;
                    }
            } else if (A.dist == 2) {                                                         // K.cu:649-655
                for (int k = 0; k < 3; k++) {
                    gxy[k][0] = 2. * p.sign * C * (p.t[k] + p.w[k]) * p.dx;
                    gxy[k][1] = 2. * p.sign * C * (p.t[k] + p.w[k]) * p.dy;
                }
            }
            double* gf = gfaces + face * 9;
            for (int k = 0; k < 3; k++) {
                if (gxy[k][0] != 0) atomicAdd(gf + 3 * k, gxy[k][0]);
                if (gxy[k][1] != 0) atomicAdd(gf + 3 * k + 1, gxy[k][1]);
                if (gz[k] != 0) atomicAdd(gf + 3 * k + 2, gz[k]);
            }
        }
    }
}

}  // namespace lasr64

using namespace lasr64;

static int check64(int N, int F, int T, int IS, int dist, int rgb, int alpha, int tex)
{
    if (N < 0 || F < 0 || T < 1 || IS < 0) return LASR_E_BADARG;
    if (dist < 0 || dist > 2 || rgb < 0 || rgb > 1 || alpha < 0 || alpha > 2 || tex < 0 || tex > 1) return LASR_E_BADMODE;
    if ((long long)N * F > 0x7fffffffLL / 64 || (long long)N * IS * IS > 0x7fffffffLL || IS > 32767) return LASR_E_BADARG;
    return LASR_OK;
}

static Args make64(int N, int F, int T, int IS, float near, float far, float eps, float sigma, int dist, float dist_eps, float gamma,
                   int rgb, int alpha, int tex, int double_side)
{
    Args A;
    A.N = N; A.F = F; A.T = T; A.res = (int)sqrt((double)T); A.IS = IS;
    A.near = near; A.far = far; A.eps = eps; A.sigma = sigma; A.gamma = gamma; A.thr = dist_eps * sigma;     // K.cu:352 (float product)
    A.dist = dist; A.rgb = rgb; A.alpha = alpha; A.tex = tex; A.double_side = double_side ? 1 : 0;
    return A;
}

extern "C" size_t lasr_sr_workspace_bytes_f64(int N, int F)
{
    if (N < 0 || F < 0) return 0;
    return (size_t)N * (size_t)F * INFO * sizeof(double) + 256;
}

static double* info_buffer(double* faces_info, void* ws, size_t ws_bytes, int N, int F)
{
    if (faces_info) return faces_info;
    if (!ws || ws_bytes < lasr_sr_workspace_bytes_f64(N, F)) return nullptr;
    return (double*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
}

extern "C" int lasr_sr_forward_f64(const double* faces, const double* textures, double* faces_info, double* aggrs_info,
                                   double* soft_colors, void* workspace, size_t workspace_bytes, int N, int F, int T, int IS,
                                   float near, float far, float eps, float sigma_val, int func_id_dist, float dist_eps,
                                   float gamma_val, int func_id_rgb, int func_id_alpha, int texture_sample_type, int double_side,
                                   void* hip_stream)
{
    int rc = check64(N, F, T, IS, func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type);
    if (rc) return rc;
    if (N == 0 || IS == 0) return LASR_OK;
    if (!aggrs_info || !soft_colors || (F > 0 && (!faces || !textures))) return LASR_E_BADARG;
    double* info = info_buffer(faces_info, workspace, workspace_bytes, N, F);
    if (!info && F > 0) return LASR_E_WORKSPACE;
    hipStream_t st = (hipStream_t)hip_stream;
    const Args A = make64(N, F, T, IS, near, far, eps, sigma_val, func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha,
                          texture_sample_type, double_side);
    if (N * F > 0) {
        ProfScope ps(K_SR_SETUP, st);
        hipLaunchKernelGGL(setup_kernel, dim3((N * F + 255) / 256), dim3(256), 0, st, faces, info, N * F);
    }
    if ((rc = launch_ok())) return rc;
    const int tiles_x = (IS + 15) / 16;
    {
        ProfScope ps(K_SR_FORWARD, st);
        hipLaunchKernelGGL(forward_kernel, dim3((unsigned)(N * tiles_x * tiles_x)), dim3(256), 0, st, A, faces, textures, info,
                           aggrs_info, soft_colors);
    }
    return launch_ok();
}

extern "C" int lasr_sr_backward_f64(const double* faces, const double* textures, const double* soft_colors,
                                    const double* faces_info, const double* aggrs_info, double* grad_faces, double* grad_textures,
                                    const double* grad_soft_colors, void* workspace, size_t workspace_bytes, int N, int F, int T,
                                    int IS, float near, float far, float eps, float sigma_val, int func_id_dist, float dist_eps,
                                    float gamma_val, int func_id_rgb, int func_id_alpha, int texture_sample_type, int double_side,
                                    void* hip_stream)
{
    int rc = check64(N, F, T, IS, func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type);
    if (rc) return rc;
    if (N == 0 || IS == 0 || F == 0) return LASR_OK;
    if (!faces || !textures || !soft_colors || !aggrs_info || !grad_faces || !grad_textures || !grad_soft_colors) return LASR_E_BADARG;
    hipStream_t st = (hipStream_t)hip_stream;
    const double* info = faces_info;
    if (!info) {                                     // no tensor from the forward pass: rebuild the per-face data in the workspace
        double* w = info_buffer(nullptr, workspace, workspace_bytes, N, F);
        if (!w) return LASR_E_WORKSPACE;
        ProfScope ps(K_SR_SETUP, st);
        hipLaunchKernelGGL(setup_kernel, dim3((N * F + 255) / 256), dim3(256), 0, st, faces, w, N * F);
        info = w;
    }
    if ((rc = launch_ok())) return rc;
    const Args A = make64(N, F, T, IS, near, far, eps, sigma_val, func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha,
                          texture_sample_type, double_side);
    const int tiles_x = (IS + 15) / 16;
    {
        ProfScope ps(K_SR_BACKWARD, st);
        hipLaunchKernelGGL(backward_kernel, dim3((unsigned)(N * tiles_x * tiles_x)), dim3(256), 0, st, A, faces, textures, info,
                           soft_colors, aggrs_info, grad_soft_colors, grad_faces, grad_textures);
    }
    return launch_ok();
}
