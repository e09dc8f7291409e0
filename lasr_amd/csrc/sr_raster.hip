// sr_raster.hip -- soft-rasteriser forward/backward for MI355X (gfx950, wave64) + the C ABI.
//
// Replaces the three reference kernels (soft_rasterize_cuda_kernel.cu "K.cu":245-305, 308-483, 486-668) with a
// different execution shape:
//
//   setup    1 thread / face : 48-float record (geometry, hoisted edge vectors, correctly rounded reciprocals, flags)
//            + the EXACT integer pixel rectangle of the reference's float bbox test (sr_device.h).
//   forward  1 workgroup / 16x16 px tile, 1 wave / 8x8 quadrant, 1 lane / pixel.
//            level 1: the workgroup scans the image's pixel rects (coalesced 8-B loads) and compacts, IN FACE-INDEX
//            ORDER (wave ballots + prefix), the faces touching the tile into LDS; level 2: each wave filters that
//            list 64 entries at a time down to its quadrant (rect overlap + conservative corner cull); walk: the
//            face index is wave-uniform, so the record is fetched through the scalar cache into SGPRs and only the
//            per-pixel state lives in VGPRs.  Index order is preserved, so the alpha product, the online
//            depth-softmax and the hard z-buffer tie-break see exactly the reference's sequence of faces.
//   backward 1 wave / (image, face), FACE-major: lanes enumerate the pixels of the face's rect; a cheap stage
//            compacts the pixels that can be within the distance threshold into an LDS ring, the heavy stage runs
//            on dense batches of 64, accumulates the 9+9 gradient components in registers, one permlane/DPP wave
//            reduction per face, one plain read-modify-write per component.  The backward pass has no cross-face
//            dependence (it only needs the finished per-pixel aggregates), so this removes every global atomic of
//            the reference and makes the gradients deterministic.  (Surface textures, T != 3 texels, scatter with
//            atomics: cold path.)
//
// Brute force in the reference is N*P*F pair tests; here a pixel only ever sees the faces binned to its quadrant.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/lasr_sr.h"
#include "sr_device.h"
#include "sr_common.h"
#include "host_common.h"

namespace lasr {

constexpr int TILE = 16;        // pixels per tile side (4 waves of 8x8)
#ifndef LASR_PREFETCH
#define LASR_PREFETCH 1         // scalar-cache prefetch of the next list entry in the forward walk (A/B: profiles/r05_prefetch_ab.txt)
#endif
#ifndef LASR_LIST_CAP
#define LASR_LIST_CAP 2048
#endif
constexpr int LIST_CAP = LASR_LIST_CAP;  // capacity of each LDS face list (u16 ids relative to the round's first face; 5 lists = 20 KB)

// ---------------------------------------------------------------------------
// One thread builds one face's record -- into LDS; the block then writes its 256 records (48 KB, contiguous in the
// workspace) with coalesced stores.  Writing the 192-B records straight from the building threads is a 192-B-strided
// scatter: the PMC pass of round 1 showed 2x the algorithmic write traffic for it (profiles/r01i_pmc.txt).
constexpr int SETUP_STRIDE = REC + 1;      // odd LDS stride: the building threads' stores spread over the banks
constexpr int GROUP = 64;                  // faces per group rect (one wave of the setup kernel, one wave-load of the forward scan)

__device__ __forceinline__ int groups_of(int F) { return (F + GROUP - 1) / GROUP; }

// Grid: one block per (image, 256 consecutive faces of it) -- groups of 64 faces never straddle images.  Besides the records
// and the per-face pixel rects it writes one UNION rect per group of 64 consecutive faces (grects [N, ceil(F/64)]): the forward
// kernel tests those first and scans only the groups that can touch its tile.  Meshes number their faces patch by patch, so
// a 16x16 tile typically meets 2-6 of the ~40 groups; an arbitrary numbering degrades to the full scan, never to a wrong list.
__global__ __launch_bounds__(256) void sr_setup_kernel(const float* __restrict__ faces, float* __restrict__ recs,
                                                       short4* __restrict__ rects, short4* __restrict__ grects,
                                                       float* __restrict__ info27, int F, int blocks_per_image,
                                                       float margin, int IS)
{
    __shared__ float s_rec[256 * SETUP_STRIDE];
    const int bn = blockIdx.x / blocks_per_image;
    const int chunk = blockIdx.x - bn * blocks_per_image;
    const int local = chunk * 256 + threadIdx.x;               // face index inside the image
    const size_t first = (size_t)bn * F + (size_t)chunk * 256;
    const size_t i = first + threadIdx.x;
    short4 r = make_short4(32767, -1, 32767, -1);               // neutral element of the union (= the empty rect)
    if (local < F) {
        build_record(faces + i * 9, s_rec + threadIdx.x * SETUP_STRIDE, &r, margin, IS, info27 ? info27 + i * 27 : nullptr);
        rects[i] = r;
    }
    // union rect of this wave's 64 faces (empty rects are neutral: x0 = 32767, x1 = -1)
    int x0 = r.x, x1 = r.y, y0 = r.z, y1 = r.w;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        x0 = min(x0, __shfl_xor(x0, d)); x1 = max(x1, __shfl_xor(x1, d));
        y0 = min(y0, __shfl_xor(y0, d)); y1 = max(y1, __shfl_xor(y1, d));
    }
    const int g = chunk * 4 + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0 && g < groups_of(F))
        grects[(size_t)bn * groups_of(F) + g] = make_short4((short)x0, (short)x1, (short)y0, (short)y1);
    __syncthreads();
    const int nflt = min(256, F - chunk * 256) * REC;
    float* __restrict__ out = recs + first * REC;
    for (int k = threadIdx.x; k < nflt; k += 256) {
        const int rr = k / REC;
        out[k] = s_rec[rr * SETUP_STRIDE + (k - rr * REC)];
    }
}

// ---------------------------------------------------------------------------
// Per-pixel forward state (K.cu:354-368)
// ---------------------------------------------------------------------------
// Block -> tile order of ONE launch, heaviest tiles first.  A tile's walk length is the number of faces whose pixel rect touches
// it (0 to ~200 on LASR's crops); the fixed centre-out spiral of tile_of_block starts crowded tiles early but knows nothing of
// THIS batch, and with 5-128 frames per launch (most calls LASR makes: nnutils/mesh_net.py:318-363) a SIMD holds only a handful
// of crowded tiles, so the launch ends when the unluckiest SIMD does.  Issued in descending weight the hardware's round-robin
// placement deals every SIMD one tile of each weight class: forward kernel -18 % at 16 frames, -18 % at 64, -2 % at 256
// (profiles/r04_tile_order_ab.txt).  Which block renders which tile does not change any tile's arithmetic: bit-identical output.
//
// sr_tile_weight_kernel, one workgroup per image: every face adds its tile rectangle to a 2-D difference array in LDS (four
// atomics per face, whatever the rect's size); a prefix pass along the rows and one down the columns turn it into the per-tile
// counts, stored as min(count, 255).
// sr_order_kernel, one workgroup per XCD: block b of the forward runs on XCD b % 8 and takes entry b / 8 of that XCD's list; the
// XCD keeps the m = N / 8 images whose records it already fetches (tile_of_block's partition).  Counting sort of the m x tiles
// keys (staged in LDS), descending; the empty tiles (three quarters of a LASR crop) are counted per wave, not per lane; ties
// land in atomic order (any order is correct).  Table: [8][N x tiles / 8], image << 16 | tile row << 8 | tile column.  A frame
// count that is not a multiple of 8 gives every XCD the same share of the (image, tile) list, an image split between two neighbours.
// Needs N x tiles % 8 == 0, at most ORDER_MAX_SIDE tiles per side and at most ORDER_MAX_ENTRIES tiles per XCD.
// (Measured and dropped, 16 / 64 frames: weights by one atomic per (face, tile) pair -- neighbouring faces hit the same counters
// -- 26 / 84 us; both steps in the per-XCD workgroups, eight CUs doing all the work: 16 / 36 us; one launch with the sort done by
// the last workgroup of each XCD to finish, a device-scope fence per workgroup: 21 / 46 us; weights from the 64-face group rects
// only: 10 / 18 us but half of the forward's gain lost.)
constexpr int ORDER_MAX_SIDE = 127;           // (side + 1)^2 ints of LDS: image sizes up to 1016 pixels
constexpr int ORDER_MAX_ENTRIES = 61440;      // 60 KB of keys per XCD: 480 frames at 256x256, 120 at 512x512
constexpr int ORDER_THREADS = 1024;

// e / d for 0 <= e < 2^23 with inv = 1.f / d (wave-uniform divisors: no integer division sequence per element)
__device__ __forceinline__ int div_small(int e, int d, float inv)
{
    int q = (int)((float)e * inv);
    const int r = e - q * d;
    q += r >= d ? 1 : 0;
    q -= r < 0 ? 1 : 0;
    return q;
}

__global__ __launch_bounds__(512) void sr_tile_weight_kernel(const short4* __restrict__ rects, int F, int t8,
                                                             unsigned char* __restrict__ keys, int* __restrict__ busy, int shift)
{
    extern __shared__ int s_d[];                      // (t8 + 1)^2 ints: 4.3 KB for a 256x256 image, 64 KB at the 1016-pixel limit
    const int bn = blockIdx.x, S = t8 + 1, tiles = t8 * t8;
    if (busy && bn == 0 && threadIdx.x == 0) *busy = 0;                          // sr_order_kernel adds up the non-empty tiles
    for (int i = threadIdx.x; i < S * S; i += 512) s_d[i] = 0;
    __syncthreads();
    const short4* __restrict__ mine = rects + (size_t)bn * F;
    for (int i = threadIdx.x; i < F; i += 4 * 512) {
        short4 r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = i + k * 512 < F ? mine[i + k * 512] : make_short4(1, 0, 1, 0);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (r[k].y < r[k].x || r[k].w < r[k].z) continue;                   // empty rect (culled face)
            const int tx0 = r[k].x >> shift, tx1 = min((int)r[k].y >> shift, t8 - 1) + 1;    // shift 3: 8x8-pixel tiles, 4: 16x16
            const int ty0 = r[k].z >> shift, ty1 = min((int)r[k].w >> shift, t8 - 1) + 1;
            atomicAdd(&s_d[ty0 * S + tx0], 1);
            atomicAdd(&s_d[ty0 * S + tx1], -1);
            atomicAdd(&s_d[ty1 * S + tx0], -1);
            atomicAdd(&s_d[ty1 * S + tx1], 1);
        }
    }
    __syncthreads();
    // prefix along the rows, then down the columns: one thread per row / column, eight cells fetched ahead of the running sum
    for (int pass = 0; pass < 2; pass++) {
        if ((int)threadIdx.x < t8) {
            const int step = pass == 0 ? 1 : S;
            int* const p = s_d + (pass == 0 ? threadIdx.x * S : threadIdx.x);
            int run = 0;
            for (int c = 0; c < t8; c += 8) {
                int v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = c + k < t8 ? p[(c + k) * step] : 0;
#pragma unroll
                for (int k = 0; k < 8; k++) { run += v[k]; if (c + k < t8) p[(c + k) * step] = run; }
            }
        }
        __syncthreads();
    }
    unsigned char* __restrict__ out = keys + (size_t)bn * tiles;
    const float inv_t8 = 1.f / (float)t8;
    for (int t = threadIdx.x; t < tiles; t += 512) {
        const int ty = div_small(t, t8, inv_t8);
        out[t] = (unsigned char)min(s_d[ty * S + (t - ty * t8)], 255);
    }
}

__global__ __launch_bounds__(ORDER_THREADS) void sr_order_kernel(const unsigned char* __restrict__ keys, int N, int t8,
                                                                 int* __restrict__ order, int* __restrict__ busy)
{
    extern __shared__ unsigned char s_key[];
    __shared__ unsigned s_hist[256], s_base[256], s_wave[4];
    // this XCD's slice of the (image, tile) keys: N / 8 whole images, or -- frame counts that are not a multiple of 8 -- the same
    // share of the list with an image split between two neighbours (its records are then fetched into both L2s)
    // (large launches: the XCD's images in G groups, one workgroup each, sorted and issued one group after the other)
    const int G = gridDim.x >> 3, x = blockIdx.x / G, grp = blockIdx.x - x * G, tiles = t8 * t8;
    const int entries = (int)(((long long)N * tiles) >> 3) / G, e_first = (x * G + grp) * entries;
    const int lane = threadIdx.x & 63;
    const unsigned char* __restrict__ mine = keys + e_first;
    if (threadIdx.x < 256) s_hist[threadIdx.x] = 0u;
    if ((entries & 3) == 0) {                                                    // every XCD's slice starts on a word
        const unsigned* __restrict__ w = (const unsigned*)mine;
        for (int i = threadIdx.x; i < entries >> 2; i += ORDER_THREADS) ((unsigned*)s_key)[i] = w[i];
    } else {
        for (int i = threadIdx.x; i < entries; i += ORDER_THREADS) s_key[i] = mine[i];
    }
    __syncthreads();
    for (int e0 = threadIdx.x - lane; e0 < entries; e0 += ORDER_THREADS) {       // wave-uniform trip count
        const int e = e0 + lane;
        const int key = e < entries ? (int)s_key[e] : -1;
        const unsigned long long empty = wave_mask(key == 0);
        if (key > 0) atomicAdd(&s_hist[key], 1u);
        if (lane == 0 && empty) atomicAdd(&s_hist[0], (unsigned)__popcll(empty));
    }
    __syncthreads();
    // first position of key k = number of entries with a larger key: suffix sums of the 256 counts, 64 per wave
    unsigned incl = 0;
    if (threadIdx.x < 256) {
        incl = s_hist[threadIdx.x];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned up = __shfl_down(incl, d);
            if (lane + d < 64) incl += up;
        }
        if (lane == 0) s_wave[threadIdx.x >> 6] = incl;                          // this wave's total
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        unsigned above = incl - s_hist[threadIdx.x];
        for (int w = (threadIdx.x >> 6) + 1; w < 4; w++) above += s_wave[w];
        s_base[threadIdx.x] = above;
    }
    if (busy && threadIdx.x == 0) atomicAdd(busy, entries - (int)s_hist[0]);    // the launch's non-empty tiles: the forward kernels' choice
    __syncthreads();
    int* __restrict__ out = order + e_first;
    const float inv_tiles = 1.f / (float)tiles, inv_t8 = 1.f / (float)t8;
    for (int e0 = threadIdx.x - lane; e0 < entries; e0 += ORDER_THREADS) {
        const int e = e0 + lane;
        const int key = e < entries ? (int)s_key[e] : -1;
        const unsigned long long empty = wave_mask(key == 0);
        unsigned pos = 0;
        if (key > 0) pos = atomicAdd(&s_base[key], 1u);
        unsigned first = 0;
        if (lane == 0 && empty) first = atomicAdd(&s_base[0], (unsigned)__popcll(empty));
        first = __builtin_amdgcn_readfirstlane(first);
        if (key == 0) pos = first + bits_below_lane(empty);
        if (key >= 0) {
            const int img = div_small(e_first + e, tiles, inv_tiles), t = e_first + e - img * tiles, ty = div_small(t, t8, inv_t8);
            out[pos] = (img << 16) | (ty << 8) | (t - ty * t8);
        }
    }
}

template <int NCH>
struct PixState {
    float c[NCH];         // colour / attribute accumulators (3 = RGB; 6 = two attribute triples in one pass)
    float a;              // alpha accumulator
    float ssum, smax;     // softmax running (sum, max)  | hard: (zbest, -)
    int fbest;
};

// uniform reciprocals for the exact division-by-reciprocal (sr_device.h); ok = all three divisors are in the safe range
struct UniRecip { float inv_sigma, inv_gamma, inv_fmn; bool ok; };
// ok also vouches for what the tame-record arithmetic of sr_device.h assumes about the launch: the sigmoid's exponent stays
// below 80 (recip64_noscale) and a depth beyond 1e15 is beyond the far plane (recip_noscale is only exact up to 2^60)
__device__ __forceinline__ UniRecip uni_recip(const RasterArgs& A)
{
    UniRecip U;
    U.inv_sigma = 1.f / A.sigma; U.inv_gamma = 1.f / A.gamma; U.inv_fmn = 1.f / (A.far - A.near);
    U.ok = recip_safe(A.sigma) && recip_safe(A.gamma) && recip_safe(A.far - A.near) && A.thr * U.inv_sigma < 80.f && A.far < 1e15f;
    return U;
}

// RX = relaxed arithmetic (opt-in per call, LASR_SR_RELAXED_MATH; LASR's mode combination only): the distance and the
// `dis >= threshold` decision stay bit-faithful -- that is where the reference is ill-conditioned -- but everything after
// it (sigmoid, alpha product, clip/normalise, depth, softmax weights) uses fp32 v_rcp / v_exp arithmetic instead of the
// reference's division / double-promotion sequence.  Rendered image within ~3e-5 of the exact path (bar: 1e-4).
template <bool LASR_FAST, bool MK, int NCH, bool RX = false, typename RP = cptr_t, typename TP = cptr_t>
__device__ __forceinline__ void forward_face(const RasterArgs& A, const Modes m, RP rec,
                                             TP tex, int fn, int lim, float xp, float yp,
                                             float w0, float w1, float w2, PixState<NCH>& s, const UniRecip& U)
{
    Frag fr;
    if (RX) {
        euclid<false, MK>(rec, xp, yp, w0, w1, w2, fr);
        fr.dis = fr.dx * fr.dx + fr.dy * fr.dy;
        if (fr.sign < 0 && fr.dis >= A.thr) return;
        fr.D = __builtin_amdgcn_rcpf(1.f + __expf(-fr.sign * fr.dis * U.inv_sigma));
    } else if (!fragment_w<false, MK>(rec, m.dist, A.thr, A.sigma, xp, yp, w0, w1, w2, fr, U.inv_sigma)) return;
    const float D = fr.D;
    // alpha first (K.cu:409-417), before the depth test
    if (m.alpha == 0) { if ((double)D > 0.5) s.a = 1.f; }
    else if (m.alpha == 1) s.a += D;
    else if (RX) s.a *= 1.f - D;
    else s.a = (float)((double)s.a * (1. - (double)D));

    float c0 = w0, c1 = w1, c2 = w2;
    clip_normalise<RX, (MK && OPT_MED3)>(c0, c1, c2);
    const float zp = RX ? __builtin_amdgcn_rcpf(c0 * rec[R_IZ + 0] + c1 * rec[R_IZ + 1] + c2 * rec[R_IZ + 2])
                        : depth_at<false, MK>(rec, c0, c1, c2);
    // tame records: the only NaN depth is recip_noscale(0) where the reference has 1 / 0 = inf, beyond any far plane
    if (MK && OPT_NOSCALE ? !(zp >= A.near && zp <= A.far) : (zp < A.near || zp > A.far)) return;

    const bool front = (__float_as_int(rec[R_FLAGS]) & 8) != 0;
    if (m.rgb == 0) {
        if (zp < s.ssum && inside_closed(w0, w1, w2) && (m.double_side || front)) {
            s.ssum = zp; s.fbest = fn;
#pragma unroll
            for (int k = 0; k < NCH; k++) s.c[k] = sample_colour(tex, c0, c1, c2, A.res, k, m.tex, lim, NCH);
        }
    } else {
        if (front || m.double_side) {
            const float fmn = A.far - A.near;
            const float zn = RX ? (A.far - zp) * U.inv_fmn
                                : (MK ? div_by_recip(A.far - zp, fmn, U.inv_fmn) : (A.far - zp) / fmn);
            // K.cu:428-446 evaluates two exponentials per fragment: exp((smax_old - zn)/gamma) rescales the history when zn is
            // the new maximum (and the fragment's own weight is then exp(0) = 1 exactly), otherwise the history keeps weight 1
            // and the fragment gets exp((zn - smax)/gamma).  One exponential of -|zn - smax|/gamma serves both cases with
            // identical bits (exp_1ulp(0) == 1, x * 1 == x).
            const bool up = zn > s.smax;
            // -|zn - smax| is (up ? smax - zn : zn - smax) bit for bit (a - b == -(b - a)); the new maximum is a v_max
            const float d = OPT_SOFTMAX ? -fabsf(zn - s.smax) : (up ? s.smax - zn : zn - s.smax);
            const float E = RX ? __expf(d * U.inv_gamma)
                               : exp_1ulp(MK ? div_by_recip(d, A.gamma, U.inv_gamma) : d / A.gamma);
            const float hist = up ? E : 1.f, wgt = up ? D : E * D;     // (rescale, ez * D) of the reference, branch-free
            s.smax = OPT_SOFTMAX && MK ? max_finite(zn, s.smax) : (up ? zn : s.smax);
            s.ssum = hist * s.ssum + wgt;
#pragma unroll
            for (int k = 0; k < NCH; k++)
                s.c[k] = hist * s.c[k] + wgt * sample_colour(tex, c0, c1, c2, A.res, k, m.tex, lim, NCH);
        }
    }
}

// NCH = attribute channels per vertex: 3 (RGB, every mode), 6 (two attribute triples rendered in ONE pass over the
// geometry -- LASR's flow renders, nnutils/mesh_net.py:85-87, rasterise the same mesh twice with two different
// per-vertex attributes; channels are independent, so the result equals the two separate renders) or 9 (the flow
// attributes plus the texture colours: the texture render of a LASR step, mesh_net.py:348-363, has the same geometry).
// Block -> (image, tile).  An image's tiles stay on one XCD (xcd_remap: its records are fetched into a single L2).  Inside an
// XCD the blocks are issued rank-major over groups of four images and CENTRE-OUT inside an image (square spiral from the middle):
// the crowded tiles -- LASR crops every frame around the object, dataloader/vidbase.py:105-135 -- start first and the empty
// border tiles fill the tail of the launch.  With few frames per launch the crowded tiles' serial walks are the critical
// path (0.2 ms for ONE frame); starting them last cost up to 40 % of a 16-frame launch.  Any order is correct; odd tile
// counts or frame counts that do not divide over the XCDs keep the plain row-major order.
__device__ __forceinline__ void tile_of_block(int b, int total, int tiles_x, int& bn, int& tx, int& ty, const int* __restrict__ order = nullptr)
{
    if (order) {                                    // the launch's own order (sr_order_kernel): image << 16 | tile row << 8 | tile column
        const int e = __builtin_amdgcn_readfirstlane(order[(b & 7) * (total >> 3) + (b >> 3)]);
        bn = e >> 16; ty = (e >> 8) & 255; tx = e & 255;
        return;
    }
    const int tiles = tiles_x * tiles_x;
    const int per = total >> 3;
#ifndef LASR_ORDER
#define LASR_ORDER 1
#endif
    if ((tiles_x & 1) == 0 && LASR_ORDER != 0) {
        int rank;
        if ((total & 7) == 0 && per % tiles == 0) {
            const int m = per / tiles;              // images per XCD
            const int i = b >> 3;                   // position in this XCD's issue order
#ifndef LASR_ILV
#define LASR_ILV 4      // images interleaved at a time: the records of 4 images (1.9 MB at 2420 faces) stay in the XCD's 4 MB L2.
#endif                  // All 32 of the XCD: forward 2.16 ms but 700 MB fetched per 256 frames; 4: 2.18 ms, 170 MB (profiles/r03_tile_order_ab.txt)
            const int g = m < LASR_ILV ? m : LASR_ILV;
            const int grp = i / (tiles * g), j = i - grp * tiles * g;
            rank = j / g;
            bn = (b & 7) * m + grp * g + (j - rank * g);
            if (m % g) { rank = i / m; bn = (b & 7) * m + (i - rank * m); }
            if (LASR_ORDER == 3) { bn = (b & 7) * m + i / tiles; rank = i % tiles; }      // measurement: spiral, image-major
        } else {                                    // fewer images than XCDs (or a ragged count): rank-major over all images
            const int n_img = total / tiles;
            rank = b / n_img;
            bn = b - rank * n_img;
        }
        // rank -> cell of the square spiral around the grid's centre: ring r holds the 8r + 4 cells at Chebyshev distance
        // r + 1/2 from the centre, 4 r^2 cells lie inside it
        if (LASR_ORDER == 2) { ty = rank / tiles_x; tx = rank - ty * tiles_x; return; }    // measurement: row-major, interleaved
        if (LASR_ORDER == 4) rank = tiles - 1 - rank;                                      // measurement: outside-in
        if (LASR_ORDER == 5) rank = (int)(((long long)rank * 167) % tiles);                // measurement: scattered (tiles = 2^k)
        if (LASR_ORDER == 6) { rank = (rank & 1) ? tiles - 1 - (rank >> 1) : (rank >> 1); }   // measurement: centre and border alternate
        int r = (int)(sqrtf((float)rank) * 0.5f);
        while (4 * r * r > rank) r--;
        while (4 * (r + 1) * (r + 1) <= rank) r++;
        const int o = rank - 4 * r * r, s = 2 * r + 1, h = tiles_x / 2;      // s = cells per side minus one
        const int side = o / s, k = o - side * s;
        const int lo = h - 1 - r, hi = h + r;
        if (side == 0) { tx = lo + k; ty = lo; }
        else if (side == 1) { tx = hi; ty = lo + k; }
        else if (side == 2) { tx = hi - k; ty = hi; }
        else { tx = lo; ty = hi - k; }
        return;
    }
    const int blk = xcd_remap(b, total);
    bn = blk / tiles;
    const int tl = blk - bn * tiles;
    ty = tl / tiles_x;
    tx = tl - ty * tiles_x;
}

// W1 = one wave per workgroup, the workgroup's tile IS the wave's 8x8 quadrant: no workgroup barrier anywhere and nothing held
// until the slowest of four waves is done.  Affordable since the group rects (level 0) made the face scan cheap: a lone wave
// tests the ~40 group rects, then scans only the groups that can touch its 64 pixels, compacting straight into its own list.
// The tile body of sr_forward_kernel as a device function (the kernel below is a thin wrapper).  s_all / s_wcnt: the workgroup's
// LDS of the four-wave form (unused by W1); mine_lds: this wave's list of LIST_CAP u16 entries.
template <bool LASR_FAST, int NCH, bool RX, bool W1, int CAP = LIST_CAP>
__device__ __forceinline__ void forward_tile_body(RasterArgs A, float* __restrict__ aggrs, float* __restrict__ colors, int bn, int tx, int ty,
                                                  unsigned short* s_all, unsigned short* mine_lds, int (*s_wcnt)[4])
{
    constexpr int TW = W1 ? 8 : TILE;
    // fast path: LASR's training configuration (euclidean, softmax, prod, vertex, double-sided)
    const Modes m = LASR_FAST ? Modes{2, 1, 2, 1, 1} : A.m;
    if (A.near_far_dev) { A.near = A.near_far_dev[0]; A.far = A.near_far_dev[1]; }

    const int IS = A.IS, P = IS * IS;
    const int tid = threadIdx.x, wave = W1 ? 0 : tid >> 6, lane = tid & 63;

    const int qx0 = tx * TW + (wave & 1) * 8, qy0 = ty * TW + (wave >> 1) * 8;       // this wave's quadrant
    const int px = qx0 + (lane & 7);
    const int py = qy0 + (lane >> 3);
    const bool valid = px < IS && py < IS;
    const int pn = py * IS + px;
    const int pxy = valid ? px | (py << 16) : (int)0xfffefffeu;    // (lies in no rect, the empty one included: rect_has)

    PixState<NCH> s;
    s.a = (m.alpha == 2) ? 1.f : 0.f;
    s.fbest = -1;
    if (m.rgb == 0) { s.ssum = 10000000.f; s.smax = 0.f; }
    else { s.ssum = expf(A.eps / A.gamma); s.smax = A.eps; }
#pragma unroll
    for (int k = 0; k < NCH; k++) {
        // the reference's caller pre-fills soft_colors with the background (soft_rasterize.py:50-53) and the kernel reads it back;
        // lasr_sr_forward_bg passes the colour itself, which saves the fill and this read
        const float bg = A.use_bg ? A.bg[k] : (valid ? colors[((size_t)bn * (NCH + 1) + k) * P + pn] : 1.f);
        s.c[k] = m.rgb == 0 ? bg : bg * s.ssum;
    }

    // ---- level 0 first: which groups of 64 consecutive faces can touch this tile (union rects written by the setup kernel).
    // Three quarters of the tiles of a LASR frame meet none: they leave here with the background, before any per-pixel set-up
    // (pixel centres, reciprocals, quadrant corners are divisions).  In a four-wave workgroup every wave evaluates the same
    // test on the same data, so the masks agree without a barrier.
    const int G = groups_of(A.F);
    const short4* __restrict__ grects = A.grects + (size_t)bn * G;
    const int tX0 = tx * TW, tX1 = tX0 + TW - 1, tY0 = ty * TW, tY1 = tY0 + TW - 1;
    unsigned long long gmask = 0;                              // touched groups among [g_mask0, g_mask0 + 64)
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
    const float* __restrict__ texs = A.textures + (size_t)bn * A.F * A.T * NCH;
    const int texstride = A.T * NCH;
    const UniRecip U = uni_recip(A);
    const int ok_bit = U.ok ? 32 : 0;
    unsigned short* mine = mine_lds;
    const float thr_pad2 = A.thr * 1.10f;
    const float q_xlo = pix_center(qx0, IS), q_xhi = pix_center(min(qx0 + 7, IS - 1), IS);
    const float q_yhi = pix_center(IS - 1 - qy0, IS), q_ylo = pix_center(IS - 1 - min(qy0 + 7, IS - 1), IS);

    // does face f (index inside the image) reach this wave's 8x8 quadrant?  Pixel-rect overlap, then a tighter cull (lanes =
    // faces): the barycentric w_k is linear in the pixel position, so if all four corners of the quadrant lie beyond edge k's
    // line by more than sqrt(1.10 thr), every pixel of the quadrant does too, i.e. the reference's `dis >= threshold` test
    // (K.cu:402) drops every one of them.  Same faces contribute, ~20 % fewer entries to walk.  Well-conditioned faces only.
    auto touches_quadrant = [&](int f) -> bool {
        const short4 q = rects[f];
        bool hit = !(q.x > qx0 + 7 || q.y < qx0 || q.z > qy0 + 7 || q.w < qy0);
        if (hit && m.dist == 2) {
            const float* R = recs + (size_t)f * REC;
            if (__float_as_int(R[R_FLAGS]) & 16) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float a = R[R_INV + 3 * k], b = R[R_INV + 3 * k + 1], c = R[R_INV + 3 * k + 2];
                    const float w00 = a * q_xlo + b * q_ylo + c, w01 = a * q_xhi + b * q_ylo + c;
                    const float w10 = a * q_xlo + b * q_yhi + c, w11 = a * q_xhi + b * q_yhi + c;
                    const float wmax = fmaxf(fmaxf(w00, w01), fmaxf(w10, w11));   // least negative corner
                    if (wmax < 0.f && wmax * wmax * R[R_HK2 + k] > thr_pad2) hit = false;
                }
            }
        }
        return hit;
    };

    int g_next = 64;                                           // first group not yet looked at (the first 64 were tested above)
    int g_mask0 = 0;
    bool more = true;
    for (int round = 0; more; round++) {                      // one round unless a tile meets more than LIST_CAP - 256 faces
        if (round > 0 && !W1) __syncthreads();                 // the previous round's lists are still being walked
        // ---- level 1 (workgroup): ordered compaction of the touched groups' faces whose rect touches the 16x16 tile; a
        // round ends when the list could overflow or the u16 ids (relative to `base`) could
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
            if constexpr (W1) {
                // one touched group per step: its 64 faces against the quadrant, survivors straight into the wave's list
                const int g = g_mask0 + __builtin_ctzll(gmask);
                if (base < 0) base = g * GROUP;
                if (count + 64 > CAP || (g + 1) * GROUP - base > 65536) break;             // walk what we have, then continue
                gmask &= gmask - 1;
                const int f = g * GROUP + lane;
                const bool hit = f < A.F && touches_quadrant(f);
                const unsigned long long mask = wave_mask(hit);
                if (hit) mine[count + bits_below_lane(mask)] = (unsigned short)(f - base);
                count += __popcll(mask);
                continue;
            }
            // the next (up to) four touched groups, one per wave, in index order
            unsigned long long mm = gmask;
            int mine_g = -1, last_g = 0, taken = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (mm) {
                    const int b = __builtin_ctzll(mm);
                    mm &= mm - 1;
                    if (k == wave) mine_g = g_mask0 + b;
                    last_g = g_mask0 + b;
                    taken++;
                }
            }
            const int first_g = g_mask0 + __builtin_ctzll(gmask);
            if (base < 0) base = first_g * GROUP;
            if (count + 256 > CAP || (last_g + 1) * GROUP - base > 65536) break;           // walk what we have, then continue
            gmask = mm;
            bool hit = false;
            const int f = mine_g * GROUP + lane;
            if (mine_g >= 0 && f < A.F) {
                const short4 q = rects[f];
                hit = !(q.x > tX1 || q.y < tX0 || q.z > tY1 || q.w < tY0);
            }
            const unsigned long long mask = wave_mask(hit);
            if (lane == 0) s_wcnt[flip][wave] = __popcll(mask);
            __syncthreads();
            const int c0 = s_wcnt[flip][0], c1 = s_wcnt[flip][1], c2 = s_wcnt[flip][2], c3 = s_wcnt[flip][3];
            const int before = wave == 0 ? 0 : wave == 1 ? c0 : wave == 2 ? c0 + c1 : c0 + c1 + c2;
            if (hit) s_all[count + before + bits_below_lane(mask)] = (unsigned short)(f - base);
            count += c0 + c1 + c2 + c3;
            flip ^= 1;
            (void)taken;
        }
        if (base < 0) base = 0;
        int n_mine = 0;
        if constexpr (!W1) {
        __syncthreads();
        // ---- level 2 (wave, no barriers from here on): 64 list entries at a time, keep those touching the quadrant
        for (int i0 = 0; i0 < count; i0 += 64) {
            bool hit = false;
            int e = 0;
            if (i0 + lane < count) {
                e = s_all[i0 + lane];
                hit = touches_quadrant(base + e);
            }
            const unsigned long long mask = wave_mask(hit);
            if (hit) mine[n_mine + bits_below_lane(mask)] = (unsigned short)e;
            n_mine += __popcll(mask);
        }
        } else n_mine = count;
        __builtin_amdgcn_wave_barrier();
        // ---- walk: every entry has at least one candidate pixel in this wave
#if defined(LASR_ABL) && LASR_ABL == 2              // measurement build: binning only
        s.a += (float)n_mine; n_mine = 0;
#endif
        for (int i0 = 0; i0 < n_mine; i0 += 64) {
            const int chunk = (i0 + lane < n_mine) ? (int)mine[i0 + lane] : 0;
            const int n = min(64, n_mine - i0);
            for (int j = 0; j < n; j++) {
                const int fn = base + __builtin_amdgcn_readlane(chunk, j);     // wave-uniform -> scalar loads
                const cptr_t rec = as_const(recs + (size_t)fn * REC);
#if LASR_PREFETCH
                // The walk is a chain of dependent scalar loads per entry: rect, wait, test; first record line, wait; the other two
                // lines and the attributes on demand, wait again -- three round trips to L2 one after the other, and the occupancy
                // sweep (profiles/r05_occupancy_sweep.txt) puts a third of the launch in exposed latency even at eight waves per
                // SIMD.  Touch the entry's other two record lines and its attribute line together with the rect load: one
                // round trip, the later loads hit the scalar cache.  The touched words are never used (no arithmetic changes).
                int pf1, pf2, pf3;
                {
                    const float* rn = recs + (size_t)fn * REC;
                    const float* tn = texs + (size_t)fn * texstride;
                    asm volatile("s_load_dword %0, %3, 0x40\n\ts_load_dword %1, %3, 0x80\n\ts_load_dword %2, %4, 0x0"
                                 : "=&s"(pf1), "=&s"(pf2), "=&s"(pf3) : "s"(rn), "s"(tn) : "memory");
                }
#endif
                // exact integer form of the bbox test K.cu:375 (see first_pixel_ge / last_pixel_le)
                const bool cand = rect_has(__float_as_int(rec[R_BB + 0]), __float_as_int(rec[R_BB + 1]), pxy);
#if LASR_PREFETCH
                asm volatile("s_waitcnt lgkmcnt(0)" : : "s"(pf1), "s"(pf2), "s"(pf3));      // (the rect test has waited already)
#endif
                float w0, w1, w2;
                barycentric(rec, xp, yp, w0, w1, w2);
                const int lim = (A.N * A.F - (bn * A.F + fn)) * A.T;   // texels to the end of the tensor
                const cptr_t tex = as_const(texs + (size_t)fn * texstride);
                const bool mk = (__float_as_int(rec[R_FLAGS]) & ok_bit) != 0;     // wave-uniform; ok_bit = U.ok ? 32 : 0 (no branch on U.ok per entry)
#if defined(LASR_ABL) && LASR_ABL == 1      // measurement build: binning + walk + reject only
                if (cand) s.a += w0 + w1 + w2;
                (void)lim; (void)tex; (void)mk;
#elif defined(LASR_ABL) && LASR_ABL == 3    // measurement build: + distance and sigmoid, nothing after them
                if (cand) { Frag fr; if (fragment_w<false, true>(rec, m.dist, A.thr, A.sigma, xp, yp, w0, w1, w2, fr, U.inv_sigma)) s.a += fr.D; }
                (void)lim; (void)tex; (void)mk;
#else
                if (cand) {
                    if (mk) forward_face<LASR_FAST, true, NCH, RX>(A, m, rec, tex, fn, lim, xp, yp, w0, w1, w2, s, U);
                    else forward_face<LASR_FAST, false, NCH>(A, m, rec, tex, fn, lim, xp, yp, w0, w1, w2, s, U);
                }
#endif
            }
        }
    }

    }   // tile meets at least one group

    if (!valid) return;
    // ---- finalise (K.cu:458-482)
    float a_out;
    if (m.alpha == 0) a_out = s.a;
    else if (m.alpha == 1) a_out = s.a / A.F;
    else a_out = (float)(1. - (double)s.a);
#if defined(LASR_ABL) && LASR_ABL == 4              // measurement build: everything but the output stores (what do they cost?)
    if (a_out != 12345.678f || s.ssum != 3.25f) return;
#endif
    colors[((size_t)bn * (NCH + 1) + NCH) * P + pn] = a_out;
    if (m.rgb == 0) {
        if (s.fbest != -1 || A.use_bg) {             // no face: the pre-filled background stays -- or is written now
#pragma unroll
            for (int k = 0; k < NCH; k++) colors[((size_t)bn * (NCH + 1) + k) * P + pn] = s.c[k];
        }
        aggrs[((size_t)bn * 2 + 0) * P + pn] = s.ssum;
        aggrs[((size_t)bn * 2 + 1) * P + pn] = (float)s.fbest;
    } else {
#pragma unroll
        for (int k = 0; k < NCH; k++) colors[((size_t)bn * (NCH + 1) + k) * P + pn] = s.c[k] / s.ssum;
        aggrs[((size_t)bn * 2 + 0) * P + pn] = s.ssum;
        aggrs[((size_t)bn * 2 + 1) * P + pn] = s.smax;
    }
}

template <bool LASR_FAST, int NCH, bool RX = false, bool W1 = false>
__global__ __launch_bounds__(256) void sr_forward_kernel(RasterArgs A, float* __restrict__ aggrs,
                                                         float* __restrict__ colors)
{
    constexpr int NW = W1 ? 1 : 4, TW = W1 ? 8 : TILE;
    __shared__ unsigned short s_all[W1 ? 1 : LIST_CAP];    // faces whose pixel rect touches the 16x16 tile, index order
    __shared__ unsigned short s_mine[NW][LIST_CAP];        // per wave: the subset touching its 8x8 quadrant, index order
    __shared__ int s_wcnt[2][4];

    if (W1 && A.choice && chosen_kernel(A) != CHOICE_ONE_WAVE) return;      // the device took the cooperative kernel for this launch
#ifdef LASR_OCC_LDS        // measurement build (profiles/r05_occupancy_sweep.txt): extra LDS per workgroup caps the waves per SIMD
    __shared__ unsigned char s_occ_pad[W1 ? LASR_OCC_LDS : 1];
    if (A.N < 0) s_occ_pad[threadIdx.x] = (unsigned char)A.F;               // never true: keeps the allocation
#endif
    const int tiles_x = (A.IS + TW - 1) / TW;
    int bn, tx, ty;
    tile_of_block(blockIdx.x, gridDim.x, tiles_x, bn, tx, ty, W1 ? A.order : nullptr);
    forward_tile_body<LASR_FAST, NCH, RX, W1>(A, aggrs, colors, bn, tx, ty, s_all, s_mine[W1 ? 0 : threadIdx.x >> 6], s_wcnt);
}


}  // namespace lasr
#include "sr_forward_coop.h"
#include "sr_forward_pairs.h"
namespace lasr {

}  // namespace lasr
#include "sr_backward.h"
namespace lasr {

// LASR-mode instantiations live in sr_backward_fast.hip (compiled with FMA contraction)
void launch_backward_fast(int nch, dim3 grid, hipStream_t st, const RasterArgs& A, const float* colors, const float* aggrs,
                          const float* gcolors, float* gfaces, float* gtex);

// Self-test of the exact division-by-reciprocal: counts pairs where it differs (bitwise) from the IEEE quotient.
__global__ __launch_bounds__(256) void selftest_div_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           int* __restrict__ mismatches, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float y = 1.f / b[i];
    if (!recip_safe(b[i])) return;
    const float q = div_by_recip(a[i], b[i], y), r = a[i] / b[i];
    if (__float_as_int(q) != __float_as_int(r) && !(q != q && r != r)) atomicAdd(mismatches, 1);
}

// Self-test of the shared-divisor division of clip_normalise (sr_device.h: div3_shared) against the IEEE quotients.
__global__ __launch_bounds__(256) void selftest_div3_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            int* __restrict__ mismatches, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float q0 = a[3 * (size_t)i], q1 = a[3 * (size_t)i + 1], q2 = a[3 * (size_t)i + 2];
    const float r0 = q0 / b[i], r1 = q1 / b[i], r2 = q2 / b[i];
    div3_shared(q0, q1, q2, b[i]);
    const int bad = (__float_as_int(q0) != __float_as_int(r0)) + (__float_as_int(q1) != __float_as_int(r1)) +
                    (__float_as_int(q2) != __float_as_int(r2));
    if (bad) atomicAdd(mismatches, bad);
}

}  // namespace lasr

// ===========================================================================
// C ABI
// ===========================================================================
using namespace lasr;


static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int groups_of_host(int F) { return (F + GROUP - 1) / GROUP; }


extern "C" size_t lasr_sr_workspace_bytes(int N, int F, int T, int IS)
{
    (void)T;
    if (N < 0 || F < 0) return 0;
    const size_t nf = (size_t)N * (size_t)F;
    const size_t ng = (size_t)N * (size_t)((F + GROUP - 1) / GROUP);
    const size_t t8 = IS > 0 ? (size_t)(IS + 7) / 8 : 0;
    return align_up(nf * REC * sizeof(float), 256) + align_up(nf * sizeof(short4), 256) + align_up(ng * sizeof(short4), 256) + 256 /* sr_choose_kernel's word */
           + align_up((size_t)N * t8 * t8 * sizeof(int), 256) /* sr_order_kernel's table */ + align_up((size_t)N * t8 * t8, 256) /* tile weights */ + 256;
}

static int check_common(int N, int F, int T, int IS, int dist, int rgb, int alpha, int tex)
{
    if (N < 0 || F < 0 || T < 1 || IS < 0) return LASR_E_BADARG;
    if (dist < 0 || dist > 2 || rgb < 0 || rgb > 1 || alpha < 0 || alpha > 2 || tex < 0 || tex > 1) return LASR_E_BADMODE;
    if ((long long)N * F > 0x7fffffffLL / 64 || (long long)N * IS * IS > 0x7fffffffLL) return LASR_E_BADARG;
    if (IS > 32767) return LASR_E_BADARG;   // pixel rects are stored as int16
    return LASR_OK;
}


static RasterArgs make_args(void* ws, const float* textures, int N, int F, int T, int IS, float near, float far,
                            float eps, float sigma, int dist, float dist_eps, float gamma, int rgb, int alpha,
                            int tex, int double_side, float** recs, short4** rects, short4** grects)
{
    const size_t nf = (size_t)N * (size_t)F;
    char* p = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    *recs = (float*)p;
    *rects = (short4*)(p + align_up(nf * REC * sizeof(float), 256));
    *grects = (short4*)((char*)*rects + align_up(nf * sizeof(short4), 256));
    RasterArgs A;
    A.recs = *recs; A.rects = *rects; A.grects = *grects; A.textures = textures;
    A.N = N; A.F = F; A.T = T; A.res = (int)sqrt((double)T); A.IS = IS;   // K.cu:696
    A.near = near; A.far = far; A.near_far_dev = nullptr; A.eps = eps; A.sigma = sigma; A.gamma = gamma;
    A.thr = dist_eps * sigma;                                              // K.cu:352 (float product)
    A.inv_is = 1.f / (float)IS;
    A.cx_a = 2.f * A.inv_is; A.cx_b = (float)(1 - IS) * A.inv_is; A.cy_b = (float)(IS - 1) * A.inv_is;
    { const float thr_pad = A.thr * 1.05f; A.far_t = -sqrtf(thr_pad); }
    A.m = Modes{dist, rgb, alpha, tex, double_side ? 1 : 0};
    A.overwrite_grads = 0;
    A.use_bg = 0;
    A.choice = nullptr;
    A.choice_max = -1;
    A.order = nullptr;
    return A;
}

// nothing in this file is mutable process state: the forward arithmetic is a per-call flag, the kernel-choice thresholds a per-call
// lasr_sr_options (their built-in defaults below are constants read once at load time)
static int default_flags() { return 0; }

static long long env_blocks(const char* name, long long dflt)
{
    const char* e = getenv(name);
    return e ? atoll(e) : dflt;
}
// Which forward kernel a launch of LASR's mode combination takes.  All sizes are 8x8-pixel tiles (frames x tiles per frame):
//   up to g_coop8_max_tiles              eight waves share a tile        (sr_forward_coop.h: latency designs for launches that
//   up to g_coop_max_tiles               four waves share a tile          cannot fill the chip)
//   up to g_choose_max_tiles             decided on the device: sr_choose_kernel estimates the BUSY tiles from the meshes' pixel
//                                        bounding boxes -- at most g_coop_max_tiles: four waves per tile, else one wave per tile
//   above                                one wave per tile (W1)
// Six and nine channels hand more through LDS per entry (fewer tiles in flight per CU): their bounds are 5/8 of the numbers,
// and they skip the device-side range (see forward_impl).  The device-side choice costs two launches of ~5 us (the chooser and
// the kernel that returns at once).
// Measured on an MI355X, forward kernel ms (profiles/r03_coop_ab.txt).  Mesh M2 (2420 faces, the object covers a third of the
// tiles) at 256x256 (1024 tiles per frame), three channels, [four waves per 16x16 tile, the choice until then] -> four / eight waves:
//   1 frame 0.209 -> 0.084 / 0.052     2 frames 0.205 -> 0.084 / 0.057     4 frames 0.211 -> 0.094 / 0.087
//   8 frames 0.223 -> 0.140 / 0.140    16 frames 0.285 -> 0.219 / 0.256    24 frames 0.355 -> 0.298 / 0.381
//   32 frames 0.424 -> 0.397 / 0.528   48 frames 0.529 -> 0.564 / 0.762    (64 frames 0.648 -> 0.78; 256 frames 2.18 -> 3.13)
// Nine channels (the render of a LASR step), 1280 faces: 2 meshes 0.136 -> 0.065 / 0.044, 4 meshes 0.135 -> 0.071 / 0.064,
// 16 meshes 0.203 -> 0.166, 32 meshes 0.287 -> 0.309; 16 meshes of 2420 faces 0.336 -> 0.284.  The SAME 16 x 1280 faces filling
// the frame (937 of 1024 tiles busy, lists of at most 39 entries: the first iterations of optimize.py): one wave per tile
// 0.149, four waves per 16x16 tile 0.151, cooperative 0.187 -- the launch is throughput bound and the chooser must say so.
// W1 against four waves per 16x16 tile: 256 frames 2.370 -> 2.163 ms, 64 frames 0.696 -> 0.649, 16 frames 0.284 -> 0.280,
// 4 frames 0.210 -> 0.221 (profiles/r03_w1_ab.txt): launches that small now take the cooperative kernel, and the four-wave
// 16x16 kernel is left to the other mode combinations.
// Environment overrides LASR_SR_COOP8_MAX_TILES / LASR_SR_COOP_MAX_TILES / LASR_SR_CHOOSE_MAX_TILES, read once at load time;
// per call: lasr_sr_options (lasr_sr_forward_opt).  The output is bit-identical whichever kernel runs.
static const long long k_coop8_max_tiles = env_blocks("LASR_SR_COOP8_MAX_TILES", 2200);
static const long long k_coop_max_tiles = env_blocks("LASR_SR_COOP_MAX_TILES", 14336);
static const long long k_choose_max_tiles = env_blocks("LASR_SR_CHOOSE_MAX_TILES", 49152);
// launches of up to this many 8x8 tiles (five frames and more, tile total a multiple of 8) issue their tiles heaviest first
static const long long k_order_max_tiles = env_blocks("LASR_SR_ORDER_MAX_TILES", 1ll << 40);
// LASR's mode combination, launches of at least this many 8x8 tiles: the pair-walk kernel (sr_forward_pairs.h), whose lanes walk
// their own pixel's (pixel, face) pairs; smaller launches keep the latency designs above.  Its output differs from theirs in the
// rounding sequence only (image within ~1e-6).  LASR_SR_PAIR_MIN_TILES at load time, lasr_sr_options.pair_min_tiles per call.
// Measured on an MI355X (profiles/experiments/README.md, r06 sections), forward kernel ms, mesh M2 at 256x256, three channels:
//   frames                      1      2      3      4      6      8      12     16     64     256
//   cooperative 8x8 kernels    .051   .054   .085   .085   .093   .095   .107     -      -      -     (one wave per tile: .188 at 16, 1.93 at 256)
//   pair walk, two teams       .068   .067   .069   .069   .079   .079   .103   .129   .459   1.86
//   pair walk, one team        .104   .103     -    .105     -    .106   .107   .115   .322   1.32
//   nine channels: the render of spot3 stage 0 (16 meshes of 1280 faces filling the frame) 104 -> 86 us (two teams: 132), camel stage 4
//   (4 meshes of 2560 faces at 512x512) 150 -> 118 us -- every channel count and face size takes the kernel from the threshold up
static const long long k_pair_min_tiles = env_blocks("LASR_SR_PAIR_MIN_TILES", 3072);
// ... with TWO teams of four waves per 16x16 tile (sr_forward_pairs.h: SPLIT) up to this many 16x16 tiles: a launch that cannot fill
// the chip takes as long as its heaviest tile's chain of dependent phases (list rounds, then stage -> classify -> walk per chunk of 64
// entries: ~105 us on the bench object whatever the frame count), and two teams walk alternate chunks (~68 us); beyond, throughput
// counts and one team's smaller footprint wins.  LASR_SR_PAIR_ONE_TEAM / LASR_SR_PAIR_TWO_TEAMS force either per call.
static const long long k_pair_teams_max_tiles = env_blocks("LASR_SR_PAIR_TEAMS_MAX_TILES", 3072);

static bool is_lasr_fast(const Modes& m) { return m.dist == 2 && m.rgb == 1 && m.alpha == 2 && m.tex == 1 && m.double_side; }

static int forward_impl(const float* faces, const float* textures, float* faces_info, float* aggrs_info,
                        float* soft_colors, void* workspace, size_t workspace_bytes, int N, int F, int T, int IS,
                        float near, float far, float eps, float sigma_val, int func_id_dist, float dist_eps,
                        float gamma_val, int func_id_rgb, int func_id_alpha, int texture_sample_type,
                        int double_side, void* hip_stream, int nch, const float* near_far_dev, int flags,
                        const float* background = nullptr, const lasr_sr_options* opt = nullptr)
{
    const long long g_coop8_max_tiles = opt && opt->coop8_max_tiles >= 0 ? opt->coop8_max_tiles : k_coop8_max_tiles;
    const long long g_coop_max_tiles = opt && opt->coop_max_tiles >= 0 ? opt->coop_max_tiles : k_coop_max_tiles;
    const long long g_choose_max_tiles = opt && opt->choose_max_tiles >= 0 ? opt->choose_max_tiles : k_choose_max_tiles;
    const long long g_order_max_tiles = opt && opt->order_max_tiles >= 0 ? opt->order_max_tiles : k_order_max_tiles;
    int rc = check_common(N, F, T, IS, func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type);
    if (rc) return rc;
    if (N == 0 || IS == 0) return LASR_OK;
    if (!aggrs_info || !soft_colors || (F > 0 && (!faces || !textures))) return LASR_E_BADARG;
    if (!workspace || workspace_bytes < lasr_sr_workspace_bytes(N, F, T, IS)) return LASR_E_WORKSPACE;
    hipStream_t st = (hipStream_t)hip_stream;
    float* recs; short4* rects; short4* grects;
    RasterArgs A = make_args(workspace, textures, N, F, T, IS, near, far, eps, sigma_val, func_id_dist, dist_eps,
                             gamma_val, func_id_rgb, func_id_alpha, texture_sample_type, double_side, &recs, &rects, &grects);
    A.near_far_dev = near_far_dev;
    if (background) {
        A.use_bg = 1;
        for (int k = 0; k < nch; k++) A.bg[k] = background[k];
    }
    const int total = N * F;
    // the 8x8-tile kernels, five frames and more: this launch's own tile order (sr_order_kernel)
    const int t8o = (IS + 7) / 8;
    const long long tiles8o = (long long)N * t8o * t8o;
    const long long g_pair_min_tiles = opt && opt->pair_min_tiles >= 0 ? opt->pair_min_tiles : k_pair_min_tiles;
    // the pair-walk kernel: LASR's modes, default arithmetic, launches from pair_min_tiles up; its tiles are 16x16 pixels
    const bool pairs = total > 0 && (nch > 3 || is_lasr_fast(A.m)) && !(flags & (LASR_SR_RELAXED_MATH | LASR_SR_SEGMENTED)) &&
                       tiles8o >= g_pair_min_tiles;
    const int tile_shift = pairs ? 4 : 3;
    const int tso = (IS + (1 << tile_shift) - 1) >> tile_shift;              // tiles per side of the order table
    const long long tileso = (long long)N * tso * tso;
    // (an XCD walks the crowded tiles of ALL its N / 8 images at once while their records fit about twice its 4 MB L2 -- 128 frames of
    // 2420 faces, 7.4 MB per XCD: forward -8 % -- beyond that the record fetch multiplies for nothing: 256 frames, 14.9 MB: FETCH_SIZE
    // 172 MB -> 1.18 GB per launch.  Larger launches sort and issue the XCD's images four at a time, the interleave of the fixed
    // order: 256 frames, forward 1.964 -> 1.915 ms + 15 us of order kernels; groups of 2 / 16: 1.924 / 1.925)
    // (below 5 frames the two order launches, 12 us, cost more than the order gains: 4 frames forward .089 -> .081 ms, step .153 -> .155)
    static const int order_min_frames = (int)env_blocks("LASR_SR_ORDER_MIN_FRAMES", 5);
    static const int order_group_images = (int)env_blocks("LASR_SR_ORDER_GROUP_IMAGES", 4);
    int order_groups = 1;
    if ((long long)((N + 7) >> 3) * F * REC * (long long)sizeof(float) > (8ll << 20) || tileso > 8ll * ORDER_MAX_ENTRIES)
        order_groups = (N & 7) == 0 && order_group_images > 0 && (N >> 3) % order_group_images == 0 ? (N >> 3) / order_group_images : 0;
    const bool use_order = total > 0 && (nch > 3 || is_lasr_fast(A.m)) && N >= order_min_frames && (tileso & 7) == 0 &&
                           tiles8o <= g_order_max_tiles && tso <= ORDER_MAX_SIDE && order_groups > 0 &&
                           tileso <= 8ll * order_groups * ORDER_MAX_ENTRIES && N < 32768;
    char* const slot = (char*)grects + align_up((size_t)N * groups_of_host(F) * sizeof(short4), 256);   // [0] sr_choose_kernel's word
    if (total > 0) {
        {
            ProfScope ps(K_SR_SETUP, st);
            const int bpi = (F + 255) / 256;
            hipLaunchKernelGGL(sr_setup_kernel, dim3((unsigned)(N * bpi)), dim3(256), 0, st, faces, recs, rects, grects,
                               faces_info, F, bpi, sqrtf(A.thr), IS);
        }
        if ((rc = launch_ok())) return rc;
    }
    const int tiles_x = (IS + TILE - 1) / TILE;
    const dim3 grid((unsigned)(N * tiles_x * tiles_x));
    // which of the 8x8-tile kernels (LASR's mode combination; six / nine channels).  0: one wave per tile, 1: four waves, 2: eight
    // waves, 3: four waves AND one wave are launched and the device chooses.
    // (six / nine channels are LASR.forward's render: its frames are cropped around the object and nearly every tile is busy, so
    // the launch size itself is the estimate and the extra launches of the device-side choice are saved)
    // With the launch's own tile order the one-wave kernel catches up earlier (tools/prof/kernel_choice_sweep.py, forward + order
    // kernels, ms, four waves | one wave: bench object, a third of the tiles busy, 8 frames .115 | .211, 16: .182 | .228, 24:
    // .264 | .237, 32: .339 | .289; object filling the frame, 8: .151 | .154, 16: .275 | .217): four waves up to 4/7 of
    // coop_max_tiles, the device decides up to 7/16 of choose_max_tiles -- on the launch's count of NON-EMPTY tiles, which
    // sr_order_kernel has anyway (at most 3/8 of coop_max_tiles: four waves), not on sr_choose_kernel's bounding-box estimate.
    const bool tile_kernels = nch > 3 || is_lasr_fast(A.m);
    const bool rx = (flags & LASR_SR_RELAXED_MATH) && is_lasr_fast(A.m);
    const long long coop8_max = nch > 3 ? g_coop8_max_tiles / 8 * 5 : g_coop8_max_tiles;
    const long long coop_max_plain = nch > 3 ? g_coop_max_tiles / 8 * 5 : g_coop_max_tiles;
    const long long coop_max = use_order && nch == 3 ? coop_max_plain / 7 * 4 : coop_max_plain;      // (nine channels, 6 | 8 | 12 frames: .146 | .164 | .238 four, .165 | .167 | .200 one)
    const long long choose_max = nch > 3 ? coop_max : use_order ? g_choose_max_tiles / 16 * 7 : g_choose_max_tiles;
    int plan = !tile_kernels || rx ? 0 : tiles8o <= coop8_max ? 2 : tiles8o <= coop_max ? 1 :
               (tiles8o <= choose_max && total > 0 && coop_max > 0) ? 3 : 0;
    const bool seg_flag = (flags & LASR_SR_SEGMENTED) != 0;
    if (pairs) plan = 5;
    (void)seg_flag;
    if (use_order) {
        int* order = (int*)(slot + 256);
        unsigned char* keys = (unsigned char*)order + align_up((size_t)tileso * sizeof(int), 256);
        int* busy = plan == 3 ? (int*)slot : nullptr;
        ProfScope po(K_SR_ORDER, st);
        hipLaunchKernelGGL(sr_tile_weight_kernel, dim3((unsigned)N), dim3(512), (size_t)(tso + 1) * (tso + 1) * sizeof(int), st, rects, F, tso, keys, busy, tile_shift);
        hipLaunchKernelGGL(sr_order_kernel, dim3(8 * order_groups), dim3(ORDER_THREADS), align_up((size_t)(tileso / 8 / order_groups), 16), st,
                           keys, N, tso, order, busy);
        A.order = order;
        if (plan == 3) {
            A.choice = busy;
            A.choice_max = (int)std::min<long long>(coop_max_plain / 8 * 3, 0x7fffffff);
        }
    }
    {
        ProfScope ps(K_SR_FORWARD, st);
        if (tile_kernels) {
            const long long tiles8 = tiles8o;
            const dim3 grid8((unsigned)tiles8);
            if (plan == 3 && !use_order) {
                int* choice = (int*)slot;
                hipLaunchKernelGGL(sr_choose_kernel, dim3(1), dim3(64), 0, st, grects, N, groups_of_host(F), IS, coop_max, choice);
                A.choice = choice;
            }
            // LASR_SR_SEGMENTED pays up to about twice the eight-wave range (measured: -17 % at one frame, -6 % at four, +8 % at
            // sixteen, sr_forward_coop.h); beyond that the flag is ignored
            const bool seg = (flags & LASR_SR_SEGMENTED) != 0 && tiles8 <= 2 * coop8_max;
            if (seg && plan == 2) {
                if (nch == 9) hipLaunchKernelGGL((sr_forward_seg_kernel<9, 8>), grid8, dim3(512), 0, st, A, aggrs_info, soft_colors);
                else if (nch == 6) hipLaunchKernelGGL((sr_forward_seg_kernel<6, 8>), grid8, dim3(512), 0, st, A, aggrs_info, soft_colors);
                else hipLaunchKernelGGL((sr_forward_seg_kernel<3, 8>), grid8, dim3(512), 0, st, A, aggrs_info, soft_colors);
            } else if (seg && (plan == 1 || plan == 3)) {
                if (nch == 9) hipLaunchKernelGGL((sr_forward_seg_kernel<9, 4>), grid8, dim3(256), 0, st, A, aggrs_info, soft_colors);
                else if (nch == 6) hipLaunchKernelGGL((sr_forward_seg_kernel<6, 4>), grid8, dim3(256), 0, st, A, aggrs_info, soft_colors);
                else hipLaunchKernelGGL((sr_forward_seg_kernel<3, 4>), grid8, dim3(256), 0, st, A, aggrs_info, soft_colors);
            }
            if (!seg && plan == 2) {
                if (nch == 9) hipLaunchKernelGGL((sr_forward_coop_kernel<9, 8, 1, 0>), grid8, dim3(512), 0, st, A, aggrs_info, soft_colors);
                else if (nch == 6) hipLaunchKernelGGL((sr_forward_coop_kernel<6, 8, 1, 0>), grid8, dim3(512), 0, st, A, aggrs_info, soft_colors);
                else hipLaunchKernelGGL((sr_forward_coop_kernel<3, 8, 1, 0>), grid8, dim3(512), 0, st, A, aggrs_info, soft_colors);
            }
            if (!seg && (plan == 1 || plan == 3)) {
                if (nch == 9) hipLaunchKernelGGL((sr_forward_coop_kernel<9, 4, 1, 0>), grid8, dim3(256), 0, st, A, aggrs_info, soft_colors);
                else if (nch == 6) hipLaunchKernelGGL((sr_forward_coop_kernel<6, 4, 1, 0>), grid8, dim3(256), 0, st, A, aggrs_info, soft_colors);
                else hipLaunchKernelGGL((sr_forward_coop_kernel<3, 4, 2, 1>), grid8, dim3(256), 0, st, A, aggrs_info, soft_colors);
            }
            if (plan == 5) {
                const int t16 = (IS + PW_TILE - 1) / PW_TILE;
                const dim3 grid16((unsigned)(N * t16 * t16));
                // two teams per tile while the launch cannot fill the chip with one (its time is then the heaviest tile's chain of
                // phases); LASR_SR_PAIR_ONE_TEAM / LASR_SR_PAIR_TWO_TEAMS force either
                const long long tiles16 = (long long)N * t16 * t16;
                const bool teams2 = (flags & LASR_SR_PAIR_TWO_TEAMS) || (!(flags & LASR_SR_PAIR_ONE_TEAM) && tiles16 <= k_pair_teams_max_tiles);
                if (teams2) {
                    if (nch == 9) hipLaunchKernelGGL((sr_forward_pairs_teams_kernel<9, 2>), grid16, dim3(512), 0, st, A, aggrs_info, soft_colors);
                    else if (nch == 6) hipLaunchKernelGGL((sr_forward_pairs_teams_kernel<6, 2>), grid16, dim3(512), 0, st, A, aggrs_info, soft_colors);
                    else hipLaunchKernelGGL((sr_forward_pairs_teams_kernel<3, 2>), grid16, dim3(512), 0, st, A, aggrs_info, soft_colors);
                } else if (nch == 9) hipLaunchKernelGGL((sr_forward_pairs_kernel<9>), grid16, dim3(256), 0, st, A, aggrs_info, soft_colors);
                else if (nch == 6) hipLaunchKernelGGL((sr_forward_pairs_kernel<6>), grid16, dim3(256), 0, st, A, aggrs_info, soft_colors);
                else hipLaunchKernelGGL(sr_forward_pairs3_kernel, grid16, dim3(256), 0, st, A, aggrs_info, soft_colors);
            }
            if (plan == 0 || plan == 3) {
                if (nch == 9 && rx) hipLaunchKernelGGL((sr_forward_kernel<true, 9, true, true>), grid8, dim3(64), 0, st, A, aggrs_info, soft_colors);
                else if (nch == 9) hipLaunchKernelGGL((sr_forward_kernel<true, 9, false, true>), grid8, dim3(64), 0, st, A, aggrs_info, soft_colors);
                else if (nch == 6 && rx) hipLaunchKernelGGL((sr_forward_kernel<true, 6, true, true>), grid8, dim3(64), 0, st, A, aggrs_info, soft_colors);
                else if (nch == 6) hipLaunchKernelGGL((sr_forward_kernel<true, 6, false, true>), grid8, dim3(64), 0, st, A, aggrs_info, soft_colors);
                else if (rx) hipLaunchKernelGGL((sr_forward_kernel<true, 3, true, true>), grid8, dim3(64), 0, st, A, aggrs_info, soft_colors);
                else hipLaunchKernelGGL((sr_forward_kernel<true, 3, false, true>), grid8, dim3(64), 0, st, A, aggrs_info, soft_colors);
            }
        } else hipLaunchKernelGGL((sr_forward_kernel<false, 3>), grid, dim3(256), 0, st, A, aggrs_info, soft_colors);
    }
    return launch_ok();
}

static int backward_impl(const float* faces, const float* textures, const float* soft_colors,
                         const float* faces_info, const float* aggrs_info, float* grad_faces,
                         float* grad_textures, const float* grad_soft_colors, void* workspace,
                         size_t workspace_bytes, int N, int F, int T, int IS, float near, float far, float eps,
                         float sigma_val, int func_id_dist, float dist_eps, float gamma_val, int func_id_rgb,
                         int func_id_alpha, int texture_sample_type, int double_side, void* hip_stream, int nch,
                         const float* near_far_dev, int flags)
{
    (void)faces_info;
    int rc = check_common(N, F, T, IS, func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type);
    if (rc) return rc;
    if (N == 0 || IS == 0 || F == 0) return LASR_OK;
    if (!faces || !textures || !soft_colors || !aggrs_info || !grad_faces || !grad_textures || !grad_soft_colors)
        return LASR_E_BADARG;
    // the backward touches the records, the rects and the group rects only: a workspace without the forward's tile-order table
    // (lasr_sr_workspace_bytes(N, F, T, 0), the size of ABI versions 1-2) is enough
    if (!workspace || workspace_bytes < lasr_sr_workspace_bytes(N, F, T, 0)) return LASR_E_WORKSPACE;
    hipStream_t st = (hipStream_t)hip_stream;
    float* recs; short4* rects; short4* grects;
    RasterArgs A = make_args(workspace, textures, N, F, T, IS, near, far, eps, sigma_val, func_id_dist, dist_eps,
                             gamma_val, func_id_rgb, func_id_alpha, texture_sample_type, double_side, &recs, &rects, &grects);
    A.near_far_dev = near_far_dev;
    A.overwrite_grads = (flags & LASR_SR_GRADS_OVERWRITE) && texture_sample_type == 1;
    const int total = N * F;
    if (!(flags & LASR_SR_RECORDS_VALID)) {      // the caller vouches that the forward's records are still in the workspace
        {
            ProfScope ps(K_SR_SETUP, st);
            const int bpi = (F + 255) / 256;
            hipLaunchKernelGGL(sr_setup_kernel, dim3((unsigned)(N * bpi)), dim3(256), 0, st, faces, recs, rects, grects,
                               (float*)nullptr, F, bpi, sqrtf(A.thr), IS);
        }
        if ((rc = launch_ok())) return rc;
    }
    // one wave (= one face) per workgroup: faces differ a lot in size, and a 4-wave workgroup holds its LDS and wave slots
    // until its largest face is done (measured: 1.65 ms vs 1.78 ms per 256 frames with 4 waves, 1.90 ms with 8)
    const dim3 grid((unsigned)total);
    {
        ProfScope ps(K_SR_BACKWARD, st);
        if (nch > 3 || is_lasr_fast(A.m))
            launch_backward_fast(nch, grid, st, A, soft_colors, aggrs_info, grad_soft_colors, grad_faces, grad_textures);
        else
            hipLaunchKernelGGL((sr_backward_kernel<false, 3>), grid, dim3(BWD_THREADS), 0, st, A, soft_colors, aggrs_info,
                               grad_soft_colors, grad_faces, grad_textures);
    }
    return launch_ok();
}

static int check_nch(int channels, int dist, int rgb, int alpha, int tex, int double_side, int T)
{
    if (channels == 3) return LASR_OK;
    // the 6- and 9-channel passes exist for LASR's renders only (two flow attribute triples; + the texture colours when the
    // three renders of a step share their geometry): euclidean / softmax / prod / vertex attributes, double sided
    if ((channels != 6 && channels != 9) || T != 3 || !(dist == 2 && rgb == 1 && alpha == 2 && tex == 1 && double_side)) return LASR_E_BADMODE;
    return LASR_OK;
}

extern "C" int lasr_sr_forward(const float* faces, const float* textures, float* faces_info, float* aggrs_info,
                               float* soft_colors, void* workspace, size_t workspace_bytes, int N, int F, int T, int IS,
                               float near, float far, float eps, float sigma_val, int func_id_dist, float dist_eps,
                               float gamma_val, int func_id_rgb, int func_id_alpha, int texture_sample_type,
                               int double_side, void* hip_stream)
{
    return forward_impl(faces, textures, faces_info, aggrs_info, soft_colors, workspace, workspace_bytes, N, F, T, IS, near,
                        far, eps, sigma_val, func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha,
                        texture_sample_type, double_side, hip_stream, 3, nullptr, default_flags());
}

extern "C" int lasr_sr_backward(const float* faces, const float* textures, const float* soft_colors,
                                const float* faces_info, const float* aggrs_info, float* grad_faces,
                                float* grad_textures, const float* grad_soft_colors, void* workspace,
                                size_t workspace_bytes, int N, int F, int T, int IS, float near, float far, float eps,
                                float sigma_val, int func_id_dist, float dist_eps, float gamma_val, int func_id_rgb,
                                int func_id_alpha, int texture_sample_type, int double_side, void* hip_stream)
{
    return backward_impl(faces, textures, soft_colors, faces_info, aggrs_info, grad_faces, grad_textures, grad_soft_colors,
                         workspace, workspace_bytes, N, F, T, IS, near, far, eps, sigma_val, func_id_dist, dist_eps,
                         gamma_val, func_id_rgb, func_id_alpha, texture_sample_type, double_side, hip_stream, 3, nullptr, 0);
}

// Multi-attribute variants: `channels` per-vertex attributes (3 or 6) interpolated and depth-blended in ONE pass over the
// geometry.  textures [N,F,3,channels], soft_colors / grad_soft_colors [N,channels+1,IS,IS] (alpha last), grad_textures
// [N,F,3,channels].  near_far_dev may be NULL (then near/far are used).  See include/lasr_sr.h.
extern "C" int lasr_sr_forward_attr(const float* faces, const float* textures, float* aggrs_info, float* soft_colors,
                                    void* workspace, size_t workspace_bytes, int N, int F, int channels, int IS,
                                    float near, float far, const float* near_far_dev, float eps, float sigma_val,
                                    int func_id_dist, float dist_eps, float gamma_val, int func_id_rgb, int func_id_alpha,
                                    int texture_sample_type, int double_side, void* hip_stream)
{
    const int rc = check_nch(channels, func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type, double_side, 3);
    if (rc) return rc;
    return forward_impl(faces, textures, nullptr, aggrs_info, soft_colors, workspace, workspace_bytes, N, F, 3, IS,
                        near, far, eps, sigma_val, func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha,
                        texture_sample_type, double_side, hip_stream, channels, near_far_dev, default_flags());
}

extern "C" int lasr_sr_backward_attr(const float* faces, const float* textures, const float* soft_colors,
                                     const float* aggrs_info, float* grad_faces, float* grad_textures,
                                     const float* grad_soft_colors, void* workspace, size_t workspace_bytes, int N, int F,
                                     int channels, int IS, float near, float far, const float* near_far_dev, float eps,
                                     float sigma_val, int func_id_dist, float dist_eps, float gamma_val, int func_id_rgb,
                                     int func_id_alpha, int texture_sample_type, int double_side, void* hip_stream)
{
    const int rc = check_nch(channels, func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type, double_side, 3);
    if (rc) return rc;
    return backward_impl(faces, textures, soft_colors, nullptr, aggrs_info, grad_faces, grad_textures,
                         grad_soft_colors, workspace, workspace_bytes, N, F, 3, IS, near, far, eps, sigma_val,
                         func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha, texture_sample_type,
                         double_side, hip_stream, channels, near_far_dev, 0);
}

// near/far taken from device memory ({near, far} as two floats): LASR recomputes them from the projected
// vertices every iteration (nnutils/mesh_net.py:304-311); the reference converts the 0-dim tensors to Python
// floats at each extension call (an implicit device->host sync per call), this variant never leaves the device.
extern "C" int lasr_sr_forward_dev(const float* faces, const float* textures, float* faces_info, float* aggrs_info,
                                   float* soft_colors, void* workspace, size_t workspace_bytes, int N, int F, int T, int IS,
                                   const float* near_far_dev, float eps, float sigma_val, int func_id_dist, float dist_eps,
                                   float gamma_val, int func_id_rgb, int func_id_alpha, int texture_sample_type,
                                   int double_side, void* hip_stream)
{
    if (!near_far_dev) return LASR_E_BADARG;
    return forward_impl(faces, textures, faces_info, aggrs_info, soft_colors, workspace, workspace_bytes, N, F, T, IS, 0.f,
                        0.f, eps, sigma_val, func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha,
                        texture_sample_type, double_side, hip_stream, 3, near_far_dev, default_flags());
}

extern "C" int lasr_sr_backward_dev(const float* faces, const float* textures, const float* soft_colors,
                                    const float* faces_info, const float* aggrs_info, float* grad_faces,
                                    float* grad_textures, const float* grad_soft_colors, void* workspace,
                                    size_t workspace_bytes, int N, int F, int T, int IS, const float* near_far_dev,
                                    float eps, float sigma_val, int func_id_dist, float dist_eps, float gamma_val,
                                    int func_id_rgb, int func_id_alpha, int texture_sample_type, int double_side,
                                    void* hip_stream)
{
    if (!near_far_dev) return LASR_E_BADARG;
    return backward_impl(faces, textures, soft_colors, faces_info, aggrs_info, grad_faces, grad_textures, grad_soft_colors,
                         workspace, workspace_bytes, N, F, T, IS, 0.f, 0.f, eps, sigma_val, func_id_dist, dist_eps,
                         gamma_val, func_id_rgb, func_id_alpha, texture_sample_type, double_side, hip_stream, 3,
                         near_far_dev, 0);
}

// Supersets of the entry points above (include/lasr_sr.h): every option is an argument, nothing is read from process state.
extern "C" int lasr_sr_forward_ex(const float* faces, const float* textures, float* faces_info, float* aggrs_info,
                                  float* soft_colors, void* workspace, size_t workspace_bytes, int N, int F, int T,
                                  int channels, int IS, float near, float far, const float* near_far_dev, float eps,
                                  float sigma_val, int func_id_dist, float dist_eps, float gamma_val, int func_id_rgb,
                                  int func_id_alpha, int texture_sample_type, int double_side, int flags, void* hip_stream)
{
    if (flags == LASR_SR_DEFAULT_FLAGS) flags = default_flags();
    if (flags & ~(LASR_SR_RELAXED_MATH | LASR_SR_SEGMENTED | LASR_SR_PAIR_ONE_TEAM | LASR_SR_PAIR_TWO_TEAMS)) return LASR_E_BADARG;
    const int rc = check_nch(channels, func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type, double_side, T);
    if (rc) return rc;
    return forward_impl(faces, textures, faces_info, aggrs_info, soft_colors, workspace, workspace_bytes, N, F, T, IS, near,
                        far, eps, sigma_val, func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha,
                        texture_sample_type, double_side, hip_stream, channels, near_far_dev, flags);
}

extern "C" int lasr_sr_forward_bg(const float* faces, const float* textures, float* faces_info, float* aggrs_info,
                                  float* soft_colors, void* workspace, size_t workspace_bytes, int N, int F, int T,
                                  int channels, int IS, float near, float far, const float* near_far_dev, float eps,
                                  float sigma_val, int func_id_dist, float dist_eps, float gamma_val, int func_id_rgb,
                                  int func_id_alpha, int texture_sample_type, int double_side, const float* background, int flags,
                                  void* hip_stream)
{
    if (!background) return LASR_E_BADARG;
    if (flags == LASR_SR_DEFAULT_FLAGS) flags = default_flags();
    if (flags & ~(LASR_SR_RELAXED_MATH | LASR_SR_SEGMENTED | LASR_SR_PAIR_ONE_TEAM | LASR_SR_PAIR_TWO_TEAMS)) return LASR_E_BADARG;
    const int rc = check_nch(channels, func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type, double_side, T);
    if (rc) return rc;
    return forward_impl(faces, textures, faces_info, aggrs_info, soft_colors, workspace, workspace_bytes, N, F, T, IS, near,
                        far, eps, sigma_val, func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha,
                        texture_sample_type, double_side, hip_stream, channels, near_far_dev, flags, background);
}

extern "C" int lasr_sr_forward_opt(const float* faces, const float* textures, float* faces_info, float* aggrs_info,
                                   float* soft_colors, void* workspace, size_t workspace_bytes, int N, int F, int T,
                                   int channels, int IS, float near, float far, const float* near_far_dev, float eps,
                                   float sigma_val, int func_id_dist, float dist_eps, float gamma_val, int func_id_rgb,
                                   int func_id_alpha, int texture_sample_type, int double_side, const float* background, int flags,
                                   const lasr_sr_options* options, void* hip_stream)
{
    if (flags == LASR_SR_DEFAULT_FLAGS) flags = default_flags();
    if (flags & ~(LASR_SR_RELAXED_MATH | LASR_SR_SEGMENTED | LASR_SR_PAIR_ONE_TEAM | LASR_SR_PAIR_TWO_TEAMS)) return LASR_E_BADARG;
    const int rc = check_nch(channels, func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type, double_side, T);
    if (rc) return rc;
    return forward_impl(faces, textures, faces_info, aggrs_info, soft_colors, workspace, workspace_bytes, N, F, T, IS, near,
                        far, eps, sigma_val, func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha,
                        texture_sample_type, double_side, hip_stream, channels, near_far_dev, flags, background, options);
}

extern "C" int lasr_sr_backward_ex(const float* faces, const float* textures, const float* soft_colors,
                                   const float* aggrs_info, float* grad_faces, float* grad_textures,
                                   const float* grad_soft_colors, void* workspace, size_t workspace_bytes, int N, int F,
                                   int T, int channels, int IS, float near, float far, const float* near_far_dev,
                                   float eps, float sigma_val, int func_id_dist, float dist_eps, float gamma_val,
                                   int func_id_rgb, int func_id_alpha, int texture_sample_type, int double_side, int flags,
                                   void* hip_stream)
{
    if (flags & ~(LASR_SR_RECORDS_VALID | LASR_SR_GRADS_OVERWRITE)) return LASR_E_BADARG;
    const int rc = check_nch(channels, func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type, double_side, T);
    if (rc) return rc;
    return backward_impl(faces, textures, soft_colors, nullptr, aggrs_info, grad_faces, grad_textures, grad_soft_colors,
                         workspace, workspace_bytes, N, F, T, IS, near, far, eps, sigma_val, func_id_dist, dist_eps,
                         gamma_val, func_id_rgb, func_id_alpha, texture_sample_type, double_side, hip_stream, channels,
                         near_far_dev, flags);
}

// Test hook: number of (a[i], b[i]) pairs for which div_by_recip(a, b, RN(1/b)) != a / b bitwise (added to *mismatches).
extern "C" int lasr_selftest_div(const float* a, const float* b, int* mismatches, int n, void* hip_stream)
{
    if (!a || !b || !mismatches || n < 0) return LASR_E_BADARG;
    if (n == 0) return LASR_OK;
    hipLaunchKernelGGL(selftest_div_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, a, b, mismatches, n);
    return launch_ok();
}

extern "C" int lasr_selftest_div3(const float* a, const float* b, int* mismatches, int n, void* hip_stream)
{
    if (!a || !b || !mismatches || n < 0) return LASR_E_BADARG;
    if (n == 0) return LASR_OK;
    hipLaunchKernelGGL(selftest_div3_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, a, b, mismatches, n);
    return launch_ok();
}

extern "C" int lasr_sr_peek_choice(const void* workspace, int N, int F, int* choice, void* hip_stream)
{
    if (!workspace || !choice || N < 0 || F < 0) return LASR_E_BADARG;
    const size_t nf = (size_t)N * (size_t)F;
    const char* p = (const char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    p += align_up(nf * REC * sizeof(float), 256) + align_up(nf * sizeof(short4), 256) +
         align_up((size_t)N * groups_of_host(F) * sizeof(short4), 256);
    if (hipMemcpyAsync(choice, p, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)hip_stream) != hipSuccess) return LASR_E_LAUNCH;
    return hipStreamSynchronize((hipStream_t)hip_stream) == hipSuccess ? LASR_OK : LASR_E_LAUNCH;
}
