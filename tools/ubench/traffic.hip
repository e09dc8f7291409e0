// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the ACCESS PATTERNS of the raster kernels (MI355X_MICROARCH.md, HBM
// section: only 16 B/lane coalesced streaming reads are calibrated there -- FETCH_SIZE reports half of their bytes; "other access
// widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel below moves a
// byte count known by construction, over buffers larger than the 256 MiB Infinity Cache; tools/prof/traffic_cal.sh runs this
// binary under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) and tools/traffic_cal_json.py turns the
// per-dispatch counters into factors  true bytes / (counter KiB x 1024)  for tools/traffic_json.py.
//
//   lasr_cal_read16     16 B/lane coalesced streaming read            (the guide's calibrated case: expect factor 2 on FETCH_SIZE)
//   lasr_cal_read4       4 B/lane coalesced read, one plane at a time  (the backward's pixel-plane gathers)
//   lasr_cal_read4_rect  4 B/lane reads of 11 x 12-pixel rects of 10 planes (the backward's real footprint: partial lines)
//   lasr_cal_read8       8 B/lane coalesced read                       (the forward's rect scans, short4)
//   lasr_cal_scalar      192-B records through the scalar cache        (the forward's record walk), each record read by ONE wave
//   lasr_cal_write4      4 B/lane coalesced streaming write
//   lasr_cal_write_tile  the forward's stores: one wave per 8x8 tile, lane = pixel, 6 planes -> 32-byte row segments
//   lasr_cal_write16     16 B/lane coalesced streaming write
//
//   hipcc -O3 --offload-arch=gfx950 -o scratch/bin/traffic tools/ubench/traffic.hip && ./scratch/bin/traffic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef const float __attribute__((address_space(4)))* cptr_t;

__global__ __launch_bounds__(256) void lasr_cal_read16(const float4* __restrict__ in, float* __restrict__ out, size_t n4)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void lasr_cal_read4(const float* __restrict__ in, float* __restrict__ out, size_t n)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += in[i];
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void lasr_cal_read8(const float2* __restrict__ in, float* __restrict__ out, size_t n2)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { const float2 v = in[i]; acc += v.x + v.y; }
    if (acc == 12345.678f) out[0] = acc;
}
// one wave per rect: 11 columns x 12 rows of a 256x256 plane, 10 planes of image `img`; rects tile the image without overlap
// (23 x 21 rects of 11 x 12 cover 253 x 252 pixels), so the unique bytes are known: rects x 132 px x 10 planes x 4 B, while the
// 64-B lines they touch are  rows x ceil-spans.  Mirrors sr_backward_kernel's stage 2 (lane = rect pixel, 10 planes per pixel).
__global__ __launch_bounds__(64) void lasr_cal_read4_rect(const float* __restrict__ in, float* __restrict__ out, int n_img)
{
    const int per = 23 * 21;
    const int img = blockIdx.x / per, r = blockIdx.x - img * per;
    if (img >= n_img) return;
    const int rx = (r % 23) * 11, ry = (r / 23) * 12;
    float acc = 0.f;
    for (int p = threadIdx.x; p < 132; p += 64) {
        const int x = rx + p % 11, y = ry + p / 11;
        for (int k = 0; k < 10; k++) acc += in[((size_t)img * 10 + k) * 65536 + y * 256 + x];
    }
    if (acc == 12345.678f) out[0] = acc;
}
// every wave walks `per_wave` records of 48 floats through scalar loads (wave-uniform index); each record is read once
__global__ __launch_bounds__(64) void lasr_cal_scalar(const float* __restrict__ recs, float* __restrict__ out, int per_wave)
{
    float acc = 0.f;
    const size_t first = (size_t)blockIdx.x * per_wave;
    for (int j = 0; j < per_wave; j++) {
        const cptr_t r = (cptr_t)(unsigned long long)(recs + (first + j) * 48);
#pragma unroll
        for (int k = 0; k < 48; k++) acc += r[k] * (float)(threadIdx.x + k);
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void lasr_cal_write4(float* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (float)i;
}
__global__ __launch_bounds__(256) void lasr_cal_write16(float4* __restrict__ out, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}
// sr_forward_kernel's stores: block = one wave = one 8x8 tile of a 256x256 image, 6 planes (4 colour + 2 aggregate), tiles
// issued in the forward's order (an image's tiles on one XCD, centre-out is irrelevant for the byte count): 32-byte row segments
__global__ __launch_bounds__(64) void lasr_cal_write_tile(float* __restrict__ out, int n_img)
{
    const int total = gridDim.x, per = total >> 3;
    const int b = (total & 7) == 0 ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;      // xcd_remap
    const int img = b >> 10, t = b & 1023;
    if (img >= n_img) return;
    const int px = (t & 31) * 8 + (threadIdx.x & 7), py = (t >> 5) * 8 + (threadIdx.x >> 3);
#pragma unroll
    for (int k = 0; k < 6; k++) out[((size_t)img * 6 + k) * 65536 + py * 256 + px] = (float)(k + px);
}

int main()
{
    const size_t GB = 1ull << 30;
    float *a, *b;
    CK(hipMalloc(&a, GB)); CK(hipMalloc(&b, GB));
    CK(hipMemset(a, 0, GB)); CK(hipMemset(b, 0, GB));
    CK(hipDeviceSynchronize());
    const int rep = 2;
    for (int it = 0; it < rep; it++) {
        hipLaunchKernelGGL(lasr_cal_read16, dim3(8192), dim3(256), 0, 0, (const float4*)a, b, GB / 16);
        hipLaunchKernelGGL(lasr_cal_read4, dim3(8192), dim3(256), 0, 0, a, b, GB / 4);
        hipLaunchKernelGGL(lasr_cal_read8, dim3(8192), dim3(256), 0, 0, (const float2*)a, b, GB / 8);
        const int n_img_r = (int)(GB / (10 * 65536 * 4));                     // 409 images of 10 planes
        hipLaunchKernelGGL(lasr_cal_read4_rect, dim3(n_img_r * 23 * 21), dim3(64), 0, 0, a, b, n_img_r);
        const int per_wave = 64, n_rec = (int)(GB / 192), waves = n_rec / per_wave;
        hipLaunchKernelGGL(lasr_cal_scalar, dim3(waves), dim3(64), 0, 0, a, b, per_wave);
        hipLaunchKernelGGL(lasr_cal_write4, dim3(8192), dim3(256), 0, 0, b, GB / 4);
        hipLaunchKernelGGL(lasr_cal_write16, dim3(8192), dim3(256), 0, 0, (float4*)b, GB / 16);
        const int n_img_w = 512;                                             // 512 x 6 planes x 256 KiB = 768 MiB
        hipLaunchKernelGGL(lasr_cal_write_tile, dim3(n_img_w * 1024), dim3(64), 0, 0, b, n_img_w);
        CK(hipDeviceSynchronize());
    }
    const int n_img_r = (int)(GB / (10 * 65536 * 4));
    const int n_rec = (int)(GB / 192);
    printf("{\"true_bytes\": {\"lasr_cal_read16\": %zu, \"lasr_cal_read4\": %zu, \"lasr_cal_read8\": %zu, \"lasr_cal_read4_rect\": %zu, "
           "\"lasr_cal_scalar\": %zu, \"lasr_cal_write4\": %zu, \"lasr_cal_write16\": %zu, \"lasr_cal_write_tile\": %zu}, "
           "\"read4_rect_line_bytes_64\": %zu}\n",
           GB, GB, GB, (size_t)n_img_r * 23 * 21 * 132 * 10 * 4, (size_t)(n_rec / 64) * 64 * 192, GB, GB, (size_t)512 * 6 * 65536 * 4,
           /* 64-B lines touched by an 11-px row segment starting at x: (x*4 .. x*4+43) */ (size_t)0);
    return 0;
}
