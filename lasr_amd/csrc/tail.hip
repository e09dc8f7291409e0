// tail.hip -- what follows backward() in the optimisation loop (nnutils/train_utils.py:282-296 of the reference), as three
// multi-tensor kernels over ALL parameter tensors instead of ~36 launches and one host sync:
//   1. sum of squares of every gradient chunk                                   (tail_sumsq_kernel)
//   2. gradient norms -> clip coefficients (mean shape: norm 1; encoder + code predictor jointly: norm 10,
//      torch.nn.utils.clip_grad_norm_ semantics) and the "every gradient is finite" flag      (tail_finalize_kernel)
//   3. clip / zero the gradients in place and apply AdamW (torch.optim.AdamW arithmetic, decoupled weight decay,
//      bias-corrected) to parameter, exp_avg, exp_avg_sq and the step counter                 (tail_adamw_kernel)
// The tensors are described by a device table the host builds once (their addresses are stable under graph replay).
#include <hip/hip_runtime.h>

#include "../../include/lasr_ops.h"
#include "ops_common.h"

namespace lasr {

constexpr int TAIL_CHUNK = 4096;            // elements per workgroup (256 threads x 16)

struct TailRow {                            // one parameter tensor: 8 x 64 bit, the layout of the host-built table
    float* p; float* g; float* m; float* v; float* step;
    long long numel, group, clip;           // clip: 0 none, 1 mean shape (max norm 1), 2 camera networks (joint max norm 10)
};

__global__ __launch_bounds__(256) void tail_sumsq_kernel(const TailRow* __restrict__ table, const int2* __restrict__ chunks,
                                                         double* __restrict__ partials)
{
    // squares and sums in double: a large but finite gradient (|g| > 1.8e19 squares to +Inf in fp32) must not look like the
    // NaN the guard is there for (nnutils/train_utils.py:289 tests isnan); the kernel streams its input once, fp64 adds are free
    __shared__ double red[4];
    const int2 ck = chunks[blockIdx.x];
    const TailRow row = table[ck.x];
    const float* __restrict__ g = row.g + ck.y;
    const int n = (int)min((long long)TAIL_CHUNK, row.numel - ck.y);
    double s = 0.;
    for (int i = threadIdx.x; i < n; i += 256) { const double x = (double)g[i]; s += x * x; }
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ctl[0] = clip coefficient of the mean shape, ctl[1] = of the camera networks, ctl[2] = 1 if every gradient is finite,
// ctl[3] = mean-shape gradient norm AFTER clipping (what the reference logs), ctl[4] = camera-network norm before clipping
__global__ __launch_bounds__(256) void tail_finalize_kernel(const TailRow* __restrict__ table, const int2* __restrict__ chunks,
                                                            int n_chunks, const double* __restrict__ partials, float max_norm_shape,
                                                            float max_norm_cam, float* __restrict__ ctl)
{
    __shared__ double acc[3][256];
    double all = 0., shape = 0., cam = 0.;
    for (int c = threadIdx.x; c < n_chunks; c += 256) {
        const double s = partials[c];
        const long long cls = table[chunks[c].x].clip;
        all += s;
        if (cls == 1) shape += s; else if (cls == 2) cam += s;
    }
    acc[0][threadIdx.x] = all; acc[1][threadIdx.x] = shape; acc[2][threadIdx.x] = cam;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) {
#pragma unroll
            for (int k = 0; k < 3; k++) acc[k][threadIdx.x] += acc[k][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float n_shape = (float)sqrt(acc[1][0]), n_cam = (float)sqrt(acc[2][0]);
        const float c_shape = fminf(max_norm_shape / (n_shape + 1e-6f), 1.f);      // clip_grad_norm_: coef clamped to 1
        const float c_cam = fminf(max_norm_cam / (n_cam + 1e-6f), 1.f);
        const bool finite = isfinite(acc[0][0]);                                // NaN or Inf anywhere poisons the sum
        ctl[0] = c_shape; ctl[1] = c_cam; ctl[2] = finite ? 1.f : 0.f; ctl[3] = n_shape * c_shape; ctl[4] = n_cam;
        ctl[5] = (float)sqrt(acc[0][0]); ctl[6] += finite ? 0.f : 1.f; ctl[7] = 0.f;     // ctl[6]: steps skipped so far (caller zeroes it)
    }
}

struct TailGroups {                         // per optimizer parameter group, by value
    float lr[LASR_TAIL_MAX_GROUPS], w1[LASR_TAIL_MAX_GROUPS] /* 1 - beta1 */, beta2[LASR_TAIL_MAX_GROUPS],
          eps[LASR_TAIL_MAX_GROUPS], wd[LASR_TAIL_MAX_GROUPS], step_size[LASR_TAIL_MAX_GROUPS] /* lr / (1 - beta1^t) */,
          bc2_sqrt[LASR_TAIL_MAX_GROUPS] /* sqrt(1 - beta2^t) */;
};

__global__ __launch_bounds__(256) void tail_adamw_kernel(const TailRow* __restrict__ table, const int2* __restrict__ chunks,
                                                         const float* __restrict__ ctl, TailGroups G)
{
    const int2 ck = chunks[blockIdx.x];
    const TailRow row = table[ck.x];
    const int n = (int)min((long long)TAIL_CHUNK, row.numel - ck.y);
    const bool zero = ctl[2] == 0.f;                     // a NaN / Inf gradient somewhere: every gradient becomes 0 (the
    const float coef = row.clip == 1 ? ctl[0] : row.clip == 2 ? ctl[1] : 1.f;   // reference's zero_grad()), AdamW still steps
    const int gi = (int)row.group;
    const float lr = G.lr[gi], w1 = G.w1[gi], beta2 = G.beta2[gi], eps = G.eps[gi], wd = G.wd[gi];
    const float step_size = G.step_size[gi], bc2_sqrt = G.bc2_sqrt[gi];
    float* __restrict__ p = row.p + ck.y;
    float* __restrict__ g = row.g + ck.y;
    float* __restrict__ m = row.m + ck.y;
    float* __restrict__ v = row.v + ck.y;
    const bool rewrite = zero || coef != 1.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        float grad = zero ? 0.f : g[i] * coef;
        if (rewrite) g[i] = grad;                        // .grad holds the clipped / zeroed values afterwards, as in the reference
        float param = p[i];
        param -= lr * wd * param;                        // decoupled weight decay
        float ea = m[i], es = v[i];
        ea = ea + w1 * (grad - ea);                      // lerp(exp_avg, grad, 1 - beta1)
        es = beta2 * es + (1.f - beta2) * grad * grad;
        const float denom = sqrtf(es) / bc2_sqrt + eps;
        param -= step_size * ea / denom;
        p[i] = param; m[i] = ea; v[i] = es;
    }
    if (ck.y == 0 && threadIdx.x == 0 && row.step) *row.step += 1.f;
}

}  // namespace lasr

using namespace lasr;

extern "C" int lasr_tail_chunk_elems(void) { return TAIL_CHUNK; }

extern "C" int lasr_tail_step(const void* table, const int* chunks, int n_chunks, double* partials, float* ctl, float max_norm_shape,
                              float max_norm_cam, const float* lr, const float* beta1, const float* beta2, const float* eps,
                              const float* weight_decay, const double* bias_correction1, const double* bias_correction2,
                              int n_groups, void* hip_stream)
{
    if (n_chunks < 0 || n_groups < 1 || n_groups > LASR_TAIL_MAX_GROUPS) return LASR_E_BADARG;
    if (n_chunks == 0) return LASR_OK;
    if (!table || !chunks || !partials || !ctl || !lr || !beta1 || !beta2 || !eps || !weight_decay || !bias_correction1 ||
        !bias_correction2)
        return LASR_E_BADARG;
    TailGroups G;
    for (int k = 0; k < n_groups; k++) {
        G.lr[k] = lr[k]; G.w1[k] = (float)(1.0 - (double)beta1[k]); G.beta2[k] = beta2[k]; G.eps[k] = eps[k];
        G.wd[k] = weight_decay[k];
        G.step_size[k] = (float)((double)lr[k] / bias_correction1[k]);
        G.bc2_sqrt[k] = (float)sqrt(bias_correction2[k]);
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const TailRow* T = (const TailRow*)table;
    const int2* C = (const int2*)chunks;
    LASR_LAUNCH(K_TAIL, tail_sumsq_kernel, dim3(n_chunks), dim3(256), 0, T, C, partials);
    int rc = launch_ok();
    if (rc) return rc;
    LASR_LAUNCH(K_TAIL, tail_finalize_kernel, dim3(1), dim3(256), 0, T, C, n_chunks, partials, max_norm_shape, max_norm_cam, ctl);
    if ((rc = launch_ok())) return rc;
    LASR_LAUNCH(K_TAIL, tail_adamw_kernel, dim3(n_chunks), dim3(256), 0, T, C, ctl, G);
    return launch_ok();
}
