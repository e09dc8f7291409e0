// VALU issue-rate micro-benchmark on gfx950: wave64 v_fma_f32 vs v_pk_fma_f32 vs v_pk_mul_f32/v_pk_add_f32 vs v_mul_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_t __attribute__((ext_vector_type(2)));
#define REP 64
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0)
{
    float x[8]; float2_t y[8];
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 1e-3f + i; y[i] = float2_t{x[i], x[i] + 0.5f}; }
    const float a = a0, b = a0 * 0.5f;
    const float2_t av = {a, a * 1.1f}, bv = {b, b * 0.9f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
                else if (MODE == 1) y[i] = __builtin_elementwise_fma(y[i], av, bv);
                else if (MODE == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                else if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(av));
                else if (MODE == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(av));
                else if (MODE == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                else if (MODE == 6) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(*(double*)&y[i]) : "v"(*(const double*)&av));
                else if (MODE == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a));
                else if (MODE == 8) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                else if (MODE == 9) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> double run(const char* name, float* d)
{
    const int iters = 200, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * 4 * iters * REP * 8;            // wave-instructions
    const double per_simd = winst / 1024.0;
    printf("%-16s %8.3f ms  %6.2f cycles/wave-instr/SIMD @2.4GHz  (%.1f G wave-instr/s)\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd, winst / ms / 1e6);
    return ms;
}
int main()
{
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_mul_f32", d); run<3>("v_pk_mul_f32", d); run<4>("v_pk_add_f32", d);
    run<5>("v_add_f32", d); run<6>("v_fma_f64", d); run<7>("v_cndmask_b32", d); run<8>("v_exp_f32", d); run<9>("v_rcp_f32", d);
    return 0;
}
