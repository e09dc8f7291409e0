// does an SGPR source operand slow a VALU instruction down?  (gfx950, wave64; 8 independent chains, 8 waves/SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 32
#define BODY(STR) for (int it = 0; it < iters; it++) { _Pragma("unroll") for (int r = 0; r < REP; r++) { _Pragma("unroll") for (int i = 0; i < 8; i++) { asm volatile(STR : "+v"(x[i]) : "v"(a), "v"(b), "s"(sa), "s"(sb)); } } }
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float sa, float sb)
{
    float x[8];
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3f + i + 1.f;
    const float a = a0, b = a0 * 0.5f;
    if (MODE == 0) BODY("v_fma_f32 %0, %0, %1, %2")
    if (MODE == 1) BODY("v_fma_f32 %0, %0, %3, %2")
    if (MODE == 2) BODY("v_fma_f32 %0, %0, %3, %3")
    if (MODE == 3) BODY("v_mul_f32_e32 %0, %1, %0")
    if (MODE == 4) BODY("v_mul_f32_e32 %0, %3, %0")
    if (MODE == 5) BODY("v_mul_f32_e64 %0, %0, %3")
    if (MODE == 6) BODY("v_mul_f32_e64 %0, %0, %1")
    if (MODE == 7) BODY("v_add_f32_e32 %0, %1, %0")
    if (MODE == 8) BODY("v_add_f32_e32 %0, %3, %0")
    if (MODE == 9) BODY("v_sub_f32_e32 %0, %1, %0")
    if (MODE == 10) BODY("v_max_f32_e32 %0, %1, %0")
    if (MODE == 11) BODY("v_fmac_f32_e32 %0, %1, %2")
    if (MODE == 12) BODY("v_fmac_f32_e32 %0, %3, %2")
    if (MODE == 13) BODY("v_mul_f32_e32 %0, 0x40490fdb, %0")
    if (MODE == 14) BODY("v_mul_f32_e32 %0, 2.0, %0")
    if (MODE == 15) BODY("v_cmp_lt_f32_e32 vcc, %1, %0")
    if (MODE == 16) BODY("v_min_f32_e32 %0, %1, %0")
    if (MODE == 17) BODY("v_mov_b32_e32 %0, %3")
    float s = 0;
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* d)
{
    const int iters = 200, blocks = 2048;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f, 0.25f);
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double per_simd = (double)blocks * 4 * iters * REP * 8 / 1024.0;
    printf("%-44s %7.3f ms  %5.2f ns-cycles@2.4GHz/instr/SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
}
int main()
{
    float* d; hipMalloc(&d, 2048 * 256 * 4);
    run<0>("v_fma_f32 v,v,v", d); run<0>("v_fma_f32 v,v,v (again)", d); run<1>("v_fma_f32 v,s,v", d); run<2>("v_fma_f32 v,s,s", d);
    run<3>("v_mul_f32_e32 v,v", d); run<4>("v_mul_f32_e32 s,v", d); run<6>("v_mul_f32_e64 v,v", d); run<5>("v_mul_f32_e64 v,s", d);
    run<7>("v_add_f32_e32 v,v", d); run<8>("v_add_f32_e32 s,v", d); run<9>("v_sub_f32_e32 v,v", d); run<10>("v_max_f32_e32 v,v", d); run<16>("v_min_f32_e32 v,v", d);
    run<11>("v_fmac_f32_e32 v,v", d); run<12>("v_fmac_f32_e32 s,v", d); run<13>("v_mul_f32_e32 literal,v", d); run<14>("v_mul_f32_e32 inline 2.0,v", d);
    run<15>("v_cmp_lt_f32_e32 vcc", d); run<17>("v_mov_b32 v,s", d);
    return 0;
}
