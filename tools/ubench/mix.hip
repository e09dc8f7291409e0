// issue cost of the instruction kinds the raster kernels use (gfx950, wave64), 8 independent chains x 8 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 32
#define BODY(STR) for (int it = 0; it < iters; it++) { _Pragma("unroll") for (int r = 0; r < REP; r++) { _Pragma("unroll") for (int i = 0; i < 8; i++) { asm volatile(STR : "+v"(x[i]) : "v"(a), "v"(b), "s"(sa)); } } }
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float sa)
{
    float x[8];
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3f + i + 1.f;
    const float a = a0, b = a0 * 0.5f;
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");
    if (MODE == 0) BODY("v_fma_f32 %0, %0, %1, %2")
    if (MODE == 1) BODY("v_cndmask_b32 %0, %0, %1, vcc")
    if (MODE == 2) BODY("v_cmp_lt_f32 vcc, %0, %1")
    if (MODE == 3) BODY("v_max_f32 %0, %0, %1")
    if (MODE == 4) BODY("v_mov_b32 %0, %1")
    if (MODE == 5) BODY("v_cvt_f64_f32 v[200:201], %0")
    if (MODE == 6) BODY("v_and_b32 %0, %0, %1")
    if (MODE == 7) BODY("v_mul_f32 %0, %0, %3")
    if (MODE == 8) BODY("v_cmp_lt_f32 s[20:21], %0, %1")
    if (MODE == 9) BODY("v_cndmask_b32 %0, %0, %1, s[20:21]")
    if (MODE == 10) BODY("v_ldexp_f32 %0, %0, %1")
    if (MODE == 11) BODY("v_rndne_f32 %0, %0")
    if (MODE == 12) BODY("v_div_scale_f32 %0, vcc, %0, %1, %0")
    if (MODE == 13) BODY("v_div_fmas_f32 %0, %0, %1, %2")
    if (MODE == 14) BODY("v_div_fixup_f32 %0, %0, %1, %2")
    if (MODE == 15) BODY("v_rcp_f64 v[200:201], v[200:201]")
    if (MODE == 16) BODY("v_mul_f64 v[200:201], v[200:201], v[202:203]")
    if (MODE == 17) BODY("v_cvt_f32_f64 %0, v[200:201]")
    if (MODE == 18) BODY("v_readlane_b32 s22, %0, 3")
    if (MODE == 19) BODY("v_mbcnt_lo_u32_b32 %0, -1, %0")
    if (MODE == 20) BODY("v_add_u32 %0, %0, %1")
    if (MODE == 21) BODY("v_mad_u32_u24 %0, %0, %1, %2")
    if (MODE == 22) BODY("v_lshlrev_b32 %0, 2, %0")
    if (MODE == 23) BODY("v_sqrt_f32 %0, %0")
    if (MODE == 24) BODY("v_log_f32 %0, %0")
    if (MODE == 25) BODY("v_cvt_i32_f32 %0, %0")
    if (MODE == 26) BODY("v_sub_f32 %0, %3, %0")
    if (MODE == 27) BODY("s_mov_b32 s22, s23")
    if (MODE == 28) BODY("s_and_b64 s[20:21], s[20:21], exec")
    if (MODE == 29) BODY("v_med3_f32 %0, %0, %1, %2")
    if (MODE == 30) BODY("v_min3_f32 %0, %0, %1, %2")
    if (MODE == 31) BODY("v_max3_f32 %0, %0, %1, %2")
    if (MODE == 32) BODY("v_pk_sub_u16 %0, %0, %1")
    if (MODE == 33) BODY("v_pk_min_u16 %0, %0, %1")
    if (MODE == 34) BODY("v_cmp_eq_u32 vcc, %0, %1")
    if (MODE == 35) BODY("v_lshl_add_u32 %0, %0, 1, %1")
    if (MODE == 36) BODY("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
    if (MODE == 37) BODY("v_cmp_lt_f32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %1, s[20:21]")
    if (MODE == 38) BODY("v_sub_u32 %0, %0, %1")
    if (MODE == 39) BODY("v_min_u32 %0, %0, %1")
    if (MODE == 40) BODY("v_mad_i32_i24 %0, %0, %1, %2")
    if (MODE == 41) BODY("v_add_f64 v[200:201], v[200:201], v[202:203]")
    if (MODE == 42) BODY("v_cvt_f32_i32 %0, %0")
    if (MODE == 43) BODY("v_add_f32 %0, |%0|, -%1")
    if (MODE == 44) BODY("v_mul_f32 %0, %0, %1 clamp")
    if (MODE == 45) BODY("v_bfe_u32 %0, %0, 3, 5")
    if (MODE == 46) BODY("v_cmp_class_f32 vcc, %0, %1")
    if (MODE == 47) BODY("v_div_scale_f64 v[200:201], vcc, v[200:201], v[202:203], v[200:201]")
    if (MODE == 48) BODY("v_div_fixup_f64 v[200:201], v[200:201], v[202:203], v[204:205]")
    if (MODE == 49) BODY("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
    if (MODE == 50) BODY("v_fma_f32 %0, %0, %1, %2\n s_nop 0")
    if (MODE == 51) BODY("v_fma_f32 %0, %0, %1, %2\n s_and_b64 s[20:21], s[20:21], exec")
    if (MODE == 52) BODY("v_fma_f32 %0, %0, %1, %2\n s_and_b64 s[20:21], s[20:21], exec\n s_or_b64 s[22:23], s[22:23], exec")
    float s = 0;
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* d, double base)
{
    const int iters = 200, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-36s %8.3f ms   %5.2f x v_fma_f32\n", name, ms, ms / base);
}
template <int MODE> double base_ms(float* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, d, 200, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, d, 200, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    float* d; hipMalloc(&d, 2048 * 256 * 4);
    const double b = base_ms<0>(d);
    run<0>("v_fma_f32", d, b); run<7>("v_mul_f32 (sgpr operand)", d, b); run<26>("v_sub_f32 (sgpr operand)", d, b);
    run<1>("v_cndmask_b32 vcc", d, b); run<9>("v_cndmask_b32 sgpr-pair", d, b); run<2>("v_cmp_lt_f32 -> vcc", d, b); run<8>("v_cmp_lt_f32 -> sgpr-pair", d, b);
    run<3>("v_max_f32", d, b); run<4>("v_mov_b32", d, b); run<6>("v_and_b32", d, b); run<20>("v_add_u32", d, b); run<21>("v_mad_u32_u24", d, b); run<22>("v_lshlrev_b32", d, b);
    run<10>("v_ldexp_f32", d, b); run<11>("v_rndne_f32", d, b); run<25>("v_cvt_i32_f32", d, b); run<23>("v_sqrt_f32", d, b); run<24>("v_log_f32", d, b);
    run<12>("v_div_scale_f32", d, b); run<13>("v_div_fmas_f32", d, b); run<14>("v_div_fixup_f32", d, b);
    run<5>("v_cvt_f64_f32", d, b); run<17>("v_cvt_f32_f64", d, b); run<16>("v_mul_f64", d, b); run<15>("v_rcp_f64", d, b);
    run<18>("v_readlane_b32", d, b); run<19>("v_mbcnt_lo", d, b); run<27>("s_mov_b32", d, b); run<28>("s_and_b64", d, b);
    run<29>("v_med3_f32", d, b); run<30>("v_min3_f32", d, b); run<31>("v_max3_f32", d, b); run<32>("v_pk_sub_u16", d, b); run<33>("v_pk_min_u16", d, b);
    run<34>("v_cmp_eq_u32 vcc", d, b); run<35>("v_lshl_add_u32", d, b); run<36>("v_cmp+v_cndmask via vcc (pair)", d, b); run<37>("v_cmp+v_cndmask via sgpr pair (pair)", d, b);
    run<38>("v_sub_u32", d, b); run<39>("v_min_u32", d, b); run<40>("v_mad_i32_i24", d, b); run<41>("v_add_f64", d, b); run<42>("v_cvt_f32_i32", d, b);
    run<43>("v_add_f32 with abs/neg modifiers", d, b); run<44>("v_mul_f32 clamp", d, b); run<45>("v_bfe_u32", d, b); run<46>("v_cmp_class_f32", d, b);
    run<47>("v_div_scale_f64", d, b); run<48>("v_div_fixup_f64", d, b); run<49>("v_mov_b32_dpp", d, b);
    run<50>("v_fma + s_nop (pair)", d, b); run<51>("v_fma + s_and_b64 (pair)", d, b); run<52>("v_fma + 2 salu (triple)", d, b);
    return 0;
}
