// Issue cost of the f64 vector instructions the renderer leans on (one wave per SIMD, independent streams).
// Build: hipcc -O2 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define STR2(x) #x
#define STR(x) STR2(x)
#define BENCH(NAME, ASM)                                                                         \
    __global__ void NAME(unsigned long long *out, double a, double b) {                          \
        double x0 = a + threadIdx.x, x1 = b + threadIdx.x, x2 = a * 3 + threadIdx.x, x3 = b * 5 + threadIdx.x; \
        unsigned u0 = threadIdx.x + 7, u1 = threadIdx.x * 3 + 1;                                 \
        double y0 = 0, y1 = 0, y2 = 0, y3 = 0;                                                   \
        unsigned z0 = 0, z1 = 0, z2 = 0, z3 = 0;                                                 \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                    \
        for (int i = 0; i < 256; i++) {                                                          \
            asm volatile(".rept " STR(REP) "\n" ASM "\n.endr"                                    \
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3) \
                         : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(u0), "v"(u1) : "vcc");                \
        }                                                                                        \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                    \
        if (threadIdx.x == 0 && blockIdx.x < 1000) out[blockIdx.x] = t1 - t0;                                         \
        if (y0 + y1 + y2 + y3 + z0 + z1 + z2 + z3 == 12345.678) out[1001] = 1;                                       \
    }
// four independent destinations per repetition
BENCH(k_fma_f64, "v_fma_f64 %0, %8, %9, %10\n v_fma_f64 %1, %9, %10, %11\n v_fma_f64 %2, %10, %11, %8\n v_fma_f64 %3, %11, %8, %9")
BENCH(k_add_f64, "v_add_f64 %0, %8, %9\n v_add_f64 %1, %9, %10\n v_add_f64 %2, %10, %11\n v_add_f64 %3, %11, %8")
BENCH(k_mul_f64, "v_mul_f64 %0, %8, %9\n v_mul_f64 %1, %9, %10\n v_mul_f64 %2, %10, %11\n v_mul_f64 %3, %11, %8")
BENCH(k_min_f64, "v_min_f64 %0, %8, %9\n v_min_f64 %1, %9, %10\n v_min_f64 %2, %10, %11\n v_min_f64 %3, %11, %8")
BENCH(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %12\n v_cvt_f64_u32 %1, %13\n v_cvt_f64_u32 %2, %12\n v_cvt_f64_u32 %3, %13")
BENCH(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %12\n v_cvt_f64_f32 %1, %13\n v_cvt_f64_f32 %2, %12\n v_cvt_f64_f32 %3, %13")
BENCH(k_rcp_f64, "v_rcp_f64 %0, %8\n v_rcp_f64 %1, %9\n v_rcp_f64 %2, %10\n v_rcp_f64 %3, %11")
BENCH(k_rsq_f64, "v_rsq_f64 %0, %8\n v_rsq_f64 %1, %9\n v_rsq_f64 %2, %10\n v_rsq_f64 %3, %11")
BENCH(k_sqrt_f64, "v_sqrt_f64 %0, %8\n v_sqrt_f64 %1, %9\n v_sqrt_f64 %2, %10\n v_sqrt_f64 %3, %11")
BENCH(k_cmp_f64, "v_cmp_le_f64 vcc, %8, %9\n v_cmp_le_f64 vcc, %9, %10\n v_cmp_le_f64 vcc, %10, %11\n v_cmp_le_f64 vcc, %11, %8")
BENCH(k_and_b32, "v_and_b32 %4, %12, %13\n v_and_b32 %5, %13, %12\n v_and_b32 %6, %12, %13\n v_and_b32 %7, %13, %12")
BENCH(k_cndmask, "v_cndmask_b32 %4, %12, %13, vcc\n v_cndmask_b32 %5, %13, %12, vcc\n v_cndmask_b32 %6, %12, %13, vcc\n v_cndmask_b32 %7, %13, %12, vcc")
BENCH(k_fma_f32, "v_fma_f32 %4, %12, %13, %12\n v_fma_f32 %5, %13, %12, %13\n v_fma_f32 %6, %12, %13, %13\n v_fma_f32 %7, %13, %12, %12")
BENCH(k_div_scale, "v_div_scale_f64 %0, vcc, %8, %9, %8\n v_div_scale_f64 %1, vcc, %9, %10, %9\n v_div_scale_f64 %2, vcc, %10, %11, %10\n v_div_scale_f64 %3, vcc, %11, %8, %11")
BENCH(k_div_fixup, "v_div_fixup_f64 %0, %8, %9, %10\n v_div_fixup_f64 %1, %9, %10, %11\n v_div_fixup_f64 %2, %10, %11, %8\n v_div_fixup_f64 %3, %11, %8, %9")
BENCH(k_div_fmas, "v_div_fmas_f64 %0, %8, %9, %10\n v_div_fmas_f64 %1, %9, %10, %11\n v_div_fmas_f64 %2, %10, %11, %8\n v_div_fmas_f64 %3, %11, %8, %9")
BENCH(k_pk_fma_f32, "v_pk_fma_f32 %0, %8, %9, %10\n v_pk_fma_f32 %1, %9, %10, %11\n v_pk_fma_f32 %2, %10, %11, %8\n v_pk_fma_f32 %3, %11, %8, %9")
BENCH(k_max3_f32, "v_max3_f32 %4, %12, %13, %12\n v_max3_f32 %5, %13, %12, %13\n v_max3_f32 %6, %12, %13, %13\n v_max3_f32 %7, %13, %12, %12")
BENCH(k_min_f32, "v_min_f32 %4, %12, %13\n v_min_f32 %5, %13, %12\n v_min_f32 %6, %12, %13\n v_min_f32 %7, %13, %12")
BENCH(k_cvt_f32_u32, "v_cvt_f32_u32 %4, %12\n v_cvt_f32_u32 %5, %13\n v_cvt_f32_u32 %6, %12\n v_cvt_f32_u32 %7, %13")
BENCH(k_cvt_f32_u32_sdwa, "v_cvt_f32_u32_sdwa %4, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %5, %13 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %6, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %7, %13 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
BENCH(k_perm_b32, "v_perm_b32 %4, %12, %13, %12\n v_perm_b32 %5, %13, %12, %13\n v_perm_b32 %6, %12, %13, %13\n v_perm_b32 %7, %13, %12, %12")
BENCH(k_cvt_f32_f64, "v_cvt_f32_f64 %4, %8\n v_cvt_f32_f64 %5, %9\n v_cvt_f32_f64 %6, %10\n v_cvt_f32_f64 %7, %11")
BENCH(k_cmp_f32, "v_cmp_le_f32 vcc, %12, %13\n v_cmp_le_f32 vcc, %13, %12\n v_cmp_le_f32 vcc, %12, %13\n v_cmp_le_f32 vcc, %13, %12")
BENCH(k_ldexp, "v_ldexp_f64 %0, %8, %12\n v_ldexp_f64 %1, %9, %13\n v_ldexp_f64 %2, %10, %12\n v_ldexp_f64 %3, %11, %13")

template <class K>
static void run(const char *name, K k, int waves_per_cu) {
    // r2: one 64-lane workgroup per wave, `waves_per_cu` x 256 of them so that every SIMD holds waves_per_cu / 4 waves
    // (r1 capped the grid at 1000 blocks, i.e. never more than one wave per SIMD).  s_memtime counts shader cycles; the
    // per-SIMD rate is (cycles one wave needs per instruction) / (waves sharing the SIMD).
    static unsigned long long h[8192];
    unsigned long long *d;
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0, sizeof(h));
    const int blocks = 256 * waves_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 1.25, 2.5);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 1.25, 2.5);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < blocks && i < 1000; i++) sum += (double)h[i];
    const double per_wave = sum / (blocks < 1000 ? blocks : 1000) / (256.0 * REP * 4);
    const double per_simd = per_wave / (waves_per_cu / 4.0 < 1 ? 1 : waves_per_cu / 4.0);
    printf("%-20s %2d waves/CU  %6.2f cycles/instr seen by a wave  %6.2f cycles/instr per SIMD   kernel %.3f ms\n", name,
           waves_per_cu, per_wave, per_simd, ms);
    hipFree(d);
}
int main() {
#define R(k) run(#k, k, 4), run(#k, k, 8), run(#k, k, 12)
    R(k_fma_f64); R(k_min_f64); R(k_cvt_f64_u32); R(k_cmp_f64); R(k_and_b32); R(k_fma_f32); R(k_pk_fma_f32); R(k_max3_f32);
    R(k_min_f32); R(k_cvt_f32_u32); R(k_cvt_f32_u32_sdwa); R(k_perm_b32); R(k_cvt_f32_f64); R(k_cmp_f32); R(k_rcp_f64);
    R(k_sqrt_f64); R(k_div_fmas);
    return 0;
}
