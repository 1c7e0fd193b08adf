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
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                         \
        if (y0 + y1 + y2 + y3 + z0 + z1 + z2 + z3 == 12345.678) out[1000] = 1;                                       \
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
BENCH(k_ldexp, "v_ldexp_f64 %0, %8, %12\n v_ldexp_f64 %1, %9, %13\n v_ldexp_f64 %2, %10, %12\n v_ldexp_f64 %3, %11, %13")

template <class K>
static void run(const char *name, K k, int waves_per_cu) {
    unsigned long long *d, h[2048];
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0, sizeof(h));
    const int blocks = 256 * waves_per_cu;  // 64-lane workgroups: the dispatcher spreads them over the SIMDs
    hipLaunchKernelGGL(k, dim3(blocks > 1000 ? 1000 : blocks), dim3(64), 0, 0, d, 1.25, 2.5);
    hipLaunchKernelGGL(k, dim3(blocks > 1000 ? 1000 : blocks), dim3(64), 0, 0, d, 1.25, 2.5);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // s_memtime ticks at 100 MHz: convert with the shader clock (2.4 GHz) -> cycles per instruction
    const double ticks = (double)h[0];
    printf("%-16s %2d wave(s)/CU  %8.2f memtime ticks/instr  (~%.1f shader cycles at 2.4 GHz / 100 MHz)\n", name, waves_per_cu,
           ticks / (256.0 * REP * 4), ticks / (256.0 * REP * 4) * 24.0);
    hipFree(d);
}
int main() {
#define R(k) run(#k, k, 1), run(#k, k, 4), run(#k, k, 12)
    R(k_fma_f64); R(k_add_f64); R(k_mul_f64); R(k_min_f64); R(k_cvt_f64_u32); R(k_cvt_f64_f32); R(k_rcp_f64); R(k_rsq_f64);
    R(k_sqrt_f64); R(k_cmp_f64); R(k_and_b32); R(k_cndmask); R(k_fma_f32); R(k_div_scale); R(k_div_fixup); R(k_div_fmas); R(k_ldexp);
    return 0;
}
