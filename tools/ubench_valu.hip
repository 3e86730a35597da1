// ubench_valu.hip -- VALU issue-rate microbenchmark for gfx950 (design input for the SSAO kernels).
// Each kernel runs ITER x 16 independent copies of one instruction per lane; 8 waves per SIMD.
// Reports wave-instructions per SIMD per microsecond and the implied cycles/instruction at the
// clock measured with s_memtime-free wall time (assumes the v_fma_f32 row as the 100% reference).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096
typedef float f2 __attribute__((ext_vector_type(2)));

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL_F32(NAME, ASM)                                                        \
    __global__ __launch_bounds__(256) void NAME(float *out, float b, float c)        \
    {                                                                                \
        float a[16];                                                                 \
        for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;                \
        for (int it = 0; it < ITER; ++it) {                                          \
            _Pragma("unroll") for (int i = 0; i < 16; ++i)                           \
                asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c));                     \
        }                                                                            \
        float s = 0; for (int i = 0; i < 16; ++i) s += a[i];                         \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                     \
    }

#define KERNEL_F64(NAME, ASM)                                                        \
    __global__ __launch_bounds__(256) void NAME(float *out, float b, float c)        \
    {                                                                                \
        f2 a[16]; f2 bb = {b, b}, cc = {c, c};                                       \
        for (int i = 0; i < 16; ++i) a[i] = f2{threadIdx.x * 0.001f + i, 1.0f};      \
        for (int it = 0; it < ITER; ++it) {                                          \
            _Pragma("unroll") for (int i = 0; i < 16; ++i)                           \
                asm volatile(ASM : "+v"(a[i]) : "v"(bb), "v"(cc));                   \
        }                                                                            \
        float s = 0; for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;              \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                     \
    }

KERNEL_F32(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL_F32(k_fma_clamp, "v_fma_f32 %0, %0, %1, %2 clamp")
KERNEL_F32(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL_F32(k_mul_clamp, "v_mul_f32_e64 %0, %0, %1 clamp")
KERNEL_F32(k_add, "v_add_f32 %0, %0, %1")
KERNEL_F32(k_max, "v_max_f32 %0, %0, %1")
KERNEL_F32(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL_F32(k_rcp, "v_rcp_f32 %0, %0")
KERNEL_F32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_F32(k_cvt_rtz, "v_cvt_pkrtz_f16_f32 %0, %0, %1")
KERNEL_F32(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %0")
KERNEL_F32(k_cmp, "v_cmp_gt_f32 vcc, %0, %1")
KERNEL_F32(k_div_scale, "v_div_scale_f32 %0, vcc, %0, %1, %2")
KERNEL_F32(k_div_fixup, "v_div_fixup_f32 %0, %0, %1, %2")
KERNEL_F32(k_cndmask_e64_vcc, "v_cndmask_b32_e64 %0, %0, %1, vcc")
KERNEL_F32(k_cndmask_e64_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL_F32(k_cndmask_lit, "v_cndmask_b32_e32 %0, 0x3f000000, %0, vcc")
KERNEL_F32(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
KERNEL_F32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL_F32(k_sub, "v_sub_f32 %0, %0, %1")
KERNEL_F32(k_add_abs, "v_add_f32_e64 %0, |%0|, %1")
KERNEL_F32(k_cvt_u32, "v_cvt_u32_f32 %0, %0")
KERNEL_F32(k_cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
KERNEL_F32(k_lshl_or, "v_lshl_or_b32 %0, %0, %1, %2")
KERNEL_F32(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL_F32(k_mov, "v_mov_b32 %0, %1")
KERNEL_F64(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL_F64(k_pk_fma_clamp, "v_pk_fma_f32 %0, %0, %1, %2 clamp")
KERNEL_F64(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
KERNEL_F64(k_pk_mul_clamp, "v_pk_mul_f32 %0, %0, %1 clamp")
KERNEL_F64(k_pk_add, "v_pk_add_f32 %0, %0, %1")

typedef void (*kfn)(float *, float, float);

int main()
{
    float *out;
    const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU -> 8 waves per SIMD
    hipMalloc(&out, blocks * 256 * sizeof(float));
    struct { const char *name; kfn fn; } ks[] = {
        {"v_fma_f32", k_fma}, {"v_fma_f32 clamp", k_fma_clamp}, {"v_mul_f32", k_mul}, {"v_mul_f32 clamp", k_mul_clamp},
        {"v_add_f32", k_add}, {"v_max_f32", k_max}, {"v_med3_f32", k_med3}, {"v_rcp_f32", k_rcp},
        {"v_cndmask_b32", k_cndmask}, {"v_cvt_pkrtz_f16_f32", k_cvt_rtz}, {"v_cvt_f32_f16", k_cvt_f32_f16},
        {"v_cmp_gt_f32", k_cmp}, {"v_div_scale_f32", k_div_scale}, {"v_div_fixup_f32", k_div_fixup},
        {"v_cndmask_b32_e64 vcc", k_cndmask_e64_vcc}, {"v_cndmask_b32_e64 sgpr", k_cndmask_e64_sgpr},
        {"v_cndmask_b32 literal", k_cndmask_lit}, {"v_bfi_b32", k_bfi}, {"v_and_or_b32", k_and_or}, {"v_sub_f32", k_sub},
        {"v_add_f32 |abs|", k_add_abs}, {"v_cvt_u32_f32", k_cvt_u32}, {"v_cvt_f32_ubyte0", k_cvt_f32_ubyte0},
        {"v_lshl_or_b32", k_lshl_or}, {"v_perm_b32", k_perm}, {"v_mov_b32", k_mov},
        {"v_pk_fma_f32", k_pk_fma}, {"v_pk_fma_f32 clamp", k_pk_fma_clamp}, {"v_pk_mul_f32", k_pk_mul},
        {"v_pk_mul_f32 clamp", k_pk_mul_clamp}, {"v_pk_add_f32", k_pk_add}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double ref = 0;
    for (auto &k : ks) {
        k.fn<<<blocks, 256>>>(out, 0.999f, 0.001f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k.fn<<<blocks, 256>>>(out, 0.999f, 0.001f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // wave-instructions per SIMD: blocks*4 waves / 1024 SIMDs * ITER*16
        const double per_simd = double(blocks) * 4 / 1024.0 * ITER * 16;
        const double ns_per_instr = ms * 1e6 / per_simd;
        if (ref == 0) ref = ns_per_instr;
        std::printf("%-22s %8.3f ms  %.3f ns/wave-instr/SIMD  (%.2fx v_fma_f32; %.2f cycles @2.4GHz)\n",
                    k.name, ms, ns_per_instr, ns_per_instr / ref, ns_per_instr * 2.4);
    }
    return 0;
}
