// ubench_issue.hip -- VALU issue cost on gfx950, measured in SHADER CYCLES (s_memtime) inside long,
// post-ramp kernels; replaces the wall-time/assumed-clock table of tools/ubench_valu.hip as the
// evidence behind DESIGN.md's "VALU floor" numbers.
//
// Method: grid = 256 CUs x k workgroups of 256 threads -> k waves per SIMD, every wave runs
// ITER x 32 copies of one instruction (16 independent registers, so no dependency stall at any
// occupancy), brackets the loop with s_memtime (shader clock) and s_memrealtime (100 MHz) and stores
// both deltas.  cycles per wave-instruction per SIMD = median(delta_cycles) / (k * ITER * 32);
// effective clock = delta_cycles / delta_realtime * 100 MHz.  Before the table the chip is warmed
// with ~150 ms of FMA work; every row is >= 8 launches of several ms each, the first two discarded.
// The same rows are also timed with HIP events, so cycles x clock can be checked against wall time.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define UNROLL16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

struct Stamp { unsigned long long cycles, realtime; };

#define PROLOGUE                                                                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                           \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                       \
    const unsigned long long t0 = __builtin_readcyclecounter();                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

#define EPILOGUE                                                                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                           \
    const unsigned long long t1 = __builtin_readcyclecounter();                           \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                    \
    if ((threadIdx.x & 63) == 0) {                                                        \
        Stamp s; s.cycles = t1 - t0; s.realtime = r1 - r0;                                \
        stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;                                  \
    }

// one instruction, 16 independent accumulators, two rounds per loop iteration
#define KERNEL_F32(NAME, ASM)                                                             \
    __global__ __launch_bounds__(256) void NAME(float *out, Stamp *stamps, int iters, float b, float c) \
    {                                                                                     \
        float a[16];                                                                      \
        for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;                     \
        PROLOGUE                                                                          \
        for (int it = 0; it < iters; ++it) {                                              \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                 \
        EPILOGUE                                                                          \
        float s = 0; for (int i = 0; i < 16; ++i) s += a[i];                              \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                          \
    }

// the same with b, c as SGPR operands (uniform kernel arguments), %1 = b, %2 = c
#define KERNEL_F32_SGPR(NAME, ASM)                                                        \
    __global__ __launch_bounds__(256) void NAME(float *out, Stamp *stamps, int iters, float b, float c) \
    {                                                                                     \
        float a[16];                                                                      \
        for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;                     \
        PROLOGUE                                                                          \
        for (int it = 0; it < iters; ++it) {                                              \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "s"(b), "s"(c)); \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "s"(b), "s"(c)); \
        }                                                                                 \
        EPILOGUE                                                                          \
        float s = 0; for (int i = 0; i < 16; ++i) s += a[i];                              \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                          \
    }

#define KERNEL_PK(NAME, ASM)                                                              \
    __global__ __launch_bounds__(256) void NAME(float *out, Stamp *stamps, int iters, float b, float c) \
    {                                                                                     \
        f2 a[16]; f2 bb = {b, b}, cc = {c, c};                                            \
        for (int i = 0; i < 16; ++i) a[i] = f2{threadIdx.x * 0.001f + i, 1.0f};           \
        PROLOGUE                                                                          \
        for (int it = 0; it < iters; ++it) {                                              \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(bb), "v"(cc)); \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(bb), "v"(cc)); \
        }                                                                                 \
        EPILOGUE                                                                          \
        float s = 0; for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;                   \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                          \
    }

KERNEL_F32(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL_F32(k_fma_clamp, "v_fma_f32 %0, %0, %1, %2 clamp")
KERNEL_F32(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL_F32(k_mul_clamp, "v_mul_f32_e64 %0, %0, %1 clamp")
KERNEL_F32(k_add, "v_add_f32 %0, %0, %1")
KERNEL_F32(k_mov, "v_mov_b32 %0, %1")
KERNEL_F32(k_max, "v_max_f32 %0, %0, %1")
KERNEL_F32(k_min3, "v_min3_f32 %0, %0, %1, %2")
KERNEL_F32(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL_F32(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL_F32(k_cmp, "v_cmp_gt_f32 vcc, %0, %1")
KERNEL_F32(k_cmp_class, "v_cmp_class_f32 vcc, %0, %1")
KERNEL_F32(k_cndmask, "v_cndmask_b32_e64 %0, %0, %1, vcc")
KERNEL_F32(k_rcp, "v_rcp_f32 %0, %0")
KERNEL_F32(k_cvt_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %0, %1")
KERNEL_F32(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %0")
KERNEL_F32(k_div_fixup, "v_div_fixup_f32 %0, %0, %1, %2")
KERNEL_F32(k_sub, "v_sub_f32 %0, %0, %1")
KERNEL_F32(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL_F32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL_F32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL_F32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL_F32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL_F32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL_F32(k_cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
KERNEL_F32(k_cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
KERNEL_F32(k_min_f32, "v_min_f32 %0, %0, %1")
KERNEL_F32_SGPR(k_add_sgpr, "v_add_f32 %0, %1, %0")
KERNEL_F32_SGPR(k_fma_sgpr, "v_fma_f32 %0, %0, %0, %2")
KERNEL_F32_SGPR(k_mul_sgpr, "v_mul_f32 %0, %1, %0")
KERNEL_F32(k_mul_literal, "v_mul_f32 %0, 0x3f7fbe77, %0")
KERNEL_PK(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 2, %1")
KERNEL_PK(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL_PK(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
KERNEL_PK(k_pk_add, "v_pk_add_f32 %0, %0, %1")

// three DIFFERENT vector sources per instruction, rotating through 16 registers (real code reads 2-3
// distinct VGPRs per instruction; the rows above re-read the same two operand registers)
__global__ __launch_bounds__(256) void k_fma_three_sources(float *out, Stamp *stamps, int iters, float b, float c)
{
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i * b + c;
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 9) & 15]), "v"(a[(i + 14) & 15]));
    }
    EPILOGUE
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the render pair mix with operands spread over 24 registers (no operand is re-read back to back)
__global__ __launch_bounds__(256) void k_render_mix_spread(float *out, Stamp *stamps, int iters, float b, float c)
{
    float s[16], ir[4], acc[4];
    for (int i = 0; i < 16; ++i) s[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 4; ++i) { ir[i] = b + i * 1e-3f; acc[i] = 0.0f; }
    const float one = 1.0f;
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float d1, d2, p1, p2, u1, u2, sum;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d1) : "v"(s[4 * p]), "v"(ir[p]), "v"(c));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d2) : "v"(s[4 * p + 1]), "v"(ir[p]), "v"(c));
            asm volatile("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(p1) : "v"(d1), "v"(b));
            asm volatile("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(p2) : "v"(d2), "v"(b));
            asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(u1) : "v"(d1), "v"(p2), "v"(one));
            asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(u2) : "v"(d2), "v"(p1), "v"(one));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(sum) : "v"(u1), "v"(u2));
            asm volatile("v_fma_f32 %0, -%1, %2, %3 clamp" : "=v"(acc[p]) : "v"(p1), "v"(p2), "v"(sum));
            s[4 * p + 2] = acc[p];
        }
    }
    EPILOGUE
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + s[2] + s[6] + s[10] + s[14];
}

// the same, with the per-term constants where the compiled kernel has them: -frontDepth and the reject
// fade-off as SGPR operands of v_fma_f32 / v_mul_f32 (one constant-bus read each)
__global__ __launch_bounds__(256) void k_render_mix_sgpr(float *out, Stamp *stamps, int iters, float b, float c)
{
    float s[16], ir[4], acc[4];
    for (int i = 0; i < 16; ++i) s[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 4; ++i) { ir[i] = b + i * 1e-3f; acc[i] = 0.0f; }
    const float one = 1.0f;
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float d1, d2, p1, p2, u1, u2, sum;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d1) : "v"(s[4 * p]), "v"(ir[p]), "s"(c));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d2) : "v"(s[4 * p + 1]), "v"(ir[p]), "s"(c));
            asm volatile("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(p1) : "v"(d1), "s"(b));
            asm volatile("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(p2) : "v"(d2), "s"(b));
            asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(u1) : "v"(d1), "v"(p2), "v"(one));
            asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(u2) : "v"(d2), "v"(p1), "v"(one));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(sum) : "v"(u1), "v"(u2));
            asm volatile("v_fma_f32 %0, -%1, %2, %3 clamp" : "=v"(acc[p]) : "v"(p1), "v"(p2), "v"(sum));
            s[4 * p + 2] = acc[p];
        }
    }
    EPILOGUE
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + s[2] + s[6] + s[10] + s[14];
}

// dependent chain: one accumulator, every instruction waits for the previous one
__global__ __launch_bounds__(256) void k_fma_dependent(float *out, Stamp *stamps, int iters, float b, float c)
{
    float a = threadIdx.x * 0.001f;
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    }
    EPILOGUE
    out[blockIdx.x * 256 + threadIdx.x] = a;
}

// The instruction mix of one TestSamplePair of the render kernel (Render.compute:60-75) as the
// compiler emits it: 2 v_fma, 2 v_mul clamp, 2 v_med3, 1 v_add, 1 v_fma clamp; 4 independent pairs per
// round (32 instructions), operands from registers only.
__global__ __launch_bounds__(256) void k_render_mix(float *out, Stamp *stamps, int iters, float b, float c)
{
    float s[8], acc[4];
    for (int i = 0; i < 8; ++i) s[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 4; ++i) acc[i] = 0.0f;
    const float one = 1.0f;
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float d1, d2, p1, p2, u1, u2, sum;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d1) : "v"(s[2 * p]), "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d2) : "v"(s[2 * p + 1]), "v"(b), "v"(c));
            asm volatile("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(p1) : "v"(d1), "v"(c));
            asm volatile("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(p2) : "v"(d2), "v"(c));
            asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(u1) : "v"(d1), "v"(p2), "v"(one));
            asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(u2) : "v"(d2), "v"(p1), "v"(one));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(sum) : "v"(u1), "v"(u2));
            asm volatile("v_fma_f32 %0, -%1, %2, %3 clamp" : "=v"(acc[p]) : "v"(p1), "v"(p2), "v"(sum));
            s[2 * p] = acc[p];
        }
    }
    EPILOGUE
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

// the same mix with the two texels of a lane packed: v_pk_fma / v_pk_mul clamp / v_pk_add, med3 scalar
__global__ __launch_bounds__(256) void k_render_mix_pk(float *out, Stamp *stamps, int iters, float b, float c)
{
    f2 s[8], acc[4];
    const f2 bb = {b, b}, cc = {c, c};
    for (int i = 0; i < 8; ++i) s[i] = f2{threadIdx.x * 0.001f + i, 1.0f};
    for (int i = 0; i < 4; ++i) acc[i] = f2{0.0f, 0.0f};
    const float one = 1.0f;
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            f2 d1, d2, p1, p2, u1, u2, sum;
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d1) : "v"(s[2 * p]), "v"(bb), "v"(cc));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d2) : "v"(s[2 * p + 1]), "v"(bb), "v"(cc));
            asm volatile("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(p1) : "v"(d1), "v"(cc));
            asm volatile("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(p2) : "v"(d2), "v"(cc));
            asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(u1.x) : "v"(d1.x), "v"(p2.x), "v"(one));
            asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(u1.y) : "v"(d1.y), "v"(p2.y), "v"(one));
            asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(u2.x) : "v"(d2.x), "v"(p1.x), "v"(one));
            asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(u2.y) : "v"(d2.y), "v"(p1.y), "v"(one));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(sum) : "v"(u1), "v"(u2));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(acc[p]) : "v"(p1), "v"(p2), "v"(sum));
            s[2 * p] = acc[p];
        }
    }
    EPILOGUE
    out[blockIdx.x * 256 + threadIdx.x] = acc[0].x + acc[1].y + acc[2].x + acc[3].y;
}

// alternating fast / slow class instructions (1 : 1)
__global__ __launch_bounds__(256) void k_fma_med3_alternating(float *out, Stamp *stamps, int iters, float b, float c)
{
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[(i + 8) & 15]) : "v"(b), "v"(c));
        }
    }
    EPILOGUE
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// VALU next to LDS reads: 8 v_fma per ds_read_b64 (the render loop has ~8 VALU per LDS read)
__global__ __launch_bounds__(256) void k_fma_with_lds(float *out, Stamp *stamps, int iters, float b, float c)
{
    __shared__ f2 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = f2{b, c};
    __syncthreads();
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    f2 acc = {0, 0};
    const unsigned addr = (threadIdx.x & 255) * 8;
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f2 v;
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(0));
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(g & 1) * 8 + i]) : "v"(b), "v"(c));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc += v;
        }
    }
    EPILOGUE
    float s = acc.x + acc.y; for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}


// ---- round 3: does the transcendental (quarter-rate) pipe overlap with full-rate VALU work? ------------
// 1 v_rcp_f32 per 7 v_fma_f32 inside every wave (the bilateral phase of the upsample kernels: 5 rcp of ~44)
__global__ __launch_bounds__(256) void k_rcp_1_in_8(float *out, Stamp *stamps, int iters, float b, float c)
{
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i + 1.0f;
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            asm volatile("v_rcp_f32 %0, %0" : "+v"(a[g]));
#pragma unroll
            for (int i = 0; i < 7; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4 + ((g * 7 + i) % 12)]) : "v"(b), "v"(c));
        }
    }
    EPILOGUE
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 1 : 3 (a denser transcendental mix)
__global__ __launch_bounds__(256) void k_rcp_1_in_4(float *out, Stamp *stamps, int iters, float b, float c)
{
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i + 1.0f;
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            asm volatile("v_rcp_f32 %0, %0" : "+v"(a[g & 3]));
#pragma unroll
            for (int i = 0; i < 3; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4 + ((g * 3 + i) % 12)]) : "v"(b), "v"(c));
        }
    }
    EPILOGUE
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// workgroups alternate: even ones run only v_rcp_f32, odd ones only v_fma_f32 (every SIMD then holds waves of both
// kinds).  Reported per instruction over BOTH kinds: 5.3 = no overlap ((8.2 + 2.4) / 2), 4.1 = the rcp waves alone
// bound it (perfect overlap), in between = partial.
__global__ __launch_bounds__(256) void k_rcp_waves_next_to_fma_waves(float *out, Stamp *stamps, int iters, float b, float c)
{
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i + 1.0f;
    PROLOGUE
    if (blockIdx.x & 1) {
        for (int it = 0; it < iters; ++it) {
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        }
    }
    EPILOGUE
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// The bilateral phase of the upsample kernels as the compiler emits it, per hi-res texel: four weights
// K / (|hd - d| + tol) as v_sub, v_add |.|, v_rcp, v_mul, 2 v_fma (K = 1: 3 after the rcp), the two sums, the exact
// quotient (v_rcp + 2 v_fma + v_mul + 2 v_fma), the UNORM8 encode; two independent texels per round: 84 instructions.
// MODE 0: every weight finished before the next one starts (source order); 1: the four v_rcp_f32 of a texel issued
// back to back (does the transcendental pipe like batches?); 2: the four v_rcp of BOTH texels back to back;
// 3: no correction steps (raw rcp * K, raw rcp * sum: what an approximate-then-verify scheme would issue; 66 instructions)
template <int MODE>
__device__ __forceinline__ void bilateral_two_texels(const float (&hd)[2], float (&d)[8], const float (&ao)[8], float (&res)[2])
{
    const float tol = 1e-12f, nine = 9.0f, three = 3.0f, one = 1.0f;
    float x[8], r[8], w[8];
    auto arg = [&](int t, int k) {
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x[4 * t + k]) : "v"(hd[t]), "v"(d[4 * t + k]));
        asm volatile("v_add_f32 %0, |%1|, %2" : "=v"(x[4 * t + k]) : "v"(x[4 * t + k]), "v"(tol));
    };
    auto rcp = [&](int t, int k) { asm volatile("v_rcp_f32 %0, %1" : "=v"(r[4 * t + k]) : "v"(x[4 * t + k])); };
    auto fin = [&](int t, int k) {
        const int i = 4 * t + k;
        float q, e;
        if (MODE == 3) {
            if (k == 2) { w[i] = r[i]; return; }
            const float K = k == 0 ? nine : three;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w[i]) : "v"(K), "v"(r[i]));
            return;
        }
        if (k == 2) {
            asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(e) : "v"(x[i]), "v"(r[i]), "v"(one));
            asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(w[i]) : "v"(e), "v"(r[i]));
        } else {
            const float K = k == 0 ? nine : three;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q) : "v"(K), "v"(r[i]));
            asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(e) : "v"(x[i]), "v"(q), "v"(K));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(w[i]) : "v"(e), "v"(r[i]), "v"(q));
        }
    };
    auto tail = [&](int t) {
        float total, sum, rr, e, q;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(total) : "v"(w[4 * t]), "v"(w[4 * t + 1]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(total) : "v"(w[4 * t + 2]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(total) : "v"(w[4 * t + 3]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(total) : "v"(one));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(sum) : "v"(ao[4 * t]), "v"(w[4 * t]));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sum) : "v"(ao[4 * t + 1]), "v"(w[4 * t + 1]));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sum) : "v"(ao[4 * t + 2]), "v"(w[4 * t + 2]));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sum) : "v"(ao[4 * t + 3]), "v"(w[4 * t + 3]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(one));
        asm volatile("v_rcp_f32 %0, %1" : "=v"(rr) : "v"(total));
        if (MODE == 3 || MODE == 4) {
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q) : "v"(sum), "v"(rr));
        } else {
            asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(e) : "v"(total), "v"(rr), "v"(one));
            asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(rr) : "v"(e));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q) : "v"(sum), "v"(rr));
            asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(e) : "v"(total), "v"(q), "v"(sum));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(q) : "v"(e), "v"(rr));
        }
        asm volatile("v_mul_f32_e64 %0, %0, %1 clamp" : "+v"(q) : "v"(one));
        asm volatile("v_mul_f32 %0, 0x437f0000, %0" : "+v"(q));
        asm volatile("v_add_f32 %0, %0, 0.5" : "+v"(q));
        asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(q));
        res[t] = q;
        d[4 * t] = q;        // loop-carried: the next round's first tap depends on this one (no hoisting)
    };
#define EACH_T _Pragma("unroll") for (int t = 0; t < 2; ++t)
#define EACH_K _Pragma("unroll") for (int k = 0; k < 4; ++k)
    if (MODE == 0) {                    // texel after texel, weight after weight
        EACH_T { EACH_K { arg(t, k); rcp(t, k); fin(t, k); } tail(t); }
    } else if (MODE == 1) {             // per texel: 4 args, 4 rcp back to back, 4 corrections
        EACH_T { EACH_K arg(t, k); EACH_K rcp(t, k); EACH_K fin(t, k); tail(t); }
    } else if (MODE == 4) {             // no correction steps, the two reciprocals of a tap pair from ONE v_rcp_f32 of their product:
                                        // 1/x0 = x1 * rcp(x0 x1), 1/x1 = x0 * rcp(x0 x1) -- 6 rcp instead of 10 per two texels, 12 more v_mul (70 instructions)
        EACH_T EACH_K arg(t, k);
        float p[4], rp[4];
        EACH_T { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p[2 * t]) : "v"(x[4 * t]), "v"(x[4 * t + 1]));
                 asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p[2 * t + 1]) : "v"(x[4 * t + 2]), "v"(x[4 * t + 3])); }
        for (int i = 0; i < 4; ++i) asm volatile("v_rcp_f32 %0, %1" : "=v"(rp[i]) : "v"(p[i]));
        EACH_T {
            float k9, k3;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(k9) : "v"(nine), "v"(rp[2 * t]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(k3) : "v"(three), "v"(rp[2 * t]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w[4 * t]) : "v"(x[4 * t + 1]), "v"(k9));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w[4 * t + 1]) : "v"(x[4 * t]), "v"(k3));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(k3) : "v"(three), "v"(rp[2 * t + 1]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w[4 * t + 2]) : "v"(x[4 * t + 3]), "v"(rp[2 * t + 1]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w[4 * t + 3]) : "v"(x[4 * t + 2]), "v"(k3));
        }
        { constexpr int SAVED = MODE; (void)SAVED; }
        EACH_T tail(t);
    } else if (MODE == 2 || MODE == 3) {  // both texels: 8 args, 8 rcp back to back, 8 corrections, tails
        EACH_T EACH_K arg(t, k);
        EACH_T EACH_K rcp(t, k);
        EACH_T EACH_K fin(t, k);
        EACH_T tail(t);
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_bilateral_mix(float *out, Stamp *stamps, int iters, float b, float c)
{
    float hd[2], d[8], ao[8], res[2];
    for (int i = 0; i < 8; ++i) { d[i] = threadIdx.x * 0.001f + i; ao[i] = 0.5f + 0.01f * i; }
    hd[0] = b; hd[1] = c; res[0] = res[1] = 0.0f;
    PROLOGUE
    for (int it = 0; it < iters; ++it) bilateral_two_texels<MODE>(hd, d, ao, res);
    EPILOGUE
    out[blockIdx.x * 256 + threadIdx.x] = res[0] + res[1];
}

// ---- round 3: v_fma_mix_f32 (an f16 source widened inside the FMA: samples could stay f16 in LDS) ------------
KERNEL_F32(k_fma_mix_lo, "v_fma_mix_f32 %0, %1, %0, %2 op_sel_hi:[1,0,0]")
KERNEL_F32(k_fma_mix_hi, "v_fma_mix_f32 %0, %1, %0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]")

// the render pair mix with both v_fma_f32 of a pair replaced by v_fma_mix_f32 on the two halves of one packed f16 pair
__global__ __launch_bounds__(256) void k_render_mix_fma_mix(float *out, Stamp *stamps, int iters, float b, float c)
{
    float s[16], ir[4], acc[4];
    for (int i = 0; i < 16; ++i) s[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 4; ++i) { ir[i] = b + i * 1e-3f; acc[i] = 0.0f; }
    const float one = 1.0f;
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float d1, d2, p1, p2, u1, u2, sum;
            asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(s[4 * p]), "v"(ir[p]), "v"(c));
            asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d2) : "v"(s[4 * p + 1]), "v"(ir[p]), "v"(c));
            asm volatile("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(p1) : "v"(d1), "v"(b));
            asm volatile("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(p2) : "v"(d2), "v"(b));
            asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(u1) : "v"(d1), "v"(p2), "v"(one));
            asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(u2) : "v"(d2), "v"(p1), "v"(one));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(sum) : "v"(u1), "v"(u2));
            asm volatile("v_fma_f32 %0, -%1, %2, %3 clamp" : "=v"(acc[p]) : "v"(p1), "v"(p2), "v"(sum));
            s[4 * p + 2] = acc[p];
        }
    }
    EPILOGUE
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + s[2] + s[6] + s[10] + s[14];
}

typedef void (*kfn)(float *, Stamp *, int, float, float);

struct Row { const char *name; kfn fn; int per_iter; };

int main(int argc, char **argv)
{
    const double target_ms = argc > 1 ? std::atof(argv[1]) : 4.0;   // per launch
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    std::printf("# device %s, %d CUs, clockRate %.0f MHz; rows: >= 8 launches of ~%.1f ms, first 2 dropped\n",
                prop.gcnArchName, cus, prop.clockRate / 1000.0, target_ms);
    float *out;
    Stamp *stamps;
    const int max_blocks = cus * 8;
    hipMalloc(&out, size_t(max_blocks) * 256 * sizeof(float));
    hipMalloc(&stamps, size_t(max_blocks) * 4 * sizeof(Stamp));
    std::vector<Stamp> host(size_t(max_blocks) * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);

    // warm-up: ~150 ms of FMA work at full occupancy
    for (int i = 0; i < 40; ++i) k_fma<<<max_blocks, 256>>>(out, stamps, 4096, 0.999f, 0.001f);
    hipDeviceSynchronize();

    const Row rows[] = {
        {"v_fma_f32", k_fma, 32}, {"v_fma_f32 clamp", k_fma_clamp, 32}, {"v_mul_f32", k_mul, 32},
        {"v_mul_f32 clamp (e64)", k_mul_clamp, 32}, {"v_add_f32", k_add, 32}, {"v_mov_b32", k_mov, 32},
        {"v_max_f32", k_max, 32}, {"v_min3_f32", k_min3, 32}, {"v_med3_f32", k_med3, 32}, {"v_or3_b32", k_or3, 32},
        {"v_cmp_gt_f32", k_cmp, 32}, {"v_cmp_class_f32", k_cmp_class, 32}, {"v_cndmask_b32 (e64)", k_cndmask, 32},
        {"v_rcp_f32", k_rcp, 32}, {"v_cvt_pkrtz_f16_f32", k_cvt_pkrtz, 32}, {"v_cvt_f32_f16", k_cvt_f32_f16, 32},
        {"v_div_fixup_f32", k_div_fixup, 32},
        {"v_sub_f32", k_sub, 32}, {"v_fmac_f32 (VOP2)", k_fmac, 32}, {"v_min_f32", k_min_f32, 32},
        {"v_add_u32", k_add_u32, 32}, {"v_lshl_add_u32", k_lshl_add_u32, 32}, {"v_mad_u32_u24", k_mad_u32_u24, 32},
        {"v_mul_lo_u32", k_mul_lo_u32, 32}, {"v_and_or_b32", k_and_or, 32}, {"v_cvt_u32_f32", k_cvt_u32_f32, 32},
        {"v_cvt_f32_ubyte0", k_cvt_f32_ubyte0, 32}, {"v_lshl_add_u64", k_lshl_add_u64, 32},
        {"v_add_f32, one SGPR source", k_add_sgpr, 32}, {"v_mul_f32, one SGPR source", k_mul_sgpr, 32},
        {"v_fma_f32, one SGPR source", k_fma_sgpr, 32}, {"v_mul_f32, 32-bit literal source", k_mul_literal, 32},
        {"v_pk_fma_f32", k_pk_fma, 32}, {"v_pk_mul_f32", k_pk_mul, 32}, {"v_pk_add_f32", k_pk_add, 32},
        {"v_fma_f32, three distinct rotating sources", k_fma_three_sources, 32},
        {"render pair mix, operands spread over 24 regs", k_render_mix_spread, 32},
        {"render pair mix, spread, constants in SGPRs", k_render_mix_sgpr, 32},
        {"v_fma_f32 dependent chain", k_fma_dependent, 32},
        {"render pair mix (8 instr / texel pair-op)", k_render_mix, 32},
        {"render pair mix, packed (10 instr / 2 texels)", k_render_mix_pk, 40},
        {"v_fma_f32 / v_med3_f32 alternating", k_fma_med3_alternating, 32},
        {"8 x v_fma_f32 per ds_read_b64 (VALU instr only)", k_fma_with_lds, 32},
        {"1 v_rcp_f32 : 7 v_fma_f32 in every wave", k_rcp_1_in_8, 32},
        {"1 v_rcp_f32 : 3 v_fma_f32 in every wave", k_rcp_1_in_4, 32},
        {"v_rcp waves next to v_fma waves (both counted)", k_rcp_waves_next_to_fma_waves, 32},
        {"v_fma_mix_f32, f16 source (low half)", k_fma_mix_lo, 32},
        {"v_fma_mix_f32, f16 source (high half)", k_fma_mix_hi, 32},
        {"render pair mix with v_fma_mix_f32 (spread)", k_render_mix_fma_mix, 32},
        {"bilateral-phase mix (2 texels: 84 instr, 10 rcp)", k_bilateral_mix<0>, 84},
        {"bilateral mix, 4 rcp of a texel back to back", k_bilateral_mix<1>, 84},
        {"bilateral mix, 8 rcp of both texels back to back", k_bilateral_mix<2>, 84},
        {"bilateral mix without correction steps (66 instr)", k_bilateral_mix<3>, 66},
        {"the same, tap-pair reciprocals from one rcp (6 rcp, 70 instr)", k_bilateral_mix<4>, 70},
    };
    const char *filter = argc > 2 ? argv[2] : nullptr;    // only rows whose name contains this
    std::printf("%-48s %5s %10s %10s %9s %10s\n", "instruction", "w/SIMD", "cyc/instr", "clock MHz", "ms/launch", "cyc(wall)");
    for (const Row &r : rows) {
        if (filter && !std::strstr(r.name, filter)) continue;
        for (int k : {1, 2, 4, 7, 8}) {
            const int blocks = cus * k;
            // calibrate iterations for ~target_ms per launch
            int iters = 2048;
            r.fn<<<blocks, 256>>>(out, stamps, iters, 0.999f, 0.001f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            r.fn<<<blocks, 256>>>(out, stamps, iters, 0.999f, 0.001f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            iters = std::max(256, int(iters * target_ms / std::max(ms, 1e-3f)));
            std::vector<double> cyc, mhz, wall;
            for (int rep = 0; rep < 8; ++rep) {
                hipEventRecord(e0);
                r.fn<<<blocks, 256>>>(out, stamps, iters, 0.999f, 0.001f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                if (rep < 2) continue;
                hipMemcpy(host.data(), stamps, size_t(blocks) * 4 * sizeof(Stamp), hipMemcpyDeviceToHost);
                std::vector<double> c, f;
                for (int w = 0; w < blocks * 4; ++w) {
                    c.push_back(double(host[w].cycles));
                    f.push_back(double(host[w].cycles) / double(host[w].realtime) * 100.0);
                }
                std::nth_element(c.begin(), c.begin() + c.size() / 2, c.end());
                std::nth_element(f.begin(), f.begin() + f.size() / 2, f.end());
                cyc.push_back(c[c.size() / 2]);
                mhz.push_back(f[f.size() / 2]);
                wall.push_back(ms);
            }
            std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end()); std::sort(wall.begin(), wall.end());
            const double n_instr = double(iters) * r.per_iter;
            const double per = cyc[cyc.size() / 2] / (n_instr * k);   // the SIMD issued k * n_instr in that time
            const double clk = mhz[mhz.size() / 2];
            const double wall_ms = wall[wall.size() / 2];
            // cross-check: wall time x measured clock / instructions per SIMD
            const double per_wall = wall_ms * 1e-3 * clk * 1e6 / (n_instr * k);
            std::printf("%-48s %5d %10.3f %10.0f %9.3f %10.3f\n", r.name, k, per, clk, wall_ms, per_wall);
        }
    }
    return 0;
}
