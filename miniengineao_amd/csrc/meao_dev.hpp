// meao_dev.hpp -- device-side helpers every kernel translation unit shares: build switches, vector types, storage conversions,
// the exact-division sequences, frame addressing, the XCD-contiguous tile mapping.  (Design notes: meao_kernels.hip.)
#pragma once

#include "meao_kernels.hpp"

#include <algorithm>
#include <type_traits>

// Every design decision below that replaced an alternative was A/B-measured on one box; the arms that
// lost (or changed nothing) were removed in round 3 -- their logs stay in profiles/ (r02_ab_*.jsonl) and
// profiles/README.md lists them.  Experimental arms of the current round live behind the MEAO_X_* switches
// of this block only (tools/build_variants.py builds variants next to the product library;
// tests/test_variants_gpu.py runs a parity smoke through every variant library it finds).
#ifndef MEAO_X_UPS_EXACT_R8
#define MEAO_X_UPS_EXACT_R8 0      // 1 = every UNORM8 bilateral result through the full exact-division sequence (the round-2 form) instead of
#endif                             // bilateral_upsample_r8: L1->L0 176 -> 196 us, L2->L1 54 -> 57 us (profiles/r03_ab_verified_r8_bilateral.txt)
#ifndef MEAO_X_BIL_WHOLE_TILE
#define MEAO_X_BIL_WHOLE_TILE 1    // 0 = no separate copy of the bilateral phase for tiles that lie wholly inside the frame (the round-3 form):
#endif                             // last kernel 296 -> 272 us, step 570 -> 551 us (profiles/r04_ab_bilateral_arms.jsonl)
#ifndef MEAO_X_BIL_PAIR_RCP
#define MEAO_X_BIL_PAIR_RCP 0      // 1 = three v_rcp_f32 per UNORM8 bilateral texel instead of five (the reciprocals of a tap pair from one reciprocal
#endif                             // of their product, bilateral_upsample_r8<PAIRED>; variant library `pair`).  Proven and device-checked like the
                                   // five-reciprocal form, -5.5 % on the instruction mix in isolation -- and nothing in the kernels: last kernel
                                   // 269.6 vs 269.4 us, full-resolution pass 170.7 vs 170.3 us, L2->L1 54.9 vs 53.5 us per 16 frames (the exact
                                   // path can no longer start from the estimate's reciprocals): profiles/r05_ab_pair_rcp.jsonl
#ifndef MEAO_X_FINAL_NT_STORE
#define MEAO_X_FINAL_NT_STORE 0    // 1 = the result texels of the full-resolution pass with non-temporal stores (rounds 2-4; variant `ntstore`).  A 64-texel
#endif                             // tile row of R8 results is HALF a 128-byte line, the other half belongs to the next tile: streamed out at once the
                                   // halves cost 1.31x their bytes in HBM writes (WRITE_SIZE 10.82 MB per 4K frame for 8.29 MB of results; temporal:
                                   // 8.30 MB -- L2 merges the halves), last kernel 268.7 -> 267.1 us: profiles/r05_result_store_write_size.txt,
                                   // r05_ab_result_stores.jsonl.  (fp16 results are whole lines per tile row: 1.00x either way.)
#ifndef MEAO_X_LOWDEPTH_FROM_RAW
#define MEAO_X_LOWDEPTH_FROM_RAW 1 // 0 = the full-resolution pass reads its LoResDB window from the LowDepth1 buffer (rounds 1-6a; variant `lowbuf`).  1: a tile
#endif                             // inside the frame never reads LowDepth1 -- the 32 x 32 interior of its window ARE the pre-rounding values of its own
                                   // even-even raw texels (DS1:64-70), which hi_depth_words has in registers, and the 3-texel apron is linearized from
                                   // raw texels of the neighbouring tiles' lines (L2 hits: they are those tiles' HiResDB): 8.3 MB per 4K frame less
#ifndef MEAO_X_HOT_PATH_ONLY
#define MEAO_X_HOT_PATH_ONLY 0  // ANALYSIS builds only (tools/kernel_isa.py -DMEAO_X_HOT_PATH_ONLY=1 --stats; never a library): the upsample
#endif                          // and render kernels keep nothing but the path an interior tile of a clean frame takes, so that the
                                // static instruction counts of the ISA are the dynamic ones of (almost) every workgroup
#ifndef MEAO_X_PHASE_CLOCKS
#define MEAO_X_PHASE_CLOCKS 0   // diagnostic build: upsample tiles stamp s_memrealtime at their phase boundaries (tools/phase_clocks.py),
#endif                          // the render launch logs start / end / CU of every workgroup (tools/render_wg_log.py)

#if MEAO_X_PHASE_CLOCKS
#ifndef MEAO_UNITY_BUILD
#error "MEAO_X_PHASE_CLOCKS needs the single-translation-unit build (meao_kernels.hip): its device counters are one set of globals"
#endif
// [phase] summed 100 MHz ticks and [32 + phase] wave counts, per upsample-tile phase (0..7 full-resolution pass,
// 8..15 blend passes); read and cleared by meao_x_phase_clocks
__device__ unsigned long long g_phase_clocks[64];
extern "C" __attribute__((visibility("default"))) int meao_x_phase_clocks(unsigned long long *out64)
{
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_phase_clocks), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    static const unsigned long long zero[64] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_clocks), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("default"))) int meao_x_wg_log_preset(void);
// per workgroup of the last logged launch: start, end (100 MHz), HW_ID, XCC_ID
__device__ unsigned long long g_wg_log[16384 * 4];
extern "C" __attribute__((visibility("default"))) int meao_x_wg_log(unsigned long long *out, int workgroups)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_log), sizeof(unsigned long long) * 4 * (workgroups < 16384 ? workgroups : 16384)) == hipSuccess ? 0 : -1;
}
extern "C" int meao_x_wg_log_preset(void)       // render log: min fields to ~0, max field to 0
{
    static unsigned long long init[16384 * 4];
    for (int i = 0; i < 16384; ++i) { init[4 * i] = ~0ull; init[4 * i + 1] = 0; init[4 * i + 2] = ~0ull; init[4 * i + 3] = 0; }
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wg_log), init, sizeof init) == hipSuccess ? 0 : -1;
}
#endif

namespace meao {
namespace {

// Phase stamps of a tile (diagnostic builds only; compiles to nothing otherwise): lane 0 of every wave adds the
// time since its previous stamp to the phase's accumulator.
struct PhaseClock {
#if MEAO_X_PHASE_CLOCKS
    // one workgroup in 32 is sampled; the others never read the clock (the read needs an s_waitcnt lgkmcnt(0))
    unsigned long long last;
    int base;
    bool on;
    __device__ __forceinline__ explicit PhaseClock(int base_) : last(0), base(base_), on((blockIdx.x & 31) == 0)
    {
        if (on) last = __builtin_amdgcn_s_memrealtime();
    }
    __device__ __forceinline__ void mark(int phase)
    {
        if (!on) return;
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&g_phase_clocks[base + phase], now - last);
            atomicAdd(&g_phase_clocks[32 + base + phase], 1ull);
        }
        last = now;
    }
#else
    __device__ __forceinline__ explicit PhaseClock(int) {}
    __device__ __forceinline__ void mark(int) {}
#endif
};

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef uint32_t uint4v __attribute__((ext_vector_type(4)));
typedef uint16_t ushort4v __attribute__((ext_vector_type(4)));
typedef uint16_t ushort2v __attribute__((ext_vector_type(2)));
typedef uint8_t uchar4v __attribute__((ext_vector_type(4)));
typedef uint8_t uchar2v __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;

// ------------------------------------------------------------------------------------------
// scalar / packed helpers

__device__ __forceinline__ float mad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// threadIdx.x behind an optimisation barrier: lane-dependent indices (LDS addresses, run numbers, texel coordinates) derived
// from it are computed where they are used instead of being hoisted to the top of the tile into long-lived registers.
__device__ __forceinline__ int thread_index_opaque()
{
    int t = static_cast<int>(threadIdx.x);
    asm volatile("" : "+v"(t));
    return t;
}
__device__ __forceinline__ float sat(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

__device__ __forceinline__ float2v splat(float x) { return float2v{x, x}; }
__device__ __forceinline__ float2v fma2(float2v a, float2v b, float2v c) { return __builtin_elementwise_fma(a, b, c); }

// f32 -> f16 store conversion (HalfUAV targets).  RTZ: v_cvt_pkrtz_f16_f32 rounds toward zero,
// so finite overflow lands on 65504; RTNE: v_cvt_f16_f32 in the default rounding mode.
template <bool RTNE>
__device__ __forceinline__ uint16_t f32_to_f16_bits(float x)
{
    if constexpr (RTNE) {
        // The value is pinned in a VGPR first: without it LLVM folds "round(a * b)" into
        // v_fma_mixlo_f16 a, b, +0, which returns +0 for a product of -0 (seen in the composite
        // kernel, caught by tests/test_composite.py) -- the conversion must stay a plain v_cvt_f16_f32.
        asm volatile("" : "+v"(x));
        const _Float16 h = static_cast<_Float16>(x);
        return __builtin_bit_cast(uint16_t, h);
    } else {
        const auto p = __builtin_amdgcn_cvt_pkrtz(x, 0.0f);
        return static_cast<uint16_t>(__builtin_bit_cast(uint32_t, p) & 0xffffu);
    }
}

__device__ __forceinline__ float f16_bits_to_f32(uint16_t b)
{
    return static_cast<float>(__builtin_bit_cast(_Float16, b));
}

// value an f16 render target returns after storing x
template <bool RTNE>
__device__ __forceinline__ float through_f16(float x) { return f16_bits_to_f32(f32_to_f16_bits<RTNE>(x)); }

// the same for two values: RTZ converts both with one v_cvt_pkrtz_f16_f32
template <bool RTNE>
__device__ __forceinline__ float2v through_f16_pair(float x, float y)
{
    if constexpr (RTNE) {
        return float2v{through_f16<true>(x), through_f16<true>(y)};
    } else {
        const auto p = __builtin_amdgcn_cvt_pkrtz(x, y);
        return float2v{static_cast<float>(p[0]), static_cast<float>(p[1])};
    }
}

// f32 -> UNORM8 (FixedUAV targets): NaN -> 0, clamp, *255, +0.5, truncate
__device__ __forceinline__ uint32_t f32_to_unorm8(float x)
{
    float s = sat(x) * 255.0f;
    s = s + 0.5f;
    return static_cast<uint32_t>(s);
}

// UNORM8 -> f32 == (float)n / 255.0f exactly, in two operations behind the conversion: fma(n, c, n * c_lo) with c = RN(1/255)
// and c_lo = RN(1/255 - c), i.e. n times a double-float 1/255 with one rounding at the end -- correctly rounded for every
// n in 0..255 (checked exhaustively with exact rational arithmetic when the constants were chosen, on the device by
// meao_selftest(2), and by tests/test_abi.py).  One operation less than quotient estimate + fused remainder step.
__device__ __forceinline__ float unorm8_to_f32(uint32_t n)
{
    const float fn = static_cast<float>(n);
    constexpr float c = 0x1.010102p-8f;              // RN(1 / 255) = 0x3b808081
    constexpr float c_lo = -0x1.fdfdfep-33f;         // RN(1 / 255 - c) = -2.3191758e-10
    return mad(fn, c, fn * c_lo);
}

// N-bit UNORM -> f32 == (float)n / (2^N - 1) exactly, same construction as unorm8_to_f32
// (tests/test_abi.py checks the sequence against IEEE division for all 2^16 and 2^24 codes on
// the CPU; fmaf is the same operation on both sides).
template <int N>
__device__ __forceinline__ float unorm_to_f32(uint32_t n)
{
    constexpr float D = static_cast<float>((1u << N) - 1u);
    const float fn = static_cast<float>(n);
    const float r = 1.0f / D;
    const float q = fn * r;
    const float e = mad(-D, q, fn);
    return mad(e, r, q);
}

template <int AOFMT>
struct AoTexel;
template <>
struct AoTexel<MEAO_AO_R8> {
    typedef uint8_t type;
    typedef uchar2v type2;
    typedef uchar4v type4;
    template <bool RTNE>
    static __device__ __forceinline__ type encode(float v) { return static_cast<uint8_t>(f32_to_unorm8(v)); }
    static __device__ __forceinline__ float decode(type t) { return unorm8_to_f32(t); }
};
template <>
struct AoTexel<MEAO_AO_F16> {
    typedef uint16_t type;
    typedef ushort2v type2;
    typedef ushort4v type4;
    template <bool RTNE>
    static __device__ __forceinline__ type encode(float v) { return f32_to_f16_bits<RTNE>(v); }
    static __device__ __forceinline__ float decode(type t) { return f16_bits_to_f32(t); }
};

// Intermediates of frame f live stride_bytes * f behind frame 0's.  Pointer arithmetic (not an
// integer round trip) so the compiler keeps the global address space and emits global_load/store.
template <typename T>
__device__ __forceinline__ T *frame_ptr(T *base, uint64_t stride_bytes, int frame)
{
    typedef typename std::conditional<std::is_const<T>::value, const char, char>::type byte_t;
    return reinterpret_cast<T *>(reinterpret_cast<byte_t *>(base) + stride_bytes * static_cast<uint64_t>(frame));
}

// Uniform base + 32-bit byte offset: the form the global_load/store "saddr" addressing mode takes (SGPR base,
// zero-extended VGPR offset), no 64-bit VALU address arithmetic.  Every intermediate of a frame is < 4 GB.
template <typename T>
__device__ __forceinline__ T *at_byte_offset(T *uniform_base, uint32_t byte_offset)
{
    typedef typename std::conditional<std::is_const<T>::value, const char, char>::type byte_t;
    return reinterpret_cast<T *>(reinterpret_cast<byte_t *>(uniform_base) + byte_offset);
}

// ------------------------------------------------------------------------------------------
// Exact division without the generic IEEE expansion.
//
// DIV_EXACT_RCP: v_rcp_f32 (1 ulp) followed by fused Newton / remainder steps.  On gfx950 these
// sequences return the correctly rounded quotient -- bit-identical to IEEE '/' -- for
//   rcp_strict(x)        every x with 2^-100 <= |x| <= 2^100          (exhaustive, 2^32 inputs)
//   div_const<3|9>(x)    every such x                                 (exhaustive)
//   div_strict(a, b)     a = 0 or 2^-60 <= |a|,|b| <= 2^60            (Markstein's theorem: the
//                        reciprocal is correctly rounded; 1.6e10 random pairs in tools/ubench_div)
// and are re-verified on the running device by meao_selftest(4..6).  The host selects this mode
// only when the operands are provably inside those ranges (RTZ depth storage, so no inf from sky
// texels; tolerances inside the component's ranges), otherwise DIV_IEEE (hipcc's expansion).
enum { DIV_EXACT_RCP = 0, DIV_IEEE = 1 };

template <int DIV>
__device__ __forceinline__ float rcp_strict(float x)
{
    if constexpr (DIV == DIV_EXACT_RCP) {
        const float r = __builtin_amdgcn_rcpf(x);
        const float e = mad(-x, r, 1.0f);
        return mad(e, r, r);
    } else {
        return 1.0f / x;
    }
}

template <int DIV, int K>
__device__ __forceinline__ float div_const(float x, float k_value = static_cast<float>(K))   // K / x, K in {1, 3, 9}; k_value == K (a register copy of it)
{
    if constexpr (DIV == DIV_EXACT_RCP) {
        if constexpr (K == 1) return rcp_strict<DIV>(x);
        const float r = __builtin_amdgcn_rcpf(x);
        const float q = k_value * r;
        const float e = mad(-x, q, k_value);
        return mad(e, r, q);
    } else {
        return static_cast<float>(K) / x;
    }
}

template <int DIV>
__device__ __forceinline__ float div_strict(float a, float b)
{
    if constexpr (DIV == DIV_EXACT_RCP) {
        const float r = rcp_strict<DIV>(b);
        const float q = a * r;
        const float e = mad(-b, q, a);
        return mad(e, r, q);
    } else {
        return a / b;
    }
}

// Workgroup ids are dealt round-robin to the 8 XCDs (id mod 8), each with its own L2.  This maps the
// ids one XCD receives to a contiguous range of tiles, so that neighbouring tiles -- which share their
// aprons -- share an L2.  Bijection of [0, n).
__device__ __forceinline__ int xcd_contiguous(int id, int n)
{
    const int q = n >> 3, r = n & 7, xcd = id & 7;
    return xcd * q + min(xcd, r) + (id >> 3);
}

// True when the downsample pass that produced this frame's depth mips saw a texel outside the
// verified operand range of the exact v_rcp_f32 sequences (see nice_denominator).
__device__ __forceinline__ bool frame_is_hostile(const uint32_t *hostile, uint32_t generation, int frame)
{
    if constexpr (MEAO_X_HOT_PATH_ONLY) return false;
    return __builtin_nontemporal_load(hostile + frame) == generation;
}

}  // namespace
}  // namespace meao
