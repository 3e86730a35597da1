// meao_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the multi-scale SSAO hot path.
//
// One workgroup = 256 threads = 4 wave64 (render: 512).  Kernels (stream order, one launch each per batch):
//   downsample_kernel  Downsample1.main + Downsample2.main   (DS1:52-81, DS2:32-51)
//   render_kernel      Render.main_interleaved, all levels   (REN:112-177)
//   render_with_composite_kernel   the same, carrying the composite of an earlier call in its texel loop
//   render_wide_kernel Render.main on LowDepth<k> (opt-in hq_levels variant)
//   upsample_kernel    Upsample.main / main_blendout [/ main_premin*]   (UPS:185-233)
//   upsample_two_level_kernel / upsample_three_level_kernel   main_blendout L3->L2 (L2->L1) with the
//                      pass(es) below evaluated inside the same launch
//   upsample_final_with_next_downsample_kernel   Upsample.main of this batch + the downsample pass
//                      of the next one (meao_prefetch_batch): streaming hidden under VALU-bound work
// (DS1/DS2/REN/UPS = Assets/MiniEngineAO/Shaders/{Downsample1,Downsample2,Render,Upsample}.compute)
//
// Numerics contract: binary32, RNE, correctly rounded '/' (exact v_rcp_f32-based
// sequences or hipcc's IEEE expansion), compiled with -ffp-contract=off; the only fused
// operations are the explicit mad()/fma2() calls, placed where the HLSL source has a*b+c in one
// expression.  Bit-exact against oracle/meao_oracle.c and against the reference's own source
// executed by oracle/{csharp,hlsl}_interp.py.
//
// MI355X design notes:
//  * The 4x4 de-interleaved TiledDepth arrays are never materialised on the hot path: a
//    workgroup that renders ALL 16 slices of a 128x32 output tile needs exactly one contiguous
//    (128+32)x(32+32) window of LowDepth<level>, so the kernel stages that window in LDS
//    (applying the per-slice clamp addressing and the atlas padding rule while filling) and
//    samples it with a stride of 4.  Output rows are then contiguous instead of a 4-byte
//    strided scatter of single R8 texels.
//  * Each lane renders horizontally adjacent texel pairs so every LDS sample is one
//    conflict-free ds_read_b64; saturate() folds into the clamp modifier of v_mul/v_fma and
//    clamp(d, p, 1) is one v_med3_f32, so a sample pair costs exactly 8 VALU ops per texel.  The reads
//    of the next pair are issued by hand before the current pair is evaluated, and the per-term constants
//    are VGPR operands: an SGPR source halves the VALU issue rate on this part (tools/ubench_issue.hip).
//  * Upsample uses 64x64 hi-res tiles in the full-resolution pass (64x32 in the blend passes):
//    1.4x apron amplification instead of the reference's 2.6x, >= 89 % of the lanes busy in both
//    blur phases, 16-byte loads of the hi-res depth and 4-byte stores of four AO texels, LDS
//    carved so that seven workgroups share a CU.
//  * vmcnt retires loads in issue order: loads whose data is needed late are issued BEHIND the ones
//    needed first (window before hi-res operands), and unrelated streaming work (the next batch's
//    downsample tile, a carried composite) puts its loads in flight inside the tile, after the tile's
//    own loads have landed, through hooks of upsample_tile / render_tile.
//  * The small blend passes are evaluated inside the launch of the pass above them by recomputation
//    (blend_window_into_lds): no inter-workgroup synchronisation, bit-identical buffers.
//  * The SIMDs issue oldest-wave-first.  A hardware-dispatched workgroup is born youngest and ages while its tile
//    progresses, so the furthest-along tile always goes first -- a software pipeline across tiles for free, and the
//    reason persistent-workgroup forms of these kernels lost to the plain launches (round 3, LABNOTES.md).
//  * v_rcp_f32 pays ~3 cycles when it follows a non-transcendental instruction: the four weight reciprocals of a
//    bilateral texel are issued back to back (bilateral_upsample_grouped).
//  * Results that are stored as UNORM8 do not need the correction steps of their divisions wherever the uncorrected
//    quotient provably converts to the same code: bilateral_upsample_r8 checks that from the estimate itself (distance
//    of the scaled value from the next rounding boundary against a proven error bound) and redoes the rare texel exactly.
//    Not in the kernel that carries the next batch's downsample tile: it waits on memory, not on VALU issue.
#include "meao_kernels.hpp"

#include <algorithm>
#include <type_traits>

// Every design decision below that replaced an alternative was A/B-measured on one box; the arms that
// lost (or changed nothing) were removed in round 3 -- their logs stay in profiles/ (r02_ab_*.jsonl) and
// profiles/README.md lists them.  Experimental arms of the current round live behind the MEAO_X_* switches
// of this block only (tests/build_variants.py builds variants next to the product library;
// tests/test_variants_gpu.py runs a parity smoke through every variant library it finds).
#ifndef MEAO_X_UPS_EXACT_R8
#define MEAO_X_UPS_EXACT_R8 0      // 1 = every UNORM8 bilateral result through the full exact-division sequence (the round-2 form) instead of
#endif                             // bilateral_upsample_r8: L1->L0 176 -> 196 us, L2->L1 54 -> 57 us (profiles/r03_ab_verified_r8_bilateral.txt)
#ifndef MEAO_X_BIL_WHOLE_TILE
#define MEAO_X_BIL_WHOLE_TILE 1    // 0 = no separate copy of the bilateral phase for tiles that lie wholly inside the frame (the round-3 form):
#endif                             // last kernel 296 -> 272 us, step 570 -> 551 us (profiles/r04_ab_bilateral_arms.jsonl)
#ifndef MEAO_X_HOT_PATH_ONLY
#define MEAO_X_HOT_PATH_ONLY 0  // ANALYSIS builds only (tools/kernel_isa.py -DMEAO_X_HOT_PATH_ONLY=1 --stats; never a library): the upsample
#endif                          // and render kernels keep nothing but the path an interior tile of a clean frame takes, so that the
                                // static instruction counts of the ISA are the dynamic ones of (almost) every workgroup
#ifndef MEAO_X_PHASE_CLOCKS
#define MEAO_X_PHASE_CLOCKS 0   // diagnostic build: upsample tiles stamp s_memrealtime at their phase boundaries (tools/phase_clocks.py),
#endif                          // the render launch logs start / end / CU of every workgroup (tools/render_wg_log.py)

#if MEAO_X_PHASE_CLOCKS
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
// DIV_FAST (MEAO_NUMERICS_FAST, not bit-exact): the raw 1-ulp v_rcp_f32 without correction steps.
enum { DIV_EXACT_RCP = 0, DIV_IEEE = 1, DIV_FAST = 2 };

template <int DIV>
__device__ __forceinline__ float rcp_strict(float x)
{
    if constexpr (DIV == DIV_EXACT_RCP) {
        const float r = __builtin_amdgcn_rcpf(x);
        const float e = mad(-x, r, 1.0f);
        return mad(e, r, r);
    } else if constexpr (DIV == DIV_FAST) {
        return __builtin_amdgcn_rcpf(x);
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
    } else if constexpr (DIV == DIV_FAST) {
        return static_cast<float>(K) * __builtin_amdgcn_rcpf(x);
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
    } else if constexpr (DIV == DIV_FAST) {
        return a * __builtin_amdgcn_rcpf(b);
    } else {
        return a / b;
    }
}

// ------------------------------------------------------------------------------------------
// Downsample: linearize + point-downsample to L1..L4.   Tile 128 x 32 full-res texels.
//
// Closed form of DS1+DS2: LinearZ = lin(x,y); DS2x[i,j] = lin(2i,2j);
// DS4x = lin(4i,4j); DS8x = lin(8i,8j); DS16x = lin(16i,16j) -- every level keeps the
// top-left texel of its block, so a lane decides what to store from its own coordinates and
// no LDS exchange is needed.

// The pass is pure streaming: its loads and its two big stores are non-temporal, so that the lines
// do not displace what the upsample tiles sharing the kernel (meao_prefetch_batch) re-read from L2
// (A/B: 344 -> 339 us for the fused kernel, no change stand-alone).

template <int DIV>
__device__ __forceinline__ float linearize(float depth, float zp0, float zp1, float sky_depth)
{
    // ZBufferParams.x * d + ZBufferParams.y lies in [1, far/near] for every depth in [0, 1]
    const float dist = rcp_strict<DIV>(mad(zp0, depth, zp1));       // DS1:40
    // DS1:41-45: depth == 0 (reversed Z) / == 1 marks the far plane; sky_depth is that constant, so the
    // test is one v_cmp + v_cndmask per texel instead of a uniform branch on the Z convention
    return depth == sky_depth ? 1e5f : dist;
}

// "Nice" depth: the denominator of Linearize lies in [2^-20, 2^24], i.e. the linear depth is a
// normal number in [2^-24, 2^20] (non-zero after the f16 store, finite, not NaN).  Every exact
// v_rcp_f32 sequence downstream (centre depth, 1 / LoResDB, the bilateral weights, the final
// quotient) has its operands inside its verified range when all texels of a frame are nice.  A frame
// with any other texel -- NaN, +-inf, negative, > 1 with a conventional Z buffer, depths below
// 2^-24 -- is marked hostile by the downsample pass and takes the IEEE-division bodies of the later
// kernels (the reference divides with IEEE '/', Downsample1.compute:37-48; inputs are never sanitised).
__device__ __forceinline__ bool nice_denominator(float den)
{
    return __builtin_amdgcn_fmed3f(den, 0x1p-20f, 0x1p24f) == den;   // false for NaN
}

// First half of a downsample tile: the raw depth texels of this lane, 4 per row in each of the 4 row passes.
// F32_ONLY: the caller has established a.depth_format == MEAO_DEPTH_F32 (no format switch in the code).
// PASSES row passes of kDsRowsPerPass rows: 4 = the 32-row tile, 1 = the 8-row tile of small calls.
// CLAMP_ROWS (f32, 16-byte loads): rows past the frame re-read its last row instead of being skipped, so that every load
// is unconditional and the one wait for them sits in front of the finish loop, not inside its first row's branch (at the
// join behind that branch the compiler otherwise waits with vmcnt(0) for the first row's STORES as well).
template <bool VEC, bool F32_ONLY = false, int PASSES = kDsTileH / kDsRowsPerPass, bool CLAMP_ROWS = false>
__device__ __forceinline__ void downsample_tile_load(const DownsampleArgs &a, int tile, int frame,
                                                     float (&v)[PASSES][4])
{
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const void *__restrict__ depth = a.depth[frame];
    const int W = a.w[0], H = a.h[0];
    const int x0 = tile_x * kDsTileW + (threadIdx.x % kDsLanesPerRow) * 4;
    const int yb = tile_y * (PASSES * kDsRowsPerPass) + (threadIdx.x / kDsLanesPerRow);
    if (x0 >= W) return;

    // The depth-copy blit of the reference (Blit.shader pass 0) is folded into this load: the
    // texel format is decoded here (wave-uniform switch), 4 texels per lane per row.
    if constexpr (CLAMP_ROWS) {
        static_assert(VEC && F32_ONLY, "the clamped form is the 16-byte f32 one");
#pragma unroll
        for (int k = 0; k < PASSES; ++k) {
            const int y = min(yb + k * kDsRowsPerPass, H - 1);
            const float4v q = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(static_cast<const float *>(depth) + static_cast<size_t>(y) * W + x0));
            v[k][0] = q.x; v[k][1] = q.y; v[k][2] = q.z; v[k][3] = q.w;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
        const int y = yb + k * kDsRowsPerPass;
        v[k][0] = v[k][1] = v[k][2] = v[k][3] = 0.5f;
        if (y < H) {
            const size_t at = static_cast<size_t>(y) * W + x0;
            if (F32_ONLY || a.depth_format == MEAO_DEPTH_F32) {
                const float *row = static_cast<const float *>(depth) + at;
                if constexpr (VEC) {
                    const float4v q = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(row));
                    v[k][0] = q.x; v[k][1] = q.y; v[k][2] = q.z; v[k][3] = q.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (x0 + e < W) v[k][e] = row[e];
                }
            } else if (a.depth_format == MEAO_DEPTH_UNORM24) {
                const uint32_t *row = static_cast<const uint32_t *>(depth) + at;
                uint32_t u[4] = {0, 0, 0, 0};
                if constexpr (VEC) {
                    const uint4v q = *reinterpret_cast<const uint4v *>(row);
                    u[0] = q.x; u[1] = q.y; u[2] = q.z; u[3] = q.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (x0 + e < W) u[e] = row[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[k][e] = unorm_to_f32<24>(u[e] & 0xffffffu);
            } else {   // 16-bit texels: UNORM16 or F16
                const uint16_t *row = static_cast<const uint16_t *>(depth) + at;
                uint16_t u[4] = {0, 0, 0, 0};
                if constexpr (VEC) {
                    const ushort4v q = *reinterpret_cast<const ushort4v *>(row);
                    u[0] = q.x; u[1] = q.y; u[2] = q.z; u[3] = q.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (x0 + e < W) u[e] = row[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[k][e] = a.depth_format == MEAO_DEPTH_UNORM16 ? unorm_to_f32<16>(u[e]) : f16_bits_to_f32(u[e]);
            }
        }
    }
}

// Second half: linearize, store LinearZ and the four point-sampled levels.
template <bool RTNE, bool VEC, int DIV, int PASSES = kDsTileH / kDsRowsPerPass>
__device__ __forceinline__ void downsample_tile_finish(const DownsampleArgs &a, int tile, int frame,
                                                       const float (&v)[PASSES][4])
{
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    uint16_t *__restrict__ linear = frame_ptr(a.linear, a.frame_stride, frame);
    float *__restrict__ low1 = frame_ptr(a.low[0], a.frame_stride, frame);
    float *__restrict__ low2 = frame_ptr(a.low[1], a.frame_stride, frame);
    float *__restrict__ low3 = frame_ptr(a.low[2], a.frame_stride, frame);
    float *__restrict__ low4 = frame_ptr(a.low[3], a.frame_stride, frame);
    const int W = a.w[0], H = a.h[0];
    const float sky_depth = a.reversed_z != 0 ? 0.0f : 1.0f;
    const int x0 = tile_x * kDsTileW + (threadIdx.x % kDsLanesPerRow) * 4;
    const int yb = tile_y * (PASSES * kDsRowsPerPass) + (threadIdx.x / kDsLanesPerRow);
    if (x0 >= W) return;
    const float zp0 = a.zp0, zp1 = a.zp1;
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
        const int y = yb + k * kDsRowsPerPass;
        if (y >= H) continue;
        float lin[4];
        if constexpr (DIV == DIV_EXACT_RCP) {
            // the exact reciprocal sequence is only valid for a "nice" denominator; anything else
            // (hostile input) is divided with IEEE '/' and marks the frame for the later kernels
            bool nice = true;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                nice = nice && nice_denominator(mad(zp0, v[k][e], zp1));
                lin[e] = linearize<DIV_EXACT_RCP>(v[k][e], zp0, zp1, sky_depth);
            }
            if (__builtin_expect(!nice, 0)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) lin[e] = linearize<DIV_IEEE>(v[k][e], zp0, zp1, sky_depth);
                a.hostile[frame] = a.generation;     // racing stores of the same value
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) lin[e] = linearize<DIV>(v[k][e], zp0, zp1, sky_depth);
        }

        uint16_t *lrow = linear + static_cast<size_t>(y) * W + x0;    // LinearZ[st] = dist (DS1:46)
        if constexpr (VEC) {
            ushort4v h;
            h.x = f32_to_f16_bits<RTNE>(lin[0]); h.y = f32_to_f16_bits<RTNE>(lin[1]);
            h.z = f32_to_f16_bits<RTNE>(lin[2]); h.w = f32_to_f16_bits<RTNE>(lin[3]);
            __builtin_nontemporal_store(h, reinterpret_cast<ushort4v *>(lrow));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (x0 + e < W) lrow[e] = f32_to_f16_bits<RTNE>(lin[e]);
        }
        if ((y & 1) == 0) {                                           // DS2x (DS1:64-70)
            float *p = low1 + static_cast<size_t>(y >> 1) * a.w[1] + (x0 >> 1);
            if constexpr (VEC) {
                __builtin_nontemporal_store(float2v{lin[0], lin[2]}, reinterpret_cast<float2v *>(p));
            } else {
                p[0] = lin[0];
                if (x0 + 2 < W) p[1] = lin[2];
            }
            if ((y & 3) == 0) {                                       // DS4x (DS1:73-77)
                low2[static_cast<size_t>(y >> 2) * a.w[2] + (x0 >> 2)] = lin[0];
                if ((y & 7) == 0 && (x0 & 7) == 0) {                  // DS8x (DS2:35-40)
                    low3[static_cast<size_t>(y >> 3) * a.w[3] + (x0 >> 3)] = lin[0];
                    if ((y & 15) == 0 && (x0 & 15) == 0)              // DS16x (DS2:43-49)
                        low4[static_cast<size_t>(y >> 4) * a.w[4] + (x0 >> 4)] = lin[0];
                }
            }
        }
    }
}

template <bool RTNE, bool VEC, int DIV, int PASSES = kDsTileH / kDsRowsPerPass>
__device__ __forceinline__ void downsample_tile(const DownsampleArgs &a, int tile, int frame)
{
    float v[PASSES][4];
    downsample_tile_load<VEC, false, PASSES>(a, tile, frame, v);
    downsample_tile_finish<RTNE, VEC, DIV, PASSES>(a, tile, frame, v);
}

template <bool RTNE, bool VEC, int DIV>
__global__ __launch_bounds__(kThreads) void downsample_kernel(const DownsampleArgs a)
{
    downsample_tile<RTNE, VEC, DIV>(a, blockIdx.x, blockIdx.z);
}

// The pass as a CO-RUNNER of the VALU-bound launches (meao_debug_set MEAO_DEBUG_DS_SIDE_STREAM): its own kernel on a second,
// low-priority stream.  Two things differ from the stand-alone pass, which waits on memory and does not care:
//  * a co-resident workgroup gets its memory-level parallelism from a deep per-lane queue (PASSES 16-byte loads in flight:
//    a 128 x 8*PASSES tile) instead of from occupancy -- the launches it runs next to leave it one wave slot per SIMD;
//  * its VALU instructions are taken from kernels that are bound by VALU issue, so there are as few as possible: ~8 per texel
//    instead of ~18.  Rows are dealt to waves so that a row's parity is wave-uniform (wave w: rows w and w + 4 of every
//    8-row pass): the waves of odd rows skip the mip stores with a scalar branch, only wave 0 ever sees L2..L4; the range
//    test of the four denominators is two unsigned min / max chains on their bit patterns (negative values and NaNs are the
//    largest unsigned words) instead of four v_med3 + four compares; the far-plane select runs only where a lane holds a
//    far-plane texel; one 32-bit byte offset per buffer, advanced by a uniform stride per row pass (saddr addressing).
// Same bits as downsample_tile (tests/test_gpu_more.py::test_next_downsample_on_the_side_stream, hostile frames included).
// PAD_VGPRS: the kernel declares 120 VGPRs whatever it uses, so that exactly one of its workgroups fits next to seven
// upsample workgroups and the registers a finishing upsample workgroup frees (56) can only go to the next upsample one.
// lane geometry of the lean tile: rows w and w + 4 of every 8-row pass for wave w (a row's parity is wave-uniform)
struct LeanDsLane {
    int wave, row, y0;
    uint32_t x0;
    __device__ __forceinline__ LeanDsLane(const DownsampleArgs &a, int tile, int passes)
    {
        const uint32_t tid = threadIdx.x;
        wave = __builtin_amdgcn_readfirstlane(static_cast<int>(tid >> 6));
        const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
        x0 = static_cast<uint32_t>(tile_x) * kDsTileW + (tid & 31u) * 4u;
        row = wave + 4 * static_cast<int>((tid >> 5) & 1u);
        y0 = tile_y * (passes * kDsRowsPerPass) + row;
    }
};

// FULL: every row of the tile is inside the frame (otherwise rows past it re-read its last row and are never used)
template <int PASSES, bool FULL>
__device__ __forceinline__ void downsample_lean_load(const DownsampleArgs &a, int tile, int frame, float4v (&q)[PASSES])
{
    const LeanDsLane L(a, tile, PASSES);
    const uint32_t W = static_cast<uint32_t>(a.w[0]);
    if (L.x0 >= W) return;
    const float *__restrict__ depth = static_cast<const float *>(a.depth[frame]);
    const uint32_t t0 = static_cast<uint32_t>(L.y0) * W + L.x0, t_step = 8u * W;       // texel index of (x0, y0 + 8k) is t0 + k * 8W
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
        uint32_t t = t0 + static_cast<uint32_t>(k) * t_step;
        if constexpr (!FULL) t = static_cast<uint32_t>(min(L.y0 + 8 * k, a.h[0] - 1)) * W + L.x0;
        q[k] = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(at_byte_offset(depth, t * 4u)));
    }
}

template <bool RTNE, int DIV, int PASSES, bool FULL>
__device__ __forceinline__ void downsample_lean_finish(const DownsampleArgs &a, int tile, int frame, const float4v (&q)[PASSES])
{
    const LeanDsLane L(a, tile, PASSES);
    const int wave = L.wave, row = L.row, y0 = L.y0;
    const uint32_t x0 = L.x0, W = static_cast<uint32_t>(a.w[0]);
    const int H = a.h[0];
    if (x0 >= W) return;
    uint16_t *__restrict__ linear = frame_ptr(a.linear, a.frame_stride, frame);
    float *__restrict__ low1 = frame_ptr(a.low[0], a.frame_stride, frame);
    float *__restrict__ low2 = frame_ptr(a.low[1], a.frame_stride, frame);
    float *__restrict__ low3 = frame_ptr(a.low[2], a.frame_stride, frame);
    float *__restrict__ low4 = frame_ptr(a.low[3], a.frame_stride, frame);
    const float zp0 = a.zp0, zp1 = a.zp1;
    const float sky_depth = a.reversed_z != 0 ? 0.0f : 1.0f;
    const uint32_t w1 = a.w[1], w2 = a.w[2], w3 = a.w[3], w4 = a.w[4];
    const uint32_t t0 = static_cast<uint32_t>(y0) * W + x0, t_step = 8u * W;
    const uint32_t o1 = (static_cast<uint32_t>(y0 >> 1) * w1 + (x0 >> 1)) * 4u, o2 = (static_cast<uint32_t>(y0 >> 2) * w2 + (x0 >> 2)) * 4u;
    const uint32_t o3 = (static_cast<uint32_t>(y0 >> 3) * w3 + (x0 >> 3)) * 4u, o4 = (static_cast<uint32_t>(y0 >> 4) * w4 + (x0 >> 4)) * 4u;
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
        if constexpr (!FULL) { if (y0 + 8 * k >= H) break; }
        const float v[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
        float lin[4];
        if constexpr (DIV == DIV_EXACT_RCP) {
            float den[4];
            uint32_t bits[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { den[e] = mad(zp0, v[e], zp1); bits[e] = __builtin_bit_cast(uint32_t, den[e]); }
            // all four denominators in [2^-20, 2^24] (nice_denominator): as unsigned words, negative values and NaNs are the largest
            const uint32_t lo = min(min(min(bits[0], bits[1]), bits[2]), bits[3]), hi = max(max(max(bits[0], bits[1]), bits[2]), bits[3]);
            if (__builtin_expect(lo >= 0x35800000u && hi <= 0x4B800000u, 1)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r = __builtin_amdgcn_rcpf(den[e]);
                    lin[e] = mad(mad(-den[e], r, 1.0f), r, r);                  // rcp_strict: DS1:40
                }
                const bool far = (v[0] == sky_depth) | (v[1] == sky_depth) | (v[2] == sky_depth) | (v[3] == sky_depth);
                if (__builtin_expect(far, 0)) {                                 // DS1:41-45
#pragma unroll
                    for (int e = 0; e < 4; ++e) lin[e] = v[e] == sky_depth ? 1e5f : lin[e];
                    asm volatile("" : "+v"(lin[0]), "+v"(lin[1]), "+v"(lin[2]), "+v"(lin[3]));   // stays a branch: rare lanes only
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) lin[e] = linearize<DIV_IEEE>(v[e], zp0, zp1, sky_depth);
                a.hostile[frame] = a.generation;     // racing stores of the same value
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) lin[e] = linearize<DIV>(v[e], zp0, zp1, sky_depth);
        }
        const uint32_t t = t0 + static_cast<uint32_t>(k) * t_step;
        typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
        uint2v h;                                                                // LinearZ[st] = dist (DS1:46)
        if constexpr (RTNE) {
            h.x = f32_to_f16_bits<true>(lin[0]) | (static_cast<uint32_t>(f32_to_f16_bits<true>(lin[1])) << 16);
            h.y = f32_to_f16_bits<true>(lin[2]) | (static_cast<uint32_t>(f32_to_f16_bits<true>(lin[3])) << 16);
        } else {
            h.x = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(lin[0], lin[1]));
            h.y = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(lin[2], lin[3]));
        }
        __builtin_nontemporal_store(h, reinterpret_cast<uint2v *>(at_byte_offset(linear, t * 2u)));
        if ((wave & 1) == 0) {                                                   // even rows (wave-uniform): DS2x (DS1:64-70)
            __builtin_nontemporal_store(float2v{lin[0], lin[2]},
                                        reinterpret_cast<float2v *>(at_byte_offset(low1, o1 + static_cast<uint32_t>(k) * (16u * w1))));
            if (wave == 0) {                                                     // rows 0, 4 of the pass: DS4x (DS1:73-77)
                *at_byte_offset(low2, o2 + static_cast<uint32_t>(k) * (8u * w2)) = lin[0];
                if (row == 0 && (x0 & 7u) == 0) {                                // DS8x (DS2:35-40)
                    *at_byte_offset(low3, o3 + static_cast<uint32_t>(k) * (4u * w3)) = lin[0];
                    if ((k & 1) == 0 && (x0 & 15u) == 0)                         // DS16x (DS2:43-49)
                        *at_byte_offset(low4, o4 + static_cast<uint32_t>(k / 2) * (4u * w4)) = lin[0];
                }
            }
        }
    }
}

template <bool RTNE, int DIV, int PASSES, bool FULL>
__device__ __forceinline__ void downsample_side_tile(const DownsampleArgs &a, int tile, int frame)
{
    float4v q[PASSES];
    downsample_lean_load<PASSES, FULL>(a, tile, frame, q);
    downsample_lean_finish<RTNE, DIV, PASSES, FULL>(a, tile, frame, q);
}

template <bool RTNE, int DIV, int PASSES, bool PAD_VGPRS>
__global__ __launch_bounds__(kThreads) void downsample_side_kernel(const DownsampleArgs a)
{
    static_assert(PASSES % 2 == 0, "tile rows are a multiple of 16 (the L4 test uses the parity of the pass)");
    if constexpr (PAD_VGPRS) asm volatile("" ::: "v119");
    const int tile = blockIdx.x, frame = blockIdx.z;
    if ((tile / a.tiles_x + 1) * (PASSES * kDsRowsPerPass) <= a.h[0]) downsample_side_tile<RTNE, DIV, PASSES, true>(a, tile, frame);
    else downsample_side_tile<RTNE, DIV, PASSES, false>(a, tile, frame);
}


// Small calls (a 1080p frame: 510 tiles of 128 x 32): tiles of one row pass, four times the workgroups, one
// load-compute-store round each instead of four in a row.
template <bool RTNE, bool VEC, int DIV>
__global__ __launch_bounds__(kThreads) void downsample_small_kernel(const DownsampleArgs a)
{
    downsample_tile<RTNE, VEC, DIV, 1>(a, blockIdx.x, blockIdx.z);
}

// ------------------------------------------------------------------------------------------
// Render: volumetric-obscurance AO, 36-sample checker set.

// TestSamplePair (REN:60-75) for one output texel, from the two signed distances d = s * invRange - front.
// saturate() folds into the clamp output modifier of v_mul/v_fma; clamp(d, p, 1) with
// 0 <= p <= 1 is v_med3_f32(d, p, 1) (same value for every input incl. NaN d -> p).
__device__ __forceinline__ float pair_from_distances(float d1, float d2, float reject)
{
    const float p1 = sat(reject * d1);
    const float p2 = sat(reject * d2);
    const float acc = __builtin_amdgcn_fmed3f(d1, p2, 1.0f) + __builtin_amdgcn_fmed3f(d2, p1, 1.0f);
    return sat(mad(-p1, p2, acc));
}

// TestSamples (REN:77-110) WITHOUT its leading 0.5 / 0.25: that exact power-of-two factor is folded
// into the term's weight on the host (RenderLevelArgs::weight), since fma(w, k*S, ao) and
// fma(k*w, S, ao) round the same real number.  (X, Y) are sample offsets in source texels; the LDS
// offset of (dx, dy) is dy*P + dx*Q.  Interleaved: one slice texel is 4 level texels (4x4 interleave),
// P = 4*pitch, Q = 4.  Wide (REN:79-82, x <<= 1): P = 2*pitch, Q = 2.
// Two horizontally adjacent texels share every LDS address: one 8-byte LDS read per sample.
//
// (A wave-uniform "all distances >= 0 => pair = saturate(d1 + d2)" fast path was built, is bit-exact and was
// measured slower on both headline workloads -- DESIGN.md 5.2, profiles/r02_render_fastpath_hitrates.txt; removed.)
template <int X, int Y, int P, int Q>
__device__ __forceinline__ float2v test_samples(const float *centre, float2v inv_depth, float inv_thickness,
                                                float front_depth, float reject)
{
    constexpr int N = (Y == 0 || X == Y) ? 2 : 4;
    constexpr int off[4] = {Y == 0 ? X * Q : (X == Y ? X * P - X * Q : Y * P + X * Q),
                            Y == 0 ? X * P : (X == Y ? X * P + X * Q : Y * P - X * Q),
                            X * P + Y * Q, X * P - Y * Q};
    const float2v inv_range = splat(inv_thickness) * inv_depth;
    const float neg_front = -front_depth;
    float2v d1[N], d2[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float2v s1 = *reinterpret_cast<const float2v *>(centre + off[i]);
        const float2v s2 = *reinterpret_cast<const float2v *>(centre - off[i]);
        d1[i] = float2v{mad(s1.x, inv_range.x, neg_front), mad(s1.y, inv_range.y, neg_front)};
        d2[i] = float2v{mad(s2.x, inv_range.x, neg_front), mad(s2.y, inv_range.y, neg_front)};
    }
    float2v r[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
        r[i] = float2v{pair_from_distances(d1[i].x, d2[i].x, reject), pair_from_distances(d1[i].y, d2[i].y, reject)};
    if constexpr (N == 2) return r[0] + r[1];
    else return ((r[0] + r[1]) + r[2]) + r[3];
}

// ao = sum over the terms of weight * TestSamples, in the reference's accumulation order:
// checker set REN:162-168 (slots 1,3,4,8,11,6,10), SAMPLE_EXHAUSTIVELY REN:146-157
// (slots 0,1,2,3,4,8,11,5,6,7,9,10).  L.weight[] etc. are already in term order; L.weight[] carries
// the 0.5 (axial, diagonal) / 0.25 (L-shaped) factor of TestSamples.
// The per-term constants of one level, held in SGPRs for the whole tile.  (Read straight from the
// kernel-argument struct the compiler re-issued the s_load_dword's inside the texel loop, three per
// term, and their s_waitcnt lgkmcnt(0) also drained the LDS reads in flight.)
template <bool EXH>
struct TermConstants {
    static constexpr int kTerms = EXH ? 12 : 7;
    float inv_thickness[kTerms], front_depth[kTerms], weight[kTerms];
    float reject_fadeoff, intensity;
    // All scalar loads first, then ONE statement that pins the values (a volatile asm per term made the compiler wait for each
    // term's loads before it issued the next ones: eight dependent scalar-memory round trips per workgroup, right behind the
    // window barrier).  render_tile calls this once its window loads are in flight, so the scalar loads' latency hides behind theirs.
    __device__ __forceinline__ TermConstants() {}
    __device__ __forceinline__ explicit TermConstants(const RenderLevelArgs &src) { load(src); }
    __device__ __forceinline__ void load(const RenderLevelArgs &src)
    {
#pragma unroll
        for (int t = 0; t < kTerms; ++t) {
            inv_thickness[t] = src.inv_thickness[t];
            front_depth[t] = src.front_depth[t];
            weight[t] = src.weight[t];
        }
        reject_fadeoff = src.reject_fadeoff;
        intensity = src.intensity;
#define MEAO_PIN3(T) "+s"(inv_thickness[T]), "+s"(front_depth[T]), "+s"(weight[T])
        asm volatile("" : MEAO_PIN3(0), MEAO_PIN3(1), MEAO_PIN3(2), MEAO_PIN3(3), MEAO_PIN3(4), MEAO_PIN3(5), MEAO_PIN3(6),
                          "+s"(reject_fadeoff), "+s"(intensity));                       // stay in SGPRs
        if constexpr (EXH) asm volatile("" : MEAO_PIN3(7), MEAO_PIN3(8), MEAO_PIN3(9), MEAO_PIN3(10), MEAO_PIN3(11));
#undef MEAO_PIN3
    }
};

template <bool EXH, int P, int Q>
__device__ __forceinline__ float2v accumulate_terms(const TermConstants<EXH> &L, const float *centre, float2v inv_depth)
{
    const float reject = L.reject_fadeoff;
    float2v ao = splat(0.0f);
#define MEAO_TERM(N, X, Y) \
    ao = fma2(splat(L.weight[N]), test_samples<X, Y, P, Q>(centre, inv_depth, L.inv_thickness[N], L.front_depth[N], reject), ao)
    if constexpr (EXH) {
        MEAO_TERM(0, 1, 0); MEAO_TERM(1, 2, 0); MEAO_TERM(2, 3, 0); MEAO_TERM(3, 4, 0);
        MEAO_TERM(4, 1, 1); MEAO_TERM(5, 2, 2); MEAO_TERM(6, 3, 3); MEAO_TERM(7, 1, 2);
        MEAO_TERM(8, 1, 3); MEAO_TERM(9, 1, 4); MEAO_TERM(10, 2, 3); MEAO_TERM(11, 2, 4);
    } else {
        MEAO_TERM(0, 2, 0); MEAO_TERM(1, 4, 0); MEAO_TERM(2, 1, 1); MEAO_TERM(3, 2, 2);
        MEAO_TERM(4, 3, 3); MEAO_TERM(5, 1, 3); MEAO_TERM(6, 2, 4);
    }
#undef MEAO_TERM
    return fma2(splat(L.intensity), ao - splat(1.0f), splat(1.0f));   // lerp(1, ao, gIntensity) REN:176
}

// ---- the same sum with the LDS reads pipelined by hand (checker set) --------------------------
// clang issues the ds_read's of a term right before their first use (s_waitcnt a few instructions
// later): every wave exposes the LDS latency 12+ times per texel pair.  Here the 18 sample pairs of
// the checker set are one flat sequence; the two 8-byte reads of pair k + DEPTH are issued before pair k
// is evaluated, as separate ds_read_b64 (the merged ds_read2_b64 form runs at half the LDS rate,
// tools/ubench_lds.hip).  The reads are inline asm, so the waits are too: LDS operations return in
// order, `s_waitcnt lgkmcnt(2 * DEPTH)` therefore means "pair k has arrived" whatever else is in flight
// behind it.  Arithmetic and its order are those of test_samples / accumulate_terms.
struct SamplePair { float2v s1, s2; };

constexpr int kCheckerTerms[7][2] = {{2, 0}, {4, 0}, {1, 1}, {2, 2}, {3, 3}, {1, 3}, {2, 4}};   // REN:162-168
constexpr int kCheckerPairs = 18;

constexpr int checker_pairs_in_term(int t) { return (kCheckerTerms[t][1] == 0 || kCheckerTerms[t][0] == kCheckerTerms[t][1]) ? 2 : 4; }
constexpr int checker_term_of_pair(int k)
{
    int t = 0;
    while (k >= checker_pairs_in_term(t)) { k -= checker_pairs_in_term(t); ++t; }
    return t;
}
constexpr int checker_index_in_term(int k)
{
    int t = 0;
    while (k >= checker_pairs_in_term(t)) { k -= checker_pairs_in_term(t); ++t; }
    return k;
}
// LDS offset (floats) of the first sample of pair i of term (X, Y); the second one is at minus that
constexpr int checker_pair_offset(int X, int Y, int P, int Q, int i)
{
    return i == 0 ? (Y == 0 ? X * Q : (X == Y ? X * P - X * Q : Y * P + X * Q))
         : i == 1 ? (Y == 0 ? X * P : (X == Y ? X * P + X * Q : Y * P - X * Q))
         : i == 2 ? X * P + Y * Q : X * P - Y * Q;
}

template <int BYTE_OFF>
__device__ __forceinline__ void lds_read_b64_async(uint32_t lds_addr, float2v &v)
{
    static_assert(BYTE_OFF >= 0 && BYTE_OFF < 65536 && BYTE_OFF % 8 == 0, "ds_read_b64 immediate offset");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(BYTE_OFF));
}

template <int K, int P, int Q>
__device__ __forceinline__ void issue_checker_pair(uint32_t base, SamplePair &into)
{
    if constexpr (K < kCheckerPairs) {
        constexpr int t = checker_term_of_pair(K);
        constexpr int off = checker_pair_offset(kCheckerTerms[t][0], kCheckerTerms[t][1], P, Q, checker_index_in_term(K));
        constexpr int centre_at = 4 * P + 4 * Q;                 // `base` is that many floats before the centre texel
        lds_read_b64_async<(centre_at + off) * 4>(base, into.s1);
        lds_read_b64_async<(centre_at - off) * 4>(base, into.s2);
    }
}

// Waits until at most PENDING LDS reads are outstanding.  The operands tie the wait into the data flow:
// the arrived pair is only readable after it, and the running sums (= the previous pair's arithmetic)
// are complete before it, so the schedule keeps one pair's arithmetic between two waits.
template <int PENDING>
__device__ __forceinline__ void wait_checker_pair(SamplePair &arrived, float2v &term_sum, float2v &ao)
{
    static_assert(PENDING == 0 || PENDING == 2 || PENDING == 4, "");
    if constexpr (PENDING == 0)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(arrived.s1), "+v"(arrived.s2), "+v"(term_sum), "+v"(ao));
    else if constexpr (PENDING == 2)
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(arrived.s1), "+v"(arrived.s2), "+v"(term_sum), "+v"(ao));
    else
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(arrived.s1), "+v"(arrived.s2), "+v"(term_sum), "+v"(ao));
}

template <int K, int P, int Q, int DEPTH>
__device__ __forceinline__ void pipelined_checker_step(const TermConstants<false> &L, uint32_t base, float2v inv_depth, float reject,
                                                       SamplePair (&ring)[DEPTH + 1], float2v &inv_range, float &neg_front,
                                                       float2v &term_sum, float2v &ao)
{
    if constexpr (K < kCheckerPairs) {
        constexpr int t = checker_term_of_pair(K), i = checker_index_in_term(K), n = checker_pairs_in_term(t);
        issue_checker_pair<K + DEPTH, P, Q>(base, ring[(K + DEPTH) % (DEPTH + 1)]);
        constexpr int behind = (K + DEPTH < kCheckerPairs ? DEPTH : kCheckerPairs - 1 - K);    // pairs issued after pair K
        SamplePair &s = ring[K % (DEPTH + 1)];
        wait_checker_pair<2 * behind>(s, term_sum, ao);
        if constexpr (i == 0) {
            inv_range = splat(L.inv_thickness[t]) * inv_depth;
            neg_front = -L.front_depth[t];
            asm volatile("" : "+v"(neg_front));      // VGPR operand: an SGPR source halves the VALU issue rate (tools/ubench_issue.hip)
        }
        const float2v d1 = float2v{mad(s.s1.x, inv_range.x, neg_front), mad(s.s1.y, inv_range.y, neg_front)};
        const float2v d2 = float2v{mad(s.s2.x, inv_range.x, neg_front), mad(s.s2.y, inv_range.y, neg_front)};
        const float2v r = float2v{pair_from_distances(d1.x, d2.x, reject), pair_from_distances(d1.y, d2.y, reject)};
        if constexpr (i == 0) term_sum = r;
        else term_sum = term_sum + r;                                       // (r0 + r1) (+ r2) (+ r3), REN:92-109
        if constexpr (i == n - 1) ao = fma2(splat(L.weight[t]), term_sum, ao);
        pipelined_checker_step<K + 1, P, Q, DEPTH>(L, base, inv_depth, reject, ring, inv_range, neg_front, term_sum, ao);
    }
}

template <int P, int Q, int DEPTH>
__device__ __forceinline__ float2v accumulate_terms_pipelined(const TermConstants<false> &L, const float *centre, float2v inv_depth)
{
    typedef __attribute__((address_space(3))) const float lds_float;
    const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_float *)(centre - (4 * P + 4 * Q))));
    SamplePair ring[DEPTH + 1];
    float2v ao = splat(0.0f), term_sum = splat(0.0f), inv_range = splat(0.0f);
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
        if (k == 0) issue_checker_pair<0, P, Q>(base, ring[0]);
        if (k == 1) issue_checker_pair<1, P, Q>(base, ring[1]);
    }
    float reject = L.reject_fadeoff, neg_front = 0.0f;
    asm volatile("" : "+v"(reject));
    pipelined_checker_step<0, P, Q, DEPTH>(L, base, inv_depth, reject, ring, inv_range, neg_front, term_sum, ao);
    return fma2(splat(L.intensity), ao - splat(1.0f), splat(1.0f));   // lerp(1, ao, gIntensity) REN:176
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

// Hook of the texel loop: begin(k) / end(k) are executed by every thread of the workgroup around iteration k
// (render_with_composite_kernel puts the loads of unrelated streaming work in flight under the arithmetic).
struct NoRenderHook {
    __device__ __forceinline__ void begin(int) {}
    __device__ __forceinline__ void end(int) {}
};

template <int AOFMT, bool RTNE, int DIV, bool EXH, typename Hook = NoRenderHook, int TILE_H = kRenTileH, int THREADS = ren_tile_w(EXH) * 4>
__device__ __forceinline__ void render_tile(const RenderArgs &a, float *tile, int frame, int block, Hook hook = Hook())
{
    typedef AoTexel<AOFMT> AO;
    constexpr int kRenTileW = ren_tile_w(EXH), kRenThreads = THREADS, kRenLdsW = kRenTileW + 2 * kRenApron;
    constexpr int kRenTileH = TILE_H, kRenLdsH = TILE_H + 2 * kRenApron;      // shadow the 32-row constants

    int b = block, lv = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < a.num_levels && b >= a.level[k].block_begin) lv = k;
    const RenderLevelArgs &L = a.level[lv];
    b -= L.block_begin;
    const int X0 = (b % L.tiles_x) * kRenTileW, Y0 = (b / L.tiles_x) * kRenTileH;
    const int lw = L.lw, lh = L.lh;
    const float *__restrict__ src = frame_ptr(L.src, a.frame_stride, frame);

    PhaseClock clk(24);      // 24: window loaded, converted, in LDS; 25: barrier; 26..29: texel-loop iterations
    __builtin_amdgcn_s_setprio(3);
    TermConstants<EXH> terms_storage;
    const TermConstants<EXH> &terms = terms_storage;
    typename AO::type *__restrict__ dst;
    // ---- stage the (64+32) x (32+32) window.  Window texel (vx,vy) (level coordinates, may
    // be outside the level) belongs to slice (vx&3, vy&3), slice texel (vx>>2, vy>>2); the
    // reference clamps the slice texel per slice (REN:118 Gather + clamp sampler) and finds
    // Linearize(out-of-range) / 0 in atlas texels beyond the level (DS1:39-46, DS2:35).
    {
        const float pad = through_f16<RTNE>(L.pad_value);
        const bool vec_ok = (lw & 3) == 0;
        constexpr int kQuadsX = kRenLdsW / 4, kQuads = kQuadsX * kRenLdsH, kRounds = (kQuads + kRenThreads - 1) / kRenThreads;
        constexpr bool kEven = kQuads % kRenThreads == 0;            // every thread fills the same number of quads (32-row tiles)
        // Phase 1: all 16-byte loads of this thread's quads are issued back to back (the plain loop
        // waited for each load before issuing the next: five dependent memory latencies per tile);
        // quads that touch the level's border take the scalar path in phase 2.
        float4v raw[kRounds];
        int row_at[kRounds];      // index of the first texel of the quad's row segment, or -1 = all padding
        bool whole[kRounds];
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int q = threadIdx.x + r * kRenThreads;
            const int qx = q % kQuadsX, qy = q / kQuadsX;
            const int px0 = clampi((X0 >> 2) - (kRenApron >> 2) + qx, 0, L.sw - 1) * 4;
            const int vy = Y0 - kRenApron + qy;
            const int py = clampi(vy >> 2, 0, L.sh - 1) * 4 + (vy & 3);
            const bool mine = kEven || q < kQuads;              // the last round of the 8-row tile is partly empty
            row_at[r] = (mine && py < lh) ? py * lw + px0 : -1;
            whole[r] = mine && py < lh && vec_ok && px0 + 3 < lw;
            if (whole[r]) raw[r] = *reinterpret_cast<const float4v *>(src + row_at[r]);
        }
        // (the texel loop's constants: fetched while the window loads are in flight, see TermConstants)
        terms_storage.load(L);
        dst = frame_ptr(static_cast<typename AO::type *>(L.dst), a.frame_stride, frame);
        asm volatile("" : "+s"(dst));
        // Phase 2: the f16 round trip the atlas store applies, then one 16-byte LDS store per quad
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int q = threadIdx.x + r * kRenThreads;
            const int qx = q % kQuadsX, qy = q / kQuadsX;
            float4v t = {pad, pad, pad, pad};
            if (whole[r]) {
                const float2v lo = through_f16_pair<RTNE>(raw[r].x, raw[r].y), hi = through_f16_pair<RTNE>(raw[r].z, raw[r].w);
                t = float4v{lo.x, lo.y, hi.x, hi.y};
            } else if (row_at[r] >= 0) {
                const float *row = src + row_at[r];
                const int px0 = row_at[r] % lw;
                if (px0 + 0 < lw) t.x = through_f16<RTNE>(row[0]);
                if (px0 + 1 < lw) t.y = through_f16<RTNE>(row[1]);
                if (px0 + 2 < lw) t.z = through_f16<RTNE>(row[2]);
                if (px0 + 3 < lw) t.w = through_f16<RTNE>(row[3]);
            }
            if (kEven || q < kQuads) *reinterpret_cast<float4v *>(&tile[qy * kRenLdsW + qx * 4]) = t;
        }
    }
    clk.mark(0);
    __syncthreads();
    clk.mark(1);
    __builtin_amdgcn_s_setprio(0);

    // ---- each lane: a texel pair (X, X+1) in each of the TILE_H / 8 iterations
    const bool pair_store = ((lw & 1) == 0);
    // a wave covers a compact 32 x 4 block (16 lanes x 4 rows) of the tile in each of the 4 iterations
    constexpr int kBlocksX = kRenTileW / 32, kWaves = kRenThreads / 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

    constexpr int kIterations = kBlocksX * (kRenTileH / 4) / kWaves;        // 32 x 4 blocks of the tile per wave
    static_assert(kIterations * kWaves == kBlocksX * (kRenTileH / 4), "the tile's blocks divide evenly among the waves");
#pragma unroll 1
    for (int k = 0; k < kIterations; ++k) {
        const int blk = k * kWaves + wave;
        const int txl = (blk % kBlocksX) * 16 + (lane & 15), ly = (blk / kBlocksX) * 4 + (lane >> 4);
        const int X = X0 + 2 * txl, Y = Y0 + ly;
        hook.begin(k);
        if (X < lw && Y < lh) {
            const float *centre = &tile[(ly + kRenApron) * kRenLdsW + 2 * txl + kRenApron];
            const float2v c = *reinterpret_cast<const float2v *>(centre);
            const float2v inv_depth = float2v{rcp_strict<DIV>(c.x), rcp_strict<DIV>(c.y)};   // REN:140
            float2v out;     // one pair in flight ahead of the one evaluated; a second one changed nothing (r02 A/B)
            if constexpr (!EXH) out = accumulate_terms_pipelined<4 * kRenLdsW, 4, 1>(terms, centre, inv_depth);
            else out = accumulate_terms<EXH, 4 * kRenLdsW, 4>(terms, centre, inv_depth);

            typename AO::type *p = dst + static_cast<size_t>(Y) * lw + X;
            const typename AO::type e0 = AO::template encode<RTNE>(out.x), e1 = AO::template encode<RTNE>(out.y);
            if (pair_store) {
                typename AO::type2 pr; pr.x = e0; pr.y = e1;
                *reinterpret_cast<typename AO::type2 *>(p) = pr;
            } else {
                p[0] = e0;
                if (X + 1 < lw) p[1] = e1;
            }
        }
        hook.end(k);
        clk.mark(2 + (k & 3));
    }
}

// 128 x 32 tiles: 40 KB window, 4 workgroups of 8 waves per CU = 8 waves per SIMD (<= 64 VGPRs).
template <int AOFMT, bool RTNE, int DIV, bool EXH>
__global__ __launch_bounds__(ren_tile_w(EXH) * 4, EXH ? 1 : 8) void render_kernel(const RenderArgs a)
{
    __shared__ __attribute__((aligned(16))) float tile[kRenLdsH * (ren_tile_w(EXH) + 2 * kRenApron)];
#if MEAO_X_PHASE_CLOCKS
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();      // every wave: the first one in and the last one out are logged
#endif
    const int frame = blockIdx.y, block = xcd_contiguous(blockIdx.x, gridDim.x);
    if constexpr (DIV == DIV_EXACT_RCP) {
        if (frame_is_hostile(a.hostile, a.generation, frame)) {       // wave-uniform, decided per frame
            render_tile<AOFMT, RTNE, DIV_IEEE, EXH>(a, tile, frame, block);
            return;
        }
    }
    render_tile<AOFMT, RTNE, DIV, EXH>(a, tile, frame, block);
#if MEAO_X_PHASE_CLOCKS
    const unsigned id = blockIdx.y * gridDim.x + blockIdx.x;
    if ((threadIdx.x & 63) == 0 && id < 16384) {
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        atomicMin(&g_wg_log[id * 4 + 0], wg_t0);           // earliest wave start (the host presets ~0)
        atomicMax(&g_wg_log[id * 4 + 1], t1);              // latest wave end
        atomicMin(&g_wg_log[id * 4 + 2], t1);              // earliest wave end: the skew inside the workgroup
        if (threadIdx.x == 0)
            g_wg_log[id * 4 + 3] = (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) & 0xFu) |
                                   (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11))) << 8);
    }
#endif
}

// (128 x 64 tiles with 1024 threads -- a 60 KB window, two workgroups = 32 waves per CU, apron share 1.875x instead of 2.5x, half
// the hand-overs per texel -- measured 170.2 vs 166.2 us per 16 frames at 4K, profiles/r04_ab_render_tile_128x64.jsonl: the
// barrier of sixteen waves and a hand-over that idles half a CU cost more than the smaller apron saves.  With 96 x 32 (r03),
// 64 x 32 (r01) and the dynamic blocks (r03) that closes the tile-shape question: render runs at 2.98 cycles per VALU
// instruction, the hand-over of a full CU's LDS is what separates it from the 2.4-2.55 of its loop, and no shape removes it.
// Nor does taking the hand-over away: persistent 1024-thread workgroups whose four loader waves fill the next tile's window while
// twelve compute waves evaluate the current one (two window buffers, one barrier per tile) run at 209-220 us -- the barrier of
// sixteen waves per tile costs more than the hand-over did: profiles/r04_ab_render_producer_consumer.jsonl.)
// One or two small frames per call (fewer 128 x 32 tiles than CUs): 128 x 8 tiles, four times the workgroups,
// one texel-loop iteration each -- the call waits for one workgroup's serial time, not for throughput.
template <int AOFMT, bool RTNE, int DIV>
__global__ __launch_bounds__(ren_tile_w(false) * 4, 6) void render_small_kernel(const RenderArgs a)
{
    __shared__ __attribute__((aligned(16))) float tile[(kRenTileHSmall + 2 * kRenApron) * (ren_tile_w(false) + 2 * kRenApron)];
    const int frame = blockIdx.y, block = xcd_contiguous(blockIdx.x, gridDim.x);
    if constexpr (DIV == DIV_EXACT_RCP) {
        if (frame_is_hostile(a.hostile, a.generation, frame)) {
            render_tile<AOFMT, RTNE, DIV_IEEE, false, NoRenderHook, kRenTileHSmall>(a, tile, frame, block);
            return;
        }
    }
    render_tile<AOFMT, RTNE, DIV, false, NoRenderHook, kRenTileHSmall>(a, tile, frame, block);
}

// ------------------------------------------------------------------------------------------
// Render.main (WIDE_SAMPLING, REN:22,27-29,46-50): the same estimator on the NON-tiled f32
// LowDepth<level>, sampling every other texel (offsets doubled, REN:79-82) out to 8 texels, one
// output texel per source texel (REN:174).  The reference's host never dispatches it; it is
// the "high quality" pass of the MiniEngine original and feeds Upsample.main_premin*.
// Tile 64 x 32 outputs, LDS window (64+16) x (32+16) of raw f32 depth with clamp addressing
// (REN:116,121 Gather on the 2D texture); no f16 round trip, no padding texels.
template <int AOFMT, bool RTNE, int DIV, bool EXH>
__device__ __forceinline__ void render_wide_tile(const RenderArgs &a, float *tile, int frame, int block)
{
    typedef AoTexel<AOFMT> AO;

    int b = block, lv = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < a.num_levels && b >= a.level[k].block_begin) lv = k;
    const RenderLevelArgs &L = a.level[lv];
    b -= L.block_begin;
    const int X0 = (b % L.tiles_x) * kWideTileW, Y0 = (b / L.tiles_x) * kRenTileH;
    const int lw = L.lw, lh = L.lh;
    const float *__restrict__ src = frame_ptr(L.src, a.frame_stride, frame);

    for (int i = threadIdx.x; i < kWideLdsW * kWideLdsH; i += kThreads) {
        const int c = i % kWideLdsW, r = i / kWideLdsW;
        const int x = clampi(X0 - kWideApron + c, 0, lw - 1), y = clampi(Y0 - kWideApron + r, 0, lh - 1);
        tile[i] = src[static_cast<size_t>(y) * lw + x];
    }
    __syncthreads();

    const int txl = threadIdx.x & 31, tyl = threadIdx.x >> 5;
    const int X = X0 + 2 * txl;
    if (X >= lw) return;
    typename AO::type *__restrict__ dst = frame_ptr(static_cast<typename AO::type *>(L.dst), a.frame_stride, frame);
    const bool pair_store = ((lw & 1) == 0);
    const TermConstants<EXH> terms(L);

#pragma unroll 1
    for (int k = 0; k < kRenTileH / 8; ++k) {
        const int ly = tyl + 8 * k, Y = Y0 + ly;
        if (Y >= lh) break;
        const float *centre = &tile[(ly + kWideApron) * kWideLdsW + 2 * txl + kWideApron];
        const float2v c = *reinterpret_cast<const float2v *>(centre);
        const float2v inv_depth = float2v{rcp_strict<DIV>(c.x), rcp_strict<DIV>(c.y)};   // REN:140
        const float2v out = accumulate_terms<EXH, 2 * kWideLdsW, 2>(terms, centre, inv_depth);

        typename AO::type *p = dst + static_cast<size_t>(Y) * lw + X;
        const typename AO::type e0 = AO::template encode<RTNE>(out.x), e1 = AO::template encode<RTNE>(out.y);
        if (pair_store) {
            typename AO::type2 pr; pr.x = e0; pr.y = e1;
            *reinterpret_cast<typename AO::type2 *>(p) = pr;
        } else {
            p[0] = e0;
            if (X + 1 < lw) p[1] = e1;
        }
    }
}

template <int AOFMT, bool RTNE, int DIV, bool EXH>
__global__ __launch_bounds__(kThreads) void render_wide_kernel(const RenderArgs a)
{
    __shared__ __attribute__((aligned(16))) float tile[kWideLdsH * kWideLdsW];
    const int frame = blockIdx.y, block = xcd_contiguous(blockIdx.x, gridDim.x);
    if constexpr (DIV == DIV_EXACT_RCP) {
        if (frame_is_hostile(a.hostile, a.generation, frame)) {
            render_wide_tile<AOFMT, RTNE, DIV_IEEE, EXH>(a, tile, frame, block);
            return;
        }
    }
    render_wide_tile<AOFMT, RTNE, DIV, EXH>(a, tile, frame, block);
}

// ------------------------------------------------------------------------------------------
// Upsample: depth-aware 5-tap separable blur of the low-res AO + bilateral 2x upsample.

template <int TILE_H>
struct UpsTile {
    static constexpr int kLowW = kUpsTileW / 2, kLowH = TILE_H / 2;   // low-res texels under the tile: 32 x 16|32
    static constexpr int kRawW = kLowW + 6, kRawH = kLowH + 6;       // raw taps: 38 x 22|38
    static constexpr int kRawPitch = 40;
    static constexpr int kBlurW = kLowW + 2, kBlurH = kLowH + 2;     // blurred texels: 34 x 18|34
    static constexpr int kBlurPitch = 36;
    // Run lengths are chosen so that each blur phase is ONE round over the 256 lanes (the phases are
    // latency-bound: a second, partly filled round costs a full LDS round trip): 64-row tiles use
    // 6 x 38 = 228 horizontal runs of 6 and 7 x 34 = 238 vertical runs of 5 (runs of 4 / 4: 342 and 306
    // items, two rounds each); 32-row tiles 9 x 22 = 198 runs of 4 and 6 x 34 = 204 runs of 3.
    static constexpr bool kLong = TILE_H == 64;          // A/B: -2.2 % on the full-resolution pass (220 -> 215 us per 16 frames)
    static constexpr int kHRun = kLong ? 6 : 4, kHSegs = (kBlurW + kHRun - 1) / kHRun;
    static constexpr int kVRun = kLong ? 5 : ((kBlurH % 3 == 0) ? 3 : 4);
    static constexpr int kVSegs = (kBlurH + kVRun - 1) / kVRun;
    // V-blur runs of the last segment may read (and produce) rows past the window: allocate them
    static constexpr int kVRows = kVSegs * kVRun;                                 // rows of s_vb
    static constexpr int kRawRows = (kVRows + 4 > kRawH) ? kVRows + 4 : kRawH;    // rows of s_ao / s_inv / s_hb
    static_assert(kUpsTileW == 64 && TILE_H % 32 == 0, "bilateral phase: 16 x 16 lanes of 4 x 2 texels per pass");
    static_assert(kHSegs * kHRun + 4 <= kRawPitch, "H-blur runs may read into the row padding only");
    static_assert(kVRows * kBlurPitch <= kRawRows * kRawPitch, "s_vb aliases s_ao");
};

struct BlurConsts { float step_size, blur_tolerance; };

// A run of N consecutive outputs of BlurHorizontally / BlurVertically (UPS:89-170) from N+4 AO
// taps a[] and inverse depths z[]: output n is centred on tap n+2.  Deltas, squared lengths and
// CompareDeltas results are shared between neighbouring outputs exactly as the reference
// shares them between the 3 (2) outputs of one lane; every output only depends on its own
// 5-tap window.  CompareDeltas UPS:83-87, SmartBlur UPS:74-81.
template <int N>
__device__ __forceinline__ void blur_run(const BlurConsts &k, const float (&a)[N + 4], const float (&z)[N + 4],
                                         float (&out)[N])
{
    float dz[N + 3], ln[N + 3];
    bool keep[N + 2];
#pragma unroll
    for (int i = 0; i < N + 3; ++i) {
        dz[i] = z[i + 1] - z[i];
        ln[i] = mad(dz[i], dz[i], k.step_size);
    }
#pragma unroll
    for (int i = 0; i < N + 2; ++i) {
        const float t = mad(dz[i], dz[i + 1], k.step_size);
        keep[i] = t * t > (ln[i] * ln[i + 1]) * k.blur_tolerance;
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const bool left = keep[n], middle = keep[n + 1], right = keep[n + 2];
        const float pc = a[n + 2];
        const float pb = (left | middle) ? a[n + 1] : pc;
        const float pa = left ? a[n] : pb;
        const float pd = (right | middle) ? a[n + 3] : pc;
        const float pe = right ? a[n + 4] : pd;
        // (pa + pe) * 0.5 is exact (power of two, operands are AO values far from underflow), so
        // fusing it into the following add rounds exactly like the reference's mul-then-add
        out[n] = ((mad(pa + pe, 0.5f, pb) + pc) + pd) * 0.25f;
    }
}

// BilateralUpsample (UPS:177-183); taps already in weight order 9,3,1,3.
// The uniform operands of the bilateral phase (SGPRs / literals; pinning them in VGPRs changed nothing here:
// this phase waits on latency, not on VALU issue, profiles/r02_ab_v14*_ups_vgpr_consts.jsonl).
struct BilateralConsts {
    float tolerance, noise, three, nine;
    __device__ __forceinline__ BilateralConsts(float upsample_tolerance, float noise_filter_strength)
        : tolerance(upsample_tolerance), noise(noise_filter_strength), three(3.0f), nine(9.0f) {}
};

template <int DIV>
__device__ __forceinline__ float bilateral_upsample(float hi_depth, float hi_ao, float d0, float d1, float d2,
                                                    float d3, float a0, float a1, float a2, float a3,
                                                    const BilateralConsts &k)
{
    const float tolerance = k.tolerance, noise = k.noise;
    const float w0 = div_const<DIV, 9>(__builtin_fabsf(hi_depth - d0) + tolerance, k.nine);
    const float w1 = div_const<DIV, 3>(__builtin_fabsf(hi_depth - d1) + tolerance, k.three);
    const float w2 = div_const<DIV, 1>(__builtin_fabsf(hi_depth - d2) + tolerance);
    const float w3 = div_const<DIV, 3>(__builtin_fabsf(hi_depth - d3) + tolerance, k.three);
    float total = ((w0 + w1) + w2) + w3;
    total = total + noise;
    float sum = a0 * w0;
    sum = mad(a1, w1, sum);
    sum = mad(a2, w2, sum);
    sum = mad(a3, w3, sum);
    sum = sum + noise;
    return div_strict<DIV>(hi_ao * sum, total);
}

// BilateralUpsample for N texels at once with the 4N weight reciprocals issued back to back (and then the N
// reciprocals of the final quotients): same operations per texel, in the same order, as bilateral_upsample<DIV_EXACT_RCP>.
template <int N>
__device__ __forceinline__ void bilateral_upsample_grouped(const float (&hi_depth)[N], const float (&hi_ao)[N], const float (&d)[N][4],
                                                           const float (&a)[N][4], const BilateralConsts &k, float (&out)[N])
{
    float x[N][4], r[N][4], w[N][4];
#pragma unroll
    for (int t = 0; t < N; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) x[t][i] = __builtin_fabsf(hi_depth[t] - d[t][i]) + k.tolerance;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < N; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_rcp_f32 %0, %1" : "=v"(r[t][i]) : "v"(x[t][i]));
    __builtin_amdgcn_sched_barrier(0);
    float total[N], sum[N], rr[N];
#pragma unroll
    for (int t = 0; t < N; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i == 2) {                                              // 1 / x: one Newton step (rcp_strict)
                const float e = mad(-x[t][i], r[t][i], 1.0f);
                w[t][i] = mad(e, r[t][i], r[t][i]);
            } else {                                                   // {9, 3} / x (div_const)
                const float kv = i == 0 ? k.nine : k.three;
                const float q = kv * r[t][i];
                const float e = mad(-x[t][i], q, kv);
                w[t][i] = mad(e, r[t][i], q);
            }
        }
        total[t] = (((w[t][0] + w[t][1]) + w[t][2]) + w[t][3]) + k.noise;
        float sm = a[t][0] * w[t][0];
        sm = mad(a[t][1], w[t][1], sm);
        sm = mad(a[t][2], w[t][2], sm);
        sm = mad(a[t][3], w[t][3], sm);
        sum[t] = hi_ao[t] * (sm + k.noise);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < N; ++t) asm volatile("v_rcp_f32 %0, %1" : "=v"(rr[t]) : "v"(total[t]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < N; ++t) {                                      // div_strict(sum, total)
        const float e0 = mad(-total[t], rr[t], 1.0f);
        const float rc = mad(e0, rr[t], rr[t]);
        const float q = sum[t] * rc;
        const float e = mad(-total[t], q, sum[t]);
        out[t] = mad(e, rc, q);
    }
}

// BilateralUpsample's result as the UNORM8 code the pass stores, for DIV_EXACT_RCP operands (a frame without hostile depth:
// every operand finite, weights and AO values >= 0, hi_ao <= 1).
//
// The code is floor(RN(RN(sat(q) * 255) + 0.5)) for the q of the correctly rounded chain (bilateral_upsample).  An estimate q~
// from the same operations with every division replaced by dividend * v_rcp_f32 differs from q by at most 35 u relatively
// (u = 2^-24):
//   v_rcp_f32 is within one ulp of the correctly rounded reciprocal (meao_selftest(4): every binary32 in range), i.e. within
//             1.5 ulp = 3u of the true one;
//   weights   RN(K * rcp(x)) against RN(K / x): 3u + u (the product) + u (the quotient's rounding) = 5u;
//   the sums  have non-negative terms only, so they inherit the largest relative error of a term plus one u per rounding in
//             either chain: total 5u + 2 * 4u = 13u, weighted sum (times hi_ao) 5u + 2 * 6u = 17u;
//   quotient  u (RN) + 4u (rcp + product) on top: 13u + 17u + 5u = 35u = 1.1 * 2^-19.
// The weighted average times hi_ao is at most 1 (+ rounding), so the estimate is off by < 5.4e-4 of a code; the reference's
// two roundings in the conversion and the fused one of the estimate add < 2.3e-5.  If v~ = fma(sat(q~), 255, 0.5) is further
// than kR8Margin = 2^-10 (1.7 x that bound) from an integer, floor(v~) IS the reference's code.  Otherwise -- 2^-9 of the texels
// of a noisy frame, none where the AO is flat (q~ = 1 -> v~ = 255.5) -- the lane runs the exact sequence.  The agreement of
// estimate and exact code is also checked on the running device for 2^32 hashed operand sets (meao_selftest(7)) and, with
// adversarial 1-ulp reciprocal errors, in numpy by tests/test_r8_estimate_bound.py.
// GROUPED: the four weight reciprocals back to back, as in bilateral_upsample_grouped.
// REUSE: the exact path starts from the estimate's x and 1 / x (14 instructions fewer on that path, 5 - 8 VGPRs more live across
// the branch); without it the whole exact sequence is run again from an opaque copy of the depth, so that nothing of the estimate
// has to stay in registers for the rare path (the nested kernels have none to spare).
constexpr float kR8Margin = 0x1p-10f;     // (1.25 * 2^-11, still above the bound, measured the same: profiles/r04_ab_r8_margin.jsonl)

// (v_cvt_pk_u8_f32, which would convert and pack in one instruction, does not truncate like v_cvt_u32_f32: tried in round 4.)
template <bool GROUPED, bool REUSE = false>
__device__ __forceinline__ uint32_t bilateral_upsample_r8(float hi_depth, float hi_ao, const float (&d)[4], const float (&a)[4],
                                                          const BilateralConsts &k)
{
    float x[4], r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = __builtin_fabsf(hi_depth - d[i]) + k.tolerance;
    if constexpr (GROUPED) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_rcp_f32 %0, %1" : "=v"(r[i]) : "v"(x[i]));
        __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = __builtin_amdgcn_rcpf(x[i]);
    }
    const float w0 = k.nine * r[0], w1 = k.three * r[1], w2 = r[2], w3 = k.three * r[3];
    const float total = (((w0 + w1) + w2) + w3) + k.noise;
    float sm = a[0] * w0;
    sm = mad(a[1], w1, sm);
    sm = mad(a[2], w2, sm);
    sm = mad(a[3], w3, sm);
    const float q = (hi_ao * (sm + k.noise)) * __builtin_amdgcn_rcpf(total);
    const float v = mad(sat(q), 255.0f, 0.5f + kR8Margin);          // v~ + margin: its floor is the code unless its fraction is < 2 margins
    uint32_t code = static_cast<uint32_t>(v);
    const bool near_boundary = __builtin_amdgcn_fractf(v) < 2.0f * kR8Margin;
    if (__builtin_expect(near_boundary, 0)) {
        if constexpr (REUSE) {
            float w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i == 2) {                                              // rcp_strict
                    const float e = mad(-x[i], r[i], 1.0f);
                    w[i] = mad(e, r[i], r[i]);
                } else {                                                   // div_const<9 | 3>
                    const float kv = i == 0 ? k.nine : k.three;
                    const float qw = kv * r[i];
                    const float e = mad(-x[i], qw, kv);
                    w[i] = mad(e, r[i], qw);
                }
            }
            const float exact_total = (((w[0] + w[1]) + w[2]) + w[3]) + k.noise;
            float s = a[0] * w[0];
            s = mad(a[1], w[1], s);
            s = mad(a[2], w[2], s);
            s = mad(a[3], w[3], s);
            code = f32_to_unorm8(div_strict<DIV_EXACT_RCP>(hi_ao * (s + k.noise), exact_total));
        } else {
            float hd = hi_depth;
            asm volatile("" : "+v"(hd));
            code = f32_to_unorm8(bilateral_upsample<DIV_EXACT_RCP>(hd, hi_ao, d[0], d[1], d[2], d[3], a[0], a[1], a[2], a[3], k));
        }
    }
    return code;
}

// One tile of Upsample.main (FINAL) / main_blendout; every thread of the workgroup must call it
// (barriers inside; lanes outside the image leave after the last one).
// (Eight workgroups per CU were measured in round 4: the LoResDB window kept in the registers its loads filled and written behind
// the V-blur into the array the H-blurred values had vacated -- 17.6 KB, 57 VGPRs, one barrier more, bit-exact -- runs at 176.0 us
// against 176.1 us: the same busy cycles, 10 % more wave-cycles, 17 % more waiting.  The pass is bound by the issue rate of its
// instruction mix, not by the number of waves that hide latency: profiles/r04_ab_final_late_depth_8_workgroups.txt.)
template <bool FINAL, int TILE_H = ups_tile_h(FINAL)>
struct UpsLds {
    typedef UpsTile<TILE_H> T;
    static constexpr int kDep0 = FINAL ? 2 : 0;                                   // first row / column kept
    static constexpr int kDepH = FINAL ? T::kLowH + 4 : T::kRawH, kDepW = FINAL ? T::kLowW + 4 : T::kRawW;
    static constexpr int kDepPitch = FINAL ? 36 : T::kRawPitch;
    static constexpr int kInvN = T::kRawH * T::kRawPitch, kHbN = T::kRawH * T::kBlurPitch, kDepN = kDepH * kDepPitch;
    static constexpr int kAoN = T::kRawH * T::kRawPitch;
    static constexpr int kFloats = kInvN + kHbN + kDepN + kAoN;
};

// The global loads of an interior tile (no horizontal clamping, 16-byte loads everywhere, no second AO input): its low-res
// window as 16-byte row quads [LX0 - 4 + 4k, +4) -- depth and AO -- and the hi-res operands of the bilateral phase
// (f16 depth in the full-resolution pass, f32 depth + AO in the blend passes).  Issued at the top of the tile.
template <int AOFMT, bool FINAL, int TILE_H>
struct UpsLoads {
    typedef AoTexel<AOFMT> AO;
    static constexpr int kItems = 10 * UpsTile<TILE_H>::kRawH, kRounds = (kItems + kThreads - 1) / kThreads, kPasses = TILE_H / 32;
    float4v wd[kRounds];
    typename AO::type4 wa[kRounds];
    ushort4v hd16[kPasses][2];
    float4v hd32[kPasses][2];
    // four AO texels as ONE integer: a <4 x i8> value is split into bytes where it is loaded, which puts the wait for it there
    typedef typename std::conditional<sizeof(typename AO::type4) == 4, uint32_t, uint64_t>::type ao_bits_t;
    ao_bits_t ha[kPasses][2];
};

template <bool FINAL, int TILE_H>
__device__ __forceinline__ bool ups_tile_is_interior(const UpsampleArgs &a, int tile)
{
    const int LX0 = ((tile % a.tiles_x) * kUpsTileW) >> 1;
    return a.vec_ok != 0 && !a.lo_ao2 && (a.lw & 3) == 0 && LX0 >= 4 && LX0 + 35 < a.lw;
}

// Hi-res operands.  CLAMPED: every lane loads (out-of-frame lanes re-read the frame's last row / quad and never use it),
// so that the code is branch-free and the compiler's s_waitcnt counts stay exact.
template <int AOFMT, bool FINAL, int TILE_H, bool CLAMPED>
__device__ __forceinline__ void ups_issue_hoisted(const UpsampleArgs &a, int tile, int frame, UpsLoads<AOFMT, FINAL, TILE_H> &L)
{
    const int tid = thread_index_opaque();
    typedef AoTexel<AOFMT> AO;
    typedef typename AO::type ao_t;
    const int HX0 = (tile % a.tiles_x) * kUpsTileW, HY0 = (tile / a.tiles_x) * TILE_H;
    const int hw = a.hw, hh = a.hh;
    const int htx = tid & 15;
    const int hhx0 = CLAMPED ? min(HX0 + 4 * htx, hw - 4) : HX0 + 4 * htx;
#pragma unroll
    for (int pass = 0; pass < TILE_H / 32; ++pass)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int hy_raw = HY0 + 2 * ((tid >> 4) + 16 * pass) + f;
            const int hy = CLAMPED ? min(hy_raw, hh - 1) : hy_raw;
            if (CLAMPED || (hhx0 < hw && hy < hh)) {
                // texel index in the level (< 2^27): 32-bit byte offsets from the frame's uniform bases (saddr addressing)
                const uint32_t hrow = static_cast<uint32_t>(hy * hw + hhx0);
                if constexpr (FINAL) {
                    L.hd16[pass][f] = __builtin_nontemporal_load(reinterpret_cast<const ushort4v *>(at_byte_offset(
                        frame_ptr(static_cast<const uint16_t *>(a.hi_depth), a.frame_stride, frame), hrow * 2u)));
                } else {
                    L.hd32[pass][f] = *reinterpret_cast<const float4v *>(at_byte_offset(
                        frame_ptr(static_cast<const float *>(a.hi_depth), a.frame_stride, frame), hrow * 4u));
                    L.ha[pass][f] = *reinterpret_cast<const typename UpsLoads<AOFMT, FINAL, TILE_H>::ao_bits_t *>(at_byte_offset(
                        frame_ptr(static_cast<const ao_t *>(a.hi_ao), a.frame_stride, frame), hrow * static_cast<uint32_t>(sizeof(ao_t))));
                }
            }
        }
}

// All loads of an interior tile: window first, hi-res operands behind them.  The window comes from L2 (written by the
// previous pass), the hi-res operands of the final pass from HBM; vmcnt retires loads in issue order, so with the hi-res
// loads in front the window wait would last an HBM latency.
template <int AOFMT, bool FINAL, int TILE_H>
__device__ __forceinline__ void ups_issue_interior_loads(const UpsampleArgs &a, int tile, int frame, UpsLoads<AOFMT, FINAL, TILE_H> &L)
{
    const int tid = thread_index_opaque();
    typedef UpsLoads<AOFMT, FINAL, TILE_H> Loads;
    typedef typename Loads::AO AO;
    typedef typename AO::type ao_t;
    const int LX0 = ((tile % a.tiles_x) * kUpsTileW) >> 1, LY0 = ((tile / a.tiles_x) * TILE_H) >> 1;
    const int lw = a.lw, lh = a.lh;
    const float *__restrict__ lo_depth = frame_ptr(a.lo_depth, a.frame_stride, frame);
    const ao_t *__restrict__ lo_ao = frame_ptr(static_cast<const ao_t *>(a.lo_ao), a.frame_stride, frame);
#pragma unroll
    for (int round = 0; round < Loads::kRounds; ++round) {
        const int i = min(tid + round * kThreads, Loads::kItems - 1);
        const int r = i / 10, k = i % 10;
        const int cy = clampi(LY0 - 3 + r, 0, lh - 1);
        const uint32_t idx = static_cast<uint32_t>(cy * lw + (LX0 - 4 + 4 * k));
        L.wd[round] = *reinterpret_cast<const float4v *>(at_byte_offset(lo_depth, idx * 4u));
        L.wa[round] = *reinterpret_cast<const typename AO::type4 *>(at_byte_offset(lo_ao, idx * static_cast<uint32_t>(sizeof(ao_t))));
    }
    __builtin_amdgcn_sched_barrier(0);          // keep the issue order: window, then hi-res
    ups_issue_hoisted<AOFMT, FINAL, TILE_H, true>(a, tile, frame, L);
    __builtin_amdgcn_sched_barrier(0);
}

// NESTED: the LoResAO1 taps (s_ao) were already produced in LDS by blend_window_into_lds (the
// previous pass of the chain evaluated inside this workgroup) instead of being read from global memory.
// Places inside an upsample tile where every thread of the workgroup can put unrelated global loads in flight:
// after_prefetch()   the tile's own low-res window is in LDS (its loads have landed); blur and bilateral follow
// before_bilateral() the hoisted hi-res operands have landed too: nothing in the bilateral phase waits on vmcnt
struct NoHook {
    static constexpr bool kBeforeBilateral = false;
    static constexpr bool kGroupReciprocals = true;      // bilateral_upsample_grouped
    static constexpr bool kEstimateR8 = true;            // bilateral_upsample_r8
    static constexpr bool kReuseEstimate = true;         // ... whose exact path starts from the estimate's reciprocals
    __device__ __forceinline__ void after_prefetch() const {}
    __device__ __forceinline__ void before_bilateral() const {}
};

template <int AOFMT, bool RTNE, bool FINAL, int DIV, bool NESTED = false, typename Hook = NoHook, int TILE_H = ups_tile_h(FINAL)>
__device__ __forceinline__ void upsample_tile(const UpsampleArgs &a, float *smem, int tile, int frame, Hook hook = Hook())
{
    const int tid = thread_index_opaque();
    typedef AoTexel<AOFMT> AO;
    typedef typename AO::type ao_t;
    constexpr int kTileH = TILE_H;
    typedef UpsTile<kTileH> T;
    // One allocation, carved so that the scratch rows the last V-blur run reads past the raw window
    // (rows kRawH .. kRawRows-1 of s_inv and s_hb; their products are never used) fall into the next
    // array instead of being allocated.  In the full-resolution pass the LoResDB window is also cut to
    // what the bilateral phase gathers (rows / columns 2 .. kLow+5): 22.3 KB per workgroup instead of
    // 24.1 KB, which lets a seventh workgroup share the CU's 160 KB (with __launch_bounds__(.., 7):
    // A/B on one box, 357 -> 343 us for the kernel that also carries the next downsample pass).
    typedef UpsLds<FINAL, TILE_H> Lds;
    constexpr int kDep0 = Lds::kDep0, kDepH = Lds::kDepH, kDepW = Lds::kDepW, kDepPitch = Lds::kDepPitch;
    constexpr int kInvN = Lds::kInvN, kHbN = Lds::kHbN, kDepN = Lds::kDepN, kAoN = Lds::kAoN;
    static_assert(kDepW <= kDepPitch && (T::kRawRows - T::kRawH) * T::kRawPitch <= kHbN &&
                  (T::kRawRows - T::kRawH) * T::kBlurPitch <= kDepN + kAoN, "scratch rows stay inside the allocation");
    static_assert(T::kVRows * T::kBlurPitch <= kAoN, "s_vb fits in s_ao");
    static_assert(kInvN % 4 == 0 && kHbN % 4 == 0 && kDepN % 4 == 0, "16-byte alignment of the carved arrays");
    float *const s_inv = smem;                       // 1 / LoResDB   (DepthCache)
    float *const s_hb = s_inv + kInvN;               // after BlurHorizontally (AOCache2)
    float *const s_dep = s_hb + kHbN;                // LoResDB       (LoDepths gather), window from (kDep0, kDep0)
    float *const s_ao = s_dep + kDepN;               // LoResAO1 taps (AOCache1 before blur)
    float *const s_vb = s_ao;                        // after BlurVertically (AOCache1): the raw taps are dead once H-blurred
    auto dep_at = [&](int r, int c) -> float & { return s_dep[(r - kDep0) * kDepPitch + (c - kDep0)]; };
    auto dep_kept = [&](int r, int c) { return !FINAL || (r >= kDep0 && r < kDep0 + kDepH && c >= kDep0 && c < kDep0 + kDepW); };

    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const int HX0 = tile_x * kUpsTileW, HY0 = tile_y * kTileH;
    const int LX0 = HX0 >> 1, LY0 = HY0 >> 1;
    const int lw = a.lw, lh = a.lh, hw = a.hw, hh = a.hh;
    const float *__restrict__ lo_depth = frame_ptr(a.lo_depth, a.frame_stride, frame);
    const ao_t *__restrict__ lo_ao = frame_ptr(static_cast<const ao_t *>(a.lo_ao), a.frame_stride, frame);
    // main_premin*: LoResAO1 = min(LoResAO1, LoResAO2) (COMBINE_LOWER_RESOLUTIONS, UPS:58-60)
    const ao_t *__restrict__ lo_ao2 = a.lo_ao2 ? frame_ptr(static_cast<const ao_t *>(a.lo_ao2), a.frame_stride, frame) : nullptr;
    const BlurConsts bk = {a.step_size, a.blur_tolerance};
    const BilateralConsts bilateral_k(a.upsample_tolerance, a.noise_filter_strength);
    // (fetched here, not where the bilateral phase first stores: a.dst[frame] is a scalar load whose latency would sit right
    // behind the last barrier)
    ao_t *__restrict__ dst = FINAL ? static_cast<ao_t *>(a.dst[frame])
                                   : frame_ptr(static_cast<ao_t *>(a.dst[0]), a.frame_stride, frame);
    asm volatile("" : "+s"(dst));

    PhaseClock clk(FINAL ? 0 : 8);
    __builtin_amdgcn_s_setprio(3);
    // The hi-res operands of the bilateral phase do not depend on anything computed here: their loads
    // are issued first, so that their latency hides behind the prefetch and blur phases.
    constexpr int kPasses = kTileH / 32;
    typedef UpsLoads<AOFMT, FINAL, TILE_H> Loads;
    Loads L;
    auto &hoist_hd16 = L.hd16;
    auto &hoist_hd32 = L.hd32;
    auto &hoist_ha = L.ha;
    const bool hoist_ok = MEAO_X_HOT_PATH_ONLY || a.vec_ok != 0;
    // interior tile, 16-byte loads everywhere, no second AO input: window loads first (ups_issue_interior_loads)
    const bool window_first = !NESTED && (MEAO_X_HOT_PATH_ONLY || ups_tile_is_interior<FINAL, TILE_H>(a, tile));
    if (hoist_ok && !window_first) ups_issue_hoisted<AOFMT, FINAL, TILE_H, false>(a, tile, frame, L);

    // ---- PrefetchData (UPS:54-72): raw window = virtual low-res texels
    // [LX0-3, LX0+34] x [LY0-3, LY0+kLowH+2], clamp addressing per texel.
    const bool interior_x = ((lw & 3) == 0) && LX0 >= 4 && LX0 + 35 < lw;
    if (window_first) {
        constexpr int kItems = Loads::kItems, kRounds = Loads::kRounds;
        if constexpr (!NESTED) ups_issue_interior_loads<AOFMT, FINAL, TILE_H>(a, tile, frame, L);
        auto &wd = L.wd;
        auto &wa = L.wa;
#pragma unroll
        for (int round = 0; round < kRounds; ++round) {
            // an unconditional use: the compiler would otherwise sink the loads of the partial last round into
            // its branch, behind the hi-res loads
            asm volatile("" : : "v"(wd[round]));
            if constexpr (!NESTED) {
                typedef typename std::conditional<sizeof(typename AO::type4) == 4, uint32_t, uint64_t>::type bits_t;
                asm volatile("" : : "v"(__builtin_bit_cast(bits_t, wa[round])));
            }
            const int i = tid + round * kThreads;
            // (storing the window as aligned 16-byte quads -- fourth column from the next lane by DPP -- removes the 4-way
            // bank conflicts of these scalar stores and changes nothing: profiles/r02_ab_v23_aligned_fill.jsonl)
            if (i < kItems) {
                const int r = i / 10, k = i % 10;
                const float dv[4] = {wd[round].x, wd[round].y, wd[round].z, wd[round].w};
                float av[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if constexpr (!NESTED) {
                    av[0] = AO::decode(wa[round].x); av[1] = AO::decode(wa[round].y);
                    av[2] = AO::decode(wa[round].z); av[3] = AO::decode(wa[round].w);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 4 * k + e - 1;
                    if (c >= 0 && c < T::kRawW) {
                        if (dep_kept(r, c)) dep_at(r, c) = dv[e];
                        s_inv[r * T::kRawPitch + c] = rcp_strict<DIV>(dv[e]);     // UPS:67
                        if constexpr (!NESTED) s_ao[r * T::kRawPitch + c] = av[e];
                    }
                }
            }
        }
    } else if (interior_x) {
        // no horizontal clamping inside this tile: one aligned 16-byte depth load (+ 4 AO texels)
        // per lane covers the 40-texel row segment [LX0-4, LX0+35]
        for (int i = tid; i < 10 * T::kRawH; i += kThreads) {
            const int r = i / 10, k = i % 10;
            const int cy = clampi(LY0 - 3 + r, 0, lh - 1);
            const size_t idx = static_cast<size_t>(cy) * lw + (LX0 - 4 + 4 * k);
            const float4v d4 = *reinterpret_cast<const float4v *>(lo_depth + idx);
            const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
            float av[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (!NESTED) {
                const typename AO::type4 a4 = *reinterpret_cast<const typename AO::type4 *>(lo_ao + idx);
                av[0] = AO::decode(a4.x); av[1] = AO::decode(a4.y); av[2] = AO::decode(a4.z); av[3] = AO::decode(a4.w);
            }
            if (!NESTED && lo_ao2) {
                const typename AO::type4 b4 = *reinterpret_cast<const typename AO::type4 *>(lo_ao2 + idx);
                av[0] = __builtin_fminf(av[0], AO::decode(b4.x)); av[1] = __builtin_fminf(av[1], AO::decode(b4.y));
                av[2] = __builtin_fminf(av[2], AO::decode(b4.z)); av[3] = __builtin_fminf(av[3], AO::decode(b4.w));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * k + e - 1;
                if (c >= 0 && c < T::kRawW) {
                    if (dep_kept(r, c)) dep_at(r, c) = dv[e];
                    s_inv[r * T::kRawPitch + c] = rcp_strict<DIV>(dv[e]);     // UPS:67
                    if constexpr (!NESTED) s_ao[r * T::kRawPitch + c] = av[e];
                }
            }
        }
    } else {
        for (int i = tid; i < T::kRawW * T::kRawH; i += kThreads) {
            const int r = i / T::kRawW, c = i % T::kRawW;
            const int cy = clampi(LY0 - 3 + r, 0, lh - 1), cx = clampi(LX0 - 3 + c, 0, lw - 1);
            const size_t idx = static_cast<size_t>(cy) * lw + cx;
            const float d = lo_depth[idx];
            if (dep_kept(r, c)) dep_at(r, c) = d;
            s_inv[r * T::kRawPitch + c] = rcp_strict<DIV>(d);             // UPS:67
            if constexpr (!NESTED) {
                float av = AO::decode(lo_ao[idx]);
                if (lo_ao2) av = __builtin_fminf(av, AO::decode(lo_ao2[idx]));
                s_ao[r * T::kRawPitch + c] = av;
            }
        }
    }
    clk.mark(0);         // 0: window loaded, converted, stored to LDS
    __syncthreads();
    clk.mark(1);         // 1: barrier
    __builtin_amdgcn_s_setprio(0);       // (3 kept through the blur phases: +10 % on the pass; rising through the phases: +2 %, r03)
    hook.after_prefetch();

    // ---- BlurHorizontally: runs of 4 outputs; output (r, c) is centred on raw column c+2.
    // (Columns 34, 35 of the last run are scratch: they read the row padding.)
    for (int i = tid; i < T::kHSegs * T::kRawH; i += kThreads) {
        const int r = i / T::kHSegs, c0 = (i % T::kHSegs) * T::kHRun;
        float av[T::kHRun + 4], zv[T::kHRun + 4], o[T::kHRun];
        if constexpr (T::kHRun == 4) {
            const float4v a0 = *reinterpret_cast<const float4v *>(&s_ao[r * T::kRawPitch + c0]);
            const float4v a1 = *reinterpret_cast<const float4v *>(&s_ao[r * T::kRawPitch + c0 + 4]);
            const float4v z0 = *reinterpret_cast<const float4v *>(&s_inv[r * T::kRawPitch + c0]);
            const float4v z1 = *reinterpret_cast<const float4v *>(&s_inv[r * T::kRawPitch + c0 + 4]);
            av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
            zv[0] = z0.x; zv[1] = z0.y; zv[2] = z0.z; zv[3] = z0.w; zv[4] = z1.x; zv[5] = z1.y; zv[6] = z1.z; zv[7] = z1.w;
        } else {    // even run length: 8-byte aligned taps
            static_assert(T::kHRun % 2 == 0, "runs start on even columns");
#pragma unroll
            for (int t = 0; t < T::kHRun + 4; t += 2) {
                const float2v a2 = *reinterpret_cast<const float2v *>(&s_ao[r * T::kRawPitch + c0 + t]);
                const float2v z2 = *reinterpret_cast<const float2v *>(&s_inv[r * T::kRawPitch + c0 + t]);
                av[t] = a2.x; av[t + 1] = a2.y; zv[t] = z2.x; zv[t + 1] = z2.y;
            }
        }
        blur_run<T::kHRun>(bk, av, zv, o);
        if constexpr (T::kHRun == 4) {
            *reinterpret_cast<float4v *>(&s_hb[r * T::kBlurPitch + c0]) = float4v{o[0], o[1], o[2], o[3]};
        } else {
#pragma unroll
            for (int n = 0; n < T::kHRun; n += 2)
                *reinterpret_cast<float2v *>(&s_hb[r * T::kBlurPitch + c0 + n]) = float2v{o[n], o[n + 1]};
        }
    }
    clk.mark(2);         // 2: H-blur
    __syncthreads();
    clk.mark(3);         // 3: barrier

    // ---- BlurVertically: runs of T::kVRun outputs; output (r, c) is centred on H-blurred row
    // r+2; depths come from the same virtual column (DepthCache[... + 2], UPS:141-146).  Rows
    // >= T::kBlurH of the last run are scratch: they read rows past the window (never used).
    // s_vb aliases s_ao, which nothing reads after the barrier above.
    for (int i = tid; i < T::kVSegs * T::kBlurW; i += kThreads) {
        const int c = i % T::kBlurW, r0 = (i / T::kBlurW) * T::kVRun;
        float av[T::kVRun + 4], zv[T::kVRun + 4], o[T::kVRun];
#pragma unroll
        for (int t = 0; t < T::kVRun + 4; ++t) {
            av[t] = s_hb[(r0 + t) * T::kBlurPitch + c];
            zv[t] = s_inv[(r0 + t) * T::kRawPitch + c + 2];
        }
        blur_run<T::kVRun>(bk, av, zv, o);
#pragma unroll
        for (int n = 0; n < T::kVRun; ++n) s_vb[(r0 + n) * T::kBlurPitch + c] = o[n];
    }
    clk.mark(4);         // 4: V-blur
    __syncthreads();
    clk.mark(5);         // 5: barrier
    if constexpr (Hook::kBeforeBilateral) {
        // vmcnt retires in order: a load issued here would sit behind nothing only if the hoisted operands
        // are waited for first -- naming them in an asm makes the compiler put that wait here
#pragma unroll
        for (int pass = 0; pass < kPasses; ++pass) {
            if constexpr (FINAL) {
                asm volatile("" : : "v"(hoist_hd16[pass][0]), "v"(hoist_hd16[pass][1]));
            } else {
                asm volatile("" : : "v"(hoist_hd32[pass][0]), "v"(hoist_hd32[pass][1]), "v"(hoist_ha[pass][0]), "v"(hoist_ha[pass][1]));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        hook.before_bilateral();
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- bilateral upsample: lane = 4 x 2 hi-res texels per pass of 64 x 32
    if constexpr (!FINAL) {
        // The hoisted AO quads (one integer each, UpsLoads) pass through an opaque statement HERE, behind the last barrier: their
        // decoding otherwise moves up to the window phase -- `s_waitcnt vmcnt(0)` in front of the first barrier, i.e. the latency
        // the hoisting was meant to hide (round 4: ISA of the L2->L1 kernel).
#pragma unroll
        for (int pass = 0; pass < kPasses; ++pass)
#pragma unroll
            for (int f = 0; f < 2; ++f) asm volatile("" : "+v"(hoist_ha[pass][f]));
    }
    const bool vec_ok_frame = MEAO_X_HOT_PATH_ONLY || a.vec_ok != 0;       // hw % 4 == 0 and (final pass) 4-texel aligned caller pointers
    // Gather component order x=(c-1,c) y=(c,c) z=(c,c-1) w=(c-1,c-1) as (col,row) offsets
    constexpr int gx[4] = {-1, 0, 0, -1}, gy[4] = {0, 0, -1, -1};
    const int tx = tid & 15;
    const int hx0 = HX0 + 4 * tx;
    // WHOLE: the tile lies inside the frame and its rows take 4-texel loads and stores -- no lane or row of it is masked
    auto bilateral_phase = [&](auto whole_tile) {
        constexpr bool WHOLE = decltype(whole_tile)::value;
        const bool vec_ok = WHOLE || vec_ok_frame;
        if (!WHOLE && hx0 >= hw) return;
#pragma unroll       // the hoisted operands live in registers: static indices
        for (int pass = 0; pass < kTileH / 32; ++pass) {
            const int ty = (tid >> 4) + 16 * pass;
            const int hy0 = HY0 + 2 * ty;
            if (!WHOLE && hy0 >= hh) return;

            float vb[3][4], dl[3][4];   // blurred AO / low depth at virtual (LY0-1+ty+rr, LX0-1+2tx+cc)
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                const float2v v0 = *reinterpret_cast<const float2v *>(&s_vb[(ty + rr) * T::kBlurPitch + 2 * tx]);
                const float2v v1 = *reinterpret_cast<const float2v *>(&s_vb[(ty + rr) * T::kBlurPitch + 2 * tx + 2]);
                const float2v d0 = *reinterpret_cast<const float2v *>(&dep_at(ty + rr + 2, 2 * tx + 2));
                const float2v d1 = *reinterpret_cast<const float2v *>(&dep_at(ty + rr + 2, 2 * tx + 4));
                vb[rr][0] = v0.x; vb[rr][1] = v0.y; vb[rr][2] = v1.x; vb[rr][3] = v1.y;
                dl[rr][0] = d0.x; dl[rr][1] = d0.y; dl[rr][2] = d1.x; dl[rr][3] = d1.y;
            }

#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int hy = hy0 + f;
                if (!WHOLE && hy >= hh) break;
                const size_t hrow = static_cast<size_t>(hy) * hw + hx0;
                float hd[4], ha[4] = {1.0f, 1.0f, 1.0f, 1.0f};                  // HiSSAOs = 1 in "main" (UPS:222)
                if constexpr (FINAL) {
                    const uint16_t *p = frame_ptr(static_cast<const uint16_t *>(a.hi_depth), a.frame_stride, frame) + hrow;
                    if (vec_ok) {
                        const ushort4v q = hoist_hd16[pass][f];
                        hd[0] = f16_bits_to_f32(q.x); hd[1] = f16_bits_to_f32(q.y);
                        hd[2] = f16_bits_to_f32(q.z); hd[3] = f16_bits_to_f32(q.w);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) hd[e] = (hx0 + e < hw) ? f16_bits_to_f32(p[e]) : 1.0f;
                    }
                } else {
                    const float *p = frame_ptr(static_cast<const float *>(a.hi_depth), a.frame_stride, frame) + hrow;
                    const ao_t *q = frame_ptr(static_cast<const ao_t *>(a.hi_ao), a.frame_stride, frame) + hrow;
                    if (vec_ok) {
                        const float4v d4 = hoist_hd32[pass][f];
                        const typename Loads::ao_bits_t a4 = hoist_ha[pass][f];
                        hd[0] = d4.x; hd[1] = d4.y; hd[2] = d4.z; hd[3] = d4.w;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ha[e] = AO::decode(static_cast<ao_t>(a4 >> (8 * sizeof(ao_t) * e)));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            hd[e] = (hx0 + e < hw) ? p[e] : 1.0f;
                            ha[e] = (hx0 + e < hw) ? AO::decode(q[e]) : 1.0f;
                        }
                    }
                }
                ao_t res[4];
                if constexpr (!MEAO_X_UPS_EXACT_R8 && Hook::kEstimateR8 && DIV == DIV_EXACT_RCP && AOFMT == MEAO_AO_R8) {
                    // UNORM8 storage: the code from the uncorrected reciprocals wherever that provably is the reference's code
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int cc = ((e + 1) >> 1) + 1, rr = f + 1;            // as below
                        const int comp = (e & 1) ? ((f & 1) ? 3 : 0) : ((f & 1) ? 2 : 1);
                        float gd[4], ga[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int g = (comp + i) & 3;
                            gd[i] = dl[rr + gy[g]][cc + gx[g]];
                            ga[i] = vb[rr + gy[g]][cc + gx[g]];
                        }
                        res[e] = static_cast<ao_t>(bilateral_upsample_r8<Hook::kGroupReciprocals, !NESTED && Hook::kReuseEstimate>(hd[e], ha[e], gd, ga, bilateral_k));
                    }
                } else if constexpr (DIV == DIV_EXACT_RCP && Hook::kGroupReciprocals) {
                    // The four weight reciprocals of a texel back to back: an isolated v_rcp_f32 costs the SIMD ~3 cycles more than
                    // one that follows another (tools/ubench_issue.hip "bilateral mix": 3.81 -> 3.55 cycles per instruction).  A/B:
                    // L2->L1 65 -> 58.5 us, L1->L0 202.5 -> 199.2 us; two texels per group: the same (profiles/r03_ab_rcp_group*.jsonl).
                    constexpr int kGroup = 1;             // texels whose reciprocals are issued together
#pragma unroll
                    for (int e0 = 0; e0 < 4; e0 += kGroup) {
                        float gd[kGroup][4], ga[kGroup][4], ghd[kGroup], gha[kGroup], gout[kGroup];
#pragma unroll
                        for (int t = 0; t < kGroup; ++t) {
                            const int e = e0 + t;
                            const int cc = ((e + 1) >> 1) + 1, rr = f + 1;
                            const int comp = (e & 1) ? ((f & 1) ? 3 : 0) : ((f & 1) ? 2 : 1);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int g = (comp + i) & 3;
                                gd[t][i] = dl[rr + gy[g]][cc + gx[g]];
                                ga[t][i] = vb[rr + gy[g]][cc + gx[g]];
                            }
                            ghd[t] = hd[e]; gha[t] = ha[e];
                        }
                        bilateral_upsample_grouped<kGroup>(ghd, gha, gd, ga, bilateral_k, gout);
#pragma unroll
                        for (int t = 0; t < kGroup; ++t) res[e0 + t] = AO::template encode<RTNE>(gout[t]);
                    }
                } else
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // hi texel (4tx+e, 2ty+f) is written by dispatch thread D = ((hx+1)>>1, (hy+1)>>1)
                    // through Gather component comp (UPS:229-232); its taps are rotated by comp.
                    const int cc = ((e + 1) >> 1) + 1, rr = f + 1;            // D in vb/dl coordinates
                    const int comp = (e & 1) ? ((f & 1) ? 3 : 0) : ((f & 1) ? 2 : 1);
                    const int g0 = comp & 3, g1 = (comp + 1) & 3, g2 = (comp + 2) & 3, g3 = (comp + 3) & 3;
                    const float v = bilateral_upsample<DIV>(
                        hd[e], ha[e],
                        dl[rr + gy[g0]][cc + gx[g0]], dl[rr + gy[g1]][cc + gx[g1]],
                        dl[rr + gy[g2]][cc + gx[g2]], dl[rr + gy[g3]][cc + gx[g3]],
                        vb[rr + gy[g0]][cc + gx[g0]], vb[rr + gy[g1]][cc + gx[g1]],
                        vb[rr + gy[g2]][cc + gx[g2]], vb[rr + gy[g3]][cc + gx[g3]],
                        bilateral_k);
                    res[e] = AO::template encode<RTNE>(v);
                }
                ao_t *o = dst + hrow;
                if (vec_ok) {
                    typename AO::type4 r4; r4.x = res[0]; r4.y = res[1]; r4.z = res[2]; r4.w = res[3];
                    // the result leaves the path; the blend passes' outputs are re-read by the next pass from L2
                    if constexpr (FINAL) __builtin_nontemporal_store(r4, reinterpret_cast<typename AO::type4 *>(o));
                    else *reinterpret_cast<typename AO::type4 *>(o) = r4;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (hx0 + e < hw) o[e] = res[e];
                }
            }
            clk.mark(6 + pass);  // 6, 7: bilateral pass 0 / 1 (64-row tiles) incl. its stores being issued
        }
    };
    // (the copy exists for clean frames only -- the IEEE-division bodies of a hostile frame are four times as long -- and not in the
    // nested launches, which have no registers for it: 3 spilled VGPRs in the two-level kernel, no gain measured there)
    if (MEAO_X_BIL_WHOLE_TILE && !NESTED && DIV == DIV_EXACT_RCP && vec_ok_frame && HX0 + kUpsTileW <= hw && HY0 + kTileH <= hh)
        bilateral_phase(std::true_type());
    else
        bilateral_phase(std::false_type());
}

// One blend pass (Upsample.main_blendout) evaluated for an arbitrary window of its OUTPUT level, into LDS:
// out[r * out_pitch + c] = what a later pass would read back from Combined<k> at virtual texel
// (vx0 + c, vy0 + r) with clamp addressing (UPS:54-72), i.e. the stored-and-decoded result.
// Every output of the pass is a pure function of the global inputs (each blurred value only depends
// on its own 5-tap window of clamped taps, the bilateral taps on the texel's parity), so evaluating it
// here gives the bits the stand-alone pass writes.  Texels of the window that fall into the "own"
// rectangle are also stored to the pass's real target, so that the buffer exists for the debug views.
// Window at most 38 x 22: low-res D range <= 21 x 13, raw taps <= 25 x 17 (scratch: 1905 floats).
constexpr int kNestLowW = 21, kNestLowH = 13, kNestRawW = kNestLowW + 4, kNestRawH = kNestLowH + 4;
constexpr int kNestScratch = 3 * kNestRawW * kNestRawH + kNestLowW * kNestRawH + kNestLowW * kNestLowH;

// The low-res texels a window of the pass's output level touches (bilateral taps D = (X+1)>>1 and D-1, X clamped
// to the level) and, two further out on every side, the raw LoResAO1 / LoResDB taps of the blur (virtual
// coordinates, clamped on load).
struct NestExtent {
    int dx_lo, dy_lo, nlw, nlh, rx0, ry0, rw, rh;
    __device__ __forceinline__ NestExtent(const UpsampleArgs &in, int vx0, int vy0, int win_w, int win_h)
    {
        const int cx_min = clampi(vx0, 0, in.hw - 1), cx_max = clampi(vx0 + win_w - 1, 0, in.hw - 1);
        const int cy_min = clampi(vy0, 0, in.hh - 1), cy_max = clampi(vy0 + win_h - 1, 0, in.hh - 1);
        dx_lo = ((cx_min + 1) >> 1) - 1; nlw = ((cx_max + 1) >> 1) - dx_lo + 1;
        dy_lo = ((cy_min + 1) >> 1) - 1; nlh = ((cy_max + 1) >> 1) - dy_lo + 1;
        rx0 = dx_lo - 2; ry0 = dy_lo - 2; rw = nlw + 4; rh = nlh + 4;
    }
};

// TAPS_IN_LDS: the raw LoResAO1 taps (scratch[r * kNestRawW + c], r < rh, c < rw of NestExtent) were produced by
// another blend_window_into_lds call (the pass below, evaluated for exactly that window) instead of being
// read from Combined<k+1> in global memory.
template <int AOFMT, bool RTNE, int DIV, bool TAPS_IN_LDS = false>
__device__ __forceinline__ void blend_window_into_lds(const UpsampleArgs &in, float *out, int out_pitch, int vx0, int vy0,
                                                      int win_w, int win_h, float *scratch, int frame, int own_x0,
                                                      int own_y0, int own_w, int own_h)
{
    typedef AoTexel<AOFMT> AO;
    typedef typename AO::type ao_t;
    float *const r_ao = scratch;                                   // raw LoResAO1 taps
    float *const r_inv = r_ao + kNestRawW * kNestRawH;             // 1 / LoResDB
    float *const r_dep = r_inv + kNestRawW * kNestRawH;            // LoResDB
    float *const hb = r_dep + kNestRawW * kNestRawH;               // after BlurHorizontally
    float *const vb = hb + kNestLowW * kNestRawH;                  // after BlurVertically
    const int lw = in.lw, lh = in.lh, hw = in.hw, hh = in.hh;
    const float *__restrict__ lo_depth = frame_ptr(in.lo_depth, in.frame_stride, frame);
    const ao_t *__restrict__ lo_ao = frame_ptr(static_cast<const ao_t *>(in.lo_ao), in.frame_stride, frame);
    const float *__restrict__ hi_depth = frame_ptr(static_cast<const float *>(in.hi_depth), in.frame_stride, frame);
    const ao_t *__restrict__ hi_ao = frame_ptr(static_cast<const ao_t *>(in.hi_ao), in.frame_stride, frame);
    ao_t *__restrict__ dst = frame_ptr(static_cast<ao_t *>(in.dst[0]), in.frame_stride, frame);
    const BlurConsts bk = {in.step_size, in.blur_tolerance};
    const BilateralConsts bilateral_k(in.upsample_tolerance, in.noise_filter_strength);

    const NestExtent ext(in, vx0, vy0, win_w, win_h);
    const int dx_lo = ext.dx_lo, dy_lo = ext.dy_lo, nlw = ext.nlw, nlh = ext.nlh;
    const int rx0 = ext.rx0, ry0 = ext.ry0, rw = ext.rw, rh = ext.rh;           // raw taps (virtual, clamped on load)

    // The hi-res operands of the bilateral step depend on nothing computed here: loaded now, used three
    // barriers later (at most 38 x 22 window texels: four per lane).
    constexpr int kHoisted = (40 * 22 + kThreads - 1) / kThreads;          // items on the output pitch: at most 40 x 22
    float hoist_d[kHoisted];
    ao_t hoist_a[kHoisted];
    // (window items on the output array's pitch, a compile-time value at every call site; see the loops below)
#pragma unroll
    for (int j = 0; j < kHoisted; ++j) {
        const int i = min(static_cast<int>(threadIdx.x) + j * kThreads, out_pitch * win_h - 1);
        const int X = clampi(vx0 + min(i % out_pitch, win_w - 1), 0, hw - 1), Y = clampi(vy0 + i / out_pitch, 0, hh - 1);
        const uint32_t at = static_cast<uint32_t>(Y * hw + X);
        hoist_d[j] = *at_byte_offset(hi_depth, at * 4u);
        hoist_a[j] = *at_byte_offset(hi_ao, at * static_cast<uint32_t>(sizeof(ao_t)));
    }

    // Work items are laid out on the arrays' compile-time pitches (item i = row i / pitch, column i % pitch; columns past the
    // extent idle): the extents are run-time values, and a division by one costs ~25 VALU instructions where a division by
    // a constant costs three -- the four loops of this function did eight of them per lane (a third of the two-level
    // kernel's instructions were integer arithmetic, profiles/r03_pmc_summary.txt).
    for (int i = threadIdx.x; i < kNestRawW * rh; i += kThreads) {
        const int r = i / kNestRawW, c = i % kNestRawW;
        if (c >= rw) continue;
        const uint32_t idx = static_cast<uint32_t>(clampi(ry0 + r, 0, lh - 1) * lw + clampi(rx0 + c, 0, lw - 1));   // a level is < 2^30 texels
        const float d = *at_byte_offset(lo_depth, idx * 4u);
        r_dep[i] = d;
        r_inv[i] = rcp_strict<DIV>(d);                                          // UPS:67
        if constexpr (!TAPS_IN_LDS) r_ao[i] = AO::decode(*at_byte_offset(lo_ao, idx * static_cast<uint32_t>(sizeof(ao_t))));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kNestLowW * rh; i += kThreads) {               // BlurHorizontally, one output per lane
        const int r = i / kNestLowW, c = i % kNestLowW;
        if (c >= nlw) continue;
        float av[5], zv[5], o[1];
#pragma unroll
        for (int t = 0; t < 5; ++t) { av[t] = r_ao[r * kNestRawW + c + t]; zv[t] = r_inv[r * kNestRawW + c + t]; }
        blur_run<1>(bk, av, zv, o);
        hb[i] = o[0];
    }
    __syncthreads();
    // BlurVertically.  The interior extent (20 x 12 = 240 outputs) is one round of the workgroup on a pitch of 20
    auto blur_vertically = [&](auto pitch_c) {
        constexpr int kPitch = decltype(pitch_c)::value;
        for (int i = threadIdx.x; i < kPitch * nlh; i += kThreads) {
            const int r = i / kPitch, c = i % kPitch;
            if (kPitch != kNestLowW || c < nlw) {
                float av[5], zv[5], o[1];
#pragma unroll
                for (int t = 0; t < 5; ++t) { av[t] = hb[(r + t) * kNestLowW + c]; zv[t] = r_inv[(r + t) * kNestRawW + c + 2]; }
                blur_run<1>(bk, av, zv, o);
                vb[r * kNestLowW + c] = o[0];
            }
        }
    };
    if (nlw == kNestLowW - 1) blur_vertically(std::integral_constant<int, kNestLowW - 1>());
    else blur_vertically(std::integral_constant<int, kNestLowW>());
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kHoisted; ++j) {
        const int i = threadIdx.x + j * kThreads;
        if (i >= out_pitch * win_h) break;
        const int wr = i / out_pitch, wc = i % out_pitch;
        if (wc >= win_w) continue;
        const int X = clampi(vx0 + wc, 0, hw - 1), Y = clampi(vy0 + wr, 0, hh - 1);
        const int Dx = (X + 1) >> 1, Dy = (Y + 1) >> 1;
        // Tap k of the texel is Gather component g = (comp + k) & 3 of dispatch thread D: texel D + (gx[g], gy[g]), i.e. the four
        // texels {Dx - 1, Dx} x {Dy - 1, Dy} in an order that rotates with the texel's parity (comp, UPS:229-232).  comp is a per-lane
        // value here (the window is dealt to lanes linearly), so the taps' byte distances below D in either array -- {4, 0, pitch * 4,
        // pitch * 4 + 4} for g = 0..3 -- sit in one word that is rotated by comp bytes: five integer operations for four addresses.
        // (Indexing gx[] / gy[] with the run-time g made the compiler put the tables in memory: eight global loads per texel.)
        const uint32_t comp = ((static_cast<uint32_t>(Y) & 1u) << 1) | (((static_cast<uint32_t>(X ^ Y)) & 1u) ^ 1u);   // (X odd, Y odd): (1,0) 0, (0,0) 1, (0,1) 2, (1,1) 3
        constexpr uint32_t kBelowVb = 4u | (0u << 8) | (static_cast<uint32_t>(kNestLowW * 4) << 16) | (static_cast<uint32_t>(kNestLowW * 4 + 4) << 24);
        constexpr uint32_t kBelowDep = 4u | (0u << 8) | (static_cast<uint32_t>(kNestRawW * 4) << 16) | (static_cast<uint32_t>(kNestRawW * 4 + 4) << 24);
        static_assert(kNestLowW * 4 + 4 < 256 && kNestRawW * 4 + 4 < 256, "byte fields");
        const uint32_t below_vb = __builtin_amdgcn_alignbit(kBelowVb, kBelowVb, comp * 8u), below_dep = __builtin_amdgcn_alignbit(kBelowDep, kBelowDep, comp * 8u);
        const char *const vb_at_d = reinterpret_cast<const char *>(vb + ((Dy - dy_lo) * kNestLowW + (Dx - dx_lo)));
        const char *const dep_at_d = reinterpret_cast<const char *>(r_dep + ((Dy - ry0) * kNestRawW + (Dx - rx0)));
        float dk[4], ak[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ak[k] = *reinterpret_cast<const float *>(vb_at_d - ((below_vb >> (8 * k)) & 0xffu));
            dk[k] = *reinterpret_cast<const float *>(dep_at_d - ((below_dep >> (8 * k)) & 0xffu));
        }
        const uint32_t at = static_cast<uint32_t>(Y * hw + X) * static_cast<uint32_t>(sizeof(ao_t));      // byte offset in the level
        float v;
        if constexpr (!MEAO_X_UPS_EXACT_R8 && DIV == DIV_EXACT_RCP && AOFMT == MEAO_AO_R8) {
            const ao_t q = static_cast<ao_t>(bilateral_upsample_r8<true>(hoist_d[j], AO::decode(hoist_a[j]), dk, ak, bilateral_k));
            out[i] = AO::decode(q);
            if (vx0 + wc == X && vy0 + wr == Y && X >= own_x0 && X < own_x0 + own_w && Y >= own_y0 && Y < own_y0 + own_h) *at_byte_offset(dst, at) = q;
            continue;
        }
        if constexpr (DIV == DIV_EXACT_RCP) {       // the four weight reciprocals back to back (see upsample_tile)
            const float ghd[1] = {hoist_d[j]}, gha[1] = {AO::decode(hoist_a[j])};
            const float gd[1][4] = {{dk[0], dk[1], dk[2], dk[3]}}, ga[1][4] = {{ak[0], ak[1], ak[2], ak[3]}};
            float gout[1];
            bilateral_upsample_grouped<1>(ghd, gha, gd, ga, bilateral_k, gout);
            v = gout[0];
        } else {
            v = bilateral_upsample<DIV>(hoist_d[j], AO::decode(hoist_a[j]), dk[0], dk[1], dk[2], dk[3], ak[0], ak[1], ak[2], ak[3],
                                        bilateral_k);
        }
        const ao_t q = AO::template encode<RTNE>(v);
        out[i] = AO::decode(q);
        if (vx0 + wc == X && vy0 + wr == Y && X >= own_x0 && X < own_x0 + own_w && Y >= own_y0 && Y < own_y0 + own_h) *at_byte_offset(dst, at) = q;
    }
    __syncthreads();
}

// Upsample.main_blendout L4 -> L3 evaluated inside the L3 -> L2 pass: the smallest pass of the chain
// (one wave of workgroups, three barriers, two memory round trips: latency-bound, and a launch of its
// own) disappears; each L3 -> L2 tile computes the 38 x 22 window of Combined3 it needs itself
// (1.6x the texels of that pass, which is 1/16 of the last pass's work).
template <int AOFMT, bool RTNE, int DIV>
__device__ __forceinline__ void upsample_two_level_tile(const UpsampleArgs &outer, const UpsampleArgs &inner, float *smem, int tile,
                                                        int frame)
{
    typedef UpsTile<ups_tile_h(false)> T;
    typedef UpsLds<false> Lds;
    static_assert(kNestScratch <= Lds::kInvN + Lds::kHbN + Lds::kDepN, "the nested pass's scratch precedes s_ao");
    float *const s_ao = smem + Lds::kInvN + Lds::kHbN + Lds::kDepN;
    const int tile_x = tile % outer.tiles_x, tile_y = tile / outer.tiles_x;
    const int LX0 = (tile_x * kUpsTileW) >> 1, LY0 = (tile_y * ups_tile_h(false)) >> 1;
    blend_window_into_lds<AOFMT, RTNE, DIV>(inner, s_ao, T::kRawPitch, LX0 - 3, LY0 - 3, T::kRawW, T::kRawH, smem, frame,
                                            LX0, LY0, T::kLowW, T::kLowH);
    upsample_tile<AOFMT, RTNE, false, DIV, true>(outer, smem, tile, frame);
}

template <int AOFMT, bool RTNE, int DIV>
__global__ __launch_bounds__(kThreads, 8) void upsample_two_level_kernel(const UpsampleArgs outer, const UpsampleArgs inner)
{
    __shared__ __attribute__((aligned(16))) float smem[UpsLds<false>::kFloats];
    const int tile = xcd_contiguous(blockIdx.x, gridDim.x), frame = blockIdx.z;
    if constexpr (DIV == DIV_EXACT_RCP) {
        if (frame_is_hostile(outer.hostile, outer.generation, frame)) {
            upsample_two_level_tile<AOFMT, RTNE, DIV_IEEE>(outer, inner, smem, tile, frame);
            return;
        }
    }
    upsample_two_level_tile<AOFMT, RTNE, DIV>(outer, inner, smem, tile, frame);
}

// The (rare) hostile-frame variant of a tile: the same code with IEEE division.
template <int AOFMT, bool RTNE, bool FINAL, int DIV, typename Hook = NoHook, int TILE_H = ups_tile_h(FINAL)>
__device__ __forceinline__ void upsample_tile_checked(const UpsampleArgs &a, float *smem, int tile, int frame, Hook hook = Hook())
{
    if constexpr (DIV == DIV_EXACT_RCP) {
        if (frame_is_hostile(a.hostile, a.generation, frame)) {       // wave-uniform, decided per frame
            upsample_tile<AOFMT, RTNE, FINAL, DIV_IEEE, false, Hook, TILE_H>(a, smem, tile, frame, hook);
            return;
        }
    }
    upsample_tile<AOFMT, RTNE, FINAL, DIV, false, Hook, TILE_H>(a, smem, tile, frame, hook);
}

template <int AOFMT, bool RTNE, bool FINAL, int DIV>
__global__ __launch_bounds__(kThreads, FINAL ? 7 : 1) void upsample_kernel(const UpsampleArgs a)
{
    __shared__ __attribute__((aligned(16))) float smem[UpsLds<FINAL>::kFloats];
    upsample_tile_checked<AOFMT, RTNE, FINAL, DIV>(a, smem, xcd_contiguous(blockIdx.x, gridDim.x), blockIdx.z);
}

// Upsample.main for calls with few tiles (one 1080p frame: 510 tiles of 64 x 64 on 256 CUs): 64 x 32 tiles, twice
// the workgroups, half the serial work in each.
template <int AOFMT, bool RTNE, int DIV>
__global__ __launch_bounds__(kThreads) void upsample_final_small_kernel(const UpsampleArgs a)
{
    __shared__ __attribute__((aligned(16))) float smem[UpsLds<true, kUpsTileHSmall>::kFloats];
    upsample_tile_checked<AOFMT, RTNE, true, DIV, NoHook, kUpsTileHSmall>(a, smem, xcd_contiguous(blockIdx.x, gridDim.x), blockIdx.z);
}

// ---- one frame per call: L4 -> L3 and L3 -> L2 inside the L2 -> L1 launch ---------------------------------
// With one or two frames per call the three blend passes are three launches of a few hundred workgroups that
// each wait out a memory round trip and three barriers; their arithmetic is nothing.  Here every L2 -> L1 tile
// evaluates the window of Combined2 it reads (as in the two-level launch), and for that the window of
// Combined3 those taps come from: inner -> the raw-tap array of mid -> the raw-tap array of the outer tile.
// ~2.6x the arithmetic of the two small passes, one launch and one latency chain instead of three; both
// intermediate buffers are still written (each tile its own 16 x 8 of Combined3 and 32 x 16 of Combined2).
template <int AOFMT, bool RTNE, int DIV>
__device__ __forceinline__ void upsample_three_level_tile(const UpsampleArgs &outer, const UpsampleArgs &mid, const UpsampleArgs &inner,
                                                          float *smem, int tile, int frame)
{
    typedef UpsTile<ups_tile_h(false)> T;
    typedef UpsLds<false> Lds;
    float *const s_ao = smem + Lds::kInvN + Lds::kHbN + Lds::kDepN;      // raw taps of the outer tile
    float *const inner_scratch = smem + Lds::kFloats;
    const int tile_x = tile % outer.tiles_x, tile_y = tile / outer.tiles_x;
    const int LX0 = (tile_x * kUpsTileW) >> 1, LY0 = (tile_y * ups_tile_h(false)) >> 1;      // L2 coordinates
    const NestExtent mid_ext(mid, LX0 - 3, LY0 - 3, T::kRawW, T::kRawH);                   // what mid reads of Combined3
    blend_window_into_lds<AOFMT, RTNE, DIV>(inner, smem, kNestRawW, mid_ext.rx0, mid_ext.ry0, mid_ext.rw, mid_ext.rh,
                                            inner_scratch, frame, LX0 >> 1, LY0 >> 1, T::kLowW / 2, T::kLowH / 2);
    blend_window_into_lds<AOFMT, RTNE, DIV, true>(mid, s_ao, T::kRawPitch, LX0 - 3, LY0 - 3, T::kRawW, T::kRawH, smem, frame,
                                                  LX0, LY0, T::kLowW, T::kLowH);
    upsample_tile<AOFMT, RTNE, false, DIV, true>(outer, smem, tile, frame);
}

template <int AOFMT, bool RTNE, int DIV>
__global__ __launch_bounds__(kThreads) void upsample_three_level_kernel(const UpsampleArgs outer, const UpsampleArgs mid,
                                                                        const UpsampleArgs inner)
{
    __shared__ __attribute__((aligned(16))) float smem[UpsLds<false>::kFloats + kNestScratch];
    const int tile = xcd_contiguous(blockIdx.x, gridDim.x), frame = blockIdx.z;
    if constexpr (DIV == DIV_EXACT_RCP) {
        if (frame_is_hostile(outer.hostile, outer.generation, frame)) {
            upsample_three_level_tile<AOFMT, RTNE, DIV_IEEE>(outer, mid, inner, smem, tile, frame);
            return;
        }
    }
    upsample_three_level_tile<AOFMT, RTNE, DIV>(outer, mid, inner, smem, tile, frame);
}

// Hook of the fused last kernel: puts the four 16-byte depth loads of the carried downsample tile in flight
// inside the upsample tile, before its bilateral phase (A/B against "tile first" and "after the prefetch":
// profiles/r02_ab_v15p..v17p_split_ds*.jsonl).
struct IssueCarriedLoads {
    static constexpr bool kBeforeBilateral = true;
    static constexpr bool kGroupReciprocals = false;     // the kernels that carry a downsample tile are short of registers: A/B +3 %
    static constexpr bool kEstimateR8 = false;           // ... and wait on memory, not on VALU issue: 10 % fewer instructions, +4 us
    static constexpr bool kReuseEstimate = false;
    const DownsampleArgs &d;
    float (&v)[kDsTileH / kDsRowsPerPass][4];
    bool mine;
    int tile, frame;
    __device__ __forceinline__ void issue() const { if (mine) downsample_tile_load<true, true>(d, tile, frame, v); }
    __device__ __forceinline__ void after_prefetch() const { if constexpr (!kBeforeBilateral) issue(); }
    __device__ __forceinline__ void before_bilateral() const { if constexpr (kBeforeBilateral) issue(); }
};


// The same for the lean tile (downsample_lean_load / _finish: wave-uniform row parity, ~8 VALU instructions per texel instead of ~18):
// what the last kernel of a pipelined step carries since round 4 (270 vs 275 us per 16 frames, profiles/r04_ab_fused_lean_tile.jsonl).
struct IssueCarriedLoadsLean {
    static constexpr bool kBeforeBilateral = true;
    // Forms of the bilateral texel (A/B with the whole-tile copy of the phase, profiles/r04_ab_fused_bilateral_forms.jsonl; before that
    // copy existed both lost here): exact sequences 272 us, UNORM8 estimate 257, grouped reciprocals 264, both 256 us per 16 frames.
    static constexpr bool kGroupReciprocals = true;
    static constexpr bool kEstimateR8 = true;
    static constexpr bool kReuseEstimate = false;        // (70 of the 72 VGPRs that seven workgroups per CU allow: reuse spills)
    const DownsampleArgs &d;
    float4v (&q)[kDsTileH / kDsRowsPerPass];
    bool mine, full;
    int tile, frame;
    __device__ __forceinline__ void after_prefetch() const {}
    __device__ __forceinline__ void before_bilateral() const
    {
        if (!mine) return;
        if (full) downsample_lean_load<kDsTileH / kDsRowsPerPass, true>(d, tile, frame, q);
        else downsample_lean_load<kDsTileH / kDsRowsPerPass, false>(d, tile, frame, q);
    }
};

// Upsample.main of this batch carrying the downsample pass of the NEXT batch (meao_prefetch_batch):
// the final upsample is VALU-bound (five exact divides per texel) and leaves HBM idle, the
// downsample is pure streaming with ~2 VALU ops per byte -- inside one kernel the streaming hides
// under the arithmetic of the other resident workgroups instead of costing a pass of its own.
// The downsample tiles (128 x 32 texels) of `d` are spread over this kernel's grid; each workgroup
// streams its share first and then does its upsample tile.
template <int AOFMT, bool RTNE, int DIV>
__global__ __launch_bounds__(kThreads, 7) void upsample_final_with_next_downsample_kernel(const UpsampleArgs a,
                                                                                       const DownsampleArgs d)
{
    __shared__ __attribute__((aligned(16))) float smem[UpsLds<true>::kFloats];
    auto carried_downsample = [&]() {
        const int ds_tiles = d.tiles_x * d.tiles_y;
        const bool vec = d.vec_ok != 0;
        for (int f = blockIdx.z; f < d.frames; f += gridDim.z)
            for (int t = blockIdx.x; t < ds_tiles; t += gridDim.x) {
                if (vec) downsample_tile<RTNE, true, DIV>(d, t, f);
                else downsample_tile<RTNE, false, DIV>(d, t, f);
            }
    };
    // One downsample tile per workgroup (the usual case: both grids tile the same frame) with 16-byte f32
    // loads: its four loads per lane go out after the upsample tile's prefetch wait -- issued earlier they
    // would sit in front of that wait (vmcnt counts in order) -- and are consumed after the bilateral phase.
    const int ds_tiles = d.tiles_x * d.tiles_y;
    const bool split = d.vec_ok != 0 && d.depth_format == MEAO_DEPTH_F32 && gridDim.x >= static_cast<unsigned>(ds_tiles) &&
                       gridDim.z >= static_cast<unsigned>(d.frames);
    if (!split) {       // (the host only moves tiles into a blend pass when the split form applies: tile_begin = 0 here)
        carried_downsample();
        upsample_tile_checked<AOFMT, RTNE, true, DIV>(a, smem, xcd_contiguous(blockIdx.x, gridDim.x), blockIdx.z);
        return;
    }
    // (tiles below d.tile_begin were carried by an earlier launch of this call: a blend pass, MEAO_DEBUG_DS_SHARE_IN_BLEND)
    const bool mine = blockIdx.x >= static_cast<unsigned>(d.tile_begin) && blockIdx.x < static_cast<unsigned>(ds_tiles) &&
                      blockIdx.z < static_cast<unsigned>(d.frames);
    constexpr int kPasses = kDsTileH / kDsRowsPerPass;
    float4v q[kPasses];
    const bool full = (static_cast<int>(blockIdx.x) / d.tiles_x + 1) * kDsTileH <= d.h[0];
    const IssueCarriedLoadsLean issue = {d, q, mine, full, static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.z)};
    upsample_tile_checked<AOFMT, RTNE, true, DIV>(a, smem, xcd_contiguous(blockIdx.x, gridDim.x), blockIdx.z, issue);
    if (mine) {
        if (full) downsample_lean_finish<RTNE, DIV, kPasses, true>(d, blockIdx.x, blockIdx.z, q);
        else downsample_lean_finish<RTNE, DIV, kPasses, false>(d, blockIdx.x, blockIdx.z, q);
    }
    // (loading the carried tile behind the first barrier and finishing it in FRONT of the bilateral phase frees 10 VGPRs there
    // and is 5 % slower: profiles/r03_ab_fused_ds_finished_before_bilateral.jsonl)
}

// Upsample.main_blendout L2 -> L1 carrying the first d.tile_end downsample tiles (per frame) of the NEXT batch: the
// blend passes wait on latency with issue slots and HBM idle, the fused last kernel is short of both
// (MEAO_DEBUG_DS_SHARE_IN_BLEND; the last kernel then starts at d.tile_begin = this launch's tile_end).
template <int AOFMT, bool RTNE, int DIV>
__global__ __launch_bounds__(kThreads) void upsample_blend_with_next_downsample_kernel(const UpsampleArgs a, const DownsampleArgs d)
{
    __shared__ __attribute__((aligned(16))) float smem[UpsLds<false>::kFloats];
    const bool mine = blockIdx.x < static_cast<unsigned>(d.tile_end) && blockIdx.z < static_cast<unsigned>(d.frames);
    float v[kDsTileH / kDsRowsPerPass][4];
    const IssueCarriedLoads issue = {d, v, mine, static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.z)};
    upsample_tile_checked<AOFMT, RTNE, false, DIV>(a, smem, xcd_contiguous(blockIdx.x, gridDim.x), blockIdx.z, issue);
    if (mine) downsample_tile_finish<RTNE, true, DIV>(d, blockIdx.x, blockIdx.z, v);
}

// ------------------------------------------------------------------------------------------
// TiledDepth<level> for the debug views: atlas texel (tx,ty) of slice s is level texel
// (4tx + (s&3), 4ty + (s>>2)) (DS1:69-71,76-78; DS2:38-40,46-48), padded beyond the level.

template <bool RTNE>
__global__ __launch_bounds__(kThreads) void tile_atlas_kernel(const TileAtlasArgs a)
{
    const int n = 16 * a.sw * a.sh;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const int s = i / (a.sw * a.sh), rem = i % (a.sw * a.sh);
        const int ty = rem / a.sw, tx = rem % a.sw;
        const int x = 4 * tx + (s & 3), y = 4 * ty + (s >> 2);
        const float v = (x < a.lw && y < a.lh) ? a.src[static_cast<size_t>(y) * a.lw + x] : a.pad_value;
        a.dst[i] = f32_to_f16_bits<RTNE>(v);
    }
}

// ------------------------------------------------------------------------------------------
// Self-tests: hardware conversions vs a bit-level software model (all inputs).

__device__ uint16_t soft_f32_to_f16(float x, bool rtne)
{
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    const uint32_t sign = (u >> 16) & 0x8000u, absu = u & 0x7fffffffu;
    if (absu >= 0x7f800000u) return static_cast<uint16_t>(sign | (absu == 0x7f800000u ? 0x7c00u : 0x7e00u));
    const int e = static_cast<int>(absu >> 23) - 127;
    const uint32_t m = absu & 0x7fffffu;
    if (e > 15) return static_cast<uint16_t>(sign | (rtne ? 0x7c00u : 0x7bffu));
    uint32_t h, rest, half;
    if (e >= -14) { h = (static_cast<uint32_t>(e + 15) << 10) | (m >> 13); rest = m & 0x1fffu; half = 0x1000u; }
    else if (e >= -25) { const uint32_t full = m | 0x800000u; const int sh = -e - 1; h = full >> sh; rest = full & ((1u << sh) - 1u); half = 1u << (sh - 1); }
    else { h = 0; rest = absu ? 1u : 0u; half = 2u; }
    if (rtne) { if (rest > half || (rest == half && (h & 1u))) h += 1u; if (h >= 0x7c00u) h = 0x7c00u; }
    return static_cast<uint16_t>(sign | h);
}

template <bool RTNE>
__global__ __launch_bounds__(kThreads) void selftest_f16_kernel(unsigned long long *count)
{
    unsigned long long bad = 0;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kThreads;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __builtin_bit_cast(float, static_cast<uint32_t>(i));
        const uint16_t hw = f32_to_f16_bits<RTNE>(x), sw = soft_f32_to_f16(x, RTNE);
        const bool both_nan = (hw & 0x7fffu) > 0x7c00u && (sw & 0x7fffu) > 0x7c00u;
        if (hw != sw && !both_nan) ++bad;
    }
    if (bad) atomicAdd(count, bad);
}

__global__ void selftest_unorm8_decode_kernel(unsigned long long *count)
{
    const uint32_t n = threadIdx.x;   // 256 threads
    const float ref = static_cast<float>(n) / 255.0f;
    if (unorm8_to_f32(n) != ref) atomicAdd(count, 1ull);
}

__global__ __launch_bounds__(kThreads) void selftest_f16_decode_kernel(unsigned long long *count)
{
    const uint32_t b = blockIdx.x * kThreads + threadIdx.x;   // 65536 inputs
    const uint32_t sign = (b & 0x8000u) << 16, e = (b >> 10) & 0x1fu, m = b & 0x3ffu;
    uint32_t ref;
    if (e == 31) ref = sign | 0x7f800000u | (m << 13);
    else if (e == 0) ref = sign | __builtin_bit_cast(uint32_t, static_cast<float>(m) * 5.9604644775390625e-8f);
    else ref = sign | ((e + 112u) << 23) | (m << 13);
    const uint32_t got = __builtin_bit_cast(uint32_t, f16_bits_to_f32(static_cast<uint16_t>(b)));
    const bool both_nan = (got & 0x7fffffffu) > 0x7f800000u && (ref & 0x7fffffffu) > 0x7f800000u;
    if (got != ref && !both_nan) atomicAdd(count, 1ull);
}

// ------------------------------------------------------------------------------------------
// Debug view (AO.cs:787-820): point-sample a buffer (or the 4x4 slice grid of a tiled array,
// Blit.shader:136-155) at the destination texel centres; integer-exact sampling positions.

template <int AOFMT, bool RTNE>
__global__ __launch_bounds__(kThreads) void debug_view_kernel(const DebugViewArgs a)
{
    typedef AoTexel<AOFMT> AO;
    const int64_t n = static_cast<int64_t>(a.w) * a.h;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * kThreads) {
        const int x = static_cast<int>(i % a.w), y = static_cast<int>(i / a.w);
        int sx, sy, sl = 0;
        if (a.slices == 1) {                         // cmd.Blit(rt, _result): uv = (x + 0.5) / W
            sx = static_cast<int>((static_cast<int64_t>(2 * x + 1) * a.sw) / (2 * a.w));
            sy = static_cast<int>((static_cast<int64_t>(2 * y + 1) * a.sh) / (2 * a.h));
        } else {                                     // uv4 = uv * 4: slice = floor(uv4), texel = frac(uv4) * dims
            const int nx = 4 * x + 2, ny = 4 * y + 2;                  // uv4 = n / W
            sl = nx / a.w + 4 * (ny / a.h);
            sx = static_cast<int>((static_cast<int64_t>(nx % a.w) * a.sw) / a.w);
            sy = static_cast<int>((static_cast<int64_t>(ny % a.h) * a.sh) / a.h);
        }
        const size_t at = (static_cast<size_t>(sl) * a.sh + sy) * a.sw + sx;
        float v;
        if (a.src_format == MEAO_FMT_F32) v = static_cast<const float *>(a.src)[at];
        else if (a.src_format == MEAO_FMT_F16) v = f16_bits_to_f32(static_cast<const uint16_t *>(a.src)[at]);
        else v = unorm8_to_f32(static_cast<const uint8_t *>(a.src)[at]);
        static_cast<typename AO::type *>(a.dst)[i] = AO::template encode<RTNE>(v);
    }
}

// ------------------------------------------------------------------------------------------
// Composite (Blit.shader:66-134): pure streaming, 17 bytes per texel (RGBA16F read + write, AO).
// One lane = 4 texels = two 16-byte colour loads/stores + one 4-byte (R8) AO load.

__device__ __forceinline__ uint16_t f32_to_f16_rtne_bits(float x) { return f32_to_f16_bits<true>(x); }

// Texel pair q (texels 2q, 2q+1) of one frame: one 16-byte colour load / store per lane.
template <int AOFMT>
__device__ __forceinline__ void composite_pair(const void *ao_base, void *color_base, void *gbuffer0_base, int64_t pixels,
                                               int32_t mode, int64_t q)
{
    typedef AoTexel<AOFMT> AO;
    typedef typename AO::type ao_t;
    const int64_t p0 = q * 2;
    const bool full = p0 + 1 < pixels;
    const ao_t *ap = static_cast<const ao_t *>(ao_base) + p0;
    float aov[2] = {1.0f, 1.0f};
    if (full) {
        const typename AO::type2 a2 = *reinterpret_cast<const typename AO::type2 *>(ap);
        aov[0] = AO::decode(a2.x); aov[1] = AO::decode(a2.y);
    } else {
        aov[0] = AO::decode(ap[0]);
    }
    uint16_t c[8] = {};
    uint16_t *cp = static_cast<uint16_t *>(color_base) + p0 * 4;
    if (full) {
        const uint4v raw = *reinterpret_cast<const uint4v *>(cp);
        c[0] = raw.x & 0xffffu; c[1] = raw.x >> 16; c[2] = raw.y & 0xffffu; c[3] = raw.y >> 16;
        c[4] = raw.z & 0xffffu; c[5] = raw.z >> 16; c[6] = raw.w & 0xffffu; c[7] = raw.w >> 16;
    } else {
        for (int k = 0; k < 4; ++k) c[k] = cp[k];
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (p0 + e >= pixels) break;
        const float ao = aov[e];
        uint16_t *t = c + 4 * e;
        if (mode == MEAO_COMPOSITE_DEBUG) {                          // pass 3: frag returns ao in every channel
            t[0] = t[1] = t[2] = t[3] = f32_to_f16_rtne_bits(ao);
        } else if (mode == MEAO_COMPOSITE_MULTIPLY) {                // pass 2: dst * src.a
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = f32_to_f16_rtne_bits(f16_bits_to_f32(t[k]) * ao);
        } else {                                                     // pass 1: dst * (1 - src), src = 1 - ao
            const float occ = 1.0f - ao;                             // Blit.shader:84
            const float keep = 1.0f - occ;                           // OneMinusSrcColor / OneMinusSrcAlpha
#pragma unroll
            for (int k = 0; k < 3; ++k) t[k] = f32_to_f16_rtne_bits(f16_bits_to_f32(t[k]) * keep);
            uint8_t *g = static_cast<uint8_t *>(gbuffer0_base) + (p0 + e) * 4 + 3;   // GBuffer0.a = occlusion
            *g = static_cast<uint8_t>(f32_to_unorm8(unorm8_to_f32(*g) * keep));
        }
    }
    if (full) {
        uint4v outv;
        outv.x = c[0] | (static_cast<uint32_t>(c[1]) << 16); outv.y = c[2] | (static_cast<uint32_t>(c[3]) << 16);
        outv.z = c[4] | (static_cast<uint32_t>(c[5]) << 16); outv.w = c[6] | (static_cast<uint32_t>(c[7]) << 16);
        *reinterpret_cast<uint4v *>(cp) = outv;
    } else {
        for (int k = 0; k < 4; ++k) cp[k] = c[k];
    }
}

template <int AOFMT>
__global__ __launch_bounds__(kThreads) void composite_kernel(const CompositeArgs a)
{
    // one lane = 2 texels = one 16-byte colour load/store; consecutive lanes are contiguous
    const int64_t pairs = (a.pixels + 1) / 2;
    for (int64_t q = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; q < pairs;
         q += static_cast<int64_t>(gridDim.x) * kThreads)
        composite_pair<AOFMT>(a.ao, a.color, a.gbuffer0, a.pixels, a.mode, q);
}

// The render pass carrying the composite of frames that an EARLIER call produced (meao_composite_enqueue):
// the composite is pure streaming (17 bytes per texel, as many bytes as the whole AO path) and render
// is VALU-bound with HBM nearly idle, so every render workgroup first streams its share of the
// composite texel pairs and then renders its tile.
// carried composite (multiply mode): two pixel pairs per lane in flight under every texel-loop iteration (three: 0.830 vs 0.834 ms, not kept)
constexpr int kCompositePerIteration = 2;
constexpr int kCompositePairsInLoop = kCompositePerIteration * (kRenTileH / 8);

// Pass 2 of Blit.shader (dst * src.a) for pixel pairs of ONE frame, as the hook of the render texel loop:
// begin(k) issues the 16-byte colour and 2/4-byte AO loads of two pairs, end(k) multiplies and stores them.
// Pair j of a lane is q = (j * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x (a workgroup touches 8 KB
// of contiguous colour per j); j < kCompositePairsInLoop here, the rest in the plain loop before the tile.
template <int AOFMT>
struct CarriedComposite {
    typedef AoTexel<AOFMT> AO;
    const typename AO::type *ao;
    uint16_t *color;
    uint32_t q0, q_step, full_pairs;          // q0 = pair of j = 0; pairs below full_pairs have both pixels
    bool active;
    uint4v col[kCompositePerIteration];
    typedef typename std::conditional<sizeof(typename AO::type) == 1, uint16_t, uint32_t>::type ao_pair_bits;
    uint32_t ao2[kCompositePerIteration];     // two AO texels, undecoded (taken apart in end(), not next to the load)
    __device__ __forceinline__ uint32_t pair_of(int k, int s) const { return q0 + static_cast<uint32_t>(kCompositePerIteration * k + s) * q_step; }
    __device__ __forceinline__ void begin(int k)
    {
        if (!active) return;
#pragma unroll
        for (int s = 0; s < kCompositePerIteration; ++s) {
            const uint32_t q = pair_of(k, s);
            if (q < full_pairs) {
                col[s] = __builtin_nontemporal_load(reinterpret_cast<const uint4v *>(at_byte_offset(color, q * 16u)));
                ao2[s] = *reinterpret_cast<const ao_pair_bits *>(at_byte_offset(ao, q * static_cast<uint32_t>(sizeof(ao_pair_bits))));
            }
        }
        __builtin_amdgcn_sched_barrier(0);        // the loads stay here; their first use is behind the texel arithmetic
    }
    __device__ __forceinline__ void end(int k)
    {
        if (!active) return;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < kCompositePerIteration; ++s) {
            const uint32_t q = pair_of(k, s);
            if (q < full_pairs) {
                constexpr int kAoBits = 8 * sizeof(typename AO::type);
                asm volatile("" : "+v"(ao2[s]));          // opaque here: nothing derived from the loaded word moves up to the load
                const float a0 = AO::decode(static_cast<typename AO::type>(ao2[s] & ((1u << kAoBits) - 1u)));
                const float a1 = AO::decode(static_cast<typename AO::type>(ao2[s] >> kAoBits));
                const uint32_t w[4] = {col[s].x, col[s].y, col[s].z, col[s].w};
                uint32_t o[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {                      // words 0, 1: pixel 0 (rg, ba); words 2, 3: pixel 1
                    const float m = h < 2 ? a0 : a1;
                    const uint32_t lo = f32_to_f16_rtne_bits(f16_bits_to_f32(static_cast<uint16_t>(w[h] & 0xffffu)) * m);
                    const uint32_t hi = f32_to_f16_rtne_bits(f16_bits_to_f32(static_cast<uint16_t>(w[h] >> 16)) * m);
                    o[h] = lo | (hi << 16);
                }
                __builtin_nontemporal_store(uint4v{o[0], o[1], o[2], o[3]}, reinterpret_cast<uint4v *>(at_byte_offset(color, q * 16u)));
            }
        }
    }
};

template <int AOFMT, bool RTNE, int DIV>
__global__ __launch_bounds__(ren_tile_w(false) * 4, 8) void render_with_composite_kernel(const RenderArgs a,
                                                                                         const CompositeBatchArgs c)
{
    __shared__ __attribute__((aligned(16))) float tile[kRenLdsH * (ren_tile_w(false) + 2 * kRenApron)];
    const int frame = blockIdx.y, block = xcd_contiguous(blockIdx.x, gridDim.x);
    // In-loop form: one composite frame per render frame, multiply mode, frames below 2^28 pairs (32-bit byte offsets)
    const bool in_loop = c.mode == MEAO_COMPOSITE_MULTIPLY && c.frames == static_cast<int32_t>(gridDim.y) &&
                         c.pixels < (int64_t(1) << 29);
    CarriedComposite<AOFMT> carried;
    carried.active = in_loop;
    if (in_loop) {
        const int64_t pairs = (c.pixels + 1) / 2;
        carried.ao = static_cast<const typename AoTexel<AOFMT>::type *>(c.ao[frame]);
        carried.color = static_cast<uint16_t *>(c.color[frame]);
        carried.q_step = gridDim.x * blockDim.x;
        carried.q0 = blockIdx.x * blockDim.x + threadIdx.x;
        carried.full_pairs = static_cast<uint32_t>(c.pixels / 2);
        // what the loop does not take: pairs j >= kCompositePairsInLoop of this lane and the half pair of an odd frame
        for (int64_t q = static_cast<int64_t>(carried.q0) + static_cast<int64_t>(kCompositePairsInLoop) * carried.q_step; q < pairs; q += carried.q_step)
            composite_pair<AOFMT>(c.ao[frame], c.color[frame], c.gbuffer0[frame], c.pixels, c.mode, q);
        if (c.pixels & 1) {     // the half pair at the end of an odd frame: the lane that owns it, if the loop would have had it
            const int64_t last = pairs - 1;
            if (last % carried.q_step == carried.q0 && last / carried.q_step < kCompositePairsInLoop)
                composite_pair<AOFMT>(c.ao[frame], c.color[frame], c.gbuffer0[frame], c.pixels, c.mode, last);
        }
    } else {
        const int64_t pairs = (c.pixels + 1) / 2, total = pairs * c.frames;
        const int64_t stride = static_cast<int64_t>(gridDim.x) * gridDim.y * blockDim.x;
        for (int64_t i = (static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
            const int f = static_cast<int>(i / pairs);
            composite_pair<AOFMT>(c.ao[f], c.color[f], c.gbuffer0[f], c.pixels, c.mode, i - f * pairs);
        }
    }
    if constexpr (DIV == DIV_EXACT_RCP) {
        if (frame_is_hostile(a.hostile, a.generation, frame)) {
            render_tile<AOFMT, RTNE, DIV_IEEE, false>(a, tile, frame, block, carried);
            return;
        }
    }
    render_tile<AOFMT, RTNE, DIV, false>(a, tile, frame, block, carried);
}

// which = 4: rcp_strict, 5: div_const<3>, div_const<9>, 6: div_strict on hashed operand pairs
__device__ __forceinline__ bool in_exact_range(float x, float lo, float hi)
{
    const float ax = __builtin_fabsf(x);
    return ax >= lo && ax <= hi;
}

__global__ __launch_bounds__(kThreads) void selftest_div_kernel(unsigned long long *count, int which)
{
    unsigned long long bad = 0;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kThreads;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __builtin_bit_cast(float, static_cast<uint32_t>(i));
        if (which == 4) {
            if (!in_exact_range(x, 0x1p-100f, 0x1p100f)) continue;
            const float exact = 1.0f / x;
            bad += rcp_strict<DIV_EXACT_RCP>(x) != exact;
            // the uncorrected v_rcp_f32 is at most one ulp from the correctly rounded reciprocal (what bilateral_upsample_r8's bound uses)
            const int32_t ulps = static_cast<int32_t>(__builtin_bit_cast(uint32_t, __builtin_amdgcn_rcpf(x))) -
                                 static_cast<int32_t>(__builtin_bit_cast(uint32_t, exact));
            bad += ulps < -1 || ulps > 1;
        } else if (which == 7) {
            // bilateral_upsample_r8 against the UNORM8 code of the exact chain on hashed operands: depths in (0, 1], the four
            // low-res depths within a random relative distance (2^-24 .. 2) of the hi-res one, AO values in [0, 1] (one in four
            // a UNORM8 code, as the unblurred taps are), tolerance and noise constants across the ranges the exact mode accepts
            uint32_t h = static_cast<uint32_t>(i) * 2654435761u + 0x9E3779B9u;
            auto next = [&h]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return h; };
            auto unit = [&next]() { return static_cast<float>(next() >> 8) * 0x1p-24f; };                  // [0, 1)
            auto pow2 = [&next](int lo, int hi) { return __builtin_bit_cast(float, static_cast<uint32_t>(127 + lo + static_cast<int>(next() % static_cast<uint32_t>(hi - lo + 1))) << 23); };
            const float hd = pow2(-12, -1) * (1.0f + unit());
            float d[4], a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = hd * (1.0f + (unit() - 0.5f) * pow2(-23, 1));
                if (!(d[k] >= 0x1p-24f)) d[k] = 0x1p-24f;
                a[k] = (next() & 3u) == 0 ? unorm8_to_f32(next() & 255u) : unit();
            }
            const float hi_ao = (next() & 1u) ? 1.0f : unorm8_to_f32(next() & 255u);
            const BilateralConsts k(pow2(-44, 20), pow2(-30, 50));
            const uint32_t want = f32_to_unorm8(bilateral_upsample<DIV_EXACT_RCP>(hd, hi_ao, d[0], d[1], d[2], d[3], a[0], a[1], a[2], a[3], k));
            bad += bilateral_upsample_r8<false, false>(hd, hi_ao, d, a, k) != want;
            bad += bilateral_upsample_r8<true, false>(hd, hi_ao, d, a, k) != want;
            bad += bilateral_upsample_r8<false, true>(hd, hi_ao, d, a, k) != want;
            bad += bilateral_upsample_r8<true, true>(hd, hi_ao, d, a, k) != want;
        } else if (which == 5) {
            if (!in_exact_range(x, 0x1p-100f, 0x1p100f)) continue;
            bad += div_const<DIV_EXACT_RCP, 3>(x) != 3.0f / x;
            bad += div_const<DIV_EXACT_RCP, 9>(x) != 9.0f / x;
        } else {
            if (!in_exact_range(x, 0x1p-60f, 0x1p60f)) continue;
            uint32_t h = static_cast<uint32_t>(i) * 2654435761u + 0x9E3779B9u;
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const uint32_t ea = 127u - 60u + (h >> 24) % 121u;     // |a| in [2^-60, 2^60]
            const float av = __builtin_bit_cast(float, (h & 0x807fffffu) | (ea << 23));
            bad += div_strict<DIV_EXACT_RCP>(av, x) != av / x;
            bad += div_strict<DIV_EXACT_RCP>(0.0f, x) != 0.0f / x;
        }
    }
    if (bad) atomicAdd(count, bad);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers

hipError_t launch_downsample(const DownsampleArgs &a, int frames, hipStream_t s)
{
    const dim3 grid(a.tiles_x * a.tiles_y, 1, frames), block(kThreads);
    const bool vec = a.vec_ok != 0;
    if (a.row_passes == 1) {
        if (a.f16_rtne) {
            if (vec) downsample_small_kernel<true, true, DIV_IEEE><<<grid, block, 0, s>>>(a);
            else downsample_small_kernel<true, false, DIV_IEEE><<<grid, block, 0, s>>>(a);
        } else if (a.exact_rcp_div == 2) {
            if (vec) downsample_small_kernel<false, true, DIV_FAST><<<grid, block, 0, s>>>(a);
            else downsample_small_kernel<false, false, DIV_FAST><<<grid, block, 0, s>>>(a);
        } else if (a.exact_rcp_div) {
            if (vec) downsample_small_kernel<false, true, DIV_EXACT_RCP><<<grid, block, 0, s>>>(a);
            else downsample_small_kernel<false, false, DIV_EXACT_RCP><<<grid, block, 0, s>>>(a);
        } else {
            if (vec) downsample_small_kernel<false, true, DIV_IEEE><<<grid, block, 0, s>>>(a);
            else downsample_small_kernel<false, false, DIV_IEEE><<<grid, block, 0, s>>>(a);
        }
        return hipGetLastError();
    }
    if (a.f16_rtne) {
        if (vec) downsample_kernel<true, true, DIV_IEEE><<<grid, block, 0, s>>>(a);
        else downsample_kernel<true, false, DIV_IEEE><<<grid, block, 0, s>>>(a);
    } else if (a.exact_rcp_div == 2) {
        if (vec) downsample_kernel<false, true, DIV_FAST><<<grid, block, 0, s>>>(a);
        else downsample_kernel<false, false, DIV_FAST><<<grid, block, 0, s>>>(a);
    } else if (a.exact_rcp_div) {
        if (vec) downsample_kernel<false, true, DIV_EXACT_RCP><<<grid, block, 0, s>>>(a);
        else downsample_kernel<false, false, DIV_EXACT_RCP><<<grid, block, 0, s>>>(a);
    } else {
        if (vec) downsample_kernel<false, true, DIV_IEEE><<<grid, block, 0, s>>>(a);
        else downsample_kernel<false, false, DIV_IEEE><<<grid, block, 0, s>>>(a);
    }
    return hipGetLastError();
}

template <int PASSES, bool PAD>
static void launch_downsample_side_t(const DownsampleArgs &a, dim3 grid, hipStream_t s)
{
    if (a.f16_rtne) downsample_side_kernel<true, DIV_IEEE, PASSES, PAD><<<grid, dim3(kThreads), 0, s>>>(a);
    else if (a.exact_rcp_div == 2) downsample_side_kernel<false, DIV_FAST, PASSES, PAD><<<grid, dim3(kThreads), 0, s>>>(a);
    else if (a.exact_rcp_div) downsample_side_kernel<false, DIV_EXACT_RCP, PASSES, PAD><<<grid, dim3(kThreads), 0, s>>>(a);
    else downsample_side_kernel<false, DIV_IEEE, PASSES, PAD><<<grid, dim3(kThreads), 0, s>>>(a);
}

// a.row_passes in {4, 8, 16} (a.tiles_y counted in tiles of 8 * row_passes rows); f32 depth, 16-byte aligned rows only
hipError_t launch_downsample_side(const DownsampleArgs &a, int frames, bool pad_vgprs, hipStream_t s)
{
    if (a.vec_ok == 0 || a.depth_format != MEAO_DEPTH_F32) return hipErrorInvalidValue;
    const dim3 grid(a.tiles_x * a.tiles_y, 1, frames);
    switch (a.row_passes) {
    case 4: pad_vgprs ? launch_downsample_side_t<4, true>(a, grid, s) : launch_downsample_side_t<4, false>(a, grid, s); break;
    case 8: pad_vgprs ? launch_downsample_side_t<8, true>(a, grid, s) : launch_downsample_side_t<8, false>(a, grid, s); break;
    case 16: pad_vgprs ? launch_downsample_side_t<16, true>(a, grid, s) : launch_downsample_side_t<16, false>(a, grid, s); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// WIDE selects render_wide_kernel; the (AOFMT, RTNE, DIV, EXH) choice is the same for both.
template <bool WIDE, int AOFMT, bool RTNE, int DIV>
static void launch_render_t(const RenderArgs &a, dim3 grid, hipStream_t s)
{
    const dim3 block(WIDE ? kThreads : ren_tile_w(a.exhaustive != 0) * 4);
    if constexpr (WIDE) {
        if (a.exhaustive) render_wide_kernel<AOFMT, RTNE, DIV, true><<<grid, block, 0, s>>>(a);
        else render_wide_kernel<AOFMT, RTNE, DIV, false><<<grid, block, 0, s>>>(a);
    } else {
        if (a.exhaustive) render_kernel<AOFMT, RTNE, DIV, true><<<grid, block, 0, s>>>(a);
        else if (a.tile_h == kRenTileHSmall) render_small_kernel<AOFMT, RTNE, DIV><<<grid, block, 0, s>>>(a);
        else render_kernel<AOFMT, RTNE, DIV, false><<<grid, block, 0, s>>>(a);
    }
}

template <bool WIDE>
static hipError_t launch_render_any(const RenderArgs &a, int ao_format, int frames, hipStream_t s)
{
    const dim3 grid(a.blocks_per_frame, frames, 1);
    if (ao_format == MEAO_AO_R8) {
        if (a.f16_rtne) launch_render_t<WIDE, MEAO_AO_R8, true, DIV_IEEE>(a, grid, s);
        else if (a.exact_rcp_div == 2) launch_render_t<WIDE, MEAO_AO_R8, false, DIV_FAST>(a, grid, s);
        else if (a.exact_rcp_div) launch_render_t<WIDE, MEAO_AO_R8, false, DIV_EXACT_RCP>(a, grid, s);
        else launch_render_t<WIDE, MEAO_AO_R8, false, DIV_IEEE>(a, grid, s);
    } else {
        if (a.f16_rtne) launch_render_t<WIDE, MEAO_AO_F16, true, DIV_IEEE>(a, grid, s);
        else if (a.exact_rcp_div == 2) launch_render_t<WIDE, MEAO_AO_F16, false, DIV_FAST>(a, grid, s);
        else if (a.exact_rcp_div) launch_render_t<WIDE, MEAO_AO_F16, false, DIV_EXACT_RCP>(a, grid, s);
        else launch_render_t<WIDE, MEAO_AO_F16, false, DIV_IEEE>(a, grid, s);
    }
    return hipGetLastError();
}

hipError_t launch_render(const RenderArgs &a, int ao_format, int frames, hipStream_t s)
{
    return launch_render_any<false>(a, ao_format, frames, s);
}

template <int AOFMT, bool RTNE, int DIV>
static void launch_render_composite_t(const RenderArgs &a, const CompositeBatchArgs &c, dim3 grid, hipStream_t s)
{
    render_with_composite_kernel<AOFMT, RTNE, DIV><<<grid, dim3(ren_tile_w(false) * 4), 0, s>>>(a, c);
}

hipError_t launch_render_with_composite(const RenderArgs &a, const CompositeBatchArgs &c, int ao_format, int frames, hipStream_t s)
{
    if (a.exhaustive) return hipErrorInvalidValue;     // the 68-sample variant keeps its own launch; the caller flushes instead
    const dim3 grid(a.blocks_per_frame, frames, 1);
    if (ao_format == MEAO_AO_R8) {
        if (a.f16_rtne) launch_render_composite_t<MEAO_AO_R8, true, DIV_IEEE>(a, c, grid, s);
        else if (a.exact_rcp_div == 2) launch_render_composite_t<MEAO_AO_R8, false, DIV_FAST>(a, c, grid, s);
        else if (a.exact_rcp_div) launch_render_composite_t<MEAO_AO_R8, false, DIV_EXACT_RCP>(a, c, grid, s);
        else launch_render_composite_t<MEAO_AO_R8, false, DIV_IEEE>(a, c, grid, s);
    } else {
        if (a.f16_rtne) launch_render_composite_t<MEAO_AO_F16, true, DIV_IEEE>(a, c, grid, s);
        else if (a.exact_rcp_div == 2) launch_render_composite_t<MEAO_AO_F16, false, DIV_FAST>(a, c, grid, s);
        else if (a.exact_rcp_div) launch_render_composite_t<MEAO_AO_F16, false, DIV_EXACT_RCP>(a, c, grid, s);
        else launch_render_composite_t<MEAO_AO_F16, false, DIV_IEEE>(a, c, grid, s);
    }
    return hipGetLastError();
}

hipError_t launch_render_wide(const RenderArgs &a, int ao_format, int frames, hipStream_t s)
{
    return launch_render_any<true>(a, ao_format, frames, s);
}

template <int AOFMT, bool RTNE, int DIV>
static void launch_upsample_t(const UpsampleArgs &a, bool final_pass, dim3 grid, hipStream_t s)
{
    if (final_pass && a.tile_h == kUpsTileHSmall) upsample_final_small_kernel<AOFMT, RTNE, DIV><<<grid, dim3(kThreads), 0, s>>>(a);
    else if (final_pass) upsample_kernel<AOFMT, RTNE, true, DIV><<<grid, dim3(kThreads), 0, s>>>(a);
    else upsample_kernel<AOFMT, RTNE, false, DIV><<<grid, dim3(kThreads), 0, s>>>(a);
}

hipError_t launch_upsample(const UpsampleArgs &a, int ao_format, bool hi_depth_f16, int frames, hipStream_t s)
{
    const dim3 grid(a.tiles_x * a.tiles_y, 1, frames);
    // exact_rcp_div is only ever set together with RTZ depth storage (no inf operands)
    if (ao_format == MEAO_AO_R8) {
        if (a.f16_rtne) launch_upsample_t<MEAO_AO_R8, true, DIV_IEEE>(a, hi_depth_f16, grid, s);
        else if (a.exact_rcp_div == 2) launch_upsample_t<MEAO_AO_R8, false, DIV_FAST>(a, hi_depth_f16, grid, s);
        else if (a.exact_rcp_div) launch_upsample_t<MEAO_AO_R8, false, DIV_EXACT_RCP>(a, hi_depth_f16, grid, s);
        else launch_upsample_t<MEAO_AO_R8, false, DIV_IEEE>(a, hi_depth_f16, grid, s);
    } else {
        if (a.f16_rtne) launch_upsample_t<MEAO_AO_F16, true, DIV_IEEE>(a, hi_depth_f16, grid, s);
        else if (a.exact_rcp_div == 2) launch_upsample_t<MEAO_AO_F16, false, DIV_FAST>(a, hi_depth_f16, grid, s);
        else if (a.exact_rcp_div) launch_upsample_t<MEAO_AO_F16, false, DIV_EXACT_RCP>(a, hi_depth_f16, grid, s);
        else launch_upsample_t<MEAO_AO_F16, false, DIV_IEEE>(a, hi_depth_f16, grid, s);
    }
    return hipGetLastError();
}

template <int AOFMT, bool RTNE, int DIV>
static void launch_upsample_two_level_t(const UpsampleArgs &outer, const UpsampleArgs &inner, dim3 grid, hipStream_t s)
{
    upsample_two_level_kernel<AOFMT, RTNE, DIV><<<grid, dim3(kThreads), 0, s>>>(outer, inner);
}

hipError_t launch_upsample_two_level(const UpsampleArgs &outer, const UpsampleArgs &inner, int ao_format, int frames, hipStream_t s)
{
    const dim3 grid(outer.tiles_x * outer.tiles_y, 1, frames);
    if (ao_format == MEAO_AO_R8) {
        if (outer.f16_rtne) launch_upsample_two_level_t<MEAO_AO_R8, true, DIV_IEEE>(outer, inner, grid, s);
        else if (outer.exact_rcp_div == 2) launch_upsample_two_level_t<MEAO_AO_R8, false, DIV_FAST>(outer, inner, grid, s);
        else if (outer.exact_rcp_div) launch_upsample_two_level_t<MEAO_AO_R8, false, DIV_EXACT_RCP>(outer, inner, grid, s);
        else launch_upsample_two_level_t<MEAO_AO_R8, false, DIV_IEEE>(outer, inner, grid, s);
    } else {
        if (outer.f16_rtne) launch_upsample_two_level_t<MEAO_AO_F16, true, DIV_IEEE>(outer, inner, grid, s);
        else if (outer.exact_rcp_div == 2) launch_upsample_two_level_t<MEAO_AO_F16, false, DIV_FAST>(outer, inner, grid, s);
        else if (outer.exact_rcp_div) launch_upsample_two_level_t<MEAO_AO_F16, false, DIV_EXACT_RCP>(outer, inner, grid, s);
        else launch_upsample_two_level_t<MEAO_AO_F16, false, DIV_IEEE>(outer, inner, grid, s);
    }
    return hipGetLastError();
}

template <int AOFMT, bool RTNE, int DIV>
static void launch_upsample_three_level_t(const UpsampleArgs &outer, const UpsampleArgs &mid, const UpsampleArgs &inner, dim3 grid,
                                          hipStream_t s)
{
    upsample_three_level_kernel<AOFMT, RTNE, DIV><<<grid, dim3(kThreads), 0, s>>>(outer, mid, inner);
}

hipError_t launch_upsample_three_level(const UpsampleArgs &outer, const UpsampleArgs &mid, const UpsampleArgs &inner, int ao_format,
                                       int frames, hipStream_t s)
{
    const dim3 grid(outer.tiles_x * outer.tiles_y, 1, frames);
    if (ao_format == MEAO_AO_R8) {
        if (outer.f16_rtne) launch_upsample_three_level_t<MEAO_AO_R8, true, DIV_IEEE>(outer, mid, inner, grid, s);
        else if (outer.exact_rcp_div == 2) launch_upsample_three_level_t<MEAO_AO_R8, false, DIV_FAST>(outer, mid, inner, grid, s);
        else if (outer.exact_rcp_div) launch_upsample_three_level_t<MEAO_AO_R8, false, DIV_EXACT_RCP>(outer, mid, inner, grid, s);
        else launch_upsample_three_level_t<MEAO_AO_R8, false, DIV_IEEE>(outer, mid, inner, grid, s);
    } else {
        if (outer.f16_rtne) launch_upsample_three_level_t<MEAO_AO_F16, true, DIV_IEEE>(outer, mid, inner, grid, s);
        else if (outer.exact_rcp_div == 2) launch_upsample_three_level_t<MEAO_AO_F16, false, DIV_FAST>(outer, mid, inner, grid, s);
        else if (outer.exact_rcp_div) launch_upsample_three_level_t<MEAO_AO_F16, false, DIV_EXACT_RCP>(outer, mid, inner, grid, s);
        else launch_upsample_three_level_t<MEAO_AO_F16, false, DIV_IEEE>(outer, mid, inner, grid, s);
    }
    return hipGetLastError();
}

template <int AOFMT, bool RTNE, int DIV>
static void launch_upsample_fused_t(const UpsampleArgs &a, const DownsampleArgs &d, dim3 grid, hipStream_t s)
{
    upsample_final_with_next_downsample_kernel<AOFMT, RTNE, DIV><<<grid, dim3(kThreads), 0, s>>>(a, d);
}

hipError_t launch_upsample_final_with_downsample(const UpsampleArgs &a, const DownsampleArgs &d, int ao_format,
                                                 int frames, hipStream_t s)
{
    const dim3 grid(a.tiles_x * a.tiles_y, 1, frames);
    if (ao_format == MEAO_AO_R8) {
        if (a.f16_rtne) launch_upsample_fused_t<MEAO_AO_R8, true, DIV_IEEE>(a, d, grid, s);
        else if (a.exact_rcp_div == 2) launch_upsample_fused_t<MEAO_AO_R8, false, DIV_FAST>(a, d, grid, s);
        else if (a.exact_rcp_div) launch_upsample_fused_t<MEAO_AO_R8, false, DIV_EXACT_RCP>(a, d, grid, s);
        else launch_upsample_fused_t<MEAO_AO_R8, false, DIV_IEEE>(a, d, grid, s);
    } else {
        if (a.f16_rtne) launch_upsample_fused_t<MEAO_AO_F16, true, DIV_IEEE>(a, d, grid, s);
        else if (a.exact_rcp_div == 2) launch_upsample_fused_t<MEAO_AO_F16, false, DIV_FAST>(a, d, grid, s);
        else if (a.exact_rcp_div) launch_upsample_fused_t<MEAO_AO_F16, false, DIV_EXACT_RCP>(a, d, grid, s);
        else launch_upsample_fused_t<MEAO_AO_F16, false, DIV_IEEE>(a, d, grid, s);
    }
    return hipGetLastError();
}

template <int AOFMT, bool RTNE, int DIV>
static void launch_upsample_blend_ds_t(const UpsampleArgs &a, const DownsampleArgs &d, dim3 grid, hipStream_t s)
{
    upsample_blend_with_next_downsample_kernel<AOFMT, RTNE, DIV><<<grid, dim3(kThreads), 0, s>>>(a, d);
}

hipError_t launch_upsample_blend_with_downsample(const UpsampleArgs &a, const DownsampleArgs &d, int ao_format, int frames, hipStream_t s)
{
    const dim3 grid(a.tiles_x * a.tiles_y, 1, frames);
    if (d.vec_ok == 0 || d.depth_format != MEAO_DEPTH_F32 || d.tile_end > static_cast<int>(grid.x) || d.frames > frames)
        return hipErrorInvalidValue;      // the caller checks the same conditions before it moves tiles here
    if (ao_format == MEAO_AO_R8) {
        if (a.f16_rtne) launch_upsample_blend_ds_t<MEAO_AO_R8, true, DIV_IEEE>(a, d, grid, s);
        else if (a.exact_rcp_div == 2) launch_upsample_blend_ds_t<MEAO_AO_R8, false, DIV_FAST>(a, d, grid, s);
        else if (a.exact_rcp_div) launch_upsample_blend_ds_t<MEAO_AO_R8, false, DIV_EXACT_RCP>(a, d, grid, s);
        else launch_upsample_blend_ds_t<MEAO_AO_R8, false, DIV_IEEE>(a, d, grid, s);
    } else {
        if (a.f16_rtne) launch_upsample_blend_ds_t<MEAO_AO_F16, true, DIV_IEEE>(a, d, grid, s);
        else if (a.exact_rcp_div == 2) launch_upsample_blend_ds_t<MEAO_AO_F16, false, DIV_FAST>(a, d, grid, s);
        else if (a.exact_rcp_div) launch_upsample_blend_ds_t<MEAO_AO_F16, false, DIV_EXACT_RCP>(a, d, grid, s);
        else launch_upsample_blend_ds_t<MEAO_AO_F16, false, DIV_IEEE>(a, d, grid, s);
    }
    return hipGetLastError();
}

hipError_t launch_tile_atlas(const TileAtlasArgs &a, hipStream_t s)
{
    const int n = 16 * a.sw * a.sh;
    const int blocks = (n + kThreads - 1) / kThreads;
    if (a.f16_rtne) tile_atlas_kernel<true><<<dim3(blocks < 4096 ? blocks : 4096), dim3(kThreads), 0, s>>>(a);
    else tile_atlas_kernel<false><<<dim3(blocks < 4096 ? blocks : 4096), dim3(kThreads), 0, s>>>(a);
    return hipGetLastError();
}

hipError_t launch_debug_view(const DebugViewArgs &a, int ao_format, hipStream_t s)
{
    const int64_t n = static_cast<int64_t>(a.w) * a.h;
    const dim3 grid(static_cast<int>(std::min<int64_t>((n + kThreads - 1) / kThreads, 256 * 32))), block(kThreads);
    if (ao_format == MEAO_AO_R8) {
        if (a.f16_rtne) debug_view_kernel<MEAO_AO_R8, true><<<grid, block, 0, s>>>(a);
        else debug_view_kernel<MEAO_AO_R8, false><<<grid, block, 0, s>>>(a);
    } else {
        if (a.f16_rtne) debug_view_kernel<MEAO_AO_F16, true><<<grid, block, 0, s>>>(a);
        else debug_view_kernel<MEAO_AO_F16, false><<<grid, block, 0, s>>>(a);
    }
    return hipGetLastError();
}

hipError_t launch_composite(const CompositeArgs &a, int ao_format, hipStream_t s)
{
    const int64_t pairs = (a.pixels + 1) / 2;
    const int blocks = static_cast<int>(std::min<int64_t>((pairs + kThreads - 1) / kThreads, 256 * 32));
    if (ao_format == MEAO_AO_R8) composite_kernel<MEAO_AO_R8><<<dim3(blocks), dim3(kThreads), 0, s>>>(a);
    else composite_kernel<MEAO_AO_F16><<<dim3(blocks), dim3(kThreads), 0, s>>>(a);
    return hipGetLastError();
}

hipError_t launch_selftest(int which, unsigned long long *count, hipStream_t s)
{
    switch (which) {
    case 0: selftest_f16_kernel<false><<<dim3(4096), dim3(kThreads), 0, s>>>(count); break;
    case 1: selftest_f16_kernel<true><<<dim3(4096), dim3(kThreads), 0, s>>>(count); break;
    case 2: selftest_unorm8_decode_kernel<<<dim3(1), dim3(256), 0, s>>>(count); break;
    case 3: selftest_f16_decode_kernel<<<dim3(65536 / kThreads), dim3(kThreads), 0, s>>>(count); break;
    case 4: case 5: case 6: case 7: selftest_div_kernel<<<dim3(4096), dim3(kThreads), 0, s>>>(count, which); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace meao
