// meao_k_misc.hip -- LinearDepth / atlas rebuild for the debug views, debug views, composite, device self-tests.
#include "meao_dev_upsample.hpp"
#include "meao_dev_downsample.hpp"
#include "meao_dev_composite.hpp"

namespace meao {
namespace {

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
// LinearDepth for debug id 1: LinearZ[st] = Linearize(Depth[st]) through the HalfUAV store (DS1:37-48, AO.cs:454).  The hot
// path never builds the buffer (the full-resolution upsample evaluates the same expression per texel); here every texel is
// divided with IEEE '/', which the exact reciprocal sequences equal wherever they are used.
template <bool RTNE>
__global__ __launch_bounds__(kThreads) void linear_depth_kernel(const LinearDepthArgs a)
{
    const float sky_depth = a.reversed_z != 0 ? 0.0f : 1.0f;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < a.pixels; i += static_cast<int64_t>(gridDim.x) * kThreads)
        a.dst[i] = f32_to_f16_bits<RTNE>(linearize<DIV_IEEE>(raw_depth_texel(a.depth, a.depth_format, static_cast<size_t>(i)), a.zp0, a.zp1, sky_depth));
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

template <int AOFMT>
__global__ __launch_bounds__(kThreads) void composite_kernel(const CompositeArgs a)
{
    // one lane = 2 texels = one 16-byte colour load/store; consecutive lanes are contiguous
    const int64_t pairs = (a.pixels + 1) / 2;
    for (int64_t q = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; q < pairs;
         q += static_cast<int64_t>(gridDim.x) * kThreads)
        composite_pair<AOFMT>(a.ao, a.color, a.gbuffer0, a.pixels, a.mode, q);
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
            bad += bilateral_upsample_r8<false, false, true>(hd, hi_ao, d, a, k) != want;        // three reciprocals per texel (round 5)
            bad += bilateral_upsample_r8<true, false, true>(hd, hi_ao, d, a, k) != want;
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

hipError_t launch_tile_atlas(const TileAtlasArgs &a, hipStream_t s)
{
    const int n = 16 * a.sw * a.sh;
    const int blocks = (n + kThreads - 1) / kThreads;
    if (a.f16_rtne) tile_atlas_kernel<true><<<dim3(blocks < 4096 ? blocks : 4096), dim3(kThreads), 0, s>>>(a);
    else tile_atlas_kernel<false><<<dim3(blocks < 4096 ? blocks : 4096), dim3(kThreads), 0, s>>>(a);
    return hipGetLastError();
}

hipError_t launch_linear_depth(const LinearDepthArgs &a, hipStream_t s)
{
    const dim3 grid(static_cast<int>(std::min<int64_t>((a.pixels + kThreads - 1) / kThreads, 256 * 32))), block(kThreads);
    if (a.f16_rtne) linear_depth_kernel<true><<<grid, block, 0, s>>>(a);
    else linear_depth_kernel<false><<<grid, block, 0, s>>>(a);
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
