// meao_dev_upsample.hpp -- the upsample tile (Upsample.main / main_blendout [/ main_premin*]): blur runs, bilateral forms, window loads, upsample_tile.
#pragma once

#include "meao_dev.hpp"
#ifndef MEAO_X_NO_REDO
#define MEAO_X_NO_REDO 0
#endif
#include "meao_dev_downsample.hpp"      // linearize, nice_denominators, raw_depth_texel: HiResDB of the final pass

namespace meao {
namespace {

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
// with every division replaced by dividend * v_rcp_f32 (and the weights' constant factors folded into fused multiply-adds, below)
// differs from q by at most 31 u relatively (u = 2^-24), both chains measured against the real-number value:
//   v_rcp_f32 is within one ulp of the correctly rounded reciprocal (meao_selftest(4): every binary32 in range), i.e. within
//             1.5 ulp = 3u of the true one;
//   the sums  have non-negative terms only, so a sum inherits the largest relative error of its terms plus one u per rounding.
//             Estimate: r1 + r3 and r2 + noise 4u, fma(3, ., .) 5u, total = fma(9, r0, .) 6u; a_i r_i 4u, fma(a3, r3, a1 r1) 5u,
//             fma(a2, r2, noise) 4u, fma(3, ., .) 6u, sum = fma(9, a0 r0, .) 7u, times hi_ao 8u; rcp(total) 6u + 3u, the product
//             u: q~ is within 18u.  Exact chain (one correctly rounded operation each): weights u, total u + 4u, weighted sum
//             times hi_ao u + 6u, quotient u: q is within 13u.  (The round-2..5 form, factors applied to the reciprocals first,
//             summed like the reference: 22u + 13u = 35u; the three-reciprocal form PAIRED: 22u + 13u = 35u with these sums.)
// The weighted average times hi_ao is at most 1 (+ rounding), so the estimate is off by < 4.8e-4 of a code; the reference's
// two roundings in the conversion and the fused one of the estimate add < 2.3e-5.  If v~ = fma(sat(q~), 255, 0.5) is further
// than kR8Margin = 2^-10 (1.7 x that bound) from an integer, floor(v~) IS the reference's code.  Otherwise -- 2^-9 of the texels
// of a noisy frame, none where the AO is flat (q~ = 1 -> v~ = 255.5) -- the lane runs the exact sequence.  The agreement of
// estimate and exact code is also checked on the running device for 2^32 hashed operand sets (meao_selftest(7)) and, with
// adversarial 1-ulp reciprocal errors, in numpy by tests/test_r8_estimate_bound.py.
// GROUPED: the four weight reciprocals back to back, as in bilateral_upsample_grouped.
// REUSE: the exact path starts from the estimate's x and 1 / x (14 instructions fewer on that path, 5 - 8 VGPRs more live across
// the branch); without it the whole exact sequence is run again from an opaque copy of the depth, so that nothing of the estimate
// has to stay in registers for the rare path (the nested kernels have none to spare).
// PAIRED (round 5): THREE v_rcp_f32 per texel instead of five -- the reciprocals of a tap pair come from one reciprocal of their
// product, 1/x0 = x1 * rcp(x0 * x1), 1/x1 = x0 * rcp(x0 * x1): a quarter-rate transcendental (8 - 11 issue cycles inside this
// mix) is traded for three full-rate multiplies.  x = |dHi - dLo| + tolerance lies in [2^-44, 2^21] (exact_rcp_div_applicable +
// nice depths), so a product of two lies in [2^-88, 2^42]: no overflow, no denormal anywhere.  Error: product u, v_rcp_f32 3u,
// multiply u = 5u per reciprocal instead of 3u: total 8u, weighted sum times hi_ao 10u, quotient 8u + 3u + u on top = 22u, + 13u
// of the exact chain = 35u = 1.1 * 2^-19 of q, < 5.4e-4 of a code (+ 2.3e-5 for the conversions) -- inside kR8Margin = 9.8e-4.
// The exact path cannot start from these reciprocals (the correctly-rounded guarantee of the Newton step is verified for the
// v_rcp_f32 seed, meao_selftest(4..6), not for a 5u one): PAIRED implies !REUSE.  Checked like the five-reciprocal form:
// tests/test_r8_estimate_bound.py (adversarial errors), meao_selftest(7) (2^32 operand sets on the device).
constexpr float kR8Margin = 0x1p-10f;     // (1.25 * 2^-11, still above the bound, measured the same: profiles/r04_ab_r8_margin.jsonl)

// (v_cvt_pk_u8_f32, which would convert and pack in one instruction, does not truncate like v_cvt_u32_f32: tried in round 4.)
template <bool GROUPED, bool REUSE = false, bool PAIRED = false>
__device__ __forceinline__ uint32_t bilateral_upsample_r8(float hi_depth, float hi_ao, const float (&d)[4], const float (&a)[4],
                                                          const BilateralConsts &k)
{
    static_assert(!(PAIRED && REUSE), "the exact path's Newton steps are only verified for v_rcp_f32 seeds");
    float x[4], r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = __builtin_fabsf(hi_depth - d[i]) + k.tolerance;
    if constexpr (PAIRED) {
        const float p01 = x[0] * x[1], p23 = x[2] * x[3];
        float r01, r23;
        if constexpr (GROUPED) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("v_rcp_f32 %0, %1" : "=v"(r01) : "v"(p01));
            asm volatile("v_rcp_f32 %0, %1" : "=v"(r23) : "v"(p23));
            __builtin_amdgcn_sched_barrier(0);
        } else {
            r01 = __builtin_amdgcn_rcpf(p01);
            r23 = __builtin_amdgcn_rcpf(p23);
        }
        r[0] = x[1] * r01; r[1] = x[0] * r01; r[2] = x[3] * r23; r[3] = x[2] * r23;
    } else if constexpr (GROUPED) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_rcp_f32 %0, %1" : "=v"(r[i]) : "v"(x[i]));
        __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = __builtin_amdgcn_rcpf(x[i]);
    }
    // The estimate only has to stay inside its error bound, not to follow the reference's operation order (round 6): the constant
    // factors of the weights are folded into fused multiply-adds -- total = 9 r0 + 3 (r1 + r3) + (r2 + noise), sum = 9 a0 r0 +
    // 3 (a1 r1 + a3 r3) + (a2 r2 + noise) -- 10 operations instead of 12 (last kernel 250.8 -> 245.6 us, full-resolution pass 191.3 ->
    // 186.0, L2 -> L1 49.9 -> 49.1 per 16 4K frames: profiles/r06_ab_bilateral_estimate_fused_weights.jsonl).
    const float total = mad(k.nine, r[0], mad(k.three, r[1] + r[3], r[2] + k.noise));
    const float sum = mad(k.nine, a[0] * r[0], mad(k.three, mad(a[3], r[3], a[1] * r[1]), mad(a[2], r[2], k.noise)));
    const float q = (hi_ao * sum) * __builtin_amdgcn_rcpf(total);
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

// HiResDB of Upsample.main for four consecutive hi-res texels: LinearZ = f16(Linearize(depth)) (DS1:37-48 through the HalfUAV
// store AO.cs:454, read back by UPS:217-223) from the raw depth texels d[] -- the buffer itself is never materialised (round 6).
// Returns false when a denominator is outside the range the exact reciprocal (and everything downstream of it in the bilateral
// step) is verified for: the caller then redoes its texels with IEEE '/' (final_lane_redo_ieee), which is what the reference
// divides with everywhere.  hd[] of such a quad is not used.
template <bool RTNE, int DIV>
__device__ __forceinline__ bool hi_depth_quad(const float (&d)[4], float zp0, float zp1, float sky_depth, float (&hd)[4])
{
    float lin[4];
    bool nice = true;
    if constexpr (DIV == DIV_EXACT_RCP) {
        float den[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) den[e] = mad(zp0, d[e], zp1);
        nice = nice_denominators(den[0], den[1], den[2], den[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float r = __builtin_amdgcn_rcpf(den[e]);
            lin[e] = mad(mad(-den[e], r, 1.0f), r, r);                      // rcp_strict: DS1:40
        }
        const bool far = (d[0] == sky_depth) | (d[1] == sky_depth) | (d[2] == sky_depth) | (d[3] == sky_depth);
        if (__builtin_expect(far, 0)) {                                     // DS1:41-45
#pragma unroll
            for (int e = 0; e < 4; ++e) lin[e] = d[e] == sky_depth ? 1e5f : lin[e];
            asm volatile("" : "+v"(lin[0]), "+v"(lin[1]), "+v"(lin[2]), "+v"(lin[3]));   // stays a branch: rare lanes only
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) lin[e] = linearize<DIV>(d[e], zp0, zp1, sky_depth);
    }
    const float2v lo = through_f16_pair<RTNE>(lin[0], lin[1]), hi = through_f16_pair<RTNE>(lin[2], lin[3]);
    hd[0] = lo.x; hd[1] = lo.y; hd[2] = hi.x; hd[3] = hi.y;
    return nice;
}

// Four raw depth texels as loaded (undecoded): f32 / UNORM24 fill all four words, the 16-bit formats the first two.
template <bool RAW_F32>
__device__ __forceinline__ void decode_raw_quad(const uint4v &q, int format, float (&d)[4])
{
    if (RAW_F32 || format == MEAO_DEPTH_F32) {
        // (the whole vector is cast: __builtin_bit_cast(float, q.y) on a vector-element lvalue reads the bytes at the address of
        // the VECTOR, i.e. element 0, with this compiler -- found by the first GPU run of round 6)
        const float4v f = __builtin_bit_cast(float4v, q);
        d[0] = f.x; d[1] = f.y; d[2] = f.z; d[3] = f.w;
    } else if (format == MEAO_DEPTH_UNORM24) {
        d[0] = unorm_to_f32<24>(q.x & 0xffffffu); d[1] = unorm_to_f32<24>(q.y & 0xffffffu);
        d[2] = unorm_to_f32<24>(q.z & 0xffffffu); d[3] = unorm_to_f32<24>(q.w & 0xffffffu);
    } else if (format == MEAO_DEPTH_UNORM16) {
        d[0] = unorm_to_f32<16>(q.x & 0xffffu); d[1] = unorm_to_f32<16>(q.x >> 16);
        d[2] = unorm_to_f32<16>(q.y & 0xffffu); d[3] = unorm_to_f32<16>(q.y >> 16);
    } else {
        d[0] = f16_bits_to_f32(static_cast<uint16_t>(q.x & 0xffffu)); d[1] = f16_bits_to_f32(static_cast<uint16_t>(q.x >> 16));
        d[2] = f16_bits_to_f32(static_cast<uint16_t>(q.y & 0xffffu)); d[3] = f16_bits_to_f32(static_cast<uint16_t>(q.y >> 16));
    }
}

// STREAMED: the frame is read once -- past the caches' LRU (the hi-res operands); the apron texels of a from-raw window are lines
// the neighbouring tiles read as THEIR hi-res operands: plain loads.
template <bool RAW_F32, bool STREAMED = true>
__device__ __forceinline__ uint4v load_raw_quad(const void *frame_base, int format, uint32_t texel)
{
    typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
    if (RAW_F32 || format == MEAO_DEPTH_F32 || format == MEAO_DEPTH_UNORM24) {
        const uint4v *p = reinterpret_cast<const uint4v *>(at_byte_offset(static_cast<const char *>(frame_base), texel * 4u));
        return STREAMED ? __builtin_nontemporal_load(p) : *p;
    }
    const uint2v *p = reinterpret_cast<const uint2v *>(at_byte_offset(static_cast<const char *>(frame_base), texel * 2u));
    const uint2v h = STREAMED ? __builtin_nontemporal_load(p) : *p;
    return uint4v{h.x, h.y, 0u, 0u};
}

// HiResDB of ALL the hi-res texels of a lane (QUADS quads of four) at once, from the hoisted raw loads, as packed f16 pairs --
// what a LinearDepth buffer would have handed the bilateral phase, so that phase stays what it was.  One straight-line block:
// the denominators and the Newton steps are packed f32 FMAs (v_pk_fma_f32: two texels per issue), the 4 * QUADS reciprocals
// are issued back to back (an isolated v_rcp_f32 costs the SIMD ~3 cycles more than one behind another), the far-plane select
// runs only in waves that hold a far-plane texel (wave-uniform branch: no exec masking inside straight-line code).
// Hostile depth: instead of range-testing 4 * QUADS denominators, the test reads the RESULT.  With a denominator the exact
// reciprocal is not verified for, the f16 word is inf / NaN / negative / -0 (den = 0, +-inf, NaN, < 0, denormal: the Newton
// step turns the first four into NaN) -- i.e. >= 0x7c00 as an unsigned 16-bit word -- or it is exactly what IEEE '/' gives:
// 65504 for every den in (0, 2^-16) whose reciprocal is finite (round toward zero), +0 for den > 2^24.  A non-negative finite
// HiResDB keeps every operand of the bilateral step inside the exact sequences' ranges (x = |hi - lo| + tolerance with lo a
// nice level texel).  So: one packed unsigned max over the words, one compare per lane.  Returns false = redo the lane (IEEE).
// even[q][j]: Linearize of texels 0 and 2 of quad q BEFORE the f16 store -- for a quad of an even row these are the LowDepth1
// texels under it (DS2x[st >> 1], DS1:64-70); the from-raw window of the full-resolution pass takes its interior from them.
template <bool RTNE, int DIV, bool RAW_F32, int QUADS>
__device__ __forceinline__ bool hi_depth_words(const uint4v (&raw)[QUADS], int format, float zp0, float zp1, float sky_depth,
                                               uint32_t (&words)[QUADS][2], float (&even)[QUADS][2])
{
    float d[QUADS][4];
#pragma unroll
    for (int q = 0; q < QUADS; ++q) decode_raw_quad<RAW_F32>(raw[q], format, d[q]);
    if constexpr (DIV == DIV_EXACT_RCP) {
        static_assert(!RTNE, "exact divisions only ever run with round-toward-zero depth storage");
        float2v den[QUADS][2], r[QUADS][2], lin[QUADS][2];
#pragma unroll
        for (int q = 0; q < QUADS; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) den[q][h] = fma2(splat(zp0), float2v{d[q][2 * h], d[q][2 * h + 1]}, splat(zp1));      // DS1:40
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < QUADS; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float rx, ry;
                asm volatile("v_rcp_f32 %0, %1" : "=v"(rx) : "v"(den[q][h].x));
                asm volatile("v_rcp_f32 %0, %1" : "=v"(ry) : "v"(den[q][h].y));
                r[q][h] = float2v{rx, ry};
            }
        __builtin_amdgcn_sched_barrier(0);
        bool far = false;      // (one min / max chain over the lane's texels and one compare instead of a compare per texel: slower, r06 A/B)
#pragma unroll
        for (int q = 0; q < QUADS; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                lin[q][h] = fma2(fma2(-den[q][h], r[q][h], splat(1.0f)), r[q][h], r[q][h]);                                      // rcp_strict
                far = far | (d[q][2 * h] == sky_depth) | (d[q][2 * h + 1] == sky_depth);
            }
        if (__builtin_amdgcn_ballot_w64(far) != 0) {                                                                              // DS1:41-45
#pragma unroll
            for (int q = 0; q < QUADS; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    lin[q][h].x = d[q][2 * h] == sky_depth ? 1e5f : lin[q][h].x;
                    lin[q][h].y = d[q][2 * h + 1] == sky_depth ? 1e5f : lin[q][h].y;
                }
        }
        ushort2v top = {0, 0};
#pragma unroll
        for (int q = 0; q < QUADS; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const auto p = __builtin_amdgcn_cvt_pkrtz(lin[q][h].x, lin[q][h].y);                                              // the HalfUAV store, AO.cs:454
                words[q][h] = __builtin_bit_cast(uint32_t, p);
                even[q][h] = lin[q][h].x;
                top = __builtin_elementwise_max(top, __builtin_bit_cast(ushort2v, p));
            }
        return top.x < 0x7c00u && top.y < 0x7c00u;
    } else {
#pragma unroll
        for (int q = 0; q < QUADS; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                even[q][h] = linearize<DIV>(d[q][2 * h], zp0, zp1, sky_depth);
                const uint32_t lo = f32_to_f16_bits<RTNE>(even[q][h]);
                const uint32_t hi = f32_to_f16_bits<RTNE>(linearize<DIV>(d[q][2 * h + 1], zp0, zp1, sky_depth));
                words[q][h] = lo | (hi << 16);
            }
        return true;
    }
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
// (the RAW depth texels of the caller's frame in the full-resolution pass, f32 depth + AO in the blend passes).  Issued at the
// top of the tile.
template <int AOFMT, bool FINAL, int TILE_H>
struct UpsLoads {
    typedef AoTexel<AOFMT> AO;
    static constexpr int kItems = 10 * UpsTile<TILE_H>::kRawH, kRounds = (kItems + kThreads - 1) / kThreads, kPasses = TILE_H / 32;
    float4v wd[kRounds];
    typename AO::type4 wa[kRounds];
    uint4v hraw[kPasses][2];            // FINAL: four raw depth texels, undecoded (load_raw_quad)
    uint4v araw[2];                     // FINAL, from-raw window: the eight raw texels over an apron item's four LowDepth1 texels
    float4v hd32[kPasses][2];
    // four AO texels as ONE integer: a <4 x i8> value is split into bytes where it is loaded, which puts the wait for it there
    typedef typename std::conditional<sizeof(typename AO::type4) == 4, uint32_t, uint64_t>::type ao_bits_t;
    ao_bits_t ha[kPasses][2];
};

// i / 10 for 0 <= i < 1024 (the window items: 10 row quads per row) as one multiply and a shift
__device__ __forceinline__ int tenth(int i) { return static_cast<int>((static_cast<uint32_t>(i) * 205u) >> 11); }

template <bool FINAL, int TILE_H>
__device__ __forceinline__ bool ups_tile_is_interior(const UpsampleArgs &a, int tile)
{
    const int LX0 = ((tile % a.tiles_x) * kUpsTileW) >> 1;
    return a.vec_ok != 0 && !a.lo_ao2 && (a.lw & 3) == 0 && LX0 >= 4 && LX0 + 35 < a.lw;
}

// Hi-res operands.  CLAMPED: every lane loads (out-of-frame lanes re-read the frame's last row / quad and never use it),
// so that the code is branch-free and the compiler's s_waitcnt counts stay exact.
template <int AOFMT, bool FINAL, int TILE_H, bool CLAMPED, bool RAW_F32>
__device__ __forceinline__ void ups_issue_hoisted(const UpsampleArgs &a, const HiDepthArgs *hi, int tile, int frame, UpsLoads<AOFMT, FINAL, TILE_H> &L)
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
            if constexpr (FINAL && !CLAMPED) L.hraw[pass][f] = uint4v{0x3f000000u, 0x3f000000u, 0x3f000000u, 0x3f000000u};   // texels past the frame: a clean depth, never used
            if (CLAMPED || (hhx0 < hw && hy < hh)) {
                // texel index in the level (< 2^27): 32-bit byte offsets from the frame's uniform bases (saddr addressing)
                const uint32_t hrow = static_cast<uint32_t>(hy * hw + hhx0);
                if constexpr (FINAL) {
                    L.hraw[pass][f] = load_raw_quad<RAW_F32>(hi->raw[frame], hi->depth_format, hrow);
                } else {
                    L.hd32[pass][f] = *reinterpret_cast<const float4v *>(at_byte_offset(
                        frame_ptr(static_cast<const float *>(a.hi_depth), a.frame_stride, frame), hrow * 4u));
                    L.ha[pass][f] = *reinterpret_cast<const typename UpsLoads<AOFMT, FINAL, TILE_H>::ao_bits_t *>(at_byte_offset(
                        frame_ptr(static_cast<const ao_t *>(a.hi_ao), a.frame_stride, frame), hrow * static_cast<uint32_t>(sizeof(ao_t))));
                }
            }
        }
}

// The same for a tile that lies inside the frame: no clamps, one 24-bit multiply -- the rows of a lane are its first one plus
// multiples of the level's width that are uniform (scalar).
template <int AOFMT, bool FINAL, int TILE_H, bool RAW_F32>
__device__ __forceinline__ void ups_issue_hoisted_inside(const UpsampleArgs &a, const HiDepthArgs *hi, int tile, int frame, UpsLoads<AOFMT, FINAL, TILE_H> &L)
{
    const int tid = thread_index_opaque();
    typedef typename AoTexel<AOFMT>::type ao_t;
    const int HX0 = (tile % a.tiles_x) * kUpsTileW, HY0 = (tile / a.tiles_x) * TILE_H;
    const uint32_t hw = static_cast<uint32_t>(a.hw);
    const uint32_t first = __umul24(static_cast<uint32_t>(HY0 + 2 * (tid >> 4)), hw) + static_cast<uint32_t>(HX0 + 4 * (tid & 15));    // rows, widths < 2^24
#pragma unroll
    for (int pass = 0; pass < TILE_H / 32; ++pass)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const uint32_t hrow = first + static_cast<uint32_t>(32 * pass + f) * hw;
            if constexpr (FINAL) {
                L.hraw[pass][f] = load_raw_quad<RAW_F32>(hi->raw[frame], hi->depth_format, hrow);
            } else {
                L.hd32[pass][f] = *reinterpret_cast<const float4v *>(at_byte_offset(
                    frame_ptr(static_cast<const float *>(a.hi_depth), a.frame_stride, frame), hrow * 4u));
                L.ha[pass][f] = *reinterpret_cast<const typename UpsLoads<AOFMT, FINAL, TILE_H>::ao_bits_t *>(at_byte_offset(
                    frame_ptr(static_cast<const ao_t *>(a.hi_ao), a.frame_stride, frame), hrow * static_cast<uint32_t>(sizeof(ao_t))));
            }
        }
}

// All loads of an interior tile: window first, hi-res operands behind them.  The window comes from L2 (written by the
// previous pass), the hi-res operands of the final pass from HBM; vmcnt retires loads in issue order, so with the hi-res
// loads in front the window wait would last an HBM latency.
template <int AOFMT, bool FINAL, int TILE_H, bool RAW_F32>
__device__ __forceinline__ void ups_issue_interior_loads(const UpsampleArgs &a, const HiDepthArgs *hi, int tile, int frame, UpsLoads<AOFMT, FINAL, TILE_H> &L,
                                                         bool inside)
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
        const int r = tenth(i), k = i - 10 * r;
        const int cy = clampi(LY0 - 3 + r, 0, lh - 1);
        const uint32_t idx = static_cast<uint32_t>(cy * lw + (LX0 - 4 + 4 * k));
        L.wd[round] = *reinterpret_cast<const float4v *>(at_byte_offset(lo_depth, idx * 4u));
        L.wa[round] = *reinterpret_cast<const typename AO::type4 *>(at_byte_offset(lo_ao, idx * static_cast<uint32_t>(sizeof(ao_t))));
    }
    __builtin_amdgcn_sched_barrier(0);          // keep the issue order: window, then hi-res
    if (inside) ups_issue_hoisted_inside<AOFMT, FINAL, TILE_H, RAW_F32>(a, hi, tile, frame, L);      // (wave-uniform; the same loads either way)
    else ups_issue_hoisted<AOFMT, FINAL, TILE_H, true, RAW_F32>(a, hi, tile, frame, L);
    __builtin_amdgcn_sched_barrier(0);
}

// The from-raw window of the full-resolution pass (MEAO_X_LOWDEPTH_FROM_RAW): LowDepth1[X, Y] = Linearize(raw[2X, 2Y]) before the f16
// store (DS1:37-48, 64-70), so a tile that lies inside the frame reads no LowDepth1 at all.  Interior texels (32 x kLowH): from the
// tile's own hi-res operands (hi_depth_words' `even`).  Apron (3 texels all round): 2 * kRawH + 48 items of four texels -- the
// row quads k = 0 and k = 9 of every window row, the quads 1..8 of the three rows above and below -- dealt to lanes 0..123 (91),
// i.e. to the first two waves only: the other two skip the conversion with a scalar branch.
template <int TILE_H>
struct UpsApron {
    typedef UpsTile<TILE_H> T;
    static constexpr int kSide = 2 * T::kRawH, kItems = kSide + 6 * 8;
    static_assert(kItems <= 128, "two waves");
    // item t -> window row r, row quad k (window columns 4k - 1 .. 4k + 2)
    // (bit selects, not ?: -- the compiler turns the nested conditionals into exec-masked branches)
    static __device__ __forceinline__ void item(int t, int &r, int &k)
    {
        const int u = t - kSide, rr = u >> 3;
        const int rows = ~(u >> 31);                                  // all ones for the items of the rows above and below
        const int r_rows = rr + (((2 - rr) >> 31) & T::kLowH), k_rows = 1 + (u & 7);
        const int r_side = t >> 1, k_side = (-(t & 1)) & 9;
        r = (r_rows & rows) | (r_side & ~rows);
        k = (k_rows & rows) | (k_side & ~rows);
    }
};

template <bool FINAL, int TILE_H>
__device__ __forceinline__ bool ups_tile_from_raw(const UpsampleArgs &a, int tile)
{
    if constexpr (!FINAL || !MEAO_X_LOWDEPTH_FROM_RAW) return false;
    return MEAO_X_HOT_PATH_ONLY || (ups_tile_is_interior<FINAL, TILE_H>(a, tile) && (tile / a.tiles_x + 1) * TILE_H <= a.hh);
}

// Loads of a from-raw tile: the AO window (L2: the previous pass wrote it), the apron's raw texels (lines of the neighbouring
// tiles' hi-res operands), the tile's own hi-res operands (HBM) -- in that order, vmcnt retires in issue order.
template <int AOFMT, int TILE_H, bool RAW_F32>
__device__ __forceinline__ void ups_issue_from_raw_loads(const UpsampleArgs &a, const HiDepthArgs *hi, int tile, int frame, UpsLoads<AOFMT, true, TILE_H> &L,
                                                         int apron_r, int apron_k)
{
    const int tid = thread_index_opaque();
    typedef UpsLoads<AOFMT, true, TILE_H> Loads;
    typedef typename Loads::AO AO;
    typedef typename AO::type ao_t;
    const int HX0 = (tile % a.tiles_x) * kUpsTileW;
    const int LX0 = HX0 >> 1, LY0 = ((tile / a.tiles_x) * TILE_H) >> 1;
    const int lw = a.lw, lh = a.lh;
    const ao_t *__restrict__ lo_ao = frame_ptr(static_cast<const ao_t *>(a.lo_ao), a.frame_stride, frame);
#pragma unroll
    for (int round = 0; round < Loads::kRounds; ++round) {
        const int i = min(tid + round * kThreads, Loads::kItems - 1);
        const int r = tenth(i), k = i - 10 * r;
        const int cy = clampi(LY0 - 3 + r, 0, lh - 1);
        const uint32_t idx = static_cast<uint32_t>(cy * lw + (LX0 - 4 + 4 * k));
        L.wa[round] = *reinterpret_cast<const typename AO::type4 *>(at_byte_offset(lo_ao, idx * static_cast<uint32_t>(sizeof(ao_t))));
    }
    {   // every lane loads (lanes past the last item repeat it), so that the code is branch-free
        const int cy = clampi(LY0 - 3 + apron_r, 0, lh - 1);
        const uint32_t at = static_cast<uint32_t>(2 * cy * a.hw + (HX0 - 8 + 8 * apron_k));      // raw texel (2X, 2Y) of LowDepth1 texel (X, Y)
        L.araw[0] = load_raw_quad<RAW_F32, false>(hi->raw[frame], hi->depth_format, at);
        L.araw[1] = load_raw_quad<RAW_F32, false>(hi->raw[frame], hi->depth_format, at + 4u);
    }
    __builtin_amdgcn_sched_barrier(0);          // keep the issue order: window, apron, hi-res
    ups_issue_hoisted_inside<AOFMT, true, TILE_H, RAW_F32>(a, hi, tile, frame, L);      // a from-raw tile lies inside the frame
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
    static constexpr bool kReuseEstimate = true;         // ... whose exact path starts from the estimate's reciprocals (five-reciprocal form only)
    static constexpr bool kPairReciprocals = MEAO_X_BIL_PAIR_RCP != 0;    // three reciprocals per texel (bilateral_upsample_r8<PAIRED>)
    __device__ __forceinline__ void after_prefetch() const {}
    __device__ __forceinline__ void before_bilateral() const {}
};

// RAW_F32 (FINAL): the caller's depth frames are f32 -- no format switch in the code (the other formats take the generic instance)
template <int AOFMT, bool RTNE, bool FINAL, int DIV, bool NESTED = false, typename Hook = NoHook, int TILE_H = ups_tile_h(FINAL), bool RAW_F32 = true>
__device__ __forceinline__ void upsample_tile(const UpsampleArgs &a, float *smem, int tile, int frame, Hook hook = Hook(),
                                              const HiDepthArgs *hi = nullptr)
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
    auto dep_at = [&](int r, int c) __attribute__((always_inline)) -> float & { return s_dep[(r - kDep0) * kDepPitch + (c - kDep0)]; };
    auto dep_kept = [&](int r, int c) __attribute__((always_inline)) { return !FINAL || (r >= kDep0 && r < kDep0 + kDepH && c >= kDep0 && c < kDep0 + kDepW); };

    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const int HX0 = tile_x * kUpsTileW, HY0 = tile_y * kTileH;
    const int LX0 = HX0 >> 1, LY0 = HY0 >> 1;
    const int lw = a.lw, lh = a.lh, hw = a.hw, hh = a.hh;
    const float *__restrict__ lo_depth = frame_ptr(a.lo_depth, a.frame_stride, frame);
    const ao_t *__restrict__ lo_ao = frame_ptr(static_cast<const ao_t *>(a.lo_ao), a.frame_stride, frame);
    // main_premin*: LoResAO1 = min(LoResAO1, LoResAO2) (COMBINE_LOWER_RESOLUTIONS, UPS:58-60)
    const ao_t *__restrict__ lo_ao2 = a.lo_ao2 ? frame_ptr(static_cast<const ao_t *>(a.lo_ao2), a.frame_stride, frame) : nullptr;
    // (SGPR operands: 207 of the 2 471 VALU instructions of the full-resolution pass's hot path read one.  Pinned in VGPRs instead
    // -- an SGPR source halves the issue rate of a full-rate instruction in isolation -- the pass runs 183.9 vs 184.8 us, the fused
    // last kernel 235.7 vs 233.8: nothing, in round 6 as in round 2; profiles/r06_ab_upsample_uniforms_in_vgprs.jsonl)
    const BlurConsts bk = {a.step_size, a.blur_tolerance};
    const BilateralConsts bilateral_k(a.upsample_tolerance, a.noise_filter_strength);
    // (fetched here, not where the bilateral phase first stores: a.dst[frame] is a scalar load whose latency would sit right
    // behind the last barrier)
    ao_t *__restrict__ dst = FINAL ? static_cast<ao_t *>(a.dst[frame])
                                   : frame_ptr(static_cast<ao_t *>(a.dst[0]), a.frame_stride, frame);
    asm volatile("" : "+s"(dst));
    // Linearize constants of the final pass's HiResDB (fetched here for the same reason)
    const float zp0 = FINAL ? hi->zp0 : 0.0f, zp1 = FINAL ? hi->zp1 : 0.0f;
    const float sky_depth = (FINAL && hi->reversed_z == 0) ? 1.0f : 0.0f;
    const int raw_format = (FINAL && !RAW_F32) ? hi->depth_format : MEAO_DEPTH_F32;

    PhaseClock clk(FINAL ? 0 : 8);
    __builtin_amdgcn_s_setprio(3);
    // The hi-res operands of the bilateral phase do not depend on anything computed here: their loads
    // are issued first, so that their latency hides behind the prefetch and blur phases.
    constexpr int kPasses = kTileH / 32;
    typedef UpsLoads<AOFMT, FINAL, TILE_H> Loads;
    Loads L;
    auto &hoist_hraw = L.hraw;
    auto &hoist_hd32 = L.hd32;
    auto &hoist_ha = L.ha;
    const bool hoist_ok = MEAO_X_HOT_PATH_ONLY || a.vec_ok != 0;
    // interior tile, 16-byte loads everywhere, no second AO input: window loads first (ups_issue_interior_loads)
    // full-resolution pass, tile inside the frame: no LowDepth1 read at all (ups_issue_from_raw_loads)
    const bool from_raw = !NESTED && ups_tile_from_raw<FINAL, TILE_H>(a, tile);
    const bool window_first = !NESTED && !from_raw && (MEAO_X_HOT_PATH_ONLY || ups_tile_is_interior<FINAL, TILE_H>(a, tile));
    if (hoist_ok && !window_first && !from_raw) ups_issue_hoisted<AOFMT, FINAL, TILE_H, false, RAW_F32>(a, hi, tile, frame, L);

    // ---- PrefetchData (UPS:54-72): raw window = virtual low-res texels
    // [LX0-3, LX0+34] x [LY0-3, LY0+kLowH+2], clamp addressing per texel.
    const bool interior_x = ((lw & 3) == 0) && LX0 >= 4 && LX0 + 35 < lw;
    if (from_raw) {
        if constexpr (FINAL && !NESTED) {
            constexpr int kItems = Loads::kItems, kRounds = Loads::kRounds;
            static_assert(kItems <= 1024, "tenth()");
            int apron_r, apron_k;      // the lane's apron item, computed once: pinned, or the compiler derives it again where it is converted
            UpsApron<TILE_H>::item(min(tid, UpsApron<TILE_H>::kItems - 1), apron_r, apron_k);
            asm volatile("" : "+v"(apron_r), "+v"(apron_k));
            ups_issue_from_raw_loads<AOFMT, TILE_H, RAW_F32>(a, hi, tile, frame, L, apron_r, apron_k);
            auto &wa = L.wa;
#pragma unroll
            for (int round = 0; round < kRounds; ++round) {
                typedef typename std::conditional<sizeof(typename AO::type4) == 4, uint32_t, uint64_t>::type bits_t;
                asm volatile("" : : "v"(__builtin_bit_cast(bits_t, wa[round])));      // (as below: keeps the partial round's loads in front)
                const int i = tid + round * kThreads;
                if (i < kItems) {
                    const int r = tenth(i), k = i - 10 * r;
                    const float av[4] = {AO::decode(wa[round].x), AO::decode(wa[round].y), AO::decode(wa[round].z), AO::decode(wa[round].w)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = 4 * k + e - 1;
                        if (c >= 0 && c < T::kRawW) s_ao[r * T::kRawPitch + c] = av[e];
                    }
                }
            }
            // the apron of LoResDB: Linearize of the raw texels under it.  Every LowDepth1 texel of a frame that is not hostile has a
            // nice denominator (that is what the flag means), so the exact sequence is what the downsample pass stored.
            asm volatile("" : : "v"(L.araw[0]), "v"(L.araw[1]));
            if (tid < UpsApron<TILE_H>::kItems) {
                const int r = apron_r, k = apron_k;
                float q0[4], q1[4];
                decode_raw_quad<RAW_F32>(L.araw[0], raw_format, q0);
                decode_raw_quad<RAW_F32>(L.araw[1], raw_format, q1);
                const float rawv[4] = {q0[0], q0[2], q1[0], q1[2]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 4 * k + e - 1;
                    if (c >= 0 && c < T::kRawW) {
                        const float d = linearize<DIV>(rawv[e], zp0, zp1, sky_depth);        // DS1:40-45, 64-70
                        if (dep_kept(r, c)) dep_at(r, c) = d;
                        s_inv[r * T::kRawPitch + c] = rcp_strict<DIV>(d);                   // UPS:67
                    }
                }
            }
        }
    } else if (window_first) {
        constexpr int kItems = Loads::kItems, kRounds = Loads::kRounds;
        if constexpr (!NESTED) ups_issue_interior_loads<AOFMT, FINAL, TILE_H, RAW_F32>(a, hi, tile, frame, L,
                                                                                       MEAO_X_HOT_PATH_ONLY || (HX0 + kUpsTileW <= hw && HY0 + kTileH <= hh));
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
                static_assert(kItems <= 1024, "tenth()");
                const int r = tenth(i), k = i - 10 * r;
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
    // ---- HiResDB of the lane's 16 (8) texels from the hoisted raw depth loads (hi_depth_words), at the end of the fill phase, in
    // front of the FIRST barrier: the bilateral phase finds packed f16 words, as it did when LinearDepth was a buffer, and only
    // those eight registers -- not the sixteen of the raw quads -- stay live across the blur phases (converted back to f32 here
    // already, sixteen registers: 236.5 vs 237.0 us, not kept).  (In front of the second /
    // third barrier instead: last kernel 247 / 248 us against 239.5, full-resolution pass 192 / 185 against 181 per 16 4K frames,
    // profiles/r06_ab_hi_depth_block_position.jsonl.)
    uint32_t hd_words[2 * kPasses][2];
    bool lane_clean = true;
    auto hi_depth_block = [&]() __attribute__((always_inline)) {
        if constexpr (FINAL) {
            if (hoist_ok) {
                uint4v rawq[2 * kPasses];
#pragma unroll
                for (int pass = 0; pass < kPasses; ++pass)
#pragma unroll
                    for (int f = 0; f < 2; ++f) rawq[2 * pass + f] = hoist_hraw[pass][f];
                float even[2 * kPasses][2];
                lane_clean = hi_depth_words<RTNE, DIV, RAW_F32, 2 * kPasses>(rawq, raw_format, zp0, zp1, sky_depth, hd_words, even);
                if (from_raw) {
                    // the interior of the LoResDB window: the lane's own even-even texels before the f16 store
                    const int r0 = 3 + (tid >> 4), c = 3 + 2 * (tid & 15);
#pragma unroll
                    for (int pass = 0; pass < kPasses; ++pass)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float d = even[2 * pass][j];
                            dep_at(r0 + 16 * pass, c + j) = d;
                            s_inv[(r0 + 16 * pass) * T::kRawPitch + c + j] = rcp_strict<DIV>(d);   // UPS:67
                        }
                }
            }
        }
    };
    hi_depth_block();      // (at s_setprio 3 like the fill it ends: at priority 0 the pass is 16 us slower per 16 4K frames, r06 A/B)
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
                // (the raw depth quads were consumed in front of the barrier: nothing of this tile is in flight any more)
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
    // (always_inline, like every helper here: whether the compiler inlined this lambda used to depend on how many kernels of the
    // translation unit instantiate the same upsample_tile -- the code of a kernel must not depend on its neighbours in the file)
    auto bilateral_phase = [&](auto whole_tile) __attribute__((always_inline)) {
        constexpr bool WHOLE = decltype(whole_tile)::value;
        const bool vec_ok = WHOLE || vec_ok_frame;
        if (!WHOLE && hx0 >= hw) return;
        bool redo = !lane_clean;   // FINAL, exact divisions: a raw depth texel of this lane is outside their verified range (hi_depth_words / _quad)
#pragma unroll       // the hoisted operands live in registers: static indices
        for (int pass = 0; pass < kTileH / 32; ++pass) {
            const int ty = (tid >> 4) + 16 * pass;
            const int hy0 = HY0 + 2 * ty;
            if (!WHOLE && hy0 >= hh) break;

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
                    // HiResDB = f16(Linearize(raw depth)), evaluated here (hi_depth_quad); LinearDepth is not a buffer
                    if (vec_ok) {
                        const uint32_t w0 = hd_words[2 * pass + f][0], w1 = hd_words[2 * pass + f][1];
                        hd[0] = f16_bits_to_f32(static_cast<uint16_t>(w0 & 0xffffu)); hd[1] = f16_bits_to_f32(static_cast<uint16_t>(w0 >> 16));
                        hd[2] = f16_bits_to_f32(static_cast<uint16_t>(w1 & 0xffffu)); hd[3] = f16_bits_to_f32(static_cast<uint16_t>(w1 >> 16));
                    } else {
                        float rawd[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) rawd[e] = (hx0 + e < hw) ? raw_depth_texel(hi->raw[frame], raw_format, hrow + e) : 0.5f;
                        redo |= !hi_depth_quad<RTNE, DIV>(rawd, zp0, zp1, sky_depth, hd);
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
                        res[e] = static_cast<ao_t>(bilateral_upsample_r8<Hook::kGroupReciprocals, !NESTED && Hook::kReuseEstimate && !Hook::kPairReciprocals,
                                                                          Hook::kPairReciprocals>(hd[e], ha[e], gd, ga, bilateral_k));
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
                    // the blend passes' outputs are re-read by the next pass from L2; the result leaves the path, but a tile row of it is half a
                    // cache line (R8): temporal stores let L2 merge it with the neighbouring tile's half (MEAO_X_FINAL_NT_STORE)
                    if constexpr (FINAL && MEAO_X_FINAL_NT_STORE) __builtin_nontemporal_store(r4, reinterpret_cast<typename AO::type4 *>(o));
                    else *reinterpret_cast<typename AO::type4 *>(o) = r4;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (hx0 + e < hw) o[e] = res[e];
                }
            }
            clk.mark(6 + pass);  // 6, 7: bilateral pass 0 / 1 (64-row tiles) incl. its stores being issued
        }
        if constexpr (FINAL && DIV == DIV_EXACT_RCP && !MEAO_X_HOT_PATH_ONLY && !MEAO_X_NO_REDO) {
            // Hostile raw depth in a frame whose LEVELS are clean (the frame flag only covers the texels the levels are made of): this
            // lane's texels once more with IEEE '/' throughout, as the reference divides -- exact sequences and IEEE '/' agree wherever
            // the former are valid, so redoing the lane's clean texels too changes nothing.  A compact loop, not unrolled: cold code.
            if (__builtin_expect(redo, 0)) {
#pragma unroll 1
                for (int t = 0; t < 8 * (kTileH / 32); ++t) {
                    const int pass = t >> 3, f = (t >> 2) & 1, e = t & 3;
                    const int ty = (tid >> 4) + 16 * pass;
                    const int hy = HY0 + 2 * ty + f, hx = hx0 + e;
                    if (hx >= hw || hy >= hh) continue;
                    const size_t at = static_cast<size_t>(hy) * hw + hx;
                    const float rawv = raw_depth_texel(hi->raw[frame], raw_format, at);
                    const float hdv = through_f16<RTNE>(linearize<DIV_IEEE>(rawv, zp0, zp1, sky_depth));
                    const int cc = ((e + 1) >> 1) + 1, rr = f + 1;                        // as above
                    const int comp = (e & 1) ? (f ? 3 : 0) : (f ? 2 : 1);
                    float gd[4], ga[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int g = (comp + i) & 3;
                        const int r = ty + rr - (g >> 1), c = 2 * tx + cc - ((g == 0 || g == 3) ? 1 : 0);      // gy[g], gx[g]
                        gd[i] = dep_at(r + 2, c + 2);
                        ga[i] = s_vb[r * T::kBlurPitch + c];
                    }
                    dst[at] = AO::template encode<RTNE>(bilateral_upsample<DIV_IEEE>(hdv, 1.0f, gd[0], gd[1], gd[2], gd[3],
                                                                                      ga[0], ga[1], ga[2], ga[3], bilateral_k));
                }
            }
        }
    };
    // (the copy exists for clean frames only -- the IEEE-division bodies of a hostile frame are four times as long -- and not in the
    // nested launches, which have no registers for it: 3 spilled VGPRs in the two-level kernel, no gain measured there)
    if (MEAO_X_BIL_WHOLE_TILE && !NESTED && DIV == DIV_EXACT_RCP && vec_ok_frame && HX0 + kUpsTileW <= hw && HY0 + kTileH <= hh)
        bilateral_phase(std::true_type());
    else
        bilateral_phase(std::false_type());
}

// The (rare) hostile-frame variant of a tile: the same code with IEEE division.
template <int AOFMT, bool RTNE, bool FINAL, int DIV, typename Hook = NoHook, int TILE_H = ups_tile_h(FINAL), bool RAW_F32 = true>
__device__ __forceinline__ void upsample_tile_checked(const UpsampleArgs &a, float *smem, int tile, int frame, Hook hook = Hook(),
                                                      const HiDepthArgs *hi = nullptr)
{
    if constexpr (DIV == DIV_EXACT_RCP) {
        if (frame_is_hostile(a.hostile, a.generation, frame)) {       // wave-uniform, decided per frame
            upsample_tile<AOFMT, RTNE, FINAL, DIV_IEEE, false, Hook, TILE_H, RAW_F32>(a, smem, tile, frame, hook, hi);
            return;
        }
    }
    upsample_tile<AOFMT, RTNE, FINAL, DIV, false, Hook, TILE_H, RAW_F32>(a, smem, tile, frame, hook, hi);
}


}  // namespace
}  // namespace meao
