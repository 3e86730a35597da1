// meao_dev_downsample.hpp -- Linearize and the downsample tile (Downsample1.main + Downsample2.main fused): the generic form of the
// stand-alone pass and the lean form the last upsample kernel of a pipelined call carries.
#pragma once

#include "meao_dev.hpp"

namespace meao {
namespace {

// ------------------------------------------------------------------------------------------
// Downsample: linearize + point-downsample to L1..L4.
//
// Closed form of DS1+DS2: LinearZ = lin(x,y); DS2x[i,j] = lin(2i,2j); DS4x = lin(4i,4j); DS8x = lin(8i,8j);
// DS16x = lin(16i,16j) -- every level keeps the top-left texel of its block, so a lane decides what to store from its own
// coordinates and no LDS exchange is needed.  Since round 6 LinearZ is not stored: its only reader, HiResDB of Upsample.main
// (UPS:217-223), evaluates lin(x,y) itself (meao_dev_upsample.hpp, hi_depth_quad), so the pass touches the EVEN rows of the frame
// only -- 27.6 MB per 4K frame instead of 60.8 -- and the 16.6 MB of LinearZ are neither written nor read back.

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
// quotient) has its operands inside its verified range when all texels it consumes are nice.  A frame
// whose LEVELS hold any other texel -- NaN, +-inf, negative, > 1 with a conventional Z buffer, depths below
// 2^-24 -- is marked hostile by the downsample pass and takes the IEEE-division bodies of the later
// kernels (the reference divides with IEEE '/', Downsample1.compute:37-48; inputs are never sanitised).
// Texels that only the full-resolution pass sees (odd rows / columns) are tested there, per lane.
__device__ __forceinline__ bool nice_denominator(float den)
{
    return __builtin_amdgcn_fmed3f(den, 0x1p-20f, 0x1p24f) == den;   // false for NaN
}

// The same for four denominators at once: two unsigned min / max chains on their bit patterns (negative values and NaNs are
// the largest unsigned words) instead of four v_med3 + four compares.
__device__ __forceinline__ bool nice_denominators(float d0, float d1, float d2, float d3)
{
    const uint32_t b0 = __builtin_bit_cast(uint32_t, d0), b1 = __builtin_bit_cast(uint32_t, d1);
    const uint32_t b2 = __builtin_bit_cast(uint32_t, d2), b3 = __builtin_bit_cast(uint32_t, d3);
    const uint32_t lo = min(min(min(b0, b1), b2), b3), hi = max(max(max(b0, b1), b2), b3);
    return lo >= 0x35800000u && hi <= 0x4B800000u;
}

// One raw depth texel of any meao_depth_format (the depth-copy blit of the reference, Blit.shader pass 0, folded into the load).
__device__ __forceinline__ float raw_depth_texel(const void *depth, int format, size_t at)
{
    if (format == MEAO_DEPTH_F32) return static_cast<const float *>(depth)[at];
    if (format == MEAO_DEPTH_UNORM24) return unorm_to_f32<24>(static_cast<const uint32_t *>(depth)[at] & 0xffffffu);
    const uint16_t u = static_cast<const uint16_t *>(depth)[at];
    return format == MEAO_DEPTH_UNORM16 ? unorm_to_f32<16>(u) : f16_bits_to_f32(u);
}

// Four consecutive raw texels from an address aligned to four texels, decoded (wave-uniform switch on the format).
__device__ __forceinline__ void raw_depth_quad(const void *depth, int format, size_t at, float (&v)[4], bool non_temporal)
{
    if (format == MEAO_DEPTH_F32) {
        const float4v *p = reinterpret_cast<const float4v *>(static_cast<const float *>(depth) + at);
        const float4v q = non_temporal ? __builtin_nontemporal_load(p) : *p;
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else if (format == MEAO_DEPTH_UNORM24) {
        const uint4v q = *reinterpret_cast<const uint4v *>(static_cast<const uint32_t *>(depth) + at);
        v[0] = unorm_to_f32<24>(q.x & 0xffffffu); v[1] = unorm_to_f32<24>(q.y & 0xffffffu);
        v[2] = unorm_to_f32<24>(q.z & 0xffffffu); v[3] = unorm_to_f32<24>(q.w & 0xffffffu);
    } else {
        const ushort4v q = *reinterpret_cast<const ushort4v *>(static_cast<const uint16_t *>(depth) + at);
        if (format == MEAO_DEPTH_UNORM16) {
            v[0] = unorm_to_f32<16>(q.x); v[1] = unorm_to_f32<16>(q.y); v[2] = unorm_to_f32<16>(q.z); v[3] = unorm_to_f32<16>(q.w);
        } else {
            v[0] = f16_bits_to_f32(q.x); v[1] = f16_bits_to_f32(q.y); v[2] = f16_bits_to_f32(q.z); v[3] = f16_bits_to_f32(q.w);
        }
    }
}

// ------------------------------------------------------------------------------------------
// The generic tile (stand-alone pass; every depth format, any width): kMipTileW x (8 * ROWS) texels of LowDepth1.  A lane takes
// 4 consecutive LowDepth1 texels -- raw texels 2j, 2j+2, 2j+4, 2j+6 of raw row 2i -- in each of its ROWS rows (i, i + 8, ...):
// with 16-byte loads two per row, all of them in flight before the first is used.
// VEC: W % 8 == 0 and every frame aligned to 4 texels.
template <bool VEC, int DIV, int ROWS>
__device__ __forceinline__ void downsample_tile(const DownsampleArgs &a, int tile, int frame)
{
    const unsigned tid = threadIdx.x;
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const int j0 = tile_x * kMipTileW + static_cast<int>(tid % kMipLanesPerRow) * 4;     // LowDepth1 column of the lane's first texel
    const int ib = tile_y * (kMipRowsPerPass * ROWS) + static_cast<int>(tid / kMipLanesPerRow);
    const int W = a.w[0], w1 = a.w[1], h1 = a.h[1];
    if (j0 >= w1) return;
    const void *__restrict__ depth = a.depth[frame];
    const int format = a.depth_format;
    float v[ROWS][4];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int i = ib + kMipRowsPerPass * k;
        v[k][0] = v[k][1] = v[k][2] = v[k][3] = 0.5f;
        if (i < h1) {
            const size_t at = static_cast<size_t>(2 * i) * W + 2 * j0;
            if constexpr (VEC) {
                float q0[4], q1[4];
                raw_depth_quad(depth, format, at, q0, true);
                raw_depth_quad(depth, format, at + 4, q1, true);
                v[k][0] = q0[0]; v[k][1] = q0[2]; v[k][2] = q1[0]; v[k][3] = q1[2];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (j0 + e < w1) v[k][e] = raw_depth_texel(depth, format, at + 2 * e);
            }
        }
    }
    float *__restrict__ low1 = frame_ptr(a.low[0], a.frame_stride, frame);
    float *__restrict__ low2 = frame_ptr(a.low[1], a.frame_stride, frame);
    float *__restrict__ low3 = frame_ptr(a.low[2], a.frame_stride, frame);
    float *__restrict__ low4 = frame_ptr(a.low[3], a.frame_stride, frame);
    const float sky_depth = a.reversed_z != 0 ? 0.0f : 1.0f;
    const float zp0 = a.zp0, zp1 = a.zp1;
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int i = ib + kMipRowsPerPass * k;
        if (i >= h1) continue;
        float lin[4];
        if constexpr (DIV == DIV_EXACT_RCP) {
            // the exact reciprocal sequence is only valid for a "nice" denominator; anything else
            // (hostile input) is divided with IEEE '/' and marks the frame for the later kernels
            if (__builtin_expect(nice_denominators(mad(zp0, v[k][0], zp1), mad(zp0, v[k][1], zp1), mad(zp0, v[k][2], zp1), mad(zp0, v[k][3], zp1)), 1)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) lin[e] = linearize<DIV_EXACT_RCP>(v[k][e], zp0, zp1, sky_depth);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) lin[e] = linearize<DIV_IEEE>(v[k][e], zp0, zp1, sky_depth);
                a.hostile[frame] = a.generation;     // racing stores of the same value
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) lin[e] = linearize<DIV>(v[k][e], zp0, zp1, sky_depth);
        }
        float *p1 = low1 + static_cast<size_t>(i) * w1 + j0;                          // DS2x (DS1:64-70)
        if constexpr (VEC) {
            __builtin_nontemporal_store(float4v{lin[0], lin[1], lin[2], lin[3]}, reinterpret_cast<float4v *>(p1));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (j0 + e < w1) p1[e] = lin[e];
        }
        if ((i & 1) == 0) {                                                           // DS4x (DS1:73-77)
            float *p2 = low2 + static_cast<size_t>(i >> 1) * a.w[2] + (j0 >> 1);
            p2[0] = lin[0];
            if (j0 + 2 < w1) p2[1] = lin[2];
            if ((i & 3) == 0) {                                                       // DS8x (DS2:35-40)
                low3[static_cast<size_t>(i >> 2) * a.w[3] + (j0 >> 2)] = lin[0];
                if ((i & 7) == 0 && (j0 & 7) == 0)                                    // DS16x (DS2:43-49)
                    low4[static_cast<size_t>(i >> 3) * a.w[4] + (j0 >> 3)] = lin[0];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The lean tile: what the last upsample kernel of a pipelined call carries for the NEXT batch (meao_prefetch_batch).  f32 depth,
// W % 8 == 0, 16-byte aligned frames.  64 x 16 texels of LowDepth1 (the even rows of 128 x 32 raw texels) = one tile per 64 x 64
// upsample tile at the usual frame sizes (1080p: 510 of either, 4K: 2 040, 8K: 8 160): a lane holds two 16-byte loads (8 VGPRs)
// across the bilateral phase of its upsample tile.  Its VALU instructions are taken from a kernel that also has arithmetic to
// do, so there are as few as possible: rows are dealt to waves so that a row's residue mod 4 is wave-uniform (wave w: rows w,
// w + 4, w + 8, w + 12) -- the waves of odd rows skip the coarser levels with a scalar branch, only wave 0 ever sees L3 and L4;
// the range test is nice_denominators; the far-plane select runs only where a lane holds a far-plane texel; one 32-bit byte
// offset per buffer (saddr addressing).  Same bits as downsample_tile.
constexpr int kLeanW = kLeanMipW, kLeanRows = kLeanMipRows;
struct LeanMipLane {
    int wave, row, i;            // i: LowDepth1 row
    uint32_t j0;                 // LowDepth1 column of the lane's first texel
    __device__ __forceinline__ LeanMipLane(const DownsampleArgs &a, int tile)
    {
        const uint32_t tid = threadIdx.x;
        wave = __builtin_amdgcn_readfirstlane(static_cast<int>(tid >> 6));
        const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
        j0 = static_cast<uint32_t>(tile_x) * kLeanW + (tid & 15u) * 4u;
        row = wave + 4 * static_cast<int>((tid >> 4) & 3u);
        i = tile_y * kLeanRows + row;
    }
};

// FULL: every row of the tile is inside the level (otherwise rows past it re-read its last row and are never used)
template <bool FULL>
__device__ __forceinline__ void downsample_lean_load(const DownsampleArgs &a, int tile, int frame, float4v (&q)[2])
{
    const LeanMipLane L(a, tile);
    const uint32_t W = static_cast<uint32_t>(a.w[0]);
    if (L.j0 >= static_cast<uint32_t>(a.w[1])) return;
    const float *__restrict__ depth = static_cast<const float *>(a.depth[frame]);
    const int i = FULL ? L.i : min(L.i, a.h[1] - 1);
    const uint32_t t = static_cast<uint32_t>(2 * i) * W + 2u * L.j0;       // texel index in the frame (< 2^30)
    q[0] = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(at_byte_offset(depth, t * 4u)));
    q[1] = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(at_byte_offset(depth, t * 4u + 16u)));
}

template <int DIV, bool FULL>
__device__ __forceinline__ void downsample_lean_finish(const DownsampleArgs &a, int tile, int frame, const float4v (&q)[2])
{
    const LeanMipLane L(a, tile);
    const int wave = L.wave, row = L.row, i = L.i;
    const uint32_t j0 = L.j0, w1 = static_cast<uint32_t>(a.w[1]);
    if (j0 >= w1) return;
    if constexpr (!FULL) { if (i >= a.h[1]) return; }
    float *__restrict__ low1 = frame_ptr(a.low[0], a.frame_stride, frame);
    float *__restrict__ low2 = frame_ptr(a.low[1], a.frame_stride, frame);
    float *__restrict__ low3 = frame_ptr(a.low[2], a.frame_stride, frame);
    float *__restrict__ low4 = frame_ptr(a.low[3], a.frame_stride, frame);
    const float zp0 = a.zp0, zp1 = a.zp1;
    const float sky_depth = a.reversed_z != 0 ? 0.0f : 1.0f;
    const float v[4] = {q[0].x, q[0].z, q[1].x, q[1].z};
    float lin[4];
    if constexpr (DIV == DIV_EXACT_RCP) {
        float den[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) den[e] = mad(zp0, v[e], zp1);
        if (__builtin_expect(nice_denominators(den[0], den[1], den[2], den[3]), 1)) {
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
    const uint32_t ui = static_cast<uint32_t>(i);
    __builtin_nontemporal_store(float4v{lin[0], lin[1], lin[2], lin[3]},                              // DS2x (DS1:64-70)
                                reinterpret_cast<float4v *>(at_byte_offset(low1, (ui * w1 + j0) * 4u)));
    if ((wave & 1) == 0) {                                                   // even rows (wave-uniform): DS4x (DS1:73-77)
        __builtin_nontemporal_store(float2v{lin[0], lin[2]}, reinterpret_cast<float2v *>(at_byte_offset(
                                        low2, ((ui >> 1) * static_cast<uint32_t>(a.w[2]) + (j0 >> 1)) * 4u)));
        if (wave == 0) {                                                     // rows 0, 4, 8, 12 of the tile: DS8x (DS2:35-40)
            *at_byte_offset(low3, ((ui >> 2) * static_cast<uint32_t>(a.w[3]) + (j0 >> 2)) * 4u) = lin[0];
            if ((row & 7) == 0 && (j0 & 7u) == 0)                            // DS16x (DS2:43-49)
                *at_byte_offset(low4, ((ui >> 3) * static_cast<uint32_t>(a.w[4]) + (j0 >> 3)) * 4u) = lin[0];
        }
    }
}


}  // namespace
}  // namespace meao
