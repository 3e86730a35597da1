// meao_dev_downsample.hpp -- the downsample tile (Downsample1.main + Downsample2.main fused): generic and lean forms, used by the
// downsample kernels and by the upsample / render kernels that carry a downsample tile.
#pragma once

#include "meao_dev.hpp"

namespace meao {
namespace {

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
                                                     float (&v)[PASSES][4], const unsigned tid = threadIdx.x)
{
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const void *__restrict__ depth = a.depth[frame];
    const int W = a.w[0], H = a.h[0];
    const int x0 = tile_x * kDsTileW + (tid % kDsLanesPerRow) * 4;
    const int yb = tile_y * (PASSES * kDsRowsPerPass) + (tid / kDsLanesPerRow);
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
                                                       const float (&v)[PASSES][4], const unsigned tid = threadIdx.x)
{
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    uint16_t *__restrict__ linear = frame_ptr(a.linear, a.frame_stride, frame);
    float *__restrict__ low1 = frame_ptr(a.low[0], a.frame_stride, frame);
    float *__restrict__ low2 = frame_ptr(a.low[1], a.frame_stride, frame);
    float *__restrict__ low3 = frame_ptr(a.low[2], a.frame_stride, frame);
    float *__restrict__ low4 = frame_ptr(a.low[3], a.frame_stride, frame);
    const int W = a.w[0], H = a.h[0];
    const float sky_depth = a.reversed_z != 0 ? 0.0f : 1.0f;
    const int x0 = tile_x * kDsTileW + (tid % kDsLanesPerRow) * 4;
    const int yb = tile_y * (PASSES * kDsRowsPerPass) + (tid / kDsLanesPerRow);
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
__device__ __forceinline__ void downsample_tile(const DownsampleArgs &a, int tile, int frame, const unsigned tid = threadIdx.x)
{
    // tid: the lane's index inside the 256-lane tile (workgroups of 512 threads process two tiles, meao_k_render_depth.hip)
    float v[PASSES][4];
    downsample_tile_load<VEC, false, PASSES>(a, tile, frame, v, tid);
    downsample_tile_finish<RTNE, VEC, DIV, PASSES>(a, tile, frame, v, tid);
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


}  // namespace
}  // namespace meao
