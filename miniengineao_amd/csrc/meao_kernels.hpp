// meao_kernels.hpp -- launch interface between the C ABI layer and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/meao.h"

namespace meao {

// ---------------------------------------------------------------------------------------
// Downsample (Downsample1.main + Downsample2.main fused; no LDS, pure streaming).  Since round 6 the pass writes the four
// point-sampled levels ONLY: LowDepth<k>[i, j] = Linearize(depth[2^k i, 2^k j]), so it reads the even rows of the frame and
// nothing else.  LinearDepth (DS1:46, f16) had one consumer, HiResDB of Upsample.main (UPS:217-223): that pass linearizes the
// raw depth itself (UpsampleArgs::hi_raw), and debug id 1 is built on demand (launch_linear_depth).
struct DownsampleArgs {
    const void *depth[MEAO_MAX_BATCH];   // caller-owned raw depth (depth_format), one pointer per frame
    int32_t depth_format;                // meao_depth_format
    float *low[4];                       // LowDepth1..4 f32, frame 0
    uint64_t frame_stride;               // bytes between consecutive frames' intermediates
    int32_t w[5], h[5];                  // mip 0..4 dims
    float zp0, zp1;                      // ZBufferParams.xy
    int32_t reversed_z;
    int32_t f16_rtne;
    int32_t exact_rcp_div;
    int32_t tiles_x, tiles_y;            // tiles of kMipTileW x (8 * rows_per_lane) LowDepth1 texels (stand-alone pass), or of
                                         // kLeanMipW x kLeanMipRows (the tile the last upsample kernel carries)
    int32_t rows_per_lane;               // kMipRowsPerLane, or 1 (stand-alone pass of small calls: twice the workgroups)
    int32_t frames;                      // used by the fused kernel only (the plain launch has grid.z = frames)
    int32_t vec_ok;                      // width % 8 == 0 and every depth pointer aligned for 4-texel loads
    // hostile[frame] = generation when a texel the LEVELS are made of is outside the range the exact v_rcp_f32 sequences are
    // verified for (NaN, inf, negative, tiny); read by the later kernels.  (The full-resolution upsample tests the texels it
    // linearizes itself, per lane.)
    uint32_t *hostile;
    uint32_t generation;
};
// a lane of the pass: 4 consecutive LowDepth1 texels (= 8 raw texels of an even row) in rows r, r + 8, ...
constexpr int kMipTileW = 128, kMipLanesPerRow = kMipTileW / 4, kMipRowsPerPass = 256 / kMipLanesPerRow, kMipRowsPerLane = 2;
constexpr int kLeanMipW = 64, kLeanMipRows = 16;        // meao_dev_downsample.hpp kLeanW / kLeanRows

// ---------------------------------------------------------------------------------------
// Render (Render.main_interleaved for all levels in one grid)
// Output texels per workgroup of the interleaved render: 128 x 32 with 512 threads (40 KB window, four
// workgroups = 8 waves per SIMD; measured 3 % faster than 64 x 32 at 6 waves per SIMD).  The 68-sample
// variant needs ~95 VGPRs, where the smaller workgroup fits more waves: 64 x 32 with 256 threads.
constexpr int ren_tile_w(bool exhaustive) { return exhaustive ? 64 : 128; }
constexpr int kRenTileH = 32;
constexpr int kRenTileHSmall = 8;               // calls with fewer 128 x 32 tiles than CUs: four times the workgroups, one texel-loop iteration each
constexpr int kWideTileW = 64;                                 // Render.main (wide) keeps 64 x 32, 256 threads
constexpr int kRenApron = 16;                   // 4 slice texels * interleave 4
constexpr int kRenLdsH = kRenTileH + 2 * kRenApron;            // rows of the staged window; columns: tile width + 2 * apron

struct RenderLevelArgs {
    const float *src;      // LowDepth<level> f32, frame 0
    void *dst;             // Occlusion<level>, frame 0
    int32_t lw, lh;        // level dims (= output dims)
    int32_t sw, sh;        // slice dims of TiledDepth<level> (mip level+2)
    int32_t tiles_x, tiles_y;
    int32_t block_begin;   // first linear workgroup id of this level
    float pad_value;       // value of atlas texels beyond the level
    float inv_thickness[12], front_depth[12];   // per term, accumulation order
    float weight[12];                           // sample weight x the 0.5 / 0.25 factor of TestSamples
    float reject_fadeoff, intensity;
};

struct RenderArgs {
    RenderLevelArgs level[4];
    uint64_t frame_stride;
    int32_t num_levels;
    int32_t blocks_per_frame;
    int32_t f16_rtne;
    int32_t exact_rcp_div;
    int32_t exhaustive;    // SAMPLE_EXHAUSTIVELY: 12 terms instead of 7
    int32_t tile_h;        // kRenTileH, or kRenTileHSmall (interleaved checker-set kernel only): the tiling `level[]` was built for
    const uint32_t *hostile;   // per frame, written by the downsample pass that produced `src`
    uint32_t generation;       // hostile[frame] == generation -> IEEE-division body for that frame
};

// Render.main (WIDE_SAMPLING, non-interleaved) on the non-tiled LowDepth<level>: same args; `src`
// is sampled directly (f32, clamp addressing), sw/sh/pad_value are unused, `level[]` holds only
// the levels that have the pass (num_levels = their count).
constexpr int kWideApron = 8;                   // 4 samples * stride 2
constexpr int kWideLdsW = kWideTileW + 2 * kWideApron, kWideLdsH = kRenTileH + 2 * kWideApron;

// ---------------------------------------------------------------------------------------
// Upsample (Upsample.main / main_blendout)
// Hi-res texels per workgroup.  The full-resolution pass (L1 -> L0, "main") uses 64 x 64: smaller
// blur aprons and better lane use in the blur phases; the three blend passes have few tiles per
// frame and run faster with 64 x 32 (measured on one MI355X, see profiles/README.md).
constexpr int kUpsTileW = 64;
constexpr int ups_tile_h(bool final_pass) { return final_pass ? 64 : 32; }
constexpr int kUpsTileHTall = 64;               // a blend pass with many tiles (UpsampleArgs::tile_h; upsample_blend_tall_kernel)
constexpr int kUpsTileHSmall = 32;              // the final pass of calls with few 64 x 64 tiles (UpsampleArgs::tile_h)

struct UpsampleArgs {
    const float *lo_depth;     // LoResDB  f32
    const void *lo_ao;         // LoResAO1
    const void *lo_ao2;        // LoResAO2 of main_premin* (min-combined in PrefetchData), or nullptr
    const void *hi_depth;      // HiResDB  f32 (blend passes); unused in the final pass, which linearizes the raw depth itself (HiDepthArgs)
    const void *hi_ao;         // HiResAO, nullptr in the final pass
    void *dst[MEAO_MAX_BATCH]; // per-frame destination (caller-owned in the final pass)
    uint64_t frame_stride;     // applies to lo_*, hi_* (context-owned intermediates)
    int32_t lw, lh, hw, hh;
    int32_t tiles_x, tiles_y;
    int32_t tile_h;            // rows of a tile: ups_tile_h(final), or kUpsTileHSmall in the final pass of a small call
    float noise_filter_strength, step_size, blur_tolerance, upsample_tolerance;
    int32_t f16_rtne;
    int32_t exact_rcp_div;     // operands proven inside the exact range of the v_rcp_f32 sequences
    int32_t vec_ok;            // hw % 4 == 0 and, in the final pass, every dst and raw depth pointer aligned for 4-texel accesses
    const uint32_t *hostile;   // as in RenderArgs
    uint32_t generation;
};

// Final pass (Upsample.main): HiResDB = LinearZ = f16(Linearize(depth)) (DS1:37-48, UPS:217-223) is evaluated from the caller's raw
// depth frame inside the bilateral phase -- same reciprocal sequence, same f16 round trip, hostile texels divided with IEEE '/'
// per lane -- instead of being read back from a LinearDepth buffer nothing else reads.  UpsampleArgs::vec_ok then also says
// that every raw[] pointer is aligned for 4-texel loads.
struct HiDepthArgs {
    const void *raw[MEAO_MAX_BATCH];   // caller-owned raw depth of the frames being upsampled
    int32_t depth_format;              // meao_depth_format
    int32_t reversed_z;
    float zp0, zp1;                    // ZBufferParams.xy
};

// ---------------------------------------------------------------------------------------
// TiledDepth<level> materialisation for the debug views (Downsample1/2 atlas stores)
struct TileAtlasArgs {
    const float *src;     // LowDepth<level> of the requested frame
    uint16_t *dst;        // [16][sh][sw] f16
    int32_t lw, lh, sw, sh;
    float pad_value;
    int32_t f16_rtne;
};

hipError_t launch_downsample(const DownsampleArgs &a, int frames, hipStream_t s);
hipError_t launch_render(const RenderArgs &a, int ao_format, int frames, hipStream_t s);
hipError_t launch_render_wide(const RenderArgs &a, int ao_format, int frames, hipStream_t s);
// hi: the raw depth frames of the final pass (Upsample.main), nullptr = a blend pass (main_blendout)
hipError_t launch_upsample(const UpsampleArgs &a, const HiDepthArgs *hi, int ao_format, int frames, hipStream_t s);
// Two blend passes in one launch: `inner` (e.g. L4 -> L3) is evaluated per tile of `outer` (L3 -> L2) for the
// window of its output that the tile reads; inner's target is still written (each tile stores its own part).
hipError_t launch_upsample_two_level(const UpsampleArgs &outer, const UpsampleArgs &inner, int ao_format, int frames,
                                     hipStream_t s);
// one or two frames per call: L4->L3 and L3->L2 inside the L2->L1 launch (outer = L2->L1, mid = L3->L2, inner = L4->L3)
hipError_t launch_upsample_three_level(const UpsampleArgs &outer, const UpsampleArgs &mid, const UpsampleArgs &inner, int ao_format,
                                       int frames, hipStream_t s);
// Upsample.main of this batch + the downsample pass of the next one in a single kernel (f32 depth, d.vec_ok, one carried
// tile per upsample tile: fused_downsample_applicable).
bool fused_downsample_applicable(const UpsampleArgs &a, const HiDepthArgs &hi, const DownsampleArgs &d, int frames);
hipError_t launch_upsample_final_with_downsample(const UpsampleArgs &a, const HiDepthArgs &hi, const DownsampleArgs &d, int ao_format,
                                                 int frames, hipStream_t s);
hipError_t launch_tile_atlas(const TileAtlasArgs &a, hipStream_t s);
// LinearDepth (debug id 1) on demand: dst[i] = f16(Linearize(depth[i])) for one frame (DS1:37-48).
struct LinearDepthArgs {
    const void *depth;
    uint16_t *dst;
    int64_t pixels;
    int32_t depth_format, reversed_z, f16_rtne;
    float zp0, zp1;
};
hipError_t launch_linear_depth(const LinearDepthArgs &a, hipStream_t s);
// Debug view (PushDebugBlitCommands): src in `src_format` (meao_format), [slices][sh][sw] -> dst AO W x H.
struct DebugViewArgs {
    const void *src;
    void *dst;
    int32_t sw, sh, slices, src_format;
    int32_t w, h;
    int32_t f16_rtne;
};
hipError_t launch_debug_view(const DebugViewArgs &a, int ao_format, hipStream_t s);
// Composite (Blit.shader passes 1-3): ao in ao_format, color RGBA16F in place, gbuffer0 RGBA8 or null.
struct CompositeArgs {
    const void *ao;
    void *color;
    void *gbuffer0;
    int64_t pixels;
    int32_t mode;
};
hipError_t launch_composite(const CompositeArgs &a, int ao_format, hipStream_t s);
// A batch of composites carried by a render launch (meao_composite_enqueue): frame f = ao[f] x color[f].
struct CompositeBatchArgs {
    const void *ao[MEAO_MAX_BATCH];
    void *color[MEAO_MAX_BATCH];
    void *gbuffer0[MEAO_MAX_BATCH];
    int64_t pixels;      // per frame
    int32_t frames;
    int32_t mode;
};
hipError_t launch_render_with_composite(const RenderArgs &a, const CompositeBatchArgs &c, int ao_format, int frames,
                                        hipStream_t s);
// Exhaustive conversion self-tests; *count (device) receives the number of mismatches.
hipError_t launch_selftest(int which, unsigned long long *count, hipStream_t s);

// meao_api.cpp, for meao_pool.cpp: meao_execute_batch that can leave the staged copies of a HOST call in flight
int execute_batch_internal(meao_ctx *ctx, int32_t n, const void *const *depth, int32_t depth_loc, void *const *ao_out,
                           int32_t out_loc, meao_stream stream, bool wait_for_host);

}  // namespace meao
