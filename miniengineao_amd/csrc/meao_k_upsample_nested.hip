// meao_k_upsample_nested.hip -- blend passes evaluated inside the launch of the pass above them (two-level, three-level).
#include "meao_dev_blend.hpp"

namespace meao {
namespace {

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


}  // namespace

// ------------------------------------------------------------------------------------------
// launchers

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
        else if (outer.exact_rcp_div) launch_upsample_two_level_t<MEAO_AO_R8, false, DIV_EXACT_RCP>(outer, inner, grid, s);
        else launch_upsample_two_level_t<MEAO_AO_R8, false, DIV_IEEE>(outer, inner, grid, s);
    } else {
        if (outer.f16_rtne) launch_upsample_two_level_t<MEAO_AO_F16, true, DIV_IEEE>(outer, inner, grid, s);
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
        else if (outer.exact_rcp_div) launch_upsample_three_level_t<MEAO_AO_R8, false, DIV_EXACT_RCP>(outer, mid, inner, grid, s);
        else launch_upsample_three_level_t<MEAO_AO_R8, false, DIV_IEEE>(outer, mid, inner, grid, s);
    } else {
        if (outer.f16_rtne) launch_upsample_three_level_t<MEAO_AO_F16, true, DIV_IEEE>(outer, mid, inner, grid, s);
        else if (outer.exact_rcp_div) launch_upsample_three_level_t<MEAO_AO_F16, false, DIV_EXACT_RCP>(outer, mid, inner, grid, s);
        else launch_upsample_three_level_t<MEAO_AO_F16, false, DIV_IEEE>(outer, mid, inner, grid, s);
    }
    return hipGetLastError();
}


}  // namespace meao
