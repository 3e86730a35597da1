// meao_k_upsample.hip -- upsample kernels: one blend pass / the full-resolution pass per launch.
#include "meao_dev_upsample.hpp"

namespace meao {
namespace {

template <int AOFMT, bool RTNE, int DIV>
__global__ __launch_bounds__(kThreads, 1) void upsample_kernel(const UpsampleArgs a)
{
    __shared__ __attribute__((aligned(16))) float smem[UpsLds<false>::kFloats];
    upsample_tile_checked<AOFMT, RTNE, false, DIV>(a, smem, xcd_contiguous(blockIdx.x, gridDim.x), blockIdx.z);
}

// Upsample.main: the full-resolution pass.  HiResDB comes from the caller's raw depth frames (HiDepthArgs); RAW_F32 = they are f32
// (the instance with the format switch is compiled for six workgroups per CU: its decode sequences need the registers).
template <int AOFMT, bool RTNE, int DIV, bool RAW_F32>
__global__ __launch_bounds__(kThreads, RAW_F32 ? 7 : 6) void upsample_final_kernel(const UpsampleArgs a, const HiDepthArgs hi)
{
    __shared__ __attribute__((aligned(16))) float smem[UpsLds<true>::kFloats];
    upsample_tile_checked<AOFMT, RTNE, true, DIV, NoHook, ups_tile_h(true), RAW_F32>(a, smem, xcd_contiguous(blockIdx.x, gridDim.x), blockIdx.z,
                                                                                  NoHook(), &hi);
}

// Upsample.main for calls with few tiles (one 1080p frame: 510 tiles of 64 x 64 on 256 CUs): 64 x 32 tiles, twice
// the workgroups, half the serial work in each.
template <int AOFMT, bool RTNE, int DIV, bool RAW_F32>
__global__ __launch_bounds__(kThreads) void upsample_final_small_kernel(const UpsampleArgs a, const HiDepthArgs hi)
{
    __shared__ __attribute__((aligned(16))) float smem[UpsLds<true, kUpsTileHSmall>::kFloats];
    upsample_tile_checked<AOFMT, RTNE, true, DIV, NoHook, kUpsTileHSmall, RAW_F32>(a, smem, xcd_contiguous(blockIdx.x, gridDim.x), blockIdx.z,
                                                                                NoHook(), &hi);
}

// Upsample.main_blendout with the full-resolution pass's 64 x 64 tiles, for launches of many tiles (L2 -> L1 of a batch: 16 320 tiles
// of 64 x 32 at 4K x 16): half the barriers and window fills per texel, apron share 1.4x instead of 1.6x; six workgroups per CU
// (23.7 KB windows, <= 80 VGPRs).  MEAO_DEBUG_BLEND_TALL_MIN_TILES.
template <int AOFMT, bool RTNE, int DIV>
__global__ __launch_bounds__(kThreads, 6) void upsample_blend_tall_kernel(const UpsampleArgs a)
{
    __shared__ __attribute__((aligned(16))) float smem[UpsLds<false, kUpsTileHTall>::kFloats];
    upsample_tile_checked<AOFMT, RTNE, false, DIV, NoHook, kUpsTileHTall>(a, smem, xcd_contiguous(blockIdx.x, gridDim.x), blockIdx.z);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers

template <int AOFMT, bool RTNE, int DIV>
static void launch_upsample_t(const UpsampleArgs &a, const HiDepthArgs *hi, dim3 grid, hipStream_t s)
{
    const dim3 block(kThreads);
    if (hi) {
        const bool f32 = hi->depth_format == MEAO_DEPTH_F32;
        if (a.tile_h == kUpsTileHSmall) {
            if (f32) upsample_final_small_kernel<AOFMT, RTNE, DIV, true><<<grid, block, 0, s>>>(a, *hi);
            else upsample_final_small_kernel<AOFMT, RTNE, DIV, false><<<grid, block, 0, s>>>(a, *hi);
        } else {
            if (f32) upsample_final_kernel<AOFMT, RTNE, DIV, true><<<grid, block, 0, s>>>(a, *hi);
            else upsample_final_kernel<AOFMT, RTNE, DIV, false><<<grid, block, 0, s>>>(a, *hi);
        }
    } else if (a.tile_h == kUpsTileHTall) {
        upsample_blend_tall_kernel<AOFMT, RTNE, DIV><<<grid, block, 0, s>>>(a);
    } else {
        upsample_kernel<AOFMT, RTNE, DIV><<<grid, block, 0, s>>>(a);
    }
}

hipError_t launch_upsample(const UpsampleArgs &a, const HiDepthArgs *hi, int ao_format, int frames, hipStream_t s)
{
    const dim3 grid(a.tiles_x * a.tiles_y, 1, frames);
    // exact_rcp_div is only ever set together with RTZ depth storage (no inf operands)
    if (ao_format == MEAO_AO_R8) {
        if (a.f16_rtne) launch_upsample_t<MEAO_AO_R8, true, DIV_IEEE>(a, hi, grid, s);
        else if (a.exact_rcp_div) launch_upsample_t<MEAO_AO_R8, false, DIV_EXACT_RCP>(a, hi, grid, s);
        else launch_upsample_t<MEAO_AO_R8, false, DIV_IEEE>(a, hi, grid, s);
    } else {
        if (a.f16_rtne) launch_upsample_t<MEAO_AO_F16, true, DIV_IEEE>(a, hi, grid, s);
        else if (a.exact_rcp_div) launch_upsample_t<MEAO_AO_F16, false, DIV_EXACT_RCP>(a, hi, grid, s);
        else launch_upsample_t<MEAO_AO_F16, false, DIV_IEEE>(a, hi, grid, s);
    }
    return hipGetLastError();
}


}  // namespace meao
