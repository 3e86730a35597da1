// meao_k_upsample.hip -- upsample kernels: one blend pass / the full-resolution pass per launch.
#include "meao_dev_upsample.hpp"

namespace meao {
namespace {

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
static void launch_upsample_t(const UpsampleArgs &a, bool final_pass, dim3 grid, hipStream_t s)
{
    if (final_pass && a.tile_h == kUpsTileHSmall) upsample_final_small_kernel<AOFMT, RTNE, DIV><<<grid, dim3(kThreads), 0, s>>>(a);
    else if (final_pass) upsample_kernel<AOFMT, RTNE, true, DIV><<<grid, dim3(kThreads), 0, s>>>(a);
    else if (a.tile_h == kUpsTileHTall) upsample_blend_tall_kernel<AOFMT, RTNE, DIV><<<grid, dim3(kThreads), 0, s>>>(a);
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


}  // namespace meao
