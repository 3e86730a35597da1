// meao_k_downsample.hip -- downsample kernels: stand-alone pass, small-call pass, side-stream co-runner.
#include "meao_dev_downsample.hpp"

namespace meao {
namespace {

template <bool RTNE, bool VEC, int DIV>
__global__ __launch_bounds__(kThreads) void downsample_kernel(const DownsampleArgs a)
{
    downsample_tile<RTNE, VEC, DIV>(a, blockIdx.x, blockIdx.z);
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


}  // namespace meao
