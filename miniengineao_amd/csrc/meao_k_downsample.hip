// meao_k_downsample.hip -- the stand-alone downsample pass (the four point-sampled levels; meao_dev_downsample.hpp).
#include "meao_dev_downsample.hpp"

namespace meao {
namespace {

// ROWS = kMipRowsPerLane: tiles of 128 x 16 LowDepth1 texels.  ROWS = 1, small calls (a 1080p frame: 272 tiles of 128 x 16):
// tiles of 128 x 8, twice the workgroups, one load-compute-store round each instead of two in a row.
template <bool VEC, int DIV, int ROWS>
__global__ __launch_bounds__(kThreads) void downsample_kernel(const DownsampleArgs a)
{
    downsample_tile<VEC, DIV, ROWS>(a, blockIdx.x, blockIdx.z);
}

template <bool VEC, int DIV>
void launch_downsample_t(const DownsampleArgs &a, dim3 grid, hipStream_t s)
{
    if (a.rows_per_lane == 1) downsample_kernel<VEC, DIV, 1><<<grid, dim3(kThreads), 0, s>>>(a);
    else downsample_kernel<VEC, DIV, kMipRowsPerLane><<<grid, dim3(kThreads), 0, s>>>(a);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers

hipError_t launch_downsample(const DownsampleArgs &a, int frames, hipStream_t s)
{
    if (a.rows_per_lane != 1 && a.rows_per_lane != kMipRowsPerLane) return hipErrorInvalidValue;
    const dim3 grid(a.tiles_x * a.tiles_y, 1, frames);
    // exact_rcp_div is only ever set together with RTZ depth storage; the pass itself stores f32 (no f16 conversion here)
    if (a.exact_rcp_div) {
        if (a.vec_ok) launch_downsample_t<true, DIV_EXACT_RCP>(a, grid, s);
        else launch_downsample_t<false, DIV_EXACT_RCP>(a, grid, s);
    } else {
        if (a.vec_ok) launch_downsample_t<true, DIV_IEEE>(a, grid, s);
        else launch_downsample_t<false, DIV_IEEE>(a, grid, s);
    }
    return hipGetLastError();
}


}  // namespace meao
