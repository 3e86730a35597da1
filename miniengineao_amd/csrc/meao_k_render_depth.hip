// meao_k_render_depth.hip -- an OPTION (meao_debug_set MEAO_DEBUG_RENDER_FROM_DEPTH, off by default), built for the reference's
// own calling pattern, one frame per call (AmbientOcclusion.cs:329-347): the render pass fills its windows from the caller's RAW
// depth buffer (render_tile<FROM_DEPTH>), so it does not wait for the downsample pass, and the downsample pass can ride in the
// SAME launch as extra workgroups behind the render ones -- render + downsample | blend passes | full-resolution pass: three
// dependent launches instead of four.  Bit-exact (every launch-structure test runs it), and measured NOT faster: a window of
// LowDepth<k> gathered from the raw frame touches 2^k times the cache lines of the stored mip (L1: every other texel of every other
// row), so the merged launch takes 33.0 us for a 4K frame where the two separate launches take 15.4 + 18.6 us, the call 61.0 us
// against 60.4 (1080p: 33.4 against 30.9); as two launches on two streams the event hand-overs cost more than the overlap gains
// (86 us).  profiles/r05_from_depth_sweep.jsonl, LABNOTES.md round 5.  Kept as a tested launch structure; the default stays the
// stored-mip sequence.
#include "meao_dev_render.hpp"
#include "meao_dev_downsample.hpp"

namespace meao {
namespace {

// Workgroups [0, a.blocks_per_frame) of a frame render (dispatched first: they are the long ones), the rest run the
// downsample pass, two 128 x (8 * d.row_passes) tiles per 512-thread workgroup; d.tile_end = 0: render only (the pass runs as
// its own launch on another stream: MEAO_DEBUG_RENDER_FROM_DEPTH 2).
// (Register budget of 6 waves per SIMD: this launch structure is chosen for calls of a few workgroups per CU; at the 64 registers
// of 8 waves the 32-row form spills four of its 20 window loads.)
template <int AOFMT, bool RTNE, int DIV, int TILE_H>
__global__ __launch_bounds__(ren_tile_w(false) * 4, 6) void render_from_depth_kernel(const RenderArgs a,
                                                                                                                const DownsampleArgs d)
{
    __shared__ __attribute__((aligned(16))) float tile[(TILE_H + 2 * kRenApron) * (ren_tile_w(false) + 2 * kRenApron)];
    const int frame = blockIdx.y;
    if (static_cast<int>(blockIdx.x) >= a.blocks_per_frame) {
        const int t = 2 * (static_cast<int>(blockIdx.x) - a.blocks_per_frame) + static_cast<int>(threadIdx.x >> 8);
        const unsigned tid = threadIdx.x & 255u;
        if (t >= d.tile_end) return;
        if (d.row_passes == 1) {
            if (d.vec_ok) downsample_tile<RTNE, true, DIV, 1>(d, t, frame, tid);
            else downsample_tile<RTNE, false, DIV, 1>(d, t, frame, tid);
        } else {
            if (d.vec_ok) downsample_tile<RTNE, true, DIV>(d, t, frame, tid);
            else downsample_tile<RTNE, false, DIV>(d, t, frame, tid);
        }
        return;
    }
    render_tile<AOFMT, RTNE, DIV, false, NoRenderHook, TILE_H, ren_tile_w(false) * 4, true>(
        a, tile, frame, xcd_contiguous(blockIdx.x, a.blocks_per_frame), NoRenderHook(), &d);
}

template <int AOFMT, bool RTNE, int DIV>
void launch_render_from_depth_t(const RenderArgs &a, const DownsampleArgs &d, dim3 grid, hipStream_t s)
{
    const dim3 block(ren_tile_w(false) * 4);
    if (a.tile_h == kRenTileHSmall) render_from_depth_kernel<AOFMT, RTNE, DIV, kRenTileHSmall><<<grid, block, 0, s>>>(a, d);
    else render_from_depth_kernel<AOFMT, RTNE, DIV, kRenTileH><<<grid, block, 0, s>>>(a, d);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers

// with_downsample: the launch also runs the downsample pass `d` describes (tiles [0, d.tile_end) of every frame); otherwise `d`
// only names the raw depth frames and the Z-buffer parameters.  f32 depth and the 36-sample set only.
hipError_t launch_render_from_depth(const RenderArgs &a, const DownsampleArgs &d_in, bool with_downsample, int ao_format, int frames,
                                    hipStream_t s)
{
    if (a.exhaustive || d_in.depth_format != MEAO_DEPTH_F32) return hipErrorInvalidValue;
    DownsampleArgs d = d_in;
    if (!with_downsample) d.tile_end = 0;
    const dim3 grid(a.blocks_per_frame + (d.tile_end + 1) / 2, frames, 1);
    if (ao_format == MEAO_AO_R8) {
        if (a.f16_rtne) launch_render_from_depth_t<MEAO_AO_R8, true, DIV_IEEE>(a, d, grid, s);
        else if (a.exact_rcp_div == 2) launch_render_from_depth_t<MEAO_AO_R8, false, DIV_FAST>(a, d, grid, s);
        else if (a.exact_rcp_div) launch_render_from_depth_t<MEAO_AO_R8, false, DIV_EXACT_RCP>(a, d, grid, s);
        else launch_render_from_depth_t<MEAO_AO_R8, false, DIV_IEEE>(a, d, grid, s);
    } else {
        if (a.f16_rtne) launch_render_from_depth_t<MEAO_AO_F16, true, DIV_IEEE>(a, d, grid, s);
        else if (a.exact_rcp_div == 2) launch_render_from_depth_t<MEAO_AO_F16, false, DIV_FAST>(a, d, grid, s);
        else if (a.exact_rcp_div) launch_render_from_depth_t<MEAO_AO_F16, false, DIV_EXACT_RCP>(a, d, grid, s);
        else launch_render_from_depth_t<MEAO_AO_F16, false, DIV_IEEE>(a, d, grid, s);
    }
    return hipGetLastError();
}

}  // namespace meao
