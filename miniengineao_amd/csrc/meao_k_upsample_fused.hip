// meao_k_upsample_fused.hip -- upsample kernels that carry the next batch's downsample pass (meao_prefetch_batch).
#include "meao_dev_upsample.hpp"
#include "meao_dev_downsample.hpp"

namespace meao {
namespace {

// Hook of the fused last kernel: puts the four 16-byte depth loads of the carried downsample tile in flight
// inside the upsample tile, before its bilateral phase (A/B against "tile first" and "after the prefetch":
// profiles/r02_ab_v15p..v17p_split_ds*.jsonl).
struct IssueCarriedLoads {
    static constexpr bool kBeforeBilateral = true;
    static constexpr bool kGroupReciprocals = false;     // the kernels that carry a downsample tile are short of registers: A/B +3 %
    static constexpr bool kEstimateR8 = false;           // ... and wait on memory, not on VALU issue: 10 % fewer instructions, +4 us
    static constexpr bool kReuseEstimate = false;
    static constexpr bool kPairReciprocals = false;
    const DownsampleArgs &d;
    float (&v)[kDsTileH / kDsRowsPerPass][4];
    bool mine;
    int tile, frame;
    __device__ __forceinline__ void issue() const { if (mine) downsample_tile_load<true, true>(d, tile, frame, v); }
    __device__ __forceinline__ void after_prefetch() const { if constexpr (!kBeforeBilateral) issue(); }
    __device__ __forceinline__ void before_bilateral() const { if constexpr (kBeforeBilateral) issue(); }
};


// The same for the lean tile (downsample_lean_load / _finish: wave-uniform row parity, ~8 VALU instructions per texel instead of ~18):
// what the last kernel of a pipelined step carries since round 4 (270 vs 275 us per 16 frames, profiles/r04_ab_fused_lean_tile.jsonl).
struct IssueCarriedLoadsLean {
    static constexpr bool kBeforeBilateral = true;
    // Forms of the bilateral texel (A/B with the whole-tile copy of the phase, profiles/r04_ab_fused_bilateral_forms.jsonl; before that
    // copy existed both lost here): exact sequences 272 us, UNORM8 estimate 257, grouped reciprocals 264, both 256 us per 16 frames.
    static constexpr bool kGroupReciprocals = true;
    static constexpr bool kEstimateR8 = true;
    static constexpr bool kReuseEstimate = false;        // (70 of the 72 VGPRs that seven workgroups per CU allow: reuse spills)
    static constexpr bool kPairReciprocals = MEAO_X_BIL_PAIR_RCP != 0;
    const DownsampleArgs &d;
    float4v (&q)[kDsTileH / kDsRowsPerPass];
    bool mine, full;
    int tile, frame;
    __device__ __forceinline__ void after_prefetch() const {}
    __device__ __forceinline__ void before_bilateral() const
    {
        if (!mine) return;
        if (full) downsample_lean_load<kDsTileH / kDsRowsPerPass, true>(d, tile, frame, q);
        else downsample_lean_load<kDsTileH / kDsRowsPerPass, false>(d, tile, frame, q);
    }
};

// Upsample.main of this batch carrying the downsample pass of the NEXT batch (meao_prefetch_batch):
// the final upsample is VALU-bound (five exact divides per texel) and leaves HBM idle, the
// downsample is pure streaming with ~2 VALU ops per byte -- inside one kernel the streaming hides
// under the arithmetic of the other resident workgroups instead of costing a pass of its own.
// The downsample tiles (128 x 32 texels) of `d` are spread over this kernel's grid; each workgroup
// streams its share first and then does its upsample tile.
template <int AOFMT, bool RTNE, int DIV>
__global__ __launch_bounds__(kThreads, 7) void upsample_final_with_next_downsample_kernel(const UpsampleArgs a,
                                                                                       const DownsampleArgs d)
{
    __shared__ __attribute__((aligned(16))) float smem[UpsLds<true>::kFloats];
    auto carried_downsample = [&]() __attribute__((always_inline)) {
        const int ds_tiles = d.tiles_x * d.tiles_y;
        const bool vec = d.vec_ok != 0;
        for (int f = blockIdx.z; f < d.frames; f += gridDim.z)
            for (int t = blockIdx.x; t < ds_tiles; t += gridDim.x) {
                if (vec) downsample_tile<RTNE, true, DIV>(d, t, f);
                else downsample_tile<RTNE, false, DIV>(d, t, f);
            }
    };
    // One downsample tile per workgroup (the usual case: both grids tile the same frame) with 16-byte f32
    // loads: its four loads per lane go out after the upsample tile's prefetch wait -- issued earlier they
    // would sit in front of that wait (vmcnt counts in order) -- and are consumed after the bilateral phase.
    const int ds_tiles = d.tiles_x * d.tiles_y;
    const bool split = d.vec_ok != 0 && d.depth_format == MEAO_DEPTH_F32 && gridDim.x >= static_cast<unsigned>(ds_tiles) &&
                       gridDim.z >= static_cast<unsigned>(d.frames);
    if (!split) {       // (the host only moves tiles into a blend pass when the split form applies: tile_begin = 0 here)
        carried_downsample();
        upsample_tile_checked<AOFMT, RTNE, true, DIV>(a, smem, xcd_contiguous(blockIdx.x, gridDim.x), blockIdx.z);
        return;
    }
    // (tiles below d.tile_begin were carried by an earlier launch of this call: a blend pass, MEAO_DEBUG_DS_SHARE_IN_BLEND)
    const bool mine = blockIdx.x >= static_cast<unsigned>(d.tile_begin) && blockIdx.x < static_cast<unsigned>(ds_tiles) &&
                      blockIdx.z < static_cast<unsigned>(d.frames);
    constexpr int kPasses = kDsTileH / kDsRowsPerPass;
    float4v q[kPasses];
    const bool full = (static_cast<int>(blockIdx.x) / d.tiles_x + 1) * kDsTileH <= d.h[0];
    const IssueCarriedLoadsLean issue = {d, q, mine, full, static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.z)};
    upsample_tile_checked<AOFMT, RTNE, true, DIV>(a, smem, xcd_contiguous(blockIdx.x, gridDim.x), blockIdx.z, issue);
    if (mine) {
        if (full) downsample_lean_finish<RTNE, DIV, kPasses, true>(d, blockIdx.x, blockIdx.z, q);
        else downsample_lean_finish<RTNE, DIV, kPasses, false>(d, blockIdx.x, blockIdx.z, q);
    }
    // (loading the carried tile behind the first barrier and finishing it in FRONT of the bilateral phase frees 10 VGPRs there
    // and is 5 % slower: profiles/r03_ab_fused_ds_finished_before_bilateral.jsonl)
}

// Upsample.main_blendout L2 -> L1 carrying the first d.tile_end downsample tiles (per frame) of the NEXT batch: the
// blend passes wait on latency with issue slots and HBM idle, the fused last kernel is short of both
// (MEAO_DEBUG_DS_SHARE_IN_BLEND; the last kernel then starts at d.tile_begin = this launch's tile_end).
template <int AOFMT, bool RTNE, int DIV>
__global__ __launch_bounds__(kThreads) void upsample_blend_with_next_downsample_kernel(const UpsampleArgs a, const DownsampleArgs d)
{
    __shared__ __attribute__((aligned(16))) float smem[UpsLds<false>::kFloats];
    const bool mine = blockIdx.x < static_cast<unsigned>(d.tile_end) && blockIdx.z < static_cast<unsigned>(d.frames);
    float v[kDsTileH / kDsRowsPerPass][4];
    const IssueCarriedLoads issue = {d, v, mine, static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.z)};
    upsample_tile_checked<AOFMT, RTNE, false, DIV>(a, smem, xcd_contiguous(blockIdx.x, gridDim.x), blockIdx.z, issue);
    if (mine) downsample_tile_finish<RTNE, true, DIV>(d, blockIdx.x, blockIdx.z, v);
}


}  // namespace

// ------------------------------------------------------------------------------------------
// launchers

template <int AOFMT, bool RTNE, int DIV>
static void launch_upsample_fused_t(const UpsampleArgs &a, const DownsampleArgs &d, dim3 grid, hipStream_t s)
{
    upsample_final_with_next_downsample_kernel<AOFMT, RTNE, DIV><<<grid, dim3(kThreads), 0, s>>>(a, d);
}

hipError_t launch_upsample_final_with_downsample(const UpsampleArgs &a, const DownsampleArgs &d, int ao_format,
                                                 int frames, hipStream_t s)
{
    const dim3 grid(a.tiles_x * a.tiles_y, 1, frames);
    if (ao_format == MEAO_AO_R8) {
        if (a.f16_rtne) launch_upsample_fused_t<MEAO_AO_R8, true, DIV_IEEE>(a, d, grid, s);
        else if (a.exact_rcp_div == 2) launch_upsample_fused_t<MEAO_AO_R8, false, DIV_FAST>(a, d, grid, s);
        else if (a.exact_rcp_div) launch_upsample_fused_t<MEAO_AO_R8, false, DIV_EXACT_RCP>(a, d, grid, s);
        else launch_upsample_fused_t<MEAO_AO_R8, false, DIV_IEEE>(a, d, grid, s);
    } else {
        if (a.f16_rtne) launch_upsample_fused_t<MEAO_AO_F16, true, DIV_IEEE>(a, d, grid, s);
        else if (a.exact_rcp_div == 2) launch_upsample_fused_t<MEAO_AO_F16, false, DIV_FAST>(a, d, grid, s);
        else if (a.exact_rcp_div) launch_upsample_fused_t<MEAO_AO_F16, false, DIV_EXACT_RCP>(a, d, grid, s);
        else launch_upsample_fused_t<MEAO_AO_F16, false, DIV_IEEE>(a, d, grid, s);
    }
    return hipGetLastError();
}

template <int AOFMT, bool RTNE, int DIV>
static void launch_upsample_blend_ds_t(const UpsampleArgs &a, const DownsampleArgs &d, dim3 grid, hipStream_t s)
{
    upsample_blend_with_next_downsample_kernel<AOFMT, RTNE, DIV><<<grid, dim3(kThreads), 0, s>>>(a, d);
}

hipError_t launch_upsample_blend_with_downsample(const UpsampleArgs &a, const DownsampleArgs &d, int ao_format, int frames, hipStream_t s)
{
    const dim3 grid(a.tiles_x * a.tiles_y, 1, frames);
    if (d.vec_ok == 0 || d.depth_format != MEAO_DEPTH_F32 || d.tile_end > static_cast<int>(grid.x) || d.frames > frames)
        return hipErrorInvalidValue;      // the caller checks the same conditions before it moves tiles here
    if (ao_format == MEAO_AO_R8) {
        if (a.f16_rtne) launch_upsample_blend_ds_t<MEAO_AO_R8, true, DIV_IEEE>(a, d, grid, s);
        else if (a.exact_rcp_div == 2) launch_upsample_blend_ds_t<MEAO_AO_R8, false, DIV_FAST>(a, d, grid, s);
        else if (a.exact_rcp_div) launch_upsample_blend_ds_t<MEAO_AO_R8, false, DIV_EXACT_RCP>(a, d, grid, s);
        else launch_upsample_blend_ds_t<MEAO_AO_R8, false, DIV_IEEE>(a, d, grid, s);
    } else {
        if (a.f16_rtne) launch_upsample_blend_ds_t<MEAO_AO_F16, true, DIV_IEEE>(a, d, grid, s);
        else if (a.exact_rcp_div == 2) launch_upsample_blend_ds_t<MEAO_AO_F16, false, DIV_FAST>(a, d, grid, s);
        else if (a.exact_rcp_div) launch_upsample_blend_ds_t<MEAO_AO_F16, false, DIV_EXACT_RCP>(a, d, grid, s);
        else launch_upsample_blend_ds_t<MEAO_AO_F16, false, DIV_IEEE>(a, d, grid, s);
    }
    return hipGetLastError();
}


}  // namespace meao
