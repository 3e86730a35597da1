// meao_k_upsample_fused.hip -- the full-resolution upsample kernel that carries the next batch's downsample pass (meao_prefetch_batch).
#include "meao_dev_upsample.hpp"
#include "meao_dev_downsample.hpp"

namespace meao {
namespace {

// Hook of the fused last kernel: puts the two 16-byte depth loads of the carried (lean) downsample tile in flight inside the
// upsample tile, before its bilateral phase -- after the tile's own hoisted operands have landed, so that nothing in the
// bilateral phase waits behind them (vmcnt retires in order) -- to be consumed after it (A/B against "tile first" and "after
// the prefetch": profiles/r02_ab_v15p..v17p_split_ds*.jsonl).
struct IssueCarriedLoadsLean {
    static constexpr bool kBeforeBilateral = true;
    // Forms of the bilateral texel (A/B with the whole-tile copy of the phase, profiles/r04_ab_fused_bilateral_forms.jsonl; before that
    // copy existed both lost here): exact sequences 272 us, UNORM8 estimate 257, grouped reciprocals 264, both 256 us per 16 frames.
    static constexpr bool kGroupReciprocals = true;
    static constexpr bool kEstimateR8 = true;
    static constexpr bool kReuseEstimate = false;        // (round 6, with registers to spare: 240.1 vs 240.3 us -- the exact path is rare; left off)
    static constexpr bool kPairReciprocals = MEAO_X_BIL_PAIR_RCP != 0;
    const DownsampleArgs &d;
    float4v (&q)[2];
    bool mine, full;
    int tile, frame;
    __device__ __forceinline__ void after_prefetch() const {}
    __device__ __forceinline__ void before_bilateral() const
    {
        if (!mine) return;
        if (full) downsample_lean_load<true>(d, tile, frame, q);
        else downsample_lean_load<false>(d, tile, frame, q);
    }
};

// Upsample.main of this batch carrying the downsample pass of the NEXT batch (meao_prefetch_batch).  The carried pass is pure
// streaming with ~2 VALU instructions per byte; inside this kernel its traffic overlaps the arithmetic of the other resident
// workgroups instead of costing a launch of its own between two VALU-bound ones.  Since round 6 both halves move fewer bytes:
// the pass writes the four levels only (it reads the even rows of the next frames: 27.6 MB per 4K frame instead of 60.8) and the
// upsample tile linearizes its HiResDB from the raw depth of ITS frames (33.2 MB read instead of 16.6 written + 16.6 read back).
// One lean downsample tile (64 x 16 LowDepth1 texels) per upsample tile (64 x 64): d.tiles_x * d.tiles_y <= gridDim.x
// (fused_downsample_applicable); each workgroup puts its tile's loads in flight inside its upsample tile and finishes it after.
template <int AOFMT, bool RTNE, int DIV>
__global__ __launch_bounds__(kThreads, 7) void upsample_final_with_next_downsample_kernel(const UpsampleArgs a, const HiDepthArgs hi,
                                                                                       const DownsampleArgs d)
{
    __shared__ __attribute__((aligned(16))) float smem[UpsLds<true>::kFloats];
    const bool mine = blockIdx.x < static_cast<unsigned>(d.tiles_x * d.tiles_y) && blockIdx.z < static_cast<unsigned>(d.frames);
    float4v q[2];
    const bool full = (static_cast<int>(blockIdx.x) / d.tiles_x + 1) * kLeanRows <= d.h[1];
    const IssueCarriedLoadsLean issue = {d, q, mine, full, static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.z)};
    upsample_tile_checked<AOFMT, RTNE, true, DIV>(a, smem, xcd_contiguous(blockIdx.x, gridDim.x), blockIdx.z, issue, &hi);
    if (mine) {
        if (full) downsample_lean_finish<DIV, true>(d, blockIdx.x, blockIdx.z, q);
        else downsample_lean_finish<DIV, false>(d, blockIdx.x, blockIdx.z, q);
    }
    // (loading the carried tile behind the first barrier and finishing it in FRONT of the bilateral phase frees its VGPRs there
    // and is 5 % slower: profiles/r03_ab_fused_ds_finished_before_bilateral.jsonl)
}


}  // namespace

// ------------------------------------------------------------------------------------------
// launchers

// The fused form needs f32 depth on both sides, frames the 16-byte loads of the lean tile apply to (d.vec_ok: W % 8 == 0, aligned)
// and a grid that has a workgroup for every carried tile.  `d` must be tiled for the lean tile: tiles of kLeanW x kLeanRows.
bool fused_downsample_applicable(const UpsampleArgs &a, const HiDepthArgs &hi, const DownsampleArgs &d, int frames)
{
    return hi.depth_format == MEAO_DEPTH_F32 && d.depth_format == MEAO_DEPTH_F32 && d.vec_ok != 0 && a.tile_h == ups_tile_h(true) &&
           d.tiles_x * d.tiles_y <= a.tiles_x * a.tiles_y && d.frames <= frames;
}

template <int AOFMT, bool RTNE, int DIV>
static void launch_upsample_fused_t(const UpsampleArgs &a, const HiDepthArgs &hi, const DownsampleArgs &d, dim3 grid, hipStream_t s)
{
    upsample_final_with_next_downsample_kernel<AOFMT, RTNE, DIV><<<grid, dim3(kThreads), 0, s>>>(a, hi, d);
}

hipError_t launch_upsample_final_with_downsample(const UpsampleArgs &a, const HiDepthArgs &hi, const DownsampleArgs &d, int ao_format,
                                                 int frames, hipStream_t s)
{
    if (!fused_downsample_applicable(a, hi, d, frames)) return hipErrorInvalidValue;      // the caller asks first
    const dim3 grid(a.tiles_x * a.tiles_y, 1, frames);
    if (ao_format == MEAO_AO_R8) {
        if (a.f16_rtne) launch_upsample_fused_t<MEAO_AO_R8, true, DIV_IEEE>(a, hi, d, grid, s);
        else if (a.exact_rcp_div) launch_upsample_fused_t<MEAO_AO_R8, false, DIV_EXACT_RCP>(a, hi, d, grid, s);
        else launch_upsample_fused_t<MEAO_AO_R8, false, DIV_IEEE>(a, hi, d, grid, s);
    } else {
        if (a.f16_rtne) launch_upsample_fused_t<MEAO_AO_F16, true, DIV_IEEE>(a, hi, d, grid, s);
        else if (a.exact_rcp_div) launch_upsample_fused_t<MEAO_AO_F16, false, DIV_EXACT_RCP>(a, hi, d, grid, s);
        else launch_upsample_fused_t<MEAO_AO_F16, false, DIV_IEEE>(a, hi, d, grid, s);
    }
    return hipGetLastError();
}


}  // namespace meao
