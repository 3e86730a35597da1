// meao_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the multi-scale SSAO hot path.
//
// One workgroup = 256 threads = 4 wave64 (render: 512).  Kernels (stream order, one launch each per batch):
//   downsample_kernel  Downsample1.main + Downsample2.main   (DS1:52-81, DS2:32-51): the four point-sampled levels.  LinearZ
//                      (DS1:46) is not stored -- its one reader, the full-resolution upsample, evaluates it from the raw depth
//   render_kernel      Render.main_interleaved, all levels   (REN:112-177)
//   render_with_composite_kernel   the same, carrying the composite of an earlier call in its texel loop
//   render_wide_kernel Render.main on LowDepth<k> (opt-in hq_levels variant)
//   upsample_kernel / upsample_final_kernel    Upsample.main_blendout / main [/ main_premin*]   (UPS:185-233)
//   upsample_two_level_kernel / upsample_three_level_kernel   main_blendout L3->L2 (L2->L1) with the
//                      pass(es) below evaluated inside the same launch
//   upsample_final_with_next_downsample_kernel   Upsample.main of this batch + the downsample pass
//                      of the next one (meao_prefetch_batch): streaming hidden under VALU-bound work
// (DS1/DS2/REN/UPS = Assets/MiniEngineAO/Shaders/{Downsample1,Downsample2,Render,Upsample}.compute)
//
// Numerics contract: binary32, RNE, correctly rounded '/' (exact v_rcp_f32-based
// sequences or hipcc's IEEE expansion), compiled with -ffp-contract=off; the only fused
// operations are the explicit mad()/fma2() calls, placed where the HLSL source has a*b+c in one
// expression.  Bit-exact against oracle/meao_oracle.c and against the reference's own source
// executed by oracle/{csharp,hlsl}_interp.py.
//
// MI355X design notes:
//  * The 4x4 de-interleaved TiledDepth arrays are never materialised on the hot path: a
//    workgroup that renders ALL 16 slices of a 128x32 output tile needs exactly one contiguous
//    (128+32)x(32+32) window of LowDepth<level>, so the kernel stages that window in LDS
//    (applying the per-slice clamp addressing and the atlas padding rule while filling) and
//    samples it with a stride of 4.  Output rows are then contiguous instead of a 4-byte
//    strided scatter of single R8 texels.
//  * Each lane renders horizontally adjacent texel pairs so every LDS sample is one
//    conflict-free ds_read_b64; saturate() folds into the clamp modifier of v_mul/v_fma and
//    clamp(d, p, 1) is one v_med3_f32, so a sample pair costs exactly 8 VALU ops per texel.  The reads
//    of the next pair are issued by hand before the current pair is evaluated, and the per-term constants
//    are VGPR operands: an SGPR source halves the VALU issue rate on this part (tools/ubench_issue.hip).
//  * Upsample uses 64x64 hi-res tiles in the full-resolution pass (64x32 in the blend passes):
//    1.4x apron amplification instead of the reference's 2.6x, >= 89 % of the lanes busy in both
//    blur phases, 16-byte loads of the hi-res depth and 4-byte stores of four AO texels, LDS
//    carved so that seven workgroups share a CU.
//  * vmcnt retires loads in issue order: loads whose data is needed late are issued BEHIND the ones
//    needed first (window before hi-res operands), and unrelated streaming work (the next batch's
//    downsample tile, a carried composite) puts its loads in flight inside the tile, after the tile's
//    own loads have landed, through hooks of upsample_tile / render_tile.
//  * The small blend passes are evaluated inside the launch of the pass above them by recomputation
//    (blend_window_into_lds): no inter-workgroup synchronisation, bit-identical buffers.
//  * The SIMDs issue oldest-wave-first.  A hardware-dispatched workgroup is born youngest and ages while its tile
//    progresses, so the furthest-along tile always goes first -- a software pipeline across tiles for free, and the
//    reason persistent-workgroup forms of these kernels lost to the plain launches (round 3, LABNOTES.md).
//  * v_rcp_f32 pays ~3 cycles when it follows a non-transcendental instruction: the four weight reciprocals of a
//    bilateral texel are issued back to back (bilateral_upsample_grouped).
//  * Results that are stored as UNORM8 do not need the correction steps of their divisions wherever the uncorrected
//    quotient provably converts to the same code: bilateral_upsample_r8 checks that from the estimate itself (distance
//    of the scaled value from the next rounding boundary against a proven error bound) and redoes the rare texel exactly.
//    Not in the kernel that carries the next batch's downsample tile: it waits on memory, not on VALU issue.
//
// This file is the UNITY form: every kernel translation unit in one (what the `clocks` variant and tools/isa_diff.py build).
// The product library compiles the parts separately and in parallel (miniengineao_amd/build.py KERNEL_UNITS).
#define MEAO_UNITY_BUILD 1
#include "meao_k_downsample.hip"
#include "meao_k_render.hip"
#include "meao_k_upsample.hip"
#include "meao_k_upsample_nested.hip"
#include "meao_k_upsample_fused.hip"
#include "meao_k_misc.hip"
