/*
 * meao.h -- C ABI of libmeao_hip.so: the MI355X-native (gfx950 / CDNA4) multi-scale
 * SSAO hot path of keijiro/MiniEngineAO.   depth in  ->  ambient-occlusion texture out.
 *
 * What this boundary replaces.  The reference has no FFI: the path sits behind the Unity
 * component MiniEngineAO.AmbientOcclusion (Assets/MiniEngineAO/AmbientOcclusion.cs, "AO.cs"
 * below), which records ten compute dispatches into a CommandBuffer
 * (AO.cs:496-531 RebuildCommandBuffers).  Every entry point below names the reference
 * lines it stands in for; INTEGRATION.md shows the C# [DllImport] stub and the
 * AmbientOcclusion wrapper a maintainer would add on the reference side.
 *
 * Conventions: plain C, no C++ or torch types; every function returns a meao_status
 * (0 = ok, negative = error) unless stated; no exception crosses the boundary; a context
 * is not thread-safe, distinct contexts are independent (no global state -- the reference's
 * shared statics AO.cs:136-137,592-593 are per-context here); one context per device.
 * The caller owns the input depth and output AO memory; the context owns every
 * intermediate and never allocates inside meao_execute*.
 */
#ifndef MEAO_H
#define MEAO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define MEAO_API __attribute__((visibility("default")))
#else
#define MEAO_API
#endif

#define MEAO_ABI_VERSION 6
#define MEAO_MAX_BATCH 64      /* frames per batched launch */
#define MEAO_NUM_PASSES 7      /* downsample, render, upsample x4, render_hq (see meao_pass) */

typedef struct meao_ctx meao_ctx;
typedef void *meao_stream;     /* hipStream_t; NULL = the context's own stream */

typedef enum meao_status {
    MEAO_OK = 0,
    MEAO_ERR_INVALID_ARGUMENT = -1,
    MEAO_ERR_HIP = -2,            /* HIP runtime error; text via meao_last_error */
    MEAO_ERR_OUT_OF_MEMORY = -3,
    MEAO_ERR_UNSUPPORTED = -4,
    MEAO_ERR_NO_DEVICE = -5,      /* no gfx950 device visible: there is no CPU fallback */
    MEAO_ERR_BUFFER_TOO_SMALL = -6
} meao_status;

/* Storage of the AO targets.  R8 = the reference's FixedUAV / RenderTextureFormat.R8
 * (AO.cs:262-273,466-475): AO is quantised to 8 bits between every pass.  F16 = fp16 AO
 * storage (BASELINE config 5). */
typedef enum meao_ao_format { MEAO_AO_R8 = 0, MEAO_AO_F16 = 1 } meao_ao_format;

/* f32 -> f16 store conversion of the HalfUAV targets (AO.cs:454).  RTZ_CLAMP (round toward
 * zero, finite overflow -> 65504) is canonical; RTNE (overflow -> inf) is the other common
 * hardware behaviour.  Only matters for sky texels (1e5) and the last mantissa bit. */
typedef enum meao_f16_rounding { MEAO_F16_RTZ_CLAMP = 0, MEAO_F16_RTNE = 1 } meao_f16_rounding;

/* Numerics (one mode, no switch): bit-exact against the CPU oracle -- correctly rounded '/', explicit mad fusion only
 * (DESIGN.md section 2).  (ABI <= 5 had a FAST mode with raw 1-ulp reciprocals; it was outside the parity bar and is gone.) */
typedef enum meao_mem { MEAO_MEM_HOST = 0, MEAO_MEM_DEVICE = 1 } meao_mem;

/* Storage of the input depth buffer.  The reference first blits _CameraDepthTexture -- whatever
 * its format -- into an RFloat copy (Blit.shader:48-64 pass 0, AO.cs:608-614) unless D3D's
 * resolved depth is available; here the downsample kernel decodes the format on load, so that
 * pass does not exist.  UNORM formats decode to v / (2^n - 1), correctly rounded (what
 * SAMPLE_DEPTH_TEXTURE returns): UNORM16 = D16, UNORM24 = the low 24 bits of a 32-bit word
 * (D24S8 / D24X8; the high byte is ignored).  F16 decodes exactly. */
typedef enum meao_depth_format {
    MEAO_DEPTH_F32 = 0, MEAO_DEPTH_UNORM16 = 1, MEAO_DEPTH_UNORM24 = 2, MEAO_DEPTH_F16 = 3
} meao_depth_format;

typedef enum meao_format { MEAO_FMT_F32 = 0, MEAO_FMT_F16 = 1, MEAO_FMT_UNORM8 = 2 } meao_format;

/* Sample set of the AO render.  CHECKER = what the reference dispatches (36 samples,
 * Render.compute:160-169).  EXHAUSTIVE = the shader's SAMPLE_EXHAUSTIVELY branch (68 samples, all
 * 12 table slots, Render.compute:144-159), which the reference's host never enables (AO.cs:709
 * "FIXME: should we support SAMPLE_EXHAUSTIVELY mode?"): the weight table is then the un-zeroed
 * AO.cs:696-707 table, normalised the same way. */
typedef enum meao_sample_set { MEAO_SAMPLES_CHECKER = 0, MEAO_SAMPLES_EXHAUSTIVE = 1 } meao_sample_set;

/* Buffers beyond the reference's 17 debug ids: the Render.main (wide) targets of the hq_levels
 * variant, OcclusionHQ1..4 (AO format, dims of L1..L4). */
#define MEAO_DEBUG_OCCLUSION_HQ1 18
#define MEAO_NUM_BUFFERS 21

/* Kernel launches of one frame/batch, in stream order. */
typedef enum meao_pass {
    MEAO_PASS_DOWNSAMPLE = 0,  /* Downsample1.main + Downsample2.main fused  (AO.cs:627-657): the four point-sampled levels.
                                * LinearDepth (Downsample1.compute:46) is not written: its one reader, Upsample.main, evaluates
                                * Linearize from the raw depth frame itself */
    MEAO_PASS_RENDER = 1,      /* Render.main_interleaved, all levels, one grid (AO.cs:519-522) */
    MEAO_PASS_UPSAMPLE_3 = 2,  /* Upsample.main_blendout L4 -> L3             (AO.cs:528).  With 4 levels and
                                * hq_levels = 0 it is evaluated inside the L3 -> L2 launch (each tile computes the
                                * window of Combined3 it reads; Combined3 is still written): no launch of its
                                * own, meao_get_pass_times reports 0 here and the joint time under UPSAMPLE_2 */
    MEAO_PASS_UPSAMPLE_2 = 3,  /* Upsample.main_blendout L3 -> L2             (AO.cs:529).  Small calls (frames x
                                * 64x32 tiles of L1 <= 1024, e.g. one 4K frame or up to four 1080p frames) evaluate it and L4 -> L3 inside
                                * the L2 -> L1 launch: 0 here and under UPSAMPLE_3, the joint time under UPSAMPLE_1 */
    MEAO_PASS_UPSAMPLE_1 = 4,  /* Upsample.main_blendout L2 -> L1             (AO.cs:530) */
    MEAO_PASS_UPSAMPLE_0 = 5,  /* Upsample.main          L1 -> L0 result      (AO.cs:531) */
    MEAO_PASS_RENDER_HQ = 6    /* Render.main (wide) on LowDepth<k>, all hq levels, one grid; launched
                                * right after MEAO_PASS_RENDER (cfg.hq_levels > 0 only) */
} meao_pass;

/* What DoLazyInitialization + RTHandle sizing fix per instance (AO.cs:440-476,276-281). */
typedef struct meao_config {
    uint32_t struct_size;   /* sizeof(meao_config), for ABI evolution */
    int32_t device;         /* HIP device ordinal */
    int32_t width, height;  /* camera.pixelWidth / pixelHeight (AO.cs:339-340) */
    int32_t num_levels;     /* 1..4; the reference always runs 4 (AO.cs:519-531) */
    int32_t ao_format;      /* meao_ao_format */
    int32_t f16_rounding;   /* meao_f16_rounding */
    int32_t max_batch;      /* 1..MEAO_MAX_BATCH frames resident per launch */
    int32_t depth_format;   /* meao_depth_format of the depth pointers given to meao_execute* */
    /* Variants present in the reference's shaders but never dispatched by its host (0 = reference): */
    int32_t hq_levels;      /* 0..num_levels.  The coarsest hq_levels levels additionally run
                             * Render.main (WIDE_SAMPLING, non-interleaved; Render.compute:22,27-29,
                             * 46-50) on LowDepth<k>, and the upsample that consumes level k becomes
                             * Upsample.main_premin / main_premin_blendout (Upsample.compute:23,25,58-60):
                             * LoResAO1 = min(LoResAO1, that render).  Wiring as in the Microsoft
                             * MiniEngine original (its quality levels = hq_levels 0..4). */
    int32_t sample_set;     /* meao_sample_set */
    int32_t pipelined;      /* 1: allocate the second set of downsample buffers at meao_create, so that
                             * meao_prefetch_batch never allocates or synchronises (streams of frames);
                             * 0 (default): the first meao_prefetch_batch call does it, once */
} meao_config;

/* The component's serialized properties (AO.cs:20-68; defaults there) and the camera terms
 * the path reads (AO.cs:561-573). */
typedef struct meao_params {
    uint32_t struct_size;
    float noise_filter_tolerance;   /* [-8, 0]   default  0     AO.cs:20 */
    float blur_tolerance;           /* [-8,-1]   default -4.6   AO.cs:28 */
    float upsample_tolerance;       /* [-12,-1]  default -12    AO.cs:36 */
    float thickness_modifier;       /* [1, 10]   default  1     AO.cs:44 */
    float intensity;                /* [0, 2]    default  1     AO.cs:52 */
    float near_clip, far_clip;      /* camera.nearClipPlane / farClipPlane  AO.cs:563 */
    float proj00;                   /* camera.projectionMatrix[0,0]         AO.cs:572 */
    int32_t reversed_z;             /* SystemInfo.usesReversedZBuffer       AO.cs:564 */
    int32_t single_pass_stereo;     /* singlePassStereoEnabled (AO.cs:392-401): the frame is the double-wide
                                     * eye pair (cfg.width = 2 * camera.pixelWidth, AO.cs:339,502) and
                                     * ThicknessMultiplier doubles (AO.cs:680) */
} meao_params;

/* Description of one of the 17 debug-visible buffers (AO.cs:789-808). */
typedef struct meao_desc {
    int32_t debug_id;       /* 1..17, or 18..21 = OcclusionHQ1..4 */
    int32_t width, height;
    int32_t slices;         /* 16 for TiledDepth1..4 (AO.cs:154), else 1 */
    int32_t format;         /* meao_format */
    uint64_t bytes;         /* width*height*slices*element size */
} meao_desc;

/* Constant buffers exactly as the reference uploads them. */
typedef struct meao_render_constants {   /* CB1 of Render.compute:38-44, AO.cs:687-734 */
    float inv_thickness_table[12];
    float sample_weight_table[12];
    float inv_slice_dimension[2];
    float reject_fadeoff;
    float intensity;
} meao_render_constants;

typedef struct meao_upsample_constants { /* CB1 of Upsample.compute:41-48, AO.cs:760-771 */
    float inv_low_resolution[2];
    float inv_high_resolution[2];
    float noise_filter_strength;
    float step_size;
    float blur_tolerance;
    float upsample_tolerance;
} meao_upsample_constants;

/* ---- library ------------------------------------------------------------------------ */
MEAO_API int32_t meao_abi_version(void);
MEAO_API const char *meao_status_string(int32_t status);
/* Fills the reference defaults (4 levels, R8, RTZ, strict, batch 1; AO.cs:20-68). */
MEAO_API void meao_default_config(meao_config *cfg);
MEAO_API void meao_default_params(meao_params *p);

/* ---- host-side plan: pure CPU, usable without a GPU ------------------------------------ */
/* RTHandle.CalculateDimensions (AO.cs:276-281): ceil(W / 2^level), level 0..6. */
MEAO_API int32_t meao_level_dims(int32_t width, int32_t height, int32_t level,
                                 int32_t *out_w, int32_t *out_h);
/* CalculateZBufferParams (AO.cs:561-568). */
MEAO_API int32_t meao_zbuffer_params(const meao_params *p, float out[4]);
/* PushRenderCommands constant math (AO.cs:660-734) for level 1..4. */
MEAO_API int32_t meao_render_constants_for(int32_t width, int32_t height, const meao_params *p,
                                           int32_t level, meao_render_constants *out);
/* The same for the variants: source_tiled = 0 -> the source is the non-tiled LowDepth<level>
 * (the !source.isTiled branch, AO.cs:679); sample_set = meao_sample_set. */
MEAO_API int32_t meao_render_constants_variant(int32_t width, int32_t height, const meao_params *p,
                                               int32_t level, int32_t source_tiled, int32_t sample_set,
                                               meao_render_constants *out);
/* PushUpsampleCommands constant math (AO.cs:750-771); low_level 1..4 is the mip of LoResDB. */
MEAO_API int32_t meao_upsample_constants_for(int32_t width, int32_t height, const meao_params *p,
                                             int32_t low_level, meao_upsample_constants *out);
/* Buffer table (AO.cs:453-475): dims/format/bytes of debug buffer 1..17 (and 18..21, OcclusionHQ1..4). */
MEAO_API int32_t meao_describe_buffer(const meao_config *cfg, int32_t debug_id, meao_desc *out);
/* Compulsory traffic of each pass in the reference's storage formats (SURVEY.md 8d,
 * BASELINE.md 3): the numerator of the roofline fraction.  bytes[MEAO_NUM_PASSES]. */
MEAO_API int32_t meao_algorithmic_bytes(const meao_config *cfg, uint64_t bytes[MEAO_NUM_PASSES]);

/* ---- context (replaces DoLazyInitialization / OnDestroy, AO.cs:440-494,357-381) --------- */
MEAO_API int32_t meao_create(const meao_config *cfg, meao_ctx **out_ctx);
MEAO_API int32_t meao_destroy(meao_ctx *ctx);
/* Screen-size change (AO.cs:338-341,501): re-plans and re-allocates the intermediates. */
MEAO_API int32_t meao_resize(meao_ctx *ctx, int32_t width, int32_t height);
/* Property change (AO.cs:104-113): recomputes the per-level constant blocks. */
MEAO_API int32_t meao_set_params(meao_ctx *ctx, const meao_params *p);
MEAO_API int32_t meao_get_params(const meao_ctx *ctx, meao_params *out);
MEAO_API int32_t meao_get_config(const meao_ctx *ctx, meao_config *out);
/* Last error text of this context ("" if none).  Never NULL; ctx may be NULL. */
MEAO_API const char *meao_last_error(const meao_ctx *ctx);

/* ---- the hot path (replaces the recorded "SSAO" CommandBuffer, AO.cs:496-531) ----------- */
/* depth: width*height raw device depth texels in cfg.depth_format (default float32), row-major,
 *        tightly packed
 *        (_CameraDepthTexture / ResolvedDepth, AO.cs:608-641).
 * ao_out: width*height AO texels in cfg.ao_format (the "AmbientOcclusion" RT, AO.cs:475).
 * Alignment of DEVICE pointers: none required.  When width % 4 == 0 and every depth pointer is
 *        aligned to 4 texels (16 bytes for F32 / UNORM24, 8 for the 16-bit formats) the last pass
 *        reads the depth frame with 4-texel vector loads (the downsample pass: width % 8 == 0), and when
 *        every ao_out pointer is also aligned to 4 texels (4 bytes R8, 8 bytes F16) it uses 4-texel stores;
 *        otherwise the scalar variants run (same results, slower).  hipMalloc / torch allocations are
 *        always aligned.
 * The depth frames are read by the FIRST and the LAST pass of the call (the downsample pass builds the four
 *        levels from their even rows; the full-resolution upsample linearizes every texel itself instead of
 *        reading it back from a LinearDepth buffer): they must stay unchanged until the call has completed.
 * Failure: meao_resize and the first meao_prefetch_batch allocate; when that fails the context is left
 *        exactly as it was (geometry, buffers, a ready prefetch).
 * Any depth value is accepted: NaN, +-inf, negative, > 1 or denormal raw depths are processed with
 *        IEEE division exactly like the reference's Linearize (Downsample1.compute:37-48) -- a frame
 *        containing such texels takes slower kernel bodies, results stay bit-exact vs the oracle.
 * *_loc: meao_mem; HOST pointers are staged through context-owned device buffers.
 * Asynchronous on `stream` for DEVICE/DEVICE; returns after completion if either is HOST. */
MEAO_API int32_t meao_execute(meao_ctx *ctx, const void *depth, int32_t depth_loc,
                              void *ao_out, int32_t out_loc, meao_stream stream);
/* n independent frames (n <= cfg.max_batch) through one launch per pass. */
MEAO_API int32_t meao_execute_batch(meao_ctx *ctx, int32_t n, const void *const *depth,
                                    int32_t depth_loc, void *const *ao_out, int32_t out_loc,
                                    meao_stream stream);
/* Pipelining for streams of frames.  Announces the DEVICE depth frames of the call after next: the
 * following meao_execute* carries their downsample pass inside its last upsample kernel (f32 depth,
 * width % 8 == 0, aligned frames; otherwise as one more launch behind it), where the pass's HBM traffic
 * overlaps arithmetic instead of costing a launch between two VALU-bound ones;
 * the meao_execute* after that, if given exactly these n pointers, skips its own downsample pass.
 * Results are identical.  Rules: the announced buffers must hold their final contents before the
 * carrying execute is submitted and stay unchanged until the consuming one has run.  Enforced, not
 * only documented: the prefetched downsample is used only by an execute on the SAME stream as the
 * one that carried it (stream order is what orders the two) and given exactly the announced
 * pointers; any other execute, meao_set_params and meao_resize simply run / re-run the pass.
 * With cfg.pipelined = 1 this call never allocates or synchronises; otherwise the first call
 * re-allocates the context's intermediates with a second set of downsample buffers (one device
 * synchronisation; on allocation failure the context is left unchanged and usable). */
MEAO_API int32_t meao_prefetch_batch(meao_ctx *ctx, int32_t n, const void *const *depth);
/* Waits for `stream`; NULL = the stream of the last meao_execute* of this context.  meao_composite /
 * meao_composite_flush do not change what NULL means: a composite issued on another stream is waited for
 * by naming that stream here (or by synchronising it directly). */
MEAO_API int32_t meao_synchronize(meao_ctx *ctx, meao_stream stream);

/* ---- observability (replaces the _debug 1..17 views, AO.cs:787-820) --------------------- */
/* Copies debug buffer `debug_id` of batch slot `frame` (as left by the last execute) to dst
 * in the reference's layout (TiledDepth: [16][h][w]).  dst may be NULL to query desc only.
 * Ids 18..21 (OcclusionHQ<k>) exist only for the levels cfg.hq_levels enables.
 * Ids 1 (LinearDepth) and 6..9 (TiledDepth<k>) are not buffers of the hot path: they are built here, on demand, bit-identical
 * to what Downsample1 / Downsample2 would have stored -- id 1 from the DEPTH FRAME the last execute was given, which must
 * therefore still be alive and unchanged (HOST frames are staged by the context and always are). */
MEAO_API int32_t meao_get_intermediate(meao_ctx *ctx, int32_t frame, int32_t debug_id,
                                       void *dst, uint64_t dst_capacity, int32_t dst_loc,
                                       meao_desc *out_desc);

/* The picture the reference's _debug = 1..17 mode shows (PushDebugBlitCommands AO.cs:787-820): the
 * chosen buffer blitted into the full-resolution AO target -- point-sampled at the destination
 * texel centres for the 2D buffers (cmd.Blit(rt, _result)), the 4x4 grid of slices for the tiled
 * arrays (Blit.shader:136-155 pass 4), the result itself for 17.  Sampling positions are evaluated
 * in exact integer arithmetic: source texel = floor((2x+1) * ws / (2W)); the store converts to
 * cfg.ao_format like every AO store.  out: width*height AO texels. */
MEAO_API int32_t meao_debug_view(meao_ctx *ctx, int32_t frame, int32_t debug_id, void *out,
                                 int32_t out_loc, meao_stream stream);

/* ---- composite: the consumer of the AO texture (PushCompositeCommands AO.cs:822-839) ----
 * The reference composites with raster blits (Blit.shader passes 1-3) into the HDR camera target
 * (ARGBHalf) and, in deferred ambient-only mode, GBuffer0 (ARGB32).  Canonical reading of the
 * fixed-function blend: operands widened to f32, one f32 multiply, result rounded to the target
 * format (f16: round-to-nearest-even, the output-merger rule; UNORM8: as the AO stores).
 *   MULTIPLY      pass 2, "Blend Zero SrcAlpha":            color.rgba *= ao
 *   AMBIENT_ONLY  pass 1, "Blend Zero OneMinusSrcColor, Zero OneMinusSrcAlpha", occ = 1 - ao:
 *                 color.rgb *= (1 - occ), color.a unchanged; gbuffer0.a *= (1 - occ), rgb unchanged
 *   DEBUG         pass 3, no blend:                         color.rgba = ao
 * ao: width*height texels in cfg.ao_format; color: width*height RGBA16F (8 bytes per texel),
 * updated in place; gbuffer0: width*height RGBA8, only for AMBIENT_ONLY (else NULL).
 * All pointers share one location `loc`. */
typedef enum meao_composite_mode {
    MEAO_COMPOSITE_MULTIPLY = 0, MEAO_COMPOSITE_AMBIENT_ONLY = 1, MEAO_COMPOSITE_DEBUG = 2
} meao_composite_mode;
MEAO_API int32_t meao_composite(meao_ctx *ctx, int32_t mode, const void *ao, void *color_rgba16f,
                                void *gbuffer0_rgba8, int32_t loc, meao_stream stream);

/* Pipelined composite ("depth in -> shaded frame out" for streams of frames).  The composite moves 17
 * bytes per texel -- as many bytes as the whole AO path -- while the render pass is VALU-bound with
 * HBM nearly idle.  meao_composite_enqueue registers the composite of n DEVICE frames (ao[f] produced
 * by an earlier meao_execute*; same formats and modes as meao_composite); the NEXT meao_execute* on
 * this context carries it inside its render kernel (every render workgroup first streams its share of
 * the texel pairs), on that call's stream, i.e. ordered behind the kernels that wrote ao[f] when the
 * same stream is used.  Results are identical to meao_composite.  One batch can wait at a time: a
 * second enqueue, meao_resize and meao_composite_flush run the waiting batch as plain composite
 * launches, on the stream of the execute that preceded its enqueue -- i.e. the one that produced
 * ao[f] -- or on the stream given to meao_composite_flush.  meao_destroy DISCARDS a batch that still
 * waits (its targets are caller memory that is usually gone by then): call meao_composite_flush
 * before destroying a context if the last enqueued composite matters.
 * ao[f], color[f] and gbuffer0[f] must stay valid and untouched until the carrying call has run. */
MEAO_API int32_t meao_composite_enqueue(meao_ctx *ctx, int32_t mode, int32_t n, const void *const *ao,
                                        void *const *color_rgba16f, void *const *gbuffer0_rgba8);
MEAO_API int32_t meao_composite_flush(meao_ctx *ctx, meao_stream stream);
/* *out_frames = frames of an enqueued composite batch that no execute / flush / resize / second enqueue has run yet
 * (0 = nothing waits).  Hosts ask the library instead of mirroring this state (a batch rides only in the members / calls
 * that actually execute; an execute that fails leaves it waiting). */
MEAO_API int32_t meao_composite_pending(const meao_ctx *ctx, int32_t *out_frames);

/* Per-pass device timing: when enabled, meao_execute* brackets every pass with HIP events on
 * the launch stream; meao_get_pass_times averages each pass over the executes that ran it since
 * the last reset (it waits for the stream of the last execute first); *out_samples = executes measured.
 * ms[MEAO_NUM_PASSES]; passes not run report 0.
 * enable: 0 = off; 1 = every execute; N > 1 = every Nth execute (the first one after the call included), the others run
 * without event records.  An event record is a marker packet between two launches; the events are created with
 * hipEventDisableSystemFence (nothing synchronizes with them but hipEventElapsedTime), so a record does not write back and
 * invalidate the caches: eight of them per 4K x 16 step cost 1 - 3 % of the step depending on the box (bench.py
 * `without_pass_events`; with fenced events it was up to 4 %, and the short launches read 2 us longer).  meao_debug_set
 * MEAO_DEBUG_PROFILE_PASS_MASK restricts the records to chosen passes. */
MEAO_API int32_t meao_set_profiling(meao_ctx *ctx, int32_t enable);
MEAO_API int32_t meao_get_pass_times(meao_ctx *ctx, float ms[MEAO_NUM_PASSES], int32_t *out_samples);

/* ---- multi-GPU: a pool of contexts, one per device, called from one host thread ------------------
 * The path shards across independent frames only (the reference keeps no temporal state,
 * AO.cs:291-308): frame f of a batch goes to member f mod G, every member owns a complete context
 * and a stream on its device, there is no data-path exchange (SURVEY.md 8e / DESIGN.md 7).  This is
 * what an in-process host (the C# component) binds; bench.py's one-process-per-GPU launch over
 * torch.distributed / RCCL is the other way to the same partition.
 * devices: num_devices HIP ordinals (repeats allowed: several members on one device), or NULL for
 * 0..num_devices-1 modulo the visible device count.  cfg.device is ignored; cfg.max_batch is per member.
 * A pool, like a context, is used by one caller thread at a time.  Inside, DEVICE batches are enqueued by one worker
 * thread per member (started by the first such batch, joined by meao_pool_destroy): a member's 4-5 launches cost
 * 10-14 us of host time, and eight members fed one after the other cannot keep up with one 4K frame per GPU
 * (tools/pool_enqueue_cost.py).  A member's context is only ever touched by one thread at a time. */
typedef struct meao_pool meao_pool;
MEAO_API int32_t meao_pool_create(const meao_config *cfg, const int32_t *devices, int32_t num_devices,
                                  meao_pool **out_pool);
MEAO_API int32_t meao_pool_destroy(meao_pool *pool);
MEAO_API int32_t meao_pool_size(const meao_pool *pool);
/* Member context (owned by the pool), e.g. for meao_get_intermediate / meao_set_profiling; NULL if out of range. */
MEAO_API meao_ctx *meao_pool_context(meao_pool *pool, int32_t member);
/* HIP ordinal that processes frame `frame` of a batch (= where DEVICE pointers of that frame must live); -1 on error. */
MEAO_API int32_t meao_pool_device_of_frame(const meao_pool *pool, int32_t frame);
MEAO_API const char *meao_pool_last_error(const meao_pool *pool);
/* meao_set_params on every member. */
MEAO_API int32_t meao_pool_set_params(meao_pool *pool, const meao_params *p);
/* Every meao_pool_* call leaves the calling thread's current HIP device as it found it.  If a member
 * fails, the other members have still been given their work (their launches are in flight and complete
 * normally); the call returns the first failing member's status and meao_pool_last_error names it. */
/* n <= max_batch * members frames; frame f runs on member f mod G, each member's share as one batched
 * launch sequence on its own stream.  DEVICE pointers of frame f must be resident on
 * meao_pool_device_of_frame(f); the call is then asynchronous (meao_pool_synchronize).  HOST pointers
 * are staged per member and the call returns after completion. */
MEAO_API int32_t meao_pool_execute_batch(meao_pool *pool, int32_t n, const void *const *depth, int32_t depth_loc,
                                         void *const *ao_out, int32_t out_loc);
/* meao_prefetch_batch for the pool: announces the n DEVICE depth frames of the call after next, dealt to
 * the members exactly like meao_pool_execute_batch deals them (frame f -> member f mod G), so that each
 * member's next execute carries its share of the next batch's downsample pass.  Create the pool with
 * cfg.pipelined = 1 to keep this call free of allocation. */
MEAO_API int32_t meao_pool_prefetch_batch(meao_pool *pool, int32_t n, const void *const *depth);
/* meao_composite_enqueue / meao_composite_flush for the pool, frames dealt f -> member f mod G: the composite
 * of frame f rides inside the next execute of the member that owns (and produced) it. */
MEAO_API int32_t meao_pool_composite_enqueue(meao_pool *pool, int32_t mode, int32_t n, const void *const *ao,
                                             void *const *color_rgba16f, void *const *gbuffer0_rgba8);
MEAO_API int32_t meao_pool_composite_flush(meao_pool *pool);
MEAO_API int32_t meao_pool_composite_pending(const meao_pool *pool, int32_t *out_frames);   /* summed over the members */
/* Copies the n DEVICE results ao_src[f] (on their owning devices) to dst[f] on dst_device with
 * hipMemcpyPeerAsync (xGMI), each on its producer's stream, i.e. ordered behind the kernels that wrote it. */
MEAO_API int32_t meao_pool_gather_to_device(meao_pool *pool, int32_t n, const void *const *ao_src,
                                            void *const *dst, int32_t dst_device);
/* How a copy from `member`'s device to dst_device travels.  meao_pool_create asks hipDeviceCanAccessPeer for
 * every ordered pair of distinct member devices and enables peer access both ways where it is offered;
 * PEER_DIRECT = enabled (device-to-device over xGMI), STAGED = not offered or not enabled (the runtime
 * bounces the copy through host memory), SAME_DEVICE = no link involved.  Negative = invalid argument. */
typedef enum meao_pool_path { MEAO_POOL_PATH_SAME_DEVICE = 0, MEAO_POOL_PATH_PEER_DIRECT = 1, MEAO_POOL_PATH_STAGED = 2 } meao_pool_path;
MEAO_API int32_t meao_pool_gather_path(const meao_pool *pool, int32_t member, int32_t dst_device);
MEAO_API int32_t meao_pool_synchronize(meao_pool *pool);
/* Host placement.  The eight GPUs of a node hang off two sockets; the thread that enqueues a GPU's launches should run on the
 * socket the GPU hangs off.  meao_device_numa_node: *out_node = NUMA node of HIP device `device` (sysfs numa_node of its PCI
 * function; -1 = the kernel knows none: single-node hosts, most VMs), cpulist (may be NULL) = that node's CPUs in the kernel's
 * "0-15,32-47" form -- what a one-process-per-GPU host (bench.py --gpus N) binds its rank to.  The pool does the same for its
 * worker threads: each binds itself to the allowed CPUs of its member's node when it starts (MEAO_POOL_BIND_NUMA, default 1; the
 * calling thread's cgroup / taskset mask is respected; nothing happens where no node is known).  MEAO_POOL_SPIN_US: how long a
 * worker spins for its next job before it sleeps on a condition variable (default 100; 0 = sleep at once; a stream of 4K steps
 * posts a job every ~60 us per member).  meao_pool_member_placement reports the member's node and whether its worker is bound. */
typedef enum meao_pool_option { MEAO_POOL_SPIN_US = 0, MEAO_POOL_BIND_NUMA = 1 } meao_pool_option;
MEAO_API int32_t meao_device_numa_node(int32_t device, int32_t *out_node, char *cpulist, uint64_t cpulist_capacity);
MEAO_API int32_t meao_pool_configure(meao_pool *pool, int32_t key, int32_t value);
MEAO_API int32_t meao_pool_member_placement(const meao_pool *pool, int32_t member, int32_t *out_numa_node, int32_t *out_worker_bound);

/* Which frames of the last meao_execute* held "hostile" depth texels (NaN, inf, negative, > 1, denormal --
 * anything outside the operand range the exact v_rcp_f32 division sequences are verified for) and therefore
 * ran the slower IEEE-division kernel bodies: bit f of *out_mask = frame f.  Results are bit-exact either
 * way; this is a performance diagnostic.  Synchronises the stream of that call. */
MEAO_API int32_t meao_hostile_frames(meao_ctx *ctx, uint64_t *out_mask);

/* Launch-structure overrides for tests and A/B runs (the library reads no environment variables).  Every
 * structure gives bit-identical results; the defaults are what measured fastest.  FUSE_COARSE_BLEND 0 = three
 * separate blend launches; *_MAX_TILES = tile-count thresholds at or below which a call uses the nested
 * three-level blend launch / 128x8 render tiles / 64x32 final tiles / one-row-per-lane downsample tiles (0 = never).
 * (ABI 6 retired the keys of launch structures that lost every A/B: DS_SHARE_IN_BLEND, DS_SIDE_STREAM, RENDER_FROM_DEPTH*;
 * their patches are kept under profiles/.  Fault injection lives in the `testhooks` variant library only.) */
typedef enum meao_debug_key {
    MEAO_DEBUG_FUSE_COARSE_BLEND = 0, MEAO_DEBUG_NESTED_MAX_TILES = 1, MEAO_DEBUG_RENDER_SMALL_MAX_TILES = 2,
    MEAO_DEBUG_FINAL_SMALL_MAX_TILES = 3, MEAO_DEBUG_DS_SMALL_MAX_TILES = 4,
    MEAO_DEBUG_BLEND_TALL_MIN_TILES = 5, /* L2 -> L1 blend launches of at least this many 64x32 tiles (frames x tiles) use 64x64 tiles, either AO
                                          * storage format; 0 = never.  Default: 4096, R8 storage only */
    MEAO_DEBUG_NEXT_DOWNSAMPLE_OWN_LAUNCH = 7, /* 1 = the announced next batch's downsample pass (meao_prefetch_batch) runs as a launch of its own
                                               * behind the last upsample kernel instead of inside it (what frames that do not take the
                                               * carried tile's 16-byte loads get anyway); default 0 */
    MEAO_DEBUG_PROFILE_PASS_MASK = 6     /* meao_set_profiling: bit k set = launch slot k (meao_pass) is bracketed with events; 0 = all (default).
                                          * Each event record is a marker packet between two launches (1 - 4 % of a batched step for all
                                          * eight): a host that wants one kernel's duration in a throughput run asks for that slot only */
} meao_debug_key;
MEAO_API int32_t meao_debug_set(meao_ctx *ctx, int32_t key, int32_t value);

/* roctx ranges ("meao:downsample", "meao:render", "meao:upsample_L1_to_L0", ...) around the launches
 * of every pass, so that rocprofv3 --marker-trace output is self-describing even where passes are
 * fused.  Off by default; librocprofiler-sdk-roctx.so (rocprofv3) or libroctx64.so is loaded on first
 * use (MEAO_ERR_UNSUPPORTED if neither is present). */
MEAO_API int32_t meao_set_tracing(meao_ctx *ctx, int32_t enable);

/* Exhaustive device self-tests of what bit-exactness rests on; *out_mismatches = number of
 * inputs whose hardware result differs from the IEEE / bit-level model.
 * which: 0 = f32->f16 RTZ_CLAMP (all 2^32), 1 = f32->f16 RTNE (all 2^32), 2 = unorm8->f32 (256),
 *        3 = f16->f32 (65536), 4 = exact reciprocal vs 1/x (all x, 2^-100<=|x|<=2^100),
 *        (and: the uncorrected v_rcp_f32 within one ulp of it), 5 = exact 3/x and 9/x (same range), 6 = exact a/b on hashed
 *        pairs (2^-60<=|a|,|b|<=2^60), 7 = the UNORM8 bilateral result that skips the correction steps away from rounding
 *        boundaries vs the code of the exact chain, 2^32 hashed operand sets. */
MEAO_API int32_t meao_selftest(meao_ctx *ctx, int32_t which, uint64_t *out_mismatches);

#ifdef __cplusplus
}
#endif
#endif /* MEAO_H */
