/*
 * meao_oracle.h -- CPU oracle for the multi-scale SSAO hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported baseline.  The
 * product path (libmeao_hip.so) never links, loads or calls it.
 *
 * PARITY PIN: the reference (keijiro/MiniEngineAO) ships no golden vectors
 * and no tests, and Unity / a D3D GPU / fxc are not available, so parity
 * against outputs of the reference *running on its own platform* is unpinned.
 * What pins the oracle instead:
 *  (1) the reference's own source text is executed: AmbientOcclusion.cs by
 *      oracle/csharp_interp.py (against recording Unity mocks) yields the
 *      buffer table, the ten dispatches, their bindings, constants and grids;
 *      the four .compute files run through oracle/hlsl_interp.py.  All 17
 *      buffers are committed as fixtures (tests/golden/ref_*.npz, generator
 *      committed) and both restatements must reproduce them bit for bit.  The
 *      interpreters supply only what the reference's platform would: the
 *      numerics contract, the resource / format semantics and Unity's API
 *      behaviour -- those remain this project's canonical reading;
 *  (2) two independently structured restatements -- this file's per-pixel
 *      gather form and meao_hlsl_emul.c's literal thread-group/LDS emulation --
 *      agree bit-for-bit on all 17 intermediates at many odd sizes and modes;
 *  (3) analytical known-answer tests derived from the reference source
 *      (tests/test_oracle_kat.py).
 *
 * Canonical numerics (DESIGN.md "Numerics contract"): IEEE-754 binary32,
 * round-to-nearest-even, correctly rounded '/' and sqrt; an HLSL expression
 * of the shape a*b+c / a*b-c / c-a*b is a single `mad` and is evaluated
 * fused (fmaf); everything else is separately rounded (-ffp-contract=off).
 */
#ifndef MEAO_ORACLE_H
#define MEAO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MEAO_ORACLE_AO_R8 = 0, MEAO_ORACLE_AO_F16 = 1 };
enum { MEAO_ORACLE_F16_RTZ = 0, MEAO_ORACLE_F16_RTNE = 1 };

/* Everything the reference component + camera feed into the path
 * (AmbientOcclusion.cs:20-68 properties; :561-573 camera terms). */
typedef struct meao_oracle_desc {
    int32_t width, height;        /* camera.pixelWidth/Height  (AO.cs:339-340) */
    int32_t num_levels;           /* 1..4; the reference always runs 4        */
    int32_t ao_format;            /* R8 = reference (AO.cs:466-475) or F16    */
    int32_t f16_rounding;         /* store conversion f32->f16                */
    int32_t reversed_z;           /* SystemInfo.usesReversedZBuffer           */
    float noise_filter_tolerance; /* AO.cs:20  default  0    */
    float blur_tolerance;         /* AO.cs:28  default -4.6  */
    float upsample_tolerance;     /* AO.cs:36  default -12   */
    float thickness_modifier;     /* AO.cs:44  default  1    */
    float intensity;              /* AO.cs:52  default  1    */
    float near_clip, far_clip;    /* AO.cs:563 */
    float proj00;                 /* camera.projectionMatrix[0,0] (AO.cs:572) */
    int32_t depth_format;         /* storage of the input depth, see below     */
    /* ---- variants the reference carries but never (or only in VR) dispatches ---- */
    int32_t single_pass_stereo;   /* AO.cs:392-401,680: double-wide frame, ThicknessMultiplier x2 */
    int32_t hq_levels;            /* 0..num_levels: the coarsest hq_levels levels additionally run
                                   * Render.main (wide, non-interleaved, REN:22,27-29,46-50) on
                                   * LowDepth<k> and min-combine it in Upsample.main_premin*
                                   * (UPS:23,25,58-60); 0 = the reference's wiring */
    int32_t sample_set;           /* 0 = 36-sample checker (REN:160-169), 1 = SAMPLE_EXHAUSTIVELY
                                   * (68 samples, all 12 weights, REN:144-159) */
} meao_oracle_desc;
enum { MEAO_ORACLE_SAMPLES_CHECKER = 0, MEAO_ORACLE_SAMPLES_EXHAUSTIVE = 1 };

/* Input depth storage.  The reference blits _CameraDepthTexture into an RFloat copy first
 * (Blit.shader:48-64 pass 0, AO.cs:608-614): UNORM texels sample as v / (2^n - 1), correctly
 * rounded; UNORM24 is the low 24 bits of a 32-bit word (D24S8/D24X8). */
enum { MEAO_ORACLE_DEPTH_F32 = 0, MEAO_ORACLE_DEPTH_UNORM16 = 1, MEAO_ORACLE_DEPTH_UNORM24 = 2,
       MEAO_ORACLE_DEPTH_F16 = 3 };
float meao_oracle_decode_depth(const void *depth, uint64_t index, int32_t depth_format);

/* The 17 debug-visible buffers (AO.cs:789-808) + nothing else.  Any pointer
 * may be NULL: the oracle then uses a private scratch buffer for it.
 * AO buffers are uint8 (R8) or uint16 f16 bit patterns (F16). */
typedef struct meao_oracle_buffers {
    uint16_t *linear_depth;    /* id 1      f16 bits, L0                 */
    float    *low_depth[4];    /* id 2..5   f32, L1..L4                  */
    uint16_t *tiled_depth[4];  /* id 6..9   f16 bits, [16][h][w] L3..L6  */
    void     *occlusion[4];    /* id 10..13 AO, L1..L4                   */
    void     *combined[3];     /* id 14..16 AO, L1..L3                   */
    void     *result;          /* id 17     AO, L0                       */
    void     *occlusion_hq[4]; /* id 18..21 AO, L1..L4: Render.main on LowDepth<k> (hq_levels) */
} meao_oracle_buffers;

/* Constant blocks, exactly what AO.cs uploads per dispatch. */
typedef struct meao_oracle_render_consts {
    float inv_thickness[12];   /* gInvThicknessTable  AO.cs:687-688 */
    float sample_weight[12];   /* gSampleWeightTable  AO.cs:696-724 */
    float inv_slice_dim[2];    /* gInvSliceDimension  AO.cs:732     */
    float reject_fadeoff;      /* AO.cs:733 */
    float intensity;           /* AO.cs:734 */
} meao_oracle_render_consts;

typedef struct meao_oracle_upsample_consts {
    float inv_low_res[2], inv_high_res[2];   /* AO.cs:766-767 */
    float noise_filter_strength;             /* AO.cs:764,768 */
    float step_size;                         /* AO.cs:760,769 */
    float blur_tolerance;                    /* AO.cs:761-762,770 */
    float upsample_tolerance;                /* AO.cs:763,771 */
} meao_oracle_upsample_consts;

/* level k dims = ceil(W / 2^k)  (AO.cs:276-281) */
void meao_oracle_level_dims(int32_t width, int32_t height, int32_t level,
                            int32_t *w, int32_t *h);
void meao_oracle_zbuffer_params(const meao_oracle_desc *d, float zp[4]);
void meao_oracle_sample_thickness(float out[12]);
/* level = 1..4: source atlas is TiledDepth<level> (dims of mip level+2). */
void meao_oracle_render_constants(const meao_oracle_desc *d, int32_t level,
                                  meao_oracle_render_consts *out);
/* level = 1..4: source is the non-tiled LowDepth<level> (PushRenderCommands with
 * !source.isTiled, AO.cs:679): ThicknessMultiplier x2, inv_slice_dim = 1 / dims(level). */
void meao_oracle_render_constants_hq(const meao_oracle_desc *d, int32_t level,
                                     meao_oracle_render_consts *out);
/* low_level = mip level of the low-res input (1..4), high = low_level-1. */
void meao_oracle_upsample_constants(const meao_oracle_desc *d, int32_t low_level,
                                    meao_oracle_upsample_consts *out);

/* Storage conversions (exposed so tests can pin them). */
uint16_t meao_oracle_f32_to_f16(float x, int32_t rounding);
float    meao_oracle_f16_to_f32(uint16_t h);
uint8_t  meao_oracle_f32_to_unorm8(float x);
float    meao_oracle_unorm8_to_f32(uint8_t v);

/* Full pipeline, gather form.  nthreads <= 1 -> scalar single core.
 * Returns 0 on success, negative on bad arguments / allocation failure. */
int32_t meao_oracle_run(const meao_oracle_desc *d, const void *depth,
                        meao_oracle_buffers *out, int32_t nthreads);

/* Composite (Blit.shader passes 1-3, AO.cs:822-839): canonical reading of the fixed-function
 * blend = operands widened to f32, one multiply, result rounded to the target format (f16 RTNE,
 * UNORM8 as the AO stores).  mode 0 multiply, 1 ambient-only (gbuffer0 RGBA8 required), 2 debug.
 * color: RGBA16F bit patterns, in place. */
int32_t meao_oracle_composite(int32_t width, int32_t height, int32_t ao_format, int32_t mode,
                              const void *ao, uint16_t *color_rgba16f, uint8_t *gbuffer0_rgba8);

/* Same contract, literal HLSL thread-group emulation (meao_hlsl_emul.c). */
int32_t meao_hlsl_emul_run(const meao_oracle_desc *d, const void *depth,
                           meao_oracle_buffers *out);

#ifdef __cplusplus
}
#endif
#endif
