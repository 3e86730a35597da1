// meao_plan.cpp -- see meao_plan.hpp.  All arithmetic is binary32 with one rounding per
// operation (this file is compiled with -ffp-contract=off), matching the float-typed C# the
// reference evaluates; Mathf.Sqrt / Mathf.Pow go through double like Unity's do.
#include "meao_plan.hpp"

#include <cmath>
#include <cstring>

namespace meao {

int render_term_slots(int sample_set, const int **slots)
{
    static const int checker[7] = {1, 3, 4, 8, 11, 6, 10};
    static const int exhaustive[12] = {0, 1, 2, 3, 4, 8, 11, 5, 6, 7, 9, 10};
    const bool all = sample_set == MEAO_SAMPLES_EXHAUSTIVE;
    *slots = all ? exhaustive : checker;
    return all ? 12 : 7;
}

Dims level_dims(int width, int height, int level)
{
    const int div = 1 << level;  // AO.cs:278
    Dims d;
    d.w = (width + (div - 1)) / div;
    d.h = (height + (div - 1)) / div;
    return d;
}

float render_term_scale(int sample_set, int term)
{
    // (x, y) of the terms in accumulation order; y == 0 axial, x == y diagonal, else L-shaped
    static const int checker[7][2] = {{2, 0}, {4, 0}, {1, 1}, {2, 2}, {3, 3}, {1, 3}, {2, 4}};
    static const int exhaustive[12][2] = {{1, 0}, {2, 0}, {3, 0}, {4, 0}, {1, 1}, {2, 2},
                                          {3, 3}, {1, 2}, {1, 3}, {1, 4}, {2, 3}, {2, 4}};
    const int *xy = sample_set == MEAO_SAMPLES_EXHAUSTIVE ? exhaustive[term] : checker[term];
    return (xy[1] == 0 || xy[0] == xy[1]) ? 0.5f : 0.25f;
}

void zbuffer_params(const meao_params &p, float out[4])
{
    const float far_over_near = p.far_clip / p.near_clip;  // AO.cs:563
    if (p.reversed_z) {
        out[0] = far_over_near - 1.0f;  // AO.cs:565
        out[1] = 1.0f;
    } else {
        out[0] = 1.0f - far_over_near;  // AO.cs:567
        out[1] = far_over_near;
    }
    out[2] = 0.0f;
    out[3] = 0.0f;
}

static float mathf_sqrt(float v) { return static_cast<float>(std::sqrt(static_cast<double>(v))); }
static float mathf_pow10(float e) { return static_cast<float>(std::pow(10.0, static_cast<double>(e))); }

void sample_thickness(float out[12])
{
    // AO.cs:577-590.  (u,v) in fifths of the sphere radius; thickness = sqrt(1 - u^2 - v^2)
    // with the squares and the subtractions done in float, left to right.
    static const int uv[12][2] = {{1, 0}, {2, 0}, {3, 0}, {4, 0}, {1, 1}, {1, 2},
                                  {1, 3}, {1, 4}, {2, 2}, {2, 3}, {2, 4}, {3, 3}};
    static const float fifth[5] = {0.0f, 0.2f, 0.4f, 0.6f, 0.8f};
    for (int i = 0; i < 12; ++i) {
        const float fu = fifth[uv[i][0]], fv = fifth[uv[i][1]];
        float r = 1.0f - fu * fu;
        if (uv[i][1] != 0) r = r - fv * fv;
        out[i] = mathf_sqrt(r);
    }
}

void render_constants(int width, int height, const meao_params &p, int level, bool source_tiled,
                      int sample_set, meao_render_constants *out)
{
    // source = TiledDepth<level> (dims of mip level+2, AO.cs:461-464) or the non-tiled LowDepth<level>
    const Dims slice = level_dims(width, height, source_tiled ? level + 2 : level);
    float thickness[12];
    sample_thickness(thickness);

    const float tan_half_fov_h = 1.0f / p.proj00;  // AO.cs:572
    // AO.cs:678: 2 * TanHalfFovH * ScreenspaceDiameter / source.width
    float multiplier = 2.0f * tan_half_fov_h;
    multiplier = multiplier * 10.0f;
    multiplier = multiplier / static_cast<float>(slice.w);
    if (!source_tiled) multiplier = multiplier * 2.0f;            // AO.cs:679
    if (p.single_pass_stereo) multiplier = multiplier * 2.0f;     // AO.cs:680
    const float inverse_range_factor = 1.0f / multiplier;  // AO.cs:683
    for (int i = 0; i < 12; ++i)
        out->inv_thickness_table[i] = inverse_range_factor / thickness[i];  // AO.cs:688

    // AO.cs:696-707 sample multiplicities; AO.cs:711-715 zero the exhaustive-only slots.
    static const float checker[12] = {0, 4, 0, 4, 4, 0, 8, 0, 4, 0, 8, 4};
    static const float exhaustive[12] = {4, 4, 4, 4, 4, 8, 8, 8, 4, 8, 8, 4};
    const float *multiplicity = sample_set == MEAO_SAMPLES_EXHAUSTIVE ? exhaustive : checker;
    float total = 0.0f;
    for (int i = 0; i < 12; ++i) {
        out->sample_weight_table[i] = multiplicity[i] == 0.0f ? 0.0f : multiplicity[i] * thickness[i];
        total += out->sample_weight_table[i];  // AO.cs:718-721, sequential float sum
    }
    for (int i = 0; i < 12; ++i) out->sample_weight_table[i] /= total;  // AO.cs:723-724

    out->inv_slice_dimension[0] = 1.0f / static_cast<float>(slice.w);  // AO.cs:171,732
    out->inv_slice_dimension[1] = 1.0f / static_cast<float>(slice.h);
    out->reject_fadeoff = -1.0f / p.thickness_modifier;  // AO.cs:733
    out->intensity = p.intensity;                        // AO.cs:734
}

void upsample_constants(int width, int height, const meao_params &p, int low_level,
                        meao_upsample_constants *out)
{
    const Dims lo = level_dims(width, height, low_level);
    const Dims hi = level_dims(width, height, low_level - 1);
    const float step_size = 1920.0f / static_cast<float>(lo.w);  // AO.cs:760 (1920 is hard-coded)
    float blur = mathf_pow10(p.blur_tolerance) * step_size;      // AO.cs:761
    blur = 1.0f - blur;
    blur = blur * blur;                                          // AO.cs:762
    const float upsample = mathf_pow10(p.upsample_tolerance);    // AO.cs:763
    float noise = mathf_pow10(p.noise_filter_tolerance) + upsample;
    noise = 1.0f / noise;                                        // AO.cs:764
    out->inv_low_resolution[0] = 1.0f / static_cast<float>(lo.w);
    out->inv_low_resolution[1] = 1.0f / static_cast<float>(lo.h);
    out->inv_high_resolution[0] = 1.0f / static_cast<float>(hi.w);
    out->inv_high_resolution[1] = 1.0f / static_cast<float>(hi.h);
    out->noise_filter_strength = noise;
    out->step_size = step_size;
    out->blur_tolerance = blur;
    out->upsample_tolerance = upsample;
}

float linearize_out_of_range(const float zp[4], bool reversed_z)
{
    // Depth[st] out of range loads 0 (Downsample1.compute:39); reversed Z then takes the
    // "depth == 0 -> 1e5" branch (:42), conventional Z yields 1 / ZBufferParams.y.
    const float depth = 0.0f;
    float dist = 1.0f / std::fmaf(zp[0], depth, zp[1]);
    if (reversed_z) dist = 1e5f;
    return dist;
}

bool params_valid(const meao_params &p)
{
    auto finite = [](float v) { return std::isfinite(v); };
    if (!finite(p.noise_filter_tolerance) || !finite(p.blur_tolerance) ||
        !finite(p.upsample_tolerance) || !finite(p.thickness_modifier) || !finite(p.intensity) ||
        !finite(p.near_clip) || !finite(p.far_clip) || !finite(p.proj00))
        return false;
    if (!(p.near_clip > 0.0f) || !(p.far_clip > p.near_clip)) return false;
    if (p.proj00 == 0.0f || p.thickness_modifier == 0.0f) return false;
    return true;
}

static void fill_terms(RenderLevelPlan &r, int sample_set)
{
    const int *slots;
    r.terms = render_term_slots(sample_set, &slots);
    for (int t = 0; t < r.terms; ++t) {
        r.inv_thickness[t] = r.cb.inv_thickness_table[slots[t]];
        r.front_depth[t] = r.inv_thickness[t] - 0.5f;  // Render.compute:85
        r.weight[t] = r.cb.sample_weight_table[slots[t]];
        r.scaled_weight[t] = r.weight[t] * render_term_scale(sample_set, t);   // power of two: exact
    }
}

void build_plan(int width, int height, int num_levels, int sample_set, const meao_params &p, Plan *out)
{
    out->width = width;
    out->height = height;
    out->num_levels = num_levels;
    for (int k = 0; k < kNumMips; ++k) out->mip[k] = level_dims(width, height, k);
    zbuffer_params(p, out->zbuffer_params);
    const float pad12 = linearize_out_of_range(out->zbuffer_params, p.reversed_z != 0);
    for (int level = 1; level <= 4; ++level) {
        RenderLevelPlan &r = out->render[level - 1];
        render_constants(width, height, p, level, true, sample_set, &r.cb);
        fill_terms(r, sample_set);
        RenderLevelPlan &q = out->render_hq[level - 1];
        render_constants(width, height, p, level, false, sample_set, &q.cb);
        fill_terms(q, sample_set);
        q.pad_value = 0.0f;                              // unused: the 2D source clamps
        // Atlas padding: TiledDepth1/2 hold Linearize(out-of-range) (Downsample1.compute:39-46,
        // 70-78), TiledDepth3/4 hold the 0 of an out-of-range DS4x load (Downsample2.compute:35).
        r.pad_value = level <= 2 ? pad12 : 0.0f;
        upsample_constants(width, height, p, level, &out->upsample[level - 1]);
    }
}

bool describe_buffer(int width, int height, int ao_format, int debug_id, meao_desc *out)
{
    if (debug_id < 1 || debug_id > MEAO_NUM_BUFFERS) return false;
    // AO.cs:453-475 in _debug order (AO.cs:789-808): id, mip level, format class, tiled
    int level, fmt, slices = 1;
    const int ao_fmt = ao_format == MEAO_AO_R8 ? MEAO_FMT_UNORM8 : MEAO_FMT_F16;
    if (debug_id == 1) { level = 0; fmt = MEAO_FMT_F16; }                              // LinearDepth
    else if (debug_id <= 5) { level = debug_id - 1; fmt = MEAO_FMT_F32; }             // LowDepth1..4
    else if (debug_id <= 9) { level = debug_id - 3; fmt = MEAO_FMT_F16; slices = 16; } // TiledDepth1..4 = L3..L6
    else if (debug_id <= 13) { level = debug_id - 9; fmt = ao_fmt; }                  // Occlusion1..4
    else if (debug_id <= 16) { level = debug_id - 13; fmt = ao_fmt; }                 // Combined1..3
    else if (debug_id == 17) { level = 0; fmt = ao_fmt; }                             // AmbientOcclusion
    else { level = debug_id - MEAO_DEBUG_OCCLUSION_HQ1 + 1; fmt = ao_fmt; }           // OcclusionHQ1..4
    const Dims d = level_dims(width, height, level);
    const uint64_t elem = fmt == MEAO_FMT_F32 ? 4 : (fmt == MEAO_FMT_F16 ? 2 : 1);
    out->debug_id = debug_id;
    out->width = d.w;
    out->height = d.h;
    out->slices = slices;
    out->format = fmt;
    out->bytes = static_cast<uint64_t>(d.w) * d.h * slices * elem;
    return true;
}

uint64_t depth_elem(int depth_format)
{
    return (depth_format == MEAO_DEPTH_F32 || depth_format == MEAO_DEPTH_UNORM24) ? 4 : 2;
}

void algorithmic_bytes(int width, int height, int num_levels, int hq_levels, int ao_format, int depth_format,
                       uint64_t bytes[MEAO_NUM_PASSES])
{
    uint64_t p[kNumMips];
    for (int k = 0; k < kNumMips; ++k) {
        const Dims d = level_dims(width, height, k);
        p[k] = static_cast<uint64_t>(d.w) * d.h;
    }
    const uint64_t a = ao_format == MEAO_AO_R8 ? 1 : 2;
    std::memset(bytes, 0, sizeof(uint64_t) * MEAO_NUM_PASSES);
    // Downsample1: read f32 L0, write f16 L0, (f32 + f16) L1, (f32 + f16) L2
    // Downsample2: read the used quarter of L2, write (f32 + f16) L3 and L4
    bytes[MEAO_PASS_DOWNSAMPLE] = depth_elem(depth_format) * p[0] + 2 * p[0] + 6 * p[1] + 6 * p[2] + 4 * p[3] + 6 * p[3] + 6 * p[4];
    for (int l = 1; l <= num_levels; ++l)  // 16 f16 slices of mip l+2 in, AO of mip l out
        bytes[MEAO_PASS_RENDER] += 2 * 16 * p[l + 2] + a * p[l];
    for (int hi = num_levels - 1; hi >= 1; --hi)  // lo (f32 + AO), hi (f32 + AO), out AO
        bytes[MEAO_PASS_UPSAMPLE_0 - hi] = (4 + a) * p[hi + 1] + (4 + a) * p[hi] + a * p[hi];
    bytes[MEAO_PASS_UPSAMPLE_0] = (4 + a) * p[1] + 2 * p[0] + a * p[0];
    for (int l = 1; l <= num_levels; ++l) {
        if (!level_has_hq(num_levels, hq_levels, l)) continue;
        bytes[MEAO_PASS_RENDER_HQ] += 4 * p[l] + a * p[l];   // f32 LowDepth<l> in, AO out
        bytes[MEAO_PASS_UPSAMPLE_0 - (l - 1)] += a * p[l];   // LoResAO2 of the pass that consumes level l
    }
}

}  // namespace meao
