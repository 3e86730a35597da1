// meao_plan.hpp -- host-side plan of the SSAO hot path: level geometry, buffer table,
// per-dispatch constant blocks and algorithmic byte counts.  Pure CPU code (no HIP).
//
// Replaces the constant math and sizing that AmbientOcclusion.cs ("AO.cs") does while it
// records its command buffers: RTHandle.CalculateDimensions (AO.cs:276-281), the buffer
// table (AO.cs:453-475), CalculateZBufferParams (AO.cs:561-568), PushRenderCommands
// (AO.cs:660-734) and PushUpsampleCommands (AO.cs:750-771).
#pragma once

#include "../../include/meao.h"

namespace meao {

constexpr int kNumMips = 7;        // Original, L1..L6  (AO.cs:124)

struct Dims { int w = 0, h = 0; };

// Render terms in accumulation order: the 36-sample checker set uses table slots
// 1,3,4,8,11,6,10 (Render.compute:162-168), SAMPLE_EXHAUSTIVELY all twelve in the order
// 0,1,2,3,4,8,11,5,6,7,9,10 (Render.compute:146-157).
constexpr int kMaxRenderTerms = 12;
int render_term_slots(int sample_set, const int **slots);   // returns the number of terms
// leading factor of TestSamples for each term (Render.compute:87-109): 0.5 axial / diagonal, 0.25 L-shaped
float render_term_scale(int sample_set, int term);

struct RenderLevelPlan {
    meao_render_constants cb;                 // the reference's constant block, verbatim
    int terms;
    float inv_thickness[kMaxRenderTerms];     // cb.inv_thickness_table[slot]
    float front_depth[kMaxRenderTerms];       // inv_thickness - 0.5   (Render.compute:85)
    float weight[kMaxRenderTerms];            // cb.sample_weight_table[slot]
    float scaled_weight[kMaxRenderTerms];     // weight * render_term_scale (exact): what the kernels multiply by
    float pad_value;                          // what an out-of-level atlas texel holds
};

struct Plan {
    int width = 0, height = 0, num_levels = 4;
    Dims mip[kNumMips];
    float zbuffer_params[4];
    RenderLevelPlan render[4];                // level 1..4 -> [0..3], source TiledDepth<level>
    RenderLevelPlan render_hq[4];             // Render.main on LowDepth<level> (cfg.hq_levels)
    meao_upsample_constants upsample[4];      // low level 1..4 -> [0..3]
};

Dims level_dims(int width, int height, int level);
void zbuffer_params(const meao_params &p, float out[4]);
void sample_thickness(float out[12]);
void render_constants(int width, int height, const meao_params &p, int level, bool source_tiled,
                      int sample_set, meao_render_constants *out);
void upsample_constants(int width, int height, const meao_params &p, int low_level,
                        meao_upsample_constants *out);
// Linearize() of an out-of-range depth load (Downsample1.compute:39-46).
float linearize_out_of_range(const float zp[4], bool reversed_z);
bool params_valid(const meao_params &p);
void build_plan(int width, int height, int num_levels, int sample_set, const meao_params &p, Plan *out);
// level k (1..4) has a Render.main pass iff it is one of the coarsest hq_levels rendered levels
inline bool level_has_hq(int num_levels, int hq_levels, int k) { return k <= num_levels && k > num_levels - hq_levels; }

bool describe_buffer(int width, int height, int ao_format, int debug_id, meao_desc *out);
uint64_t depth_elem(int depth_format);   // bytes per input depth texel
void algorithmic_bytes(int width, int height, int num_levels, int hq_levels, int ao_format, int depth_format,
                       uint64_t bytes[MEAO_NUM_PASSES]);

}  // namespace meao
