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
constexpr int kNumDebugBuffers = 17;

struct Dims { int w = 0, h = 0; };

// Render taps actually used by the 36-sample checker set, in accumulation order
// (Render.compute:162-168): table slots 1,3,4,8,11,6,10.
constexpr int kNumRenderTerms = 7;
extern const int kRenderTermSlot[kNumRenderTerms];

struct RenderLevelPlan {
    meao_render_constants cb;                 // the reference's constant block, verbatim
    float inv_thickness[kNumRenderTerms];     // cb.inv_thickness_table[slot]
    float front_depth[kNumRenderTerms];       // inv_thickness - 0.5   (Render.compute:85)
    float weight[kNumRenderTerms];            // cb.sample_weight_table[slot]
    float pad_value;                          // what an out-of-level atlas texel holds
};

struct Plan {
    int width = 0, height = 0, num_levels = 4;
    Dims mip[kNumMips];
    float zbuffer_params[4];
    RenderLevelPlan render[4];                // level 1..4 -> [0..3]
    meao_upsample_constants upsample[4];      // low level 1..4 -> [0..3]
};

Dims level_dims(int width, int height, int level);
void zbuffer_params(const meao_params &p, float out[4]);
void sample_thickness(float out[12]);
void render_constants(int width, int height, const meao_params &p, int level,
                      meao_render_constants *out);
void upsample_constants(int width, int height, const meao_params &p, int low_level,
                        meao_upsample_constants *out);
// Linearize() of an out-of-range depth load (Downsample1.compute:39-46).
float linearize_out_of_range(const float zp[4], bool reversed_z);
bool params_valid(const meao_params &p);
void build_plan(int width, int height, int num_levels, const meao_params &p, Plan *out);

bool describe_buffer(int width, int height, int ao_format, int debug_id, meao_desc *out);
uint64_t depth_elem(int depth_format);   // bytes per input depth texel
void algorithmic_bytes(int width, int height, int num_levels, int ao_format, int depth_format,
                       uint64_t bytes[MEAO_NUM_PASSES]);

}  // namespace meao
