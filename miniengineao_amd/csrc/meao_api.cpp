// meao_api.cpp -- the C ABI of libmeao_hip.so (include/meao.h): context management, the
// per-frame launch sequence, intermediates, profiling.  No CPU fallback exists anywhere in
// this library: without a gfx950 device meao_create fails with MEAO_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "meao_kernels.hpp"
#include "meao_plan.hpp"

using namespace meao;

namespace {

#ifndef MEAO_TESTING
#define MEAO_TESTING 0      // 1: the `testhooks` variant library -- exports meao_test_* fault injection, never the product
#endif

thread_local std::string g_last_error;   // for failures that have no context (meao_create)

constexpr uint64_t kAlign = 256;
constexpr int kProfileRing = 256;         // executes buffered before timings are folded
constexpr int kProfSlots = MEAO_NUM_PASSES;   // launch slots of one execute: one start / end event pair each
inline uint64_t align_up(uint64_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

}  // namespace

struct meao_ctx {
    meao_config cfg{};
    meao_params prm{};
    Plan plan{};
    hipStream_t own_stream = nullptr;
    hipStream_t last_stream = nullptr;

    // context-owned intermediates: max_batch identical slots inside one arena
    char *arena = nullptr;
    uint64_t slot_bytes = 0;
    uint64_t off_occ[4] = {}, off_comb[3] = {};
    // Downsample outputs (LowDepth1..4; LinearDepth is never materialised on the hot path).  With meao_prefetch_batch in use the
    // slot holds two such sets: the passes of a call read set `ds_cur` while its last kernel fills the other one with the next
    // batch's downsample.
    uint64_t off_ds_low[4] = {}, ds_set_bytes = 0;
    bool two_ds_sets = false;
    int ds_cur = 0;
    uint64_t off_low_of(int set, int k) const { return off_ds_low[k] + ds_set_bytes * set; }
    int next_n = 0;                               // announced by meao_prefetch_batch, consumed by the next execute
    const void *next_depth[MEAO_MAX_BATCH] = {};
    int ready_n = 0, ready_set = 0;               // a set already downsampled from exactly these frames
    const void *ready_depth[MEAO_MAX_BATCH] = {};
    hipStream_t ready_stream = nullptr;           // the stream the carrying execute ran on
    // Hostile-depth flags (meao_dev_downsample.hpp nice_denominator): [set][frame] words the downsample pass
    // stamps with its generation when a frame's levels hold texels outside the exact-division range.
    uint32_t *hostile = nullptr;
    uint32_t gen_counter = 0, set_gen[2] = {0, 0};
    uint32_t *hostile_of(int set) const { return hostile + set * MEAO_MAX_BATCH; }
    uint64_t off_hq[4] = {};                  // OcclusionHQ<k>: only the levels cfg.hq_levels enables

    // lazily allocated: staging for HOST in/out, scratch for the buffers built on demand (LinearDepth, TiledDepth<k>), selftest counter
    char *stage_depth = nullptr, *stage_out = nullptr, *stage_view = nullptr, *atlas_scratch = nullptr;
    uint64_t stage_depth_frame = 0, stage_out_frame = 0, atlas_scratch_bytes = 0;
    unsigned long long *counter = nullptr;

    // operands of every divide provably inside the exact range of the v_rcp_f32 sequences
    // (meao_dev.hpp "Exact division"); recomputed by update_plan()
    int exact_rcp_div = 0;

    // Launch structures with identical results, chosen by call size; meao_debug_set overrides the thresholds (tests, A/B runs) --
    // the library reads no environment variables.
    bool fuse_coarse_blend = true;     // Upsample L4->L3 evaluated inside the L3->L2 launch (upsample_two_level_kernel)
    int ds_small_max_tiles = 640;      // stand-alone downsample pass: calls with at most this many 128x16 (LowDepth1 texels) tiles use 128x8 tiles
    int final_small_max_tiles = 2048;  // plain final pass: calls with at most this many 64x64 tiles (one 4K frame: 2040) use 64x32 tiles (r04 sweep: 60.3 vs 60.8 us)
    int render_small_max_tiles = 256;  // calls with at most this many 128x32 render tiles (frames x tiles) use 128x8 tiles
    int nested_max_tiles = 1024;       // calls with at most this many L2->L1 tiles (frames x tiles; one 4K frame: 1020) run the three blend passes as one launch
                                       // (with the round-4 blend_window_into_lds: 55.9 vs 56.6 us per pipelined 4K frame, a tie unpipelined; 512 before)
    // L2 -> L1 launches of at least this many 64x32 tiles (frames x tiles; 4K: 1020 per frame) use 64x64 tiles with R8 AO storage
    // (upsample_blend_tall_kernel): 53.9 -> 52.5 us per 16 frames at 4K, 57.7 -> 56.3 at 1080p x 64, fp16 storage +-0
    // (profiles/r05_ab_blend_tall.jsonl).  MEAO_DEBUG_BLEND_TALL_MIN_TILES overrides it for both storage formats.
    int blend_tall_min_tiles = 4096;
    bool blend_tall_forced = false;
    // The announced next batch's downsample pass (meao_prefetch_batch) always as a launch of its own behind the last kernel instead of
    // inside it (MEAO_DEBUG_NEXT_DOWNSAMPLE_OWN_LAUNCH; what calls whose frames do not take the carried tile's 16-byte loads do anyway)
    bool next_ds_own_launch = false;

    // a composite batch waiting to ride inside the next execute's render kernel (meao_composite_enqueue),
    // and the stream its AO frames were produced on (where a flush that is not given a stream runs it)
    CompositeBatchArgs pending_comp{};
    hipStream_t pending_stream = nullptr;

#if MEAO_TESTING
    int debug_fail_allocs = 0;         // meao_test_fail_next_allocs: arena allocations still to fail (testhooks variant only)
#endif

    const void *last_out[MEAO_MAX_BATCH] = {};   // device address of the last results (debug id 17)
    const void *last_depth[MEAO_MAX_BATCH] = {}; // device address of the last call's raw depth frames (debug id 1 is built from them)
    int last_frames = 0;

    // profiling: a ring of per-execute event sets (one start/end pair per launch slot); each entry
    // remembers which slots it used
    bool profiling = false;
    uint32_t profile_mask = ~0u;                 // MEAO_DEBUG_PROFILE_PASS_MASK: bit k = launch slot k is bracketed with events
    uint32_t profile_period = 1, profile_phase = 0;   // meao_set_profiling(N > 1): every Nth execute is bracketed with events, the others run bare
    std::vector<hipEvent_t> events;              // kProfileRing * kProfSlots * 2
    int ring_fill = 0;
    uint32_t ran_mask[kProfileRing] = {};        // bit k: launch slot k ran in that execute
    double pass_ms_sum[MEAO_NUM_PASSES] = {};
    int pass_samples[MEAO_NUM_PASSES] = {};      // executes that ran pass k
    int executes_profiled = 0;

    // roctx ranges around every pass (meao_set_tracing); libroctx64.so is loaded on first use
    bool tracing = false;
    void *roctx_lib = nullptr;
    int (*roctx_push)(const char *) = nullptr;
    int (*roctx_pop)() = nullptr;

    std::string err;
};

namespace {

int fail(meao_ctx *ctx, int status, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    g_last_error = msg;
    return status;
}

int fail_hip(meao_ctx *ctx, hipError_t e, const char *what)
{
    (void)hipGetLastError();
    char buf[256];
    std::snprintf(buf, sizeof buf, "%s: %s (%d)", what, hipGetErrorString(e), static_cast<int>(e));
    return fail(ctx, e == hipErrorOutOfMemory ? MEAO_ERR_OUT_OF_MEMORY : MEAO_ERR_HIP, buf);
}

#define MEAO_HIP(ctx, expr)                                      \
    do {                                                         \
        hipError_t e_ = (expr);                                  \
        if (e_ != hipSuccess) return fail_hip((ctx), e_, #expr); \
    } while (0)

bool config_valid(const meao_config &c, std::string *why)
{
    if (c.width < 1 || c.height < 1 || c.width > 32768 || c.height > 32768) { *why = "width/height out of range [1, 32768]"; return false; }
    if (c.num_levels < 1 || c.num_levels > 4) { *why = "num_levels must be 1..4"; return false; }
    if (c.ao_format != MEAO_AO_R8 && c.ao_format != MEAO_AO_F16) { *why = "unknown ao_format"; return false; }
    if (c.f16_rounding != MEAO_F16_RTZ_CLAMP && c.f16_rounding != MEAO_F16_RTNE) { *why = "unknown f16_rounding"; return false; }
    if (c.max_batch < 1 || c.max_batch > MEAO_MAX_BATCH) { *why = "max_batch must be 1..MEAO_MAX_BATCH"; return false; }
    if (c.depth_format < MEAO_DEPTH_F32 || c.depth_format > MEAO_DEPTH_F16) { *why = "unknown depth_format"; return false; }
    if (c.hq_levels < 0 || c.hq_levels > c.num_levels) { *why = "hq_levels must be 0..num_levels"; return false; }
    if (c.sample_set != MEAO_SAMPLES_CHECKER && c.sample_set != MEAO_SAMPLES_EXHAUSTIVE) { *why = "unknown sample_set"; return false; }
    if (c.pipelined != 0 && c.pipelined != 1) { *why = "pipelined must be 0 or 1"; return false; }
    return true;
}

uint64_t ao_elem(const meao_config &c) { return c.ao_format == MEAO_AO_R8 ? 1 : 2; }

// Where the intermediates of one frame live inside its slot; a pure function of (plan, cfg, two_ds_sets).
struct SlotLayout {
    uint64_t off_ds_low[4] = {}, ds_set_bytes = 0, off_occ[4] = {}, off_comb[3] = {}, off_hq[4] = {}, slot_bytes = 0;
};

SlotLayout layout_slot(const Plan &p, const meao_config &cfg, bool two_ds_sets)
{
    SlotLayout l;
    uint64_t off = 0;
    auto take = [&](uint64_t bytes) { const uint64_t o = off; off = align_up(off + bytes); return o; };
    auto px = [&](int k) { return static_cast<uint64_t>(p.mip[k].w) * p.mip[k].h; };
    for (int k = 1; k <= 4; ++k) l.off_ds_low[k - 1] = take(px(k) * 4);
    l.ds_set_bytes = off;
    if (two_ds_sets) off = 2 * off;          // second set: same layout, ds_set_bytes further
    for (int k = 1; k <= 4; ++k) l.off_occ[k - 1] = take(px(k) * ao_elem(cfg));
    for (int k = 1; k <= 3; ++k) l.off_comb[k - 1] = take(px(k) * ao_elem(cfg));
    for (int k = 1; k <= 4; ++k) l.off_hq[k - 1] = level_has_hq(cfg.num_levels, cfg.hq_levels, k) ? take(px(k) * ao_elem(cfg)) : 0;
    l.slot_bytes = off;
    return l;
}

void apply_layout(meao_ctx *ctx, const SlotLayout &l)
{
    ctx->ds_set_bytes = l.ds_set_bytes;
    ctx->slot_bytes = l.slot_bytes;
    std::memcpy(ctx->off_ds_low, l.off_ds_low, sizeof l.off_ds_low);
    std::memcpy(ctx->off_occ, l.off_occ, sizeof l.off_occ);
    std::memcpy(ctx->off_comb, l.off_comb, sizeof l.off_comb);
    std::memcpy(ctx->off_hq, l.off_hq, sizeof l.off_hq);
}

// Divides on the path: 1/LoResDB, 1/centre depth, {9,3,1,3}/(|dHi-dLo| + tol), (HiAO*sum)/total.
// The exact v_rcp_f32 sequences need their operands inside verified ranges.  The DATA side of that
// (every linear depth in [2^-24, 2^20] or the sky value, finite, not NaN) is checked on the device: per frame
// by the downsample pass for the texels the levels are made of (nice_denominator; hostile frames take the IEEE
// bodies), per lane by the full-resolution upsample for the texels it linearizes itself.  The
// PARAMETER side is checked here: weights are <= 9/tol, total and sum are >= noise strength.
bool exact_rcp_div_applicable(const meao_config &c, const meao_params &p, const Plan &plan)
{
    if (c.f16_rounding != MEAO_F16_RTZ_CLAMP) return false;               // RTNE stores inf for sky
    (void)p;
    for (int k = 0; k < 4; ++k) {
        const meao_upsample_constants &u = plan.upsample[k];
        if (!(u.upsample_tolerance >= 0x1p-44f && u.upsample_tolerance <= 0x1p20f)) return false;  // weights <= 9 * 2^44
        if (!(u.noise_filter_strength >= 0x1p-30f && u.noise_filter_strength <= 0x1p50f)) return false;
    }
    return true;
}

void drop_prefetch(meao_ctx *ctx) { ctx->next_n = 0; ctx->ready_n = 0; ctx->ready_stream = nullptr; }

void update_plan(meao_ctx *ctx)
{
    drop_prefetch(ctx); // a prefetched downsample was computed with the old Z-buffer parameters
    build_plan(ctx->cfg.width, ctx->cfg.height, ctx->cfg.num_levels, ctx->cfg.sample_set, ctx->prm, &ctx->plan);
    ctx->exact_rcp_div = exact_rcp_div_applicable(ctx->cfg, ctx->prm, ctx->plan) ? 1 : 0;
}

void release_staging(meao_ctx *ctx)
{
    if (ctx->stage_depth) (void)hipFree(ctx->stage_depth);
    if (ctx->stage_out) (void)hipFree(ctx->stage_out);
    if (ctx->stage_view) (void)hipFree(ctx->stage_view);
    if (ctx->atlas_scratch) (void)hipFree(ctx->atlas_scratch);
    ctx->stage_depth = ctx->stage_out = ctx->stage_view = ctx->atlas_scratch = nullptr;
    ctx->atlas_scratch_bytes = 0;
}

void release_buffers(meao_ctx *ctx)
{
    if (ctx->arena) (void)hipFree(ctx->arena);
    ctx->arena = nullptr;
    release_staging(ctx);
    ctx->last_frames = 0;
}

// (Re)plans for cfg/two_ds_sets and replaces the arena.  The new geometry is planned on the side and its
// arena allocated BEFORE anything of the context changes: on failure the context is untouched -- geometry,
// buffers and a ready prefetch all stay as they were.
int reallocate(meao_ctx *ctx, const meao_config &cfg, bool two_ds_sets)
{
    Plan plan{};
    build_plan(cfg.width, cfg.height, cfg.num_levels, cfg.sample_set, ctx->prm, &plan);
    const SlotLayout lay = layout_slot(plan, cfg, two_ds_sets);
    char *fresh = nullptr;
    hipError_t e;
#if MEAO_TESTING
    if (ctx->debug_fail_allocs > 0) {      // fault injection: exists only in the `testhooks` variant library (-DMEAO_TESTING=1)
        --ctx->debug_fail_allocs;
        e = hipErrorOutOfMemory;
    } else
#endif
    {
        e = hipMalloc(reinterpret_cast<void **>(&fresh), lay.slot_bytes * cfg.max_batch);
    }
    if (e != hipSuccess) return fail_hip(ctx, e, "hipMalloc (intermediates)");
    ctx->cfg = cfg;
    ctx->two_ds_sets = two_ds_sets;
    update_plan(ctx);           // drops a ready prefetch: it refers to the old arena
    apply_layout(ctx, lay);
    release_buffers(ctx);
    ctx->arena = fresh;
    return MEAO_OK;
}

int use_device(meao_ctx *ctx)
{
    MEAO_HIP(ctx, hipSetDevice(ctx->cfg.device));
    return MEAO_OK;
}

template <typename T>
T *slot_ptr(meao_ctx *ctx, uint64_t off) { return reinterpret_cast<T *>(ctx->arena + off); }

void fold_profile(meao_ctx *ctx)
{
    if (ctx->ring_fill == 0) return;
    for (int r = 0; r < ctx->ring_fill; ++r) {
        for (int k = 0; k < kProfSlots; ++k) {
            if (!(ctx->ran_mask[r] >> k & 1u)) continue;
            hipEvent_t a = ctx->events[(r * kProfSlots + k) * 2], b = ctx->events[(r * kProfSlots + k) * 2 + 1];
            (void)hipEventSynchronize(b);
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, a, b) != hipSuccess) continue;
            ctx->pass_ms_sum[k] += ms;
            ++ctx->pass_samples[k];
        }
        ++ctx->executes_profiled;
    }
    ctx->ring_fill = 0;
}

// Runs a pending composite batch as plain composite launches (one per frame) on `stream`.
int flush_pending_composite(meao_ctx *ctx, hipStream_t stream)
{
    CompositeBatchArgs &pc = ctx->pending_comp;
    const int frames = pc.frames;
    pc.frames = 0;
    for (int f = 0; f < frames; ++f) {
        CompositeArgs ca{};
        ca.ao = pc.ao[f]; ca.color = pc.color[f]; ca.gbuffer0 = pc.gbuffer0[f];
        ca.pixels = pc.pixels; ca.mode = pc.mode;
        MEAO_HIP(ctx, launch_composite(ca, ctx->cfg.ao_format, stream));
    }
    return MEAO_OK;
}

struct TraceRange {   // roctx range around one pass (no-op unless meao_set_tracing enabled it)
    meao_ctx *ctx;
    TraceRange(meao_ctx *c, const char *name) : ctx(c) { if (ctx->tracing && ctx->roctx_push) ctx->roctx_push(name); }
    ~TraceRange() { if (ctx->tracing && ctx->roctx_pop) ctx->roctx_pop(); }
};

bool aligned_to(const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// ------------------------------------------------------------------------------------------
// One call = plan -> launch list -> submit.  The launch structure of a batch is DATA (a LaunchList); choosing it
// (plan_launches), building the kernel arguments of a launch (ArgBuilder) and issuing it (submit_launches) are separate steps.

// What RebuildCommandBuffers records (AO.cs:511-531), in the shapes this implementation launches it in.
enum class Step {
    Downsample,               // Downsample1 + Downsample2 (AO.cs:604-658): the four levels of THIS call's frames
    Render,                   // Render.main_interleaved x levels, one grid (AO.cs:519-522)
    RenderWithComposite,      // ... carrying the composite of an earlier call's frames (meao_composite_enqueue)
    RenderHq,                 // Render.main (wide) for the levels cfg.hq_levels enables
    Blend,                    // Upsample.main_blendout writing level `hi` (AO.cs:528-530)
    BlendTwoLevel,            // L4 -> L3 evaluated inside the L3 -> L2 launch
    BlendThreeLevel,          // L4 -> L3 and L3 -> L2 evaluated inside the L2 -> L1 launch (small calls)
    Final,                    // Upsample.main: the result (AO.cs:531)
    FinalWithNextDownsample,  // ... carrying the downsample pass of the announced next batch (meao_prefetch_batch)
    DownsampleNext            // the announced batch's pass as a launch of its own behind the final one (where the fused form does not apply)
};

struct Launch {
    Step step;
    int slot;                 // meao_pass the launch is timed under; -1 = not timed
    int hi;                   // Blend*: the level written
    const char *range;        // roctx range name
};

struct LaunchList {
    Launch v[12];
    int n = 0;
    void add(Step step, int slot, int hi, const char *range) { v[n++] = Launch{step, slot, hi, range}; }
};

struct BatchShape {           // what the structure of a call depends on
    int frames;
    bool prefetched;          // an earlier call carried this batch's downsample pass
    bool carry_composite;     // a composite batch waits for a render launch to ride in
    int next;                 // 0 = nothing announced; the announced pass 1 = rides in the final kernel, 2 = runs as its own launch
};

LaunchList plan_launches(const meao_ctx *ctx, const BatchShape &b)
{
    const meao_config &c = ctx->cfg;
    const Plan &p = ctx->plan;
    static const char *const kBlendRange[4] = {nullptr, "meao:upsample_L2_to_L1", "meao:upsample_L3_to_L2", "meao:upsample_L4_to_L3"};
    LaunchList l;
    if (!b.prefetched) l.add(Step::Downsample, MEAO_PASS_DOWNSAMPLE, 0, "meao:downsample");
    if (b.carry_composite) l.add(Step::RenderWithComposite, MEAO_PASS_RENDER, 0, "meao:render+composite_of_previous_call");
    else l.add(Step::Render, MEAO_PASS_RENDER, 0, "meao:render");
    if (c.hq_levels > 0) l.add(Step::RenderHq, MEAO_PASS_RENDER_HQ, 0, "meao:render_hq");
    const bool nestable = ctx->fuse_coarse_blend && c.num_levels == 4 && c.hq_levels == 0;
    const int l1_tiles = ((p.mip[1].w + kUpsTileW - 1) / kUpsTileW) * ((p.mip[1].h + ups_tile_h(false) - 1) / ups_tile_h(false));
    if (nestable && b.frames * l1_tiles <= ctx->nested_max_tiles) {
        // a small frame or two per call (at most two workgroups per CU): all three blend passes in one launch -- their latency
        // chains, not their arithmetic, are what such a call waits for (1080p: 39.2 -> 36.9 us per frame; at 4K, 1020 tiles, it
        // is a wash).  Combined3 and Combined2 are still written
        l.add(Step::BlendThreeLevel, MEAO_PASS_UPSAMPLE_1, 1, "meao:upsample_L4_to_L3+L3_to_L2+L2_to_L1");
    } else {
        // L4 -> L3 inside the L3 -> L2 launch: one launch, one latency-bound pass less (Combined3 is still written)
        if (nestable) l.add(Step::BlendTwoLevel, MEAO_PASS_UPSAMPLE_2, 2, "meao:upsample_L4_to_L3+L3_to_L2");
        else for (int hi = c.num_levels - 1; hi >= 2; --hi) l.add(Step::Blend, MEAO_PASS_UPSAMPLE_0 - hi, hi, kBlendRange[hi]);
        if (c.num_levels >= 2) l.add(Step::Blend, MEAO_PASS_UPSAMPLE_1, 1, kBlendRange[1]);
    }
    if (b.next == 1) l.add(Step::FinalWithNextDownsample, MEAO_PASS_UPSAMPLE_0, 0, "meao:upsample_L1_to_L0+downsample_next");
    else l.add(Step::Final, MEAO_PASS_UPSAMPLE_0, 0, "meao:upsample_L1_to_L0");
    // (timed in the DOWNSAMPLE slot only in calls that did not run a pass of their own there)
    if (b.next == 2) l.add(Step::DownsampleNext, b.prefetched ? MEAO_PASS_DOWNSAMPLE : -1, 0, "meao:downsample_next");
    return l;
}

// Kernel arguments of the launches of one call.
struct ArgBuilder {
    meao_ctx *ctx;
    int n;
    const void *const *depth_dev;
    void *const *out_dev;
    const uint32_t *hostile;      // flags and generation of the downsample set this call reads
    uint32_t generation;

    const Plan &p() const { return ctx->plan; }
    const meao_config &c() const { return ctx->cfg; }
    int rtne() const { return ctx->cfg.f16_rounding == MEAO_F16_RTNE; }
    // 4-texel vector loads / stores need 16-byte (f32, UNORM24), 8-byte (16-bit) aligned depth rows and 4- (R8) / 8-byte (F16)
    // aligned AO rows: width % 4 == 0 (% 8 for the downsample pass, whose lanes take 8 raw texels) and aligned base pointers
    // (include/meao.h); anything else takes the scalar variants.
    uintptr_t depth_align() const { return 4 * depth_elem(ctx->cfg.depth_format); }
    uintptr_t out_align() const { return 4 * ao_elem(ctx->cfg); }

    // ---- PushDownsampleCommands (AO.cs:604-658).  lean: tiled for the tile the final kernel carries (kLeanMipW x kLeanMipRows),
    // else for the stand-alone pass (small_ok: calls with few tiles use the one-row-per-lane tile)
    DownsampleArgs downsample(int frames, const void *const *depth, int set, uint32_t gen, bool lean, bool small_ok) const
    {
        DownsampleArgs ds{};
        bool aligned = (p().mip[0].w & 7) == 0;
        for (int f = 0; f < frames; ++f) {
            ds.depth[f] = depth[f];
            aligned = aligned && aligned_to(depth[f], depth_align());
        }
        ds.vec_ok = aligned;
        ds.frames = frames;
        ds.depth_format = c().depth_format;
        for (int k = 0; k < 4; ++k) ds.low[k] = slot_ptr<float>(ctx, ctx->off_low_of(set, k));
        ds.frame_stride = ctx->slot_bytes;
        for (int k = 0; k < 5; ++k) { ds.w[k] = p().mip[k].w; ds.h[k] = p().mip[k].h; }
        ds.zp0 = p().zbuffer_params[0];
        ds.zp1 = p().zbuffer_params[1];
        ds.reversed_z = ctx->prm.reversed_z != 0;
        ds.f16_rtne = rtne();
        ds.exact_rcp_div = ctx->exact_rcp_div;
        if (lean) {
            ds.rows_per_lane = 1;
            ds.tiles_x = (ds.w[1] + kLeanMipW - 1) / kLeanMipW;
            ds.tiles_y = (ds.h[1] + kLeanMipRows - 1) / kLeanMipRows;
        } else {
            ds.rows_per_lane = kMipRowsPerLane;
            ds.tiles_x = (ds.w[1] + kMipTileW - 1) / kMipTileW;
            ds.tiles_y = (ds.h[1] + kMipRowsPerPass * kMipRowsPerLane - 1) / (kMipRowsPerPass * kMipRowsPerLane);
            if (small_ok && frames * ds.tiles_x * ds.tiles_y <= ctx->ds_small_max_tiles) {
                ds.rows_per_lane = 1;
                ds.tiles_y = (ds.h[1] + kMipRowsPerPass - 1) / kMipRowsPerPass;
            }
        }
        ds.hostile = ctx->hostile_of(set);
        ds.generation = gen;
        return ds;
    }

    // ---- PushRenderCommands x num_levels (AO.cs:519-522): the levels [first, last] as one grid
    RenderArgs render(int first, int last, bool wide, bool allow_small) const
    {
        RenderArgs rn{};
        int blocks = 0, count = 0;
        // few tiles (a 1080p frame or two): 128 x 8 tiles instead of 128 x 32 (render_small_kernel)
        int tile_h = kRenTileH;
        if (allow_small && !wide && c().sample_set != MEAO_SAMPLES_EXHAUSTIVE) {
            int tiles32 = 0;
            for (int l = first; l <= last; ++l)
                tiles32 += ((p().mip[l].w + ren_tile_w(false) - 1) / ren_tile_w(false)) * ((p().mip[l].h + kRenTileH - 1) / kRenTileH);
            if (n * tiles32 <= ctx->render_small_max_tiles) tile_h = kRenTileHSmall;
        }
        rn.tile_h = tile_h;
        for (int l = first; l <= last; ++l) {
            if (wide && !level_has_hq(c().num_levels, c().hq_levels, l)) continue;
            RenderLevelArgs &L = rn.level[count++];
            const RenderLevelPlan &rp = wide ? p().render_hq[l - 1] : p().render[l - 1];
            L.src = slot_ptr<float>(ctx, ctx->off_low_of(ctx->ds_cur, l - 1));
            L.dst = slot_ptr<void>(ctx, wide ? ctx->off_hq[l - 1] : ctx->off_occ[l - 1]);
            L.lw = p().mip[l].w; L.lh = p().mip[l].h;
            L.sw = p().mip[l + 2].w; L.sh = p().mip[l + 2].h;
            const int tile_w = wide ? kWideTileW : ren_tile_w(c().sample_set == MEAO_SAMPLES_EXHAUSTIVE);
            L.tiles_x = (L.lw + tile_w - 1) / tile_w;
            L.tiles_y = (L.lh + tile_h - 1) / tile_h;
            L.block_begin = blocks;
            blocks += L.tiles_x * L.tiles_y;
            L.pad_value = rp.pad_value;
            for (int t = 0; t < rp.terms; ++t) {
                L.inv_thickness[t] = rp.inv_thickness[t];
                L.front_depth[t] = rp.front_depth[t];
                L.weight[t] = rp.scaled_weight[t];
            }
            L.reject_fadeoff = rp.cb.reject_fadeoff;
            L.intensity = rp.cb.intensity;
        }
        rn.frame_stride = ctx->slot_bytes;
        rn.num_levels = count;
        rn.blocks_per_frame = blocks;
        rn.f16_rtne = rtne();
        rn.exact_rcp_div = ctx->exact_rcp_div;
        rn.exhaustive = c().sample_set == MEAO_SAMPLES_EXHAUSTIVE;
        rn.hostile = hostile;
        rn.generation = generation;
        return rn;
    }

    // ---- PushUpsampleCommands (AO.cs:750-785): the pass that writes level `hi`.  carrying: the final pass of a call whose
    // last kernel carries the next batch's downsample pass (always 64 x 64 tiles)
    UpsampleArgs upsample(int hi, bool carrying = false) const
    {
        UpsampleArgs up{};
        const meao_upsample_constants &k = p().upsample[hi];   // low level = hi + 1
        up.lo_depth = slot_ptr<float>(ctx, ctx->off_low_of(ctx->ds_cur, hi));
        // LoResAO1: the coarsest level's Occlusion, else the Combined buffer of the previous pass
        up.lo_ao = slot_ptr<void>(ctx, hi == c().num_levels - 1 ? ctx->off_occ[hi] : ctx->off_comb[hi]);
        // main_premin*: the Render.main output of the low level is min-combined in PrefetchData
        up.lo_ao2 = level_has_hq(c().num_levels, c().hq_levels, hi + 1) ? slot_ptr<void>(ctx, ctx->off_hq[hi]) : nullptr;
        up.frame_stride = ctx->slot_bytes;
        up.lw = p().mip[hi + 1].w; up.lh = p().mip[hi + 1].h;
        up.hw = p().mip[hi].w; up.hh = p().mip[hi].h;
        up.tiles_x = (up.hw + kUpsTileW - 1) / kUpsTileW;
        up.tile_h = ups_tile_h(hi == 0);
        // few tiles (one 1080p frame): the plain final pass runs 64 x 32 tiles (upsample_final_small_kernel)
        if (hi == 0 && !carrying && n * up.tiles_x * ((up.hh + up.tile_h - 1) / up.tile_h) <= ctx->final_small_max_tiles)
            up.tile_h = kUpsTileHSmall;
        // L2 -> L1 of a large batch: 64 x 64 tiles like the full-resolution pass (upsample_blend_tall_kernel)
        if (hi == 1 && (c().ao_format == MEAO_AO_R8 || ctx->blend_tall_forced) &&
            static_cast<int64_t>(n) * up.tiles_x * ((up.hh + up.tile_h - 1) / up.tile_h) >= ctx->blend_tall_min_tiles)
            up.tile_h = kUpsTileHTall;
        up.tiles_y = (up.hh + up.tile_h - 1) / up.tile_h;
        up.noise_filter_strength = k.noise_filter_strength;
        up.step_size = k.step_size;
        up.blur_tolerance = k.blur_tolerance;
        up.upsample_tolerance = k.upsample_tolerance;
        up.f16_rtne = rtne();
        up.exact_rcp_div = ctx->exact_rcp_div;
        up.hostile = hostile;
        up.generation = generation;
        bool vec_ok = (up.hw & 3) == 0;
        if (hi > 0) {   // main_blendout: blend with Occlusion<hi>, write Combined<hi>
            up.hi_depth = slot_ptr<float>(ctx, ctx->off_low_of(ctx->ds_cur, hi - 1));
            up.hi_ao = slot_ptr<void>(ctx, ctx->off_occ[hi - 1]);
            up.dst[0] = slot_ptr<void>(ctx, ctx->off_comb[hi - 1]);
        } else {        // main: HiResDB from the raw depth frames (hi_depth()), no HiResAO, write the result
            up.hi_depth = nullptr;
            up.hi_ao = nullptr;
            for (int f = 0; f < n; ++f) {
                up.dst[f] = out_dev[f];
                vec_ok = vec_ok && aligned_to(out_dev[f], out_align()) && aligned_to(depth_dev[f], depth_align());
            }
        }
        up.vec_ok = vec_ok;
        return up;
    }

    // HiResDB of Upsample.main = LinearZ (DS1:37-48), which the final pass evaluates from this call's raw depth frames
    HiDepthArgs hi_depth() const
    {
        HiDepthArgs hd{};
        for (int f = 0; f < n; ++f) hd.raw[f] = depth_dev[f];
        hd.depth_format = c().depth_format;
        hd.reversed_z = ctx->prm.reversed_z != 0;
        hd.zp0 = p().zbuffer_params[0];
        hd.zp1 = p().zbuffer_params[1];
        return hd;
    }
};

uint32_t next_generation(meao_ctx *ctx) { if (++ctx->gen_counter == 0) ++ctx->gen_counter; return ctx->gen_counter; }   // never 0

// The launch sequence of one batch.
int run_batch(meao_ctx *ctx, int n, const void *const *depth_dev, void *const *out_dev, hipStream_t stream)
{
    const meao_config &c = ctx->cfg;
    hipEvent_t *ev = nullptr;
    uint32_t ran = 0;
    if (ctx->profiling) {
        if (ctx->profile_phase == 0) {
            if (ctx->ring_fill == kProfileRing) fold_profile(ctx);
            ev = &ctx->events[ctx->ring_fill * kProfSlots * 2];
        }
        if (++ctx->profile_phase >= ctx->profile_period) ctx->profile_phase = 0;
    }

    // A previous call may already have downsampled exactly these frames (meao_prefetch_batch).  The
    // prefetched set is only valid on the stream of the execute that carried it: stream order is what
    // orders that kernel before this call's readers.
    BatchShape shape{};
    shape.frames = n;
    shape.prefetched = ctx->ready_n == n && ctx->ready_stream == stream && std::memcmp(ctx->ready_depth, depth_dev, sizeof(void *) * n) == 0;
    ctx->ds_cur = shape.prefetched ? ctx->ready_set : 0;
    ctx->ready_n = 0;
    if (!shape.prefetched) ctx->set_gen[ctx->ds_cur] = next_generation(ctx);      // direct launches take a fresh generation per pass: no flag clearing
    if (ctx->pending_comp.frames > 0 && c.sample_set == MEAO_SAMPLES_EXHAUSTIVE) {
        const int rc = flush_pending_composite(ctx, stream);     // the 68-sample render kernel carries nothing
        if (rc != MEAO_OK) return rc;
    }
    shape.carry_composite = ctx->pending_comp.frames > 0;

    const ArgBuilder args{ctx, n, depth_dev, out_dev, ctx->hostile_of(ctx->ds_cur), ctx->set_gen[ctx->ds_cur]};
    // The announced next batch: its pass rides in this call's last kernel where the fused form applies (f32 depth, 16-byte
    // loads, a workgroup per carried tile), else it runs as a launch of its own behind it.  Either way the next call finds it done.
    // (Inside the render launch instead -- CarriedMips in the texel loop, round 6 -- it costs the same 56-60 us per 16 4K frames:
    // profiles/r06_ab_next_downsample_in_render_vs_final_vs_own_launch.jsonl, r06_scripts/r06_downsample_in_render.patch.)
    const int other = 1 - ctx->ds_cur;
    DownsampleArgs next_ds{};
    UpsampleArgs final_up = args.upsample(0);
    const HiDepthArgs hi_depth = args.hi_depth();
    if (ctx->next_n > 0) {
        ctx->set_gen[other] = next_generation(ctx);
        next_ds = args.downsample(ctx->next_n, ctx->next_depth, other, ctx->set_gen[other], true, false);
        const UpsampleArgs carrying = args.upsample(0, true);
        if (!ctx->next_ds_own_launch && fused_downsample_applicable(carrying, hi_depth, next_ds, n)) {
            shape.next = 1;
            final_up = carrying;
        } else {
            shape.next = 2;
            next_ds = args.downsample(ctx->next_n, ctx->next_depth, other, ctx->set_gen[other], false, true);
        }
    }

    const LaunchList list = plan_launches(ctx, shape);
    for (int i = 0; i < list.n; ++i) {
        const Launch &L = list.v[i];
        TraceRange tr(ctx, L.range);
        // one launch = one profiling slot: events right before and after it on its stream
        const bool timed = L.slot >= 0 && (ctx->profile_mask >> L.slot & 1u);
        if (timed && ev) MEAO_HIP(ctx, hipEventRecord(ev[L.slot * 2], stream));
        switch (L.step) {
        case Step::Downsample:
            MEAO_HIP(ctx, launch_downsample(args.downsample(n, depth_dev, ctx->ds_cur, ctx->set_gen[ctx->ds_cur], false, true), n, stream));
            break;
        case Step::Render:
            MEAO_HIP(ctx, launch_render(args.render(1, c.num_levels, false, true), c.ao_format, n, stream));
            break;
        case Step::RenderWithComposite:
            // the composite of frames an earlier call produced streams under this (VALU-bound) kernel
            MEAO_HIP(ctx, launch_render_with_composite(args.render(1, c.num_levels, false, false), ctx->pending_comp, c.ao_format, n, stream));
            ctx->pending_comp.frames = 0;
            break;
        case Step::RenderHq:
            MEAO_HIP(ctx, launch_render_wide(args.render(1, c.num_levels, true, false), c.ao_format, n, stream));
            break;
        case Step::Blend:
            MEAO_HIP(ctx, launch_upsample(args.upsample(L.hi), nullptr, c.ao_format, n, stream));
            break;
        case Step::BlendTwoLevel:
            MEAO_HIP(ctx, launch_upsample_two_level(args.upsample(2), args.upsample(3), c.ao_format, n, stream));
            break;
        case Step::BlendThreeLevel: {
            UpsampleArgs outer = args.upsample(1);
            if (outer.tile_h != ups_tile_h(false)) {        // the nested launch tiles L2 -> L1 with 64 x 32
                outer.tile_h = ups_tile_h(false);
                outer.tiles_y = (outer.hh + outer.tile_h - 1) / outer.tile_h;
            }
            MEAO_HIP(ctx, launch_upsample_three_level(outer, args.upsample(2), args.upsample(3), c.ao_format, n, stream));
            break;
        }
        case Step::Final:
            MEAO_HIP(ctx, launch_upsample(final_up, &hi_depth, c.ao_format, n, stream));
            break;
        case Step::FinalWithNextDownsample:
            MEAO_HIP(ctx, launch_upsample_final_with_downsample(final_up, hi_depth, next_ds, c.ao_format, n, stream));
            break;
        case Step::DownsampleNext:
            MEAO_HIP(ctx, launch_downsample(next_ds, ctx->next_n, stream));
            break;
        }
        if (timed) {
            ran |= 1u << L.slot;
            if (ev) MEAO_HIP(ctx, hipEventRecord(ev[L.slot * 2 + 1], stream));
        }
    }
    if (shape.next != 0) {
        ctx->ready_n = ctx->next_n;
        ctx->ready_set = other;
        ctx->ready_stream = stream;
        std::memcpy(ctx->ready_depth, ctx->next_depth, sizeof ctx->ready_depth);
        ctx->next_n = 0;
    }
    if (ev) ctx->ran_mask[ctx->ring_fill++] = ran;
    for (int f = 0; f < n; ++f) { ctx->last_out[f] = out_dev[f]; ctx->last_depth[f] = depth_dev[f]; }
    ctx->last_frames = n;
    ctx->last_stream = stream;
    return MEAO_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" {

int32_t meao_abi_version(void) { return MEAO_ABI_VERSION; }

const char *meao_status_string(int32_t status)
{
    switch (status) {
    case MEAO_OK: return "ok";
    case MEAO_ERR_INVALID_ARGUMENT: return "invalid argument";
    case MEAO_ERR_HIP: return "HIP runtime error";
    case MEAO_ERR_OUT_OF_MEMORY: return "out of memory";
    case MEAO_ERR_UNSUPPORTED: return "unsupported";
    case MEAO_ERR_NO_DEVICE: return "no gfx950 device (there is no CPU fallback)";
    case MEAO_ERR_BUFFER_TOO_SMALL: return "destination buffer too small";
    default: return "unknown status";
    }
}

void meao_default_config(meao_config *cfg)
{
    if (!cfg) return;
    std::memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = sizeof *cfg;
    cfg->device = 0;
    cfg->width = 1920;
    cfg->height = 1080;
    cfg->num_levels = 4;
    cfg->ao_format = MEAO_AO_R8;
    cfg->f16_rounding = MEAO_F16_RTZ_CLAMP;
    cfg->max_batch = 1;
    cfg->depth_format = MEAO_DEPTH_F32;
    cfg->pipelined = 0;
}

void meao_default_params(meao_params *p)
{
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->struct_size = sizeof *p;
    p->noise_filter_tolerance = 0.0f;   // AO.cs:20
    p->blur_tolerance = -4.6f;          // AO.cs:28
    p->upsample_tolerance = -12.0f;     // AO.cs:36
    p->thickness_modifier = 1.0f;       // AO.cs:44
    p->intensity = 1.0f;                // AO.cs:52
    p->near_clip = 0.3f;                // Unity camera defaults
    p->far_clip = 1000.0f;
    p->proj00 = 0.9742786f;             // fovY 60 deg at 16:9
    p->reversed_z = 1;
}

int32_t meao_level_dims(int32_t width, int32_t height, int32_t level, int32_t *out_w, int32_t *out_h)
{
    if (width < 1 || height < 1 || level < 0 || level >= kNumMips || !out_w || !out_h)
        return MEAO_ERR_INVALID_ARGUMENT;
    const Dims d = level_dims(width, height, level);
    *out_w = d.w;
    *out_h = d.h;
    return MEAO_OK;
}

int32_t meao_zbuffer_params(const meao_params *p, float out[4])
{
    if (!p || !out || !params_valid(*p)) return MEAO_ERR_INVALID_ARGUMENT;
    zbuffer_params(*p, out);
    return MEAO_OK;
}

int32_t meao_render_constants_for(int32_t width, int32_t height, const meao_params *p, int32_t level,
                                  meao_render_constants *out)
{
    if (width < 1 || height < 1 || !p || !out || level < 1 || level > 4 || !params_valid(*p))
        return MEAO_ERR_INVALID_ARGUMENT;
    render_constants(width, height, *p, level, true, MEAO_SAMPLES_CHECKER, out);
    return MEAO_OK;
}

int32_t meao_render_constants_variant(int32_t width, int32_t height, const meao_params *p, int32_t level,
                                      int32_t source_tiled, int32_t sample_set, meao_render_constants *out)
{
    if (width < 1 || height < 1 || !p || !out || level < 1 || level > 4 || !params_valid(*p))
        return MEAO_ERR_INVALID_ARGUMENT;
    if (sample_set != MEAO_SAMPLES_CHECKER && sample_set != MEAO_SAMPLES_EXHAUSTIVE) return MEAO_ERR_INVALID_ARGUMENT;
    render_constants(width, height, *p, level, source_tiled != 0, sample_set, out);
    return MEAO_OK;
}

int32_t meao_upsample_constants_for(int32_t width, int32_t height, const meao_params *p, int32_t low_level,
                                    meao_upsample_constants *out)
{
    if (width < 1 || height < 1 || !p || !out || low_level < 1 || low_level > 4 || !params_valid(*p))
        return MEAO_ERR_INVALID_ARGUMENT;
    upsample_constants(width, height, *p, low_level, out);
    return MEAO_OK;
}

int32_t meao_describe_buffer(const meao_config *cfg, int32_t debug_id, meao_desc *out)
{
    std::string why;
    if (!cfg || !out || !config_valid(*cfg, &why)) return MEAO_ERR_INVALID_ARGUMENT;
    return describe_buffer(cfg->width, cfg->height, cfg->ao_format, debug_id, out) ? MEAO_OK
                                                                                   : MEAO_ERR_INVALID_ARGUMENT;
}

int32_t meao_algorithmic_bytes(const meao_config *cfg, uint64_t bytes[MEAO_NUM_PASSES])
{
    std::string why;
    if (!cfg || !bytes || !config_valid(*cfg, &why)) return MEAO_ERR_INVALID_ARGUMENT;
    algorithmic_bytes(cfg->width, cfg->height, cfg->num_levels, cfg->hq_levels, cfg->ao_format, cfg->depth_format, bytes);
    return MEAO_OK;
}

int32_t meao_create(const meao_config *cfg, meao_ctx **out_ctx)
{
    if (out_ctx) *out_ctx = nullptr;
    if (!cfg || !out_ctx) return fail(nullptr, MEAO_ERR_INVALID_ARGUMENT, "meao_create: null argument");
    if (cfg->struct_size != sizeof(meao_config))
        return fail(nullptr, MEAO_ERR_INVALID_ARGUMENT, "meao_create: struct_size mismatch (ABI)");
    std::string why;
    if (!config_valid(*cfg, &why)) return fail(nullptr, MEAO_ERR_INVALID_ARGUMENT, "meao_create: " + why);

    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, MEAO_ERR_NO_DEVICE, "meao_create: no HIP device visible (no CPU fallback exists)");
    }
    if (cfg->device < 0 || cfg->device >= count)
        return fail(nullptr, MEAO_ERR_INVALID_ARGUMENT, "meao_create: device ordinal out of range");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, cfg->device)) != hipSuccess) return fail_hip(nullptr, e, "hipGetDeviceProperties");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, MEAO_ERR_NO_DEVICE,
                    std::string("meao_create: device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");

    meao_ctx *ctx = new (std::nothrow) meao_ctx();
    if (!ctx) return fail(nullptr, MEAO_ERR_OUT_OF_MEMORY, "meao_create: host allocation failed");
    ctx->cfg = *cfg;
    meao_default_params(&ctx->prm);
    int rc = use_device(ctx);
    if (rc == MEAO_OK) {
        e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
        if (e != hipSuccess) rc = fail_hip(ctx, e, "hipStreamCreateWithFlags");
    }
    if (rc == MEAO_OK) {
        // hostile-depth flags: 2 downsample sets x MEAO_MAX_BATCH frames, zero = never hostile (generations start at 1)
        const size_t bytes = 2 * MEAO_MAX_BATCH * sizeof(uint32_t);
        e = hipMalloc(reinterpret_cast<void **>(&ctx->hostile), bytes);
        if (e == hipSuccess) e = hipMemset(ctx->hostile, 0, bytes);
        if (e != hipSuccess) rc = fail_hip(ctx, e, "hipMalloc (hostile flags)");
    }
    // cfg.pipelined: the second downsample set exists from the start, so meao_prefetch_batch never re-allocates
    if (rc == MEAO_OK) rc = reallocate(ctx, *cfg, cfg->pipelined != 0);
    if (rc != MEAO_OK) {
        g_last_error = ctx->err;
        meao_destroy(ctx);
        return rc;
    }
    ctx->last_stream = ctx->own_stream;
    *out_ctx = ctx;
    return MEAO_OK;
}

int32_t meao_destroy(meao_ctx *ctx)
{
    if (!ctx) return MEAO_OK;
    int prev_device = -1;
    (void)hipGetDevice(&prev_device);
    (void)hipSetDevice(ctx->cfg.device);
    // A composite batch still waiting is DISCARDED (meao.h): its targets are caller memory whose lifetime has
    // typically ended by now (free the frames, then destroy); hosts that want it call meao_composite_flush first.
    ctx->pending_comp.frames = 0;
    (void)hipDeviceSynchronize();
    release_buffers(ctx);
    if (ctx->counter) (void)hipFree(ctx->counter);
    if (ctx->hostile) (void)hipFree(ctx->hostile);
    if (ctx->roctx_lib) (void)dlclose(ctx->roctx_lib);
    for (hipEvent_t ev : ctx->events) (void)hipEventDestroy(ev);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    if (prev_device >= 0) (void)hipSetDevice(prev_device);
    return MEAO_OK;
}

int32_t meao_resize(meao_ctx *ctx, int32_t width, int32_t height)
{
    if (!ctx) return MEAO_ERR_INVALID_ARGUMENT;
    meao_config c = ctx->cfg;
    c.width = width;
    c.height = height;
    std::string why;
    if (!config_valid(c, &why)) return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_resize: " + why);
    int rc = use_device(ctx);
    if (rc != MEAO_OK) return rc;
    if (ctx->pending_comp.frames > 0) {        // sized for the old geometry: run it now, where its AO frames were produced
        rc = flush_pending_composite(ctx, ctx->pending_stream);
        if (rc != MEAO_OK) return rc;
    }
    MEAO_HIP(ctx, hipDeviceSynchronize());
    // on failure (e.g. out of memory) the context keeps its previous size and buffers
    return reallocate(ctx, c, ctx->two_ds_sets);
}

int32_t meao_set_params(meao_ctx *ctx, const meao_params *p)
{
    if (!ctx || !p) return MEAO_ERR_INVALID_ARGUMENT;
    if (p->struct_size != sizeof(meao_params)) return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_set_params: struct_size mismatch (ABI)");
    if (!params_valid(*p)) return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_set_params: non-finite or degenerate parameter");
    ctx->prm = *p;
    update_plan(ctx);
    return MEAO_OK;
}

int32_t meao_get_params(const meao_ctx *ctx, meao_params *out)
{
    if (!ctx || !out) return MEAO_ERR_INVALID_ARGUMENT;
    *out = ctx->prm;
    return MEAO_OK;
}

int32_t meao_get_config(const meao_ctx *ctx, meao_config *out)
{
    if (!ctx || !out) return MEAO_ERR_INVALID_ARGUMENT;
    *out = ctx->cfg;
    return MEAO_OK;
}

const char *meao_last_error(const meao_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

}  // extern "C"

// meao_execute_batch; wait_for_host = false (pool members only) leaves the staged copies of a HOST call in
// flight on `stream_` -- the caller synchronises the stream before it touches the host buffers.
int meao::execute_batch_internal(meao_ctx *ctx, int32_t n, const void *const *depth, int32_t depth_loc, void *const *ao_out,
                                 int32_t out_loc, meao_stream stream_, bool wait_for_host)
{
    if (!ctx || !depth || !ao_out) return MEAO_ERR_INVALID_ARGUMENT;
    if (n < 1 || n > ctx->cfg.max_batch) return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_execute_batch: n must be 1..max_batch");
    if ((depth_loc != MEAO_MEM_HOST && depth_loc != MEAO_MEM_DEVICE) || (out_loc != MEAO_MEM_HOST && out_loc != MEAO_MEM_DEVICE))
        return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_execute_batch: bad memory location");
    for (int f = 0; f < n; ++f)
        if (!depth[f] || !ao_out[f]) return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_execute_batch: null frame pointer");
    if (!ctx->arena) return fail(ctx, MEAO_ERR_OUT_OF_MEMORY, "meao_execute_batch: the context has no intermediates");
    int rc = use_device(ctx);
    if (rc != MEAO_OK) return rc;
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : ctx->own_stream;

    const uint64_t px = static_cast<uint64_t>(ctx->cfg.width) * ctx->cfg.height;
    const uint64_t depth_bytes = px * depth_elem(ctx->cfg.depth_format), out_bytes = px * ao_elem(ctx->cfg);
    const void *depth_dev[MEAO_MAX_BATCH];
    void *out_dev[MEAO_MAX_BATCH];
    if (depth_loc == MEAO_MEM_HOST) {
        if (!ctx->stage_depth) {
            ctx->stage_depth_frame = align_up(depth_bytes);
            MEAO_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->stage_depth), ctx->stage_depth_frame * ctx->cfg.max_batch));
        }
        for (int f = 0; f < n; ++f) {
            char *d = ctx->stage_depth + ctx->stage_depth_frame * f;
            MEAO_HIP(ctx, hipMemcpyAsync(d, depth[f], depth_bytes, hipMemcpyHostToDevice, stream));
            depth_dev[f] = d;
        }
    } else {
        for (int f = 0; f < n; ++f) depth_dev[f] = depth[f];
    }
    if (out_loc == MEAO_MEM_HOST) {
        if (!ctx->stage_out) {
            ctx->stage_out_frame = align_up(out_bytes);
            MEAO_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->stage_out), ctx->stage_out_frame * ctx->cfg.max_batch));
        }
        for (int f = 0; f < n; ++f) out_dev[f] = ctx->stage_out + ctx->stage_out_frame * f;
    } else {
        for (int f = 0; f < n; ++f) out_dev[f] = ao_out[f];
    }

    rc = run_batch(ctx, n, depth_dev, out_dev, stream);
    if (rc != MEAO_OK) return rc;

    if (out_loc == MEAO_MEM_HOST)
        for (int f = 0; f < n; ++f)
            MEAO_HIP(ctx, hipMemcpyAsync(ao_out[f], out_dev[f], out_bytes, hipMemcpyDeviceToHost, stream));
    if (wait_for_host && (out_loc == MEAO_MEM_HOST || depth_loc == MEAO_MEM_HOST)) MEAO_HIP(ctx, hipStreamSynchronize(stream));
    return MEAO_OK;
}

extern "C" {

int32_t meao_execute_batch(meao_ctx *ctx, int32_t n, const void *const *depth, int32_t depth_loc,
                           void *const *ao_out, int32_t out_loc, meao_stream stream_)
{
    return meao::execute_batch_internal(ctx, n, depth, depth_loc, ao_out, out_loc, stream_, true);
}

int32_t meao_prefetch_batch(meao_ctx *ctx, int32_t n, const void *const *depth)
{
    if (!ctx || !depth) return MEAO_ERR_INVALID_ARGUMENT;
    if (n < 1 || n > ctx->cfg.max_batch) return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_prefetch_batch: n must be 1..max_batch");
    for (int f = 0; f < n; ++f)
        if (!depth[f]) return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_prefetch_batch: null frame pointer");
    int rc = use_device(ctx);
    if (rc != MEAO_OK) return rc;
    if (!ctx->two_ds_sets) {
        // Created without cfg.pipelined: the first announcement re-lays the slots out with a second
        // downsample set (device-wide synchronisation + allocation, once).  On failure the context is unchanged.
        MEAO_HIP(ctx, hipDeviceSynchronize());
        rc = reallocate(ctx, ctx->cfg, true);
        if (rc != MEAO_OK) return rc;
    }
    ctx->next_n = n;
    for (int f = 0; f < n; ++f) ctx->next_depth[f] = depth[f];
    return MEAO_OK;
}

int32_t meao_execute(meao_ctx *ctx, const void *depth, int32_t depth_loc, void *ao_out, int32_t out_loc,
                     meao_stream stream)
{
    const void *d[1] = {depth};
    void *o[1] = {ao_out};
    return meao_execute_batch(ctx, 1, d, depth_loc, o, out_loc, stream);
}

int32_t meao_synchronize(meao_ctx *ctx, meao_stream stream)
{
    if (!ctx) return MEAO_ERR_INVALID_ARGUMENT;
    int rc = use_device(ctx);
    if (rc != MEAO_OK) return rc;
    MEAO_HIP(ctx, hipStreamSynchronize(stream ? static_cast<hipStream_t>(stream) : ctx->last_stream));
    return MEAO_OK;
}

// Scratch for the buffers the hot path never materialises (LinearDepth, TiledDepth<level>): at least `bytes`.
static int reserve_scratch(meao_ctx *ctx, uint64_t bytes)
{
    if (ctx->atlas_scratch_bytes >= bytes) return MEAO_OK;
    if (ctx->atlas_scratch) (void)hipFree(ctx->atlas_scratch);
    ctx->atlas_scratch = nullptr;
    ctx->atlas_scratch_bytes = 0;
    MEAO_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->atlas_scratch), bytes));
    ctx->atlas_scratch_bytes = bytes;
    return MEAO_OK;
}

// Device address of debug buffer `debug_id` of batch slot `frame` (LinearDepth and TiledDepth are built on demand).
static int locate_debug_buffer(meao_ctx *ctx, int32_t frame, int32_t debug_id, const meao_desc &d, hipStream_t s,
                               const void **out_src)
{
    const char *slot = ctx->arena + ctx->slot_bytes * frame;
    const int nl = ctx->cfg.num_levels;
    if (debug_id == 1) {
        // LinearDepth: materialised on demand from the raw depth frame of the last call (its one consumer on the hot path,
        // the full-resolution upsample, evaluates Linearize itself).  The caller's depth frame must still be alive.
        const int rc = reserve_scratch(ctx, d.bytes);
        if (rc != MEAO_OK) return rc;
        LinearDepthArgs la{};
        la.depth = ctx->last_depth[frame];
        la.dst = reinterpret_cast<uint16_t *>(ctx->atlas_scratch);
        la.pixels = static_cast<int64_t>(d.width) * d.height;
        la.depth_format = ctx->cfg.depth_format;
        la.reversed_z = ctx->prm.reversed_z != 0;
        la.f16_rtne = ctx->cfg.f16_rounding == MEAO_F16_RTNE;
        la.zp0 = ctx->plan.zbuffer_params[0];
        la.zp1 = ctx->plan.zbuffer_params[1];
        MEAO_HIP(ctx, launch_linear_depth(la, s));
        *out_src = ctx->atlas_scratch;
    } else if (debug_id <= 5) *out_src = slot + ctx->off_low_of(ctx->ds_cur, debug_id - 2);
    else if (debug_id <= 9) {
        // TiledDepth<level>: materialised on demand from LowDepth<level> (the hot path samples
        // LowDepth directly and never builds the de-interleaved arrays).
        const int level = debug_id - 5;
        const int rc = reserve_scratch(ctx, d.bytes);
        if (rc != MEAO_OK) return rc;
        TileAtlasArgs ta{};
        ta.src = reinterpret_cast<const float *>(slot + ctx->off_low_of(ctx->ds_cur, level - 1));
        ta.dst = reinterpret_cast<uint16_t *>(ctx->atlas_scratch);
        ta.lw = ctx->plan.mip[level].w; ta.lh = ctx->plan.mip[level].h;
        ta.sw = d.width; ta.sh = d.height;
        ta.pad_value = ctx->plan.render[level - 1].pad_value;
        ta.f16_rtne = ctx->cfg.f16_rounding == MEAO_F16_RTNE;
        MEAO_HIP(ctx, launch_tile_atlas(ta, s));
        *out_src = ctx->atlas_scratch;
    } else if (debug_id <= 13) {
        if (debug_id - 9 > nl) return fail(ctx, MEAO_ERR_UNSUPPORTED, "debug buffer: level not rendered (num_levels)");
        *out_src = slot + ctx->off_occ[debug_id - 10];
    } else if (debug_id <= 16) {
        if (debug_id - 13 > nl - 1) return fail(ctx, MEAO_ERR_UNSUPPORTED, "debug buffer: level not combined (num_levels)");
        *out_src = slot + ctx->off_comb[debug_id - 14];
    } else if (debug_id == 17) {
        *out_src = ctx->last_out[frame];
    } else {
        const int level = debug_id - MEAO_DEBUG_OCCLUSION_HQ1 + 1;
        if (!level_has_hq(nl, ctx->cfg.hq_levels, level))
            return fail(ctx, MEAO_ERR_UNSUPPORTED, "debug buffer: this level has no Render.main pass (hq_levels)");
        *out_src = slot + ctx->off_hq[level - 1];
    }
    return MEAO_OK;
}

int32_t meao_get_intermediate(meao_ctx *ctx, int32_t frame, int32_t debug_id, void *dst, uint64_t dst_capacity,
                              int32_t dst_loc, meao_desc *out_desc)
{
    if (!ctx) return MEAO_ERR_INVALID_ARGUMENT;
    meao_desc d{};
    if (!describe_buffer(ctx->cfg.width, ctx->cfg.height, ctx->cfg.ao_format, debug_id, &d))
        return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_get_intermediate: debug_id must be 1..21");
    if (out_desc) *out_desc = d;
    if (!dst) return MEAO_OK;
    if (!ctx->arena) return fail(ctx, MEAO_ERR_OUT_OF_MEMORY, "meao_get_intermediate: the context has no intermediates");
    if (frame < 0 || frame >= ctx->last_frames)
        return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_get_intermediate: frame not produced by the last execute");
    if (dst_capacity < d.bytes) return fail(ctx, MEAO_ERR_BUFFER_TOO_SMALL, "meao_get_intermediate: dst_capacity < desc.bytes");
    if (dst_loc != MEAO_MEM_HOST && dst_loc != MEAO_MEM_DEVICE) return MEAO_ERR_INVALID_ARGUMENT;
    int rc = use_device(ctx);
    if (rc != MEAO_OK) return rc;
    hipStream_t s = ctx->last_stream;
    const void *src = nullptr;
    rc = locate_debug_buffer(ctx, frame, debug_id, d, s, &src);
    if (rc != MEAO_OK) return rc;
    MEAO_HIP(ctx, hipMemcpyAsync(dst, src, d.bytes,
                                 dst_loc == MEAO_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s));
    MEAO_HIP(ctx, hipStreamSynchronize(s));
    return MEAO_OK;
}

int32_t meao_debug_view(meao_ctx *ctx, int32_t frame, int32_t debug_id, void *out, int32_t out_loc, meao_stream stream_)
{
    if (!ctx || !out) return MEAO_ERR_INVALID_ARGUMENT;
    meao_desc d{};
    if (!describe_buffer(ctx->cfg.width, ctx->cfg.height, ctx->cfg.ao_format, debug_id, &d))
        return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_debug_view: debug_id must be 1..21");
    if (!ctx->arena) return fail(ctx, MEAO_ERR_OUT_OF_MEMORY, "meao_debug_view: the context has no intermediates");
    if (frame < 0 || frame >= ctx->last_frames)
        return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_debug_view: frame not produced by the last execute");
    if (out_loc != MEAO_MEM_HOST && out_loc != MEAO_MEM_DEVICE) return MEAO_ERR_INVALID_ARGUMENT;
    int rc = use_device(ctx);
    if (rc != MEAO_OK) return rc;
    hipStream_t s = stream_ ? static_cast<hipStream_t>(stream_) : ctx->last_stream;
    const void *src = nullptr;
    rc = locate_debug_buffer(ctx, frame, debug_id, d, s, &src);
    if (rc != MEAO_OK) return rc;
    const uint64_t out_bytes = static_cast<uint64_t>(ctx->cfg.width) * ctx->cfg.height * ao_elem(ctx->cfg);
    void *dev_out = out;
    if (out_loc == MEAO_MEM_HOST) {   // own staging buffer: stage_out may hold the results (debug id 17)
        if (!ctx->stage_view) MEAO_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->stage_view), align_up(out_bytes)));
        dev_out = ctx->stage_view;
    }
    DebugViewArgs dv{};
    dv.src = src; dv.dst = dev_out;
    dv.sw = d.width; dv.sh = d.height; dv.slices = d.slices; dv.src_format = d.format;
    dv.w = ctx->cfg.width; dv.h = ctx->cfg.height;
    dv.f16_rtne = ctx->cfg.f16_rounding == MEAO_F16_RTNE;
    MEAO_HIP(ctx, launch_debug_view(dv, ctx->cfg.ao_format, s));
    if (out_loc == MEAO_MEM_HOST) {
        MEAO_HIP(ctx, hipMemcpyAsync(out, dev_out, out_bytes, hipMemcpyDeviceToHost, s));
        MEAO_HIP(ctx, hipStreamSynchronize(s));
    }
    return MEAO_OK;
}

int32_t meao_set_profiling(meao_ctx *ctx, int32_t enable)
{
    if (!ctx) return MEAO_ERR_INVALID_ARGUMENT;
    int rc = use_device(ctx);
    if (rc != MEAO_OK) return rc;
    if (enable && ctx->events.empty()) {
        const int count = kProfileRing * kProfSlots * 2;
        ctx->events.reserve(count);
        for (int i = 0; i < count; ++i) {
            hipEvent_t ev;
            // timing only: no system-scope fence (cache write-back + invalidate) when a record completes -- nothing synchronizes
            // with these events but hipEventElapsedTime
            MEAO_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableSystemFence));
            ctx->events.push_back(ev);
        }
    }
    if (enable) {   // (re)start a measurement window
        if (ctx->ring_fill) fold_profile(ctx);
        std::memset(ctx->pass_ms_sum, 0, sizeof ctx->pass_ms_sum);
        std::memset(ctx->pass_samples, 0, sizeof ctx->pass_samples);
        ctx->executes_profiled = 0;
    }
    ctx->profiling = enable != 0;
    ctx->profile_period = enable > 1 ? enable : 1;
    ctx->profile_phase = 0;
    return MEAO_OK;
}

int32_t meao_get_pass_times(meao_ctx *ctx, float ms[MEAO_NUM_PASSES], int32_t *out_samples)
{
    if (!ctx || !ms) return MEAO_ERR_INVALID_ARGUMENT;
    int rc = use_device(ctx);
    if (rc != MEAO_OK) return rc;
    // the documented contract: the call returns after the stream of the last execute has drained (executes that recorded no
    // events -- meao_set_profiling(N > 1), PROFILE_PASS_MASK -- included; the timing events themselves carry no fence)
    MEAO_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    fold_profile(ctx);
    // mean over the executes that actually ran the pass (a prefetched downsample does not dilute it)
    for (int k = 0; k < MEAO_NUM_PASSES; ++k)
        ms[k] = ctx->pass_samples[k] ? static_cast<float>(ctx->pass_ms_sum[k] / ctx->pass_samples[k]) : 0.0f;
    if (out_samples) *out_samples = ctx->executes_profiled;
    return MEAO_OK;
}

int32_t meao_set_tracing(meao_ctx *ctx, int32_t enable)
{
    if (!ctx) return MEAO_ERR_INVALID_ARGUMENT;
    if (enable && !ctx->roctx_lib) {
        // rocprofv3 (rocprofiler-sdk) intercepts the roctx API of its own library; libroctx64.so is the
        // older roctracer one (rocprof v1/v2) with the same entry points
        static const char *const kCandidates[] = {"librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                                                  "libroctx64.so", "/opt/rocm/lib/libroctx64.so"};
        void *lib = nullptr;
        for (const char *name : kCandidates)
            if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
        if (!lib) return fail(ctx, MEAO_ERR_UNSUPPORTED, "meao_set_tracing: no roctx library found (librocprofiler-sdk-roctx.so / libroctx64.so)");
        ctx->roctx_push = reinterpret_cast<int (*)(const char *)>(dlsym(lib, "roctxRangePushA"));
        ctx->roctx_pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        if (!ctx->roctx_push || !ctx->roctx_pop) {
            (void)dlclose(lib);
            ctx->roctx_push = nullptr; ctx->roctx_pop = nullptr;
            return fail(ctx, MEAO_ERR_UNSUPPORTED, "meao_set_tracing: roctxRangePushA / roctxRangePop missing");
        }
        ctx->roctx_lib = lib;
    }
    ctx->tracing = enable != 0;
    return MEAO_OK;
}

int32_t meao_composite(meao_ctx *ctx, int32_t mode, const void *ao, void *color_rgba16f, void *gbuffer0_rgba8,
                       int32_t loc, meao_stream stream_)
{
    if (!ctx || !ao || !color_rgba16f) return MEAO_ERR_INVALID_ARGUMENT;
    if (mode < MEAO_COMPOSITE_MULTIPLY || mode > MEAO_COMPOSITE_DEBUG) return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_composite: unknown mode");
    if (mode == MEAO_COMPOSITE_AMBIENT_ONLY && !gbuffer0_rgba8)
        return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_composite: AMBIENT_ONLY needs the GBuffer0 target");
    if (loc != MEAO_MEM_HOST && loc != MEAO_MEM_DEVICE) return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_composite: bad memory location");
    int rc = use_device(ctx);
    if (rc != MEAO_OK) return rc;
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : ctx->own_stream;
    const uint64_t px = static_cast<uint64_t>(ctx->cfg.width) * ctx->cfg.height;
    const uint64_t ao_bytes = px * ao_elem(ctx->cfg), color_bytes = px * 8, g_bytes = px * 4;
    CompositeArgs ca{};
    ca.pixels = static_cast<int64_t>(px);
    ca.mode = mode;
    char *scratch = nullptr;
    if (loc == MEAO_MEM_HOST) {     // tools / tests: stage through one temporary device buffer
        MEAO_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&scratch), align_up(ao_bytes) + align_up(color_bytes) + g_bytes));
        char *d_ao = scratch, *d_color = scratch + align_up(ao_bytes), *d_g = d_color + align_up(color_bytes);
        hipError_t e = hipMemcpyAsync(d_ao, ao, ao_bytes, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_color, color_rgba16f, color_bytes, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess && gbuffer0_rgba8) e = hipMemcpyAsync(d_g, gbuffer0_rgba8, g_bytes, hipMemcpyHostToDevice, stream);
        ca.ao = d_ao; ca.color = d_color; ca.gbuffer0 = gbuffer0_rgba8 ? d_g : nullptr;
        if (e == hipSuccess) e = launch_composite(ca, ctx->cfg.ao_format, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(color_rgba16f, d_color, color_bytes, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess && gbuffer0_rgba8) e = hipMemcpyAsync(gbuffer0_rgba8, d_g, g_bytes, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(scratch);
        if (e != hipSuccess) return fail_hip(ctx, e, "meao_composite (host staging)");
        return MEAO_OK;
    }
    ca.ao = ao; ca.color = color_rgba16f; ca.gbuffer0 = gbuffer0_rgba8;
    MEAO_HIP(ctx, launch_composite(ca, ctx->cfg.ao_format, stream));
    return MEAO_OK;
}

int32_t meao_composite_enqueue(meao_ctx *ctx, int32_t mode, int32_t n, const void *const *ao, void *const *color_rgba16f,
                               void *const *gbuffer0_rgba8)
{
    if (!ctx || !ao || !color_rgba16f) return MEAO_ERR_INVALID_ARGUMENT;
    if (mode < MEAO_COMPOSITE_MULTIPLY || mode > MEAO_COMPOSITE_DEBUG) return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_composite_enqueue: unknown mode");
    if (n < 1 || n > MEAO_MAX_BATCH) return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_composite_enqueue: n must be 1..MEAO_MAX_BATCH");
    if (mode == MEAO_COMPOSITE_AMBIENT_ONLY && !gbuffer0_rgba8)
        return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_composite_enqueue: AMBIENT_ONLY needs the GBuffer0 targets");
    for (int f = 0; f < n; ++f)
        if (!ao[f] || !color_rgba16f[f] || (mode == MEAO_COMPOSITE_AMBIENT_ONLY && !gbuffer0_rgba8[f]))
            return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_composite_enqueue: null frame pointer");
    int rc = use_device(ctx);
    if (rc != MEAO_OK) return rc;
    if (ctx->pending_comp.frames > 0) {        // one batch can wait at a time: the older one runs now, in order
        rc = flush_pending_composite(ctx, ctx->pending_stream);
        if (rc != MEAO_OK) return rc;
    }
    ctx->pending_stream = ctx->last_stream;    // the stream of the execute that (by contract) produced ao[f]
    CompositeBatchArgs &pc = ctx->pending_comp;
    for (int f = 0; f < n; ++f) {
        pc.ao[f] = ao[f];
        pc.color[f] = color_rgba16f[f];
        pc.gbuffer0[f] = gbuffer0_rgba8 ? gbuffer0_rgba8[f] : nullptr;
    }
    pc.pixels = static_cast<int64_t>(ctx->cfg.width) * ctx->cfg.height;
    pc.mode = mode;
    pc.frames = n;
    return MEAO_OK;
}

int32_t meao_composite_pending(const meao_ctx *ctx, int32_t *out_frames)
{
    if (!ctx || !out_frames) return MEAO_ERR_INVALID_ARGUMENT;
    *out_frames = ctx->pending_comp.frames;
    return MEAO_OK;
}

int32_t meao_composite_flush(meao_ctx *ctx, meao_stream stream_)
{
    if (!ctx) return MEAO_ERR_INVALID_ARGUMENT;
    int rc = use_device(ctx);
    if (rc != MEAO_OK) return rc;
    if (ctx->pending_comp.frames == 0) return MEAO_OK;
    return flush_pending_composite(ctx, stream_ ? static_cast<hipStream_t>(stream_) : ctx->pending_stream);
}

int32_t meao_hostile_frames(meao_ctx *ctx, uint64_t *out_mask)
{
    if (!ctx || !out_mask) return MEAO_ERR_INVALID_ARGUMENT;
    *out_mask = 0;
    if (ctx->last_frames == 0) return MEAO_OK;
    int rc = use_device(ctx);
    if (rc != MEAO_OK) return rc;
    uint32_t words[MEAO_MAX_BATCH];
    MEAO_HIP(ctx, hipMemcpyAsync(words, ctx->hostile_of(ctx->ds_cur), sizeof(uint32_t) * ctx->last_frames, hipMemcpyDeviceToHost,
                                 ctx->last_stream));
    MEAO_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    for (int f = 0; f < ctx->last_frames; ++f)
        if (words[f] == ctx->set_gen[ctx->ds_cur]) *out_mask |= uint64_t(1) << f;
    return MEAO_OK;
}

int32_t meao_debug_set(meao_ctx *ctx, int32_t key, int32_t value)
{
    if (!ctx) return MEAO_ERR_INVALID_ARGUMENT;
    switch (key) {
    case MEAO_DEBUG_FUSE_COARSE_BLEND: ctx->fuse_coarse_blend = value != 0; break;
    case MEAO_DEBUG_NESTED_MAX_TILES: ctx->nested_max_tiles = value; break;
    case MEAO_DEBUG_RENDER_SMALL_MAX_TILES: ctx->render_small_max_tiles = value; break;
    case MEAO_DEBUG_FINAL_SMALL_MAX_TILES: ctx->final_small_max_tiles = value; break;
    case MEAO_DEBUG_DS_SMALL_MAX_TILES: ctx->ds_small_max_tiles = value; break;
    case MEAO_DEBUG_NEXT_DOWNSAMPLE_OWN_LAUNCH: ctx->next_ds_own_launch = value != 0; break;
    case MEAO_DEBUG_PROFILE_PASS_MASK: ctx->profile_mask = value == 0 ? ~0u : static_cast<uint32_t>(value); break;
    case MEAO_DEBUG_BLEND_TALL_MIN_TILES:
        ctx->blend_tall_min_tiles = value <= 0 ? 0x7fffffff : value;
        ctx->blend_tall_forced = true;
        break;
    default: return fail(ctx, MEAO_ERR_INVALID_ARGUMENT, "meao_debug_set: unknown key");
    }
    return MEAO_OK;
}

#if MEAO_TESTING
// Fault injection for the resize / first-announcement error paths (tests/test_gpu_more.py): the next n allocations of
// intermediates fail with MEAO_ERR_OUT_OF_MEMORY.  Not declared in include/meao.h and not compiled into libmeao_hip.so:
// only the `testhooks` variant library (miniengineao_amd/build.py VARIANTS, -DMEAO_TESTING=1) exports it.
__attribute__((visibility("default"))) int32_t meao_test_fail_next_allocs(meao_ctx *ctx, int32_t n)
{
    if (!ctx) return MEAO_ERR_INVALID_ARGUMENT;
    ctx->debug_fail_allocs = n < 0 ? 0 : n;
    return MEAO_OK;
}
#endif

int32_t meao_selftest(meao_ctx *ctx, int32_t which, uint64_t *out_mismatches)
{
    if (!ctx || !out_mismatches || which < 0 || which > 7) return MEAO_ERR_INVALID_ARGUMENT;
    int rc = use_device(ctx);
    if (rc != MEAO_OK) return rc;
    if (!ctx->counter) MEAO_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->counter), sizeof(unsigned long long)));
    MEAO_HIP(ctx, hipMemsetAsync(ctx->counter, 0, sizeof(unsigned long long), ctx->own_stream));
    MEAO_HIP(ctx, launch_selftest(which, ctx->counter, ctx->own_stream));
    unsigned long long host = 0;
    MEAO_HIP(ctx, hipMemcpyAsync(&host, ctx->counter, sizeof host, hipMemcpyDeviceToHost, ctx->own_stream));
    MEAO_HIP(ctx, hipStreamSynchronize(ctx->own_stream));
    *out_mismatches = host;
    return MEAO_OK;
}

}  // extern "C"
