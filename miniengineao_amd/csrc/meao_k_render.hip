// meao_k_render.hip -- render kernels: interleaved (all levels, one grid), small tiles, wide (Render.main), and the form that carries a composite.
#include "meao_dev_render.hpp"
#include "meao_dev_composite.hpp"

namespace meao {
namespace {

// 128 x 32 tiles: 40 KB window, 4 workgroups of 8 waves per CU = 8 waves per SIMD (<= 64 VGPRs).
template <int AOFMT, bool RTNE, int DIV, bool EXH>
__global__ __launch_bounds__(ren_tile_w(EXH) * 4, EXH ? 1 : 8) void render_kernel(const RenderArgs a)
{
    __shared__ __attribute__((aligned(16))) float tile[kRenLdsH * (ren_tile_w(EXH) + 2 * kRenApron)];
#if MEAO_X_PHASE_CLOCKS
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();      // every wave: the first one in and the last one out are logged
#endif
    const int frame = blockIdx.y, block = xcd_contiguous(blockIdx.x, gridDim.x);
    if constexpr (DIV == DIV_EXACT_RCP) {
        if (frame_is_hostile(a.hostile, a.generation, frame)) {       // wave-uniform, decided per frame
            render_tile<AOFMT, RTNE, DIV_IEEE, EXH>(a, tile, frame, block);
            return;
        }
    }
    render_tile<AOFMT, RTNE, DIV, EXH>(a, tile, frame, block);
#if MEAO_X_PHASE_CLOCKS
    const unsigned id = blockIdx.y * gridDim.x + blockIdx.x;
    if ((threadIdx.x & 63) == 0 && id < 16384) {
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        atomicMin(&g_wg_log[id * 4 + 0], wg_t0);           // earliest wave start (the host presets ~0)
        atomicMax(&g_wg_log[id * 4 + 1], t1);              // latest wave end
        atomicMin(&g_wg_log[id * 4 + 2], t1);              // earliest wave end: the skew inside the workgroup
        if (threadIdx.x == 0)
            g_wg_log[id * 4 + 3] = (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) & 0xFu) |
                                   (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11))) << 8);
    }
#endif
}

// (128 x 64 tiles with 1024 threads -- a 60 KB window, two workgroups = 32 waves per CU, apron share 1.875x instead of 2.5x, half
// the hand-overs per texel -- measured 170.2 vs 166.2 us per 16 frames at 4K, profiles/r04_ab_render_tile_128x64.jsonl: the
// barrier of sixteen waves and a hand-over that idles half a CU cost more than the smaller apron saves.  With 96 x 32 (r03),
// 64 x 32 (r01) and the dynamic blocks (r03) that closes the tile-shape question: render runs at 2.98 cycles per VALU
// instruction, the hand-over of a full CU's LDS is what separates it from the 2.4-2.55 of its loop, and no shape removes it.
// Nor does taking the hand-over away: persistent 1024-thread workgroups whose four loader waves fill the next tile's window while
// twelve compute waves evaluate the current one (two window buffers, one barrier per tile) run at 209-220 us -- the barrier of
// sixteen waves per tile costs more than the hand-over did: profiles/r04_ab_render_producer_consumer.jsonl.)
// One or two small frames per call (fewer 128 x 32 tiles than CUs): 128 x 8 tiles, four times the workgroups,
// one texel-loop iteration each -- the call waits for one workgroup's serial time, not for throughput.
template <int AOFMT, bool RTNE, int DIV>
__global__ __launch_bounds__(ren_tile_w(false) * 4, 6) void render_small_kernel(const RenderArgs a)
{
    __shared__ __attribute__((aligned(16))) float tile[(kRenTileHSmall + 2 * kRenApron) * (ren_tile_w(false) + 2 * kRenApron)];
    const int frame = blockIdx.y, block = xcd_contiguous(blockIdx.x, gridDim.x);
    if constexpr (DIV == DIV_EXACT_RCP) {
        if (frame_is_hostile(a.hostile, a.generation, frame)) {
            render_tile<AOFMT, RTNE, DIV_IEEE, false, NoRenderHook, kRenTileHSmall>(a, tile, frame, block);
            return;
        }
    }
    render_tile<AOFMT, RTNE, DIV, false, NoRenderHook, kRenTileHSmall>(a, tile, frame, block);
}

// ------------------------------------------------------------------------------------------
// Render.main (WIDE_SAMPLING, REN:22,27-29,46-50): the same estimator on the NON-tiled f32
// LowDepth<level>, sampling every other texel (offsets doubled, REN:79-82) out to 8 texels, one
// output texel per source texel (REN:174).  The reference's host never dispatches it; it is
// the "high quality" pass of the MiniEngine original and feeds Upsample.main_premin*.
// Tile 64 x 32 outputs, LDS window (64+16) x (32+16) of raw f32 depth with clamp addressing
// (REN:116,121 Gather on the 2D texture); no f16 round trip, no padding texels.
template <int AOFMT, bool RTNE, int DIV, bool EXH>
__device__ __forceinline__ void render_wide_tile(const RenderArgs &a, float *tile, int frame, int block)
{
    typedef AoTexel<AOFMT> AO;

    int b = block, lv = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < a.num_levels && b >= a.level[k].block_begin) lv = k;
    const RenderLevelArgs &L = a.level[lv];
    b -= L.block_begin;
    const int X0 = (b % L.tiles_x) * kWideTileW, Y0 = (b / L.tiles_x) * kRenTileH;
    const int lw = L.lw, lh = L.lh;
    const float *__restrict__ src = frame_ptr(L.src, a.frame_stride, frame);

    for (int i = threadIdx.x; i < kWideLdsW * kWideLdsH; i += kThreads) {
        const int c = i % kWideLdsW, r = i / kWideLdsW;
        const int x = clampi(X0 - kWideApron + c, 0, lw - 1), y = clampi(Y0 - kWideApron + r, 0, lh - 1);
        tile[i] = src[static_cast<size_t>(y) * lw + x];
    }
    __syncthreads();

    const int txl = threadIdx.x & 31, tyl = threadIdx.x >> 5;
    const int X = X0 + 2 * txl;
    if (X >= lw) return;
    typename AO::type *__restrict__ dst = frame_ptr(static_cast<typename AO::type *>(L.dst), a.frame_stride, frame);
    const bool pair_store = ((lw & 1) == 0);
    const TermConstants<EXH> terms(L);

#pragma unroll 1
    for (int k = 0; k < kRenTileH / 8; ++k) {
        const int ly = tyl + 8 * k, Y = Y0 + ly;
        if (Y >= lh) break;
        const float *centre = &tile[(ly + kWideApron) * kWideLdsW + 2 * txl + kWideApron];
        const float2v c = *reinterpret_cast<const float2v *>(centre);
        const float2v inv_depth = float2v{rcp_strict<DIV>(c.x), rcp_strict<DIV>(c.y)};   // REN:140
        const float2v out = accumulate_terms<EXH, 2 * kWideLdsW, 2>(terms, centre, inv_depth);

        typename AO::type *p = dst + static_cast<size_t>(Y) * lw + X;
        const typename AO::type e0 = AO::template encode<RTNE>(out.x), e1 = AO::template encode<RTNE>(out.y);
        if (pair_store) {
            typename AO::type2 pr; pr.x = e0; pr.y = e1;
            *reinterpret_cast<typename AO::type2 *>(p) = pr;
        } else {
            p[0] = e0;
            if (X + 1 < lw) p[1] = e1;
        }
    }
}

template <int AOFMT, bool RTNE, int DIV, bool EXH>
__global__ __launch_bounds__(kThreads) void render_wide_kernel(const RenderArgs a)
{
    __shared__ __attribute__((aligned(16))) float tile[kWideLdsH * kWideLdsW];
    const int frame = blockIdx.y, block = xcd_contiguous(blockIdx.x, gridDim.x);
    if constexpr (DIV == DIV_EXACT_RCP) {
        if (frame_is_hostile(a.hostile, a.generation, frame)) {
            render_wide_tile<AOFMT, RTNE, DIV_IEEE, EXH>(a, tile, frame, block);
            return;
        }
    }
    render_wide_tile<AOFMT, RTNE, DIV, EXH>(a, tile, frame, block);
}

// The render pass carrying the composite of frames that an EARLIER call produced (meao_composite_enqueue):
// the composite is pure streaming (17 bytes per texel, as many bytes as the whole AO path) and render
// is VALU-bound with HBM nearly idle, so every render workgroup first streams its share of the
// composite texel pairs and then renders its tile.
// carried composite (multiply mode): two pixel pairs per lane in flight under every texel-loop iteration (three: 0.830 vs 0.834 ms, not kept)
constexpr int kCompositePerIteration = 2;
constexpr int kCompositePairsInLoop = kCompositePerIteration * (kRenTileH / 8);

// Pass 2 of Blit.shader (dst * src.a) for pixel pairs of ONE frame, as the hook of the render texel loop:
// begin(k) issues the 16-byte colour and 2/4-byte AO loads of two pairs, end(k) multiplies and stores them.
// Pair j of a lane is q = (j * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x (a workgroup touches 8 KB
// of contiguous colour per j); j < kCompositePairsInLoop here, the rest in the plain loop before the tile.
template <int AOFMT>
struct CarriedComposite {
    typedef AoTexel<AOFMT> AO;
    const typename AO::type *ao;
    uint16_t *color;
    uint32_t q0, q_step, full_pairs;          // q0 = pair of j = 0; pairs below full_pairs have both pixels
    bool active;
    uint4v col[kCompositePerIteration];
    typedef typename std::conditional<sizeof(typename AO::type) == 1, uint16_t, uint32_t>::type ao_pair_bits;
    uint32_t ao2[kCompositePerIteration];     // two AO texels, undecoded (taken apart in end(), not next to the load)
    __device__ __forceinline__ uint32_t pair_of(int k, int s) const { return q0 + static_cast<uint32_t>(kCompositePerIteration * k + s) * q_step; }
    __device__ __forceinline__ void begin(int k)
    {
        if (!active) return;
#pragma unroll
        for (int s = 0; s < kCompositePerIteration; ++s) {
            const uint32_t q = pair_of(k, s);
            if (q < full_pairs) {
                col[s] = __builtin_nontemporal_load(reinterpret_cast<const uint4v *>(at_byte_offset(color, q * 16u)));
                ao2[s] = *reinterpret_cast<const ao_pair_bits *>(at_byte_offset(ao, q * static_cast<uint32_t>(sizeof(ao_pair_bits))));
            }
        }
        __builtin_amdgcn_sched_barrier(0);        // the loads stay here; their first use is behind the texel arithmetic
    }
    __device__ __forceinline__ void end(int k)
    {
        if (!active) return;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < kCompositePerIteration; ++s) {
            const uint32_t q = pair_of(k, s);
            if (q < full_pairs) {
                constexpr int kAoBits = 8 * sizeof(typename AO::type);
                asm volatile("" : "+v"(ao2[s]));          // opaque here: nothing derived from the loaded word moves up to the load
                const float a0 = AO::decode(static_cast<typename AO::type>(ao2[s] & ((1u << kAoBits) - 1u)));
                const float a1 = AO::decode(static_cast<typename AO::type>(ao2[s] >> kAoBits));
                const uint32_t w[4] = {col[s].x, col[s].y, col[s].z, col[s].w};
                uint32_t o[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {                      // words 0, 1: pixel 0 (rg, ba); words 2, 3: pixel 1
                    const float m = h < 2 ? a0 : a1;
                    const uint32_t lo = f32_to_f16_rtne_bits(f16_bits_to_f32(static_cast<uint16_t>(w[h] & 0xffffu)) * m);
                    const uint32_t hi = f32_to_f16_rtne_bits(f16_bits_to_f32(static_cast<uint16_t>(w[h] >> 16)) * m);
                    o[h] = lo | (hi << 16);
                }
                __builtin_nontemporal_store(uint4v{o[0], o[1], o[2], o[3]}, reinterpret_cast<uint4v *>(at_byte_offset(color, q * 16u)));
            }
        }
    }
};

template <int AOFMT, bool RTNE, int DIV>
__global__ __launch_bounds__(ren_tile_w(false) * 4, 8) void render_with_composite_kernel(const RenderArgs a,
                                                                                         const CompositeBatchArgs c)
{
    __shared__ __attribute__((aligned(16))) float tile[kRenLdsH * (ren_tile_w(false) + 2 * kRenApron)];
    const int frame = blockIdx.y, block = xcd_contiguous(blockIdx.x, gridDim.x);
    // In-loop form: one composite frame per render frame, multiply mode, frames below 2^28 pairs (32-bit byte offsets)
    const bool in_loop = c.mode == MEAO_COMPOSITE_MULTIPLY && c.frames == static_cast<int32_t>(gridDim.y) &&
                         c.pixels < (int64_t(1) << 29);
    CarriedComposite<AOFMT> carried;
    carried.active = in_loop;
    if (in_loop) {
        const int64_t pairs = (c.pixels + 1) / 2;
        carried.ao = static_cast<const typename AoTexel<AOFMT>::type *>(c.ao[frame]);
        carried.color = static_cast<uint16_t *>(c.color[frame]);
        carried.q_step = gridDim.x * blockDim.x;
        carried.q0 = blockIdx.x * blockDim.x + threadIdx.x;
        carried.full_pairs = static_cast<uint32_t>(c.pixels / 2);
        // what the loop does not take: pairs j >= kCompositePairsInLoop of this lane and the half pair of an odd frame
        for (int64_t q = static_cast<int64_t>(carried.q0) + static_cast<int64_t>(kCompositePairsInLoop) * carried.q_step; q < pairs; q += carried.q_step)
            composite_pair<AOFMT>(c.ao[frame], c.color[frame], c.gbuffer0[frame], c.pixels, c.mode, q);
        if (c.pixels & 1) {     // the half pair at the end of an odd frame: the lane that owns it, if the loop would have had it
            const int64_t last = pairs - 1;
            if (last % carried.q_step == carried.q0 && last / carried.q_step < kCompositePairsInLoop)
                composite_pair<AOFMT>(c.ao[frame], c.color[frame], c.gbuffer0[frame], c.pixels, c.mode, last);
        }
    } else {
        const int64_t pairs = (c.pixels + 1) / 2, total = pairs * c.frames;
        const int64_t stride = static_cast<int64_t>(gridDim.x) * gridDim.y * blockDim.x;
        for (int64_t i = (static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
            const int f = static_cast<int>(i / pairs);
            composite_pair<AOFMT>(c.ao[f], c.color[f], c.gbuffer0[f], c.pixels, c.mode, i - f * pairs);
        }
    }
    if constexpr (DIV == DIV_EXACT_RCP) {
        if (frame_is_hostile(a.hostile, a.generation, frame)) {
            render_tile<AOFMT, RTNE, DIV_IEEE, false>(a, tile, frame, block, carried);
            return;
        }
    }
    render_tile<AOFMT, RTNE, DIV, false>(a, tile, frame, block, carried);
}


}  // namespace

// ------------------------------------------------------------------------------------------
// launchers

// WIDE selects render_wide_kernel; the (AOFMT, RTNE, DIV, EXH) choice is the same for both.
template <bool WIDE, int AOFMT, bool RTNE, int DIV>
static void launch_render_t(const RenderArgs &a, dim3 grid, hipStream_t s)
{
    const dim3 block(WIDE ? kThreads : ren_tile_w(a.exhaustive != 0) * 4);
    if constexpr (WIDE) {
        if (a.exhaustive) render_wide_kernel<AOFMT, RTNE, DIV, true><<<grid, block, 0, s>>>(a);
        else render_wide_kernel<AOFMT, RTNE, DIV, false><<<grid, block, 0, s>>>(a);
    } else {
        if (a.exhaustive) render_kernel<AOFMT, RTNE, DIV, true><<<grid, block, 0, s>>>(a);
        else if (a.tile_h == kRenTileHSmall) render_small_kernel<AOFMT, RTNE, DIV><<<grid, block, 0, s>>>(a);
        else render_kernel<AOFMT, RTNE, DIV, false><<<grid, block, 0, s>>>(a);
    }
}

template <bool WIDE>
static hipError_t launch_render_any(const RenderArgs &a, int ao_format, int frames, hipStream_t s)
{
    const dim3 grid(a.blocks_per_frame, frames, 1);
    if (ao_format == MEAO_AO_R8) {
        if (a.f16_rtne) launch_render_t<WIDE, MEAO_AO_R8, true, DIV_IEEE>(a, grid, s);
        else if (a.exact_rcp_div) launch_render_t<WIDE, MEAO_AO_R8, false, DIV_EXACT_RCP>(a, grid, s);
        else launch_render_t<WIDE, MEAO_AO_R8, false, DIV_IEEE>(a, grid, s);
    } else {
        if (a.f16_rtne) launch_render_t<WIDE, MEAO_AO_F16, true, DIV_IEEE>(a, grid, s);
        else if (a.exact_rcp_div) launch_render_t<WIDE, MEAO_AO_F16, false, DIV_EXACT_RCP>(a, grid, s);
        else launch_render_t<WIDE, MEAO_AO_F16, false, DIV_IEEE>(a, grid, s);
    }
    return hipGetLastError();
}

hipError_t launch_render(const RenderArgs &a, int ao_format, int frames, hipStream_t s)
{
    return launch_render_any<false>(a, ao_format, frames, s);
}

template <int AOFMT, bool RTNE, int DIV>
static void launch_render_composite_t(const RenderArgs &a, const CompositeBatchArgs &c, dim3 grid, hipStream_t s)
{
    render_with_composite_kernel<AOFMT, RTNE, DIV><<<grid, dim3(ren_tile_w(false) * 4), 0, s>>>(a, c);
}

hipError_t launch_render_with_composite(const RenderArgs &a, const CompositeBatchArgs &c, int ao_format, int frames, hipStream_t s)
{
    if (a.exhaustive) return hipErrorInvalidValue;     // the 68-sample variant keeps its own launch; the caller flushes instead
    const dim3 grid(a.blocks_per_frame, frames, 1);
    if (ao_format == MEAO_AO_R8) {
        if (a.f16_rtne) launch_render_composite_t<MEAO_AO_R8, true, DIV_IEEE>(a, c, grid, s);
        else if (a.exact_rcp_div) launch_render_composite_t<MEAO_AO_R8, false, DIV_EXACT_RCP>(a, c, grid, s);
        else launch_render_composite_t<MEAO_AO_R8, false, DIV_IEEE>(a, c, grid, s);
    } else {
        if (a.f16_rtne) launch_render_composite_t<MEAO_AO_F16, true, DIV_IEEE>(a, c, grid, s);
        else if (a.exact_rcp_div) launch_render_composite_t<MEAO_AO_F16, false, DIV_EXACT_RCP>(a, c, grid, s);
        else launch_render_composite_t<MEAO_AO_F16, false, DIV_IEEE>(a, c, grid, s);
    }
    return hipGetLastError();
}

hipError_t launch_render_wide(const RenderArgs &a, int ao_format, int frames, hipStream_t s)
{
    return launch_render_any<true>(a, ao_format, frames, s);
}


}  // namespace meao
