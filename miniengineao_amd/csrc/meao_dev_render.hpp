// meao_dev_render.hpp -- the interleaved render tile (Render.main_interleaved): sample arithmetic, pipelined LDS reads, window fill, texel loop.
#pragma once

#include "meao_dev.hpp"

namespace meao {
namespace {

// ------------------------------------------------------------------------------------------
// Render: volumetric-obscurance AO, 36-sample checker set.

// TestSamplePair (REN:60-75) for one output texel, from the two signed distances d = s * invRange - front.
// saturate() folds into the clamp output modifier of v_mul/v_fma; clamp(d, p, 1) with
// 0 <= p <= 1 is v_med3_f32(d, p, 1) (same value for every input incl. NaN d -> p).
__device__ __forceinline__ float pair_from_distances(float d1, float d2, float reject)
{
    const float p1 = sat(reject * d1);
    const float p2 = sat(reject * d2);
    const float acc = __builtin_amdgcn_fmed3f(d1, p2, 1.0f) + __builtin_amdgcn_fmed3f(d2, p1, 1.0f);
    return sat(mad(-p1, p2, acc));
}

// TestSamples (REN:77-110) WITHOUT its leading 0.5 / 0.25: that exact power-of-two factor is folded
// into the term's weight on the host (RenderLevelArgs::weight), since fma(w, k*S, ao) and
// fma(k*w, S, ao) round the same real number.  (X, Y) are sample offsets in source texels; the LDS
// offset of (dx, dy) is dy*P + dx*Q.  Interleaved: one slice texel is 4 level texels (4x4 interleave),
// P = 4*pitch, Q = 4.  Wide (REN:79-82, x <<= 1): P = 2*pitch, Q = 2.
// Two horizontally adjacent texels share every LDS address: one 8-byte LDS read per sample.
//
// (A wave-uniform "all distances >= 0 => pair = saturate(d1 + d2)" fast path was built, is bit-exact and was
// measured slower on both headline workloads -- DESIGN.md 5.2, profiles/r02_render_fastpath_hitrates.txt; removed.)
template <int X, int Y, int P, int Q>
__device__ __forceinline__ float2v test_samples(const float *centre, float2v inv_depth, float inv_thickness,
                                                float front_depth, float reject)
{
    constexpr int N = (Y == 0 || X == Y) ? 2 : 4;
    constexpr int off[4] = {Y == 0 ? X * Q : (X == Y ? X * P - X * Q : Y * P + X * Q),
                            Y == 0 ? X * P : (X == Y ? X * P + X * Q : Y * P - X * Q),
                            X * P + Y * Q, X * P - Y * Q};
    const float2v inv_range = splat(inv_thickness) * inv_depth;
    const float neg_front = -front_depth;
    float2v d1[N], d2[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float2v s1 = *reinterpret_cast<const float2v *>(centre + off[i]);
        const float2v s2 = *reinterpret_cast<const float2v *>(centre - off[i]);
        d1[i] = float2v{mad(s1.x, inv_range.x, neg_front), mad(s1.y, inv_range.y, neg_front)};
        d2[i] = float2v{mad(s2.x, inv_range.x, neg_front), mad(s2.y, inv_range.y, neg_front)};
    }
    float2v r[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
        r[i] = float2v{pair_from_distances(d1[i].x, d2[i].x, reject), pair_from_distances(d1[i].y, d2[i].y, reject)};
    if constexpr (N == 2) return r[0] + r[1];
    else return ((r[0] + r[1]) + r[2]) + r[3];
}

// ao = sum over the terms of weight * TestSamples, in the reference's accumulation order:
// checker set REN:162-168 (slots 1,3,4,8,11,6,10), SAMPLE_EXHAUSTIVELY REN:146-157
// (slots 0,1,2,3,4,8,11,5,6,7,9,10).  L.weight[] etc. are already in term order; L.weight[] carries
// the 0.5 (axial, diagonal) / 0.25 (L-shaped) factor of TestSamples.
// The per-term constants of one level, held in SGPRs for the whole tile.  (Read straight from the
// kernel-argument struct the compiler re-issued the s_load_dword's inside the texel loop, three per
// term, and their s_waitcnt lgkmcnt(0) also drained the LDS reads in flight.)
template <bool EXH>
struct TermConstants {
    static constexpr int kTerms = EXH ? 12 : 7;
    float inv_thickness[kTerms], front_depth[kTerms], weight[kTerms];
    float reject_fadeoff, intensity;
    // All scalar loads first, then ONE statement that pins the values (a volatile asm per term made the compiler wait for each
    // term's loads before it issued the next ones: eight dependent scalar-memory round trips per workgroup, right behind the
    // window barrier).  render_tile calls this once its window loads are in flight, so the scalar loads' latency hides behind theirs.
    __device__ __forceinline__ TermConstants() {}
    __device__ __forceinline__ explicit TermConstants(const RenderLevelArgs &src) { load(src); }
    __device__ __forceinline__ void load(const RenderLevelArgs &src)
    {
#pragma unroll
        for (int t = 0; t < kTerms; ++t) {
            inv_thickness[t] = src.inv_thickness[t];
            front_depth[t] = src.front_depth[t];
            weight[t] = src.weight[t];
        }
        reject_fadeoff = src.reject_fadeoff;
        intensity = src.intensity;
#define MEAO_PIN3(T) "+s"(inv_thickness[T]), "+s"(front_depth[T]), "+s"(weight[T])
        asm volatile("" : MEAO_PIN3(0), MEAO_PIN3(1), MEAO_PIN3(2), MEAO_PIN3(3), MEAO_PIN3(4), MEAO_PIN3(5), MEAO_PIN3(6),
                          "+s"(reject_fadeoff), "+s"(intensity));                       // stay in SGPRs
        if constexpr (EXH) asm volatile("" : MEAO_PIN3(7), MEAO_PIN3(8), MEAO_PIN3(9), MEAO_PIN3(10), MEAO_PIN3(11));
#undef MEAO_PIN3
    }
};

template <bool EXH, int P, int Q>
__device__ __forceinline__ float2v accumulate_terms(const TermConstants<EXH> &L, const float *centre, float2v inv_depth)
{
    const float reject = L.reject_fadeoff;
    float2v ao = splat(0.0f);
#define MEAO_TERM(N, X, Y) \
    ao = fma2(splat(L.weight[N]), test_samples<X, Y, P, Q>(centre, inv_depth, L.inv_thickness[N], L.front_depth[N], reject), ao)
    if constexpr (EXH) {
        MEAO_TERM(0, 1, 0); MEAO_TERM(1, 2, 0); MEAO_TERM(2, 3, 0); MEAO_TERM(3, 4, 0);
        MEAO_TERM(4, 1, 1); MEAO_TERM(5, 2, 2); MEAO_TERM(6, 3, 3); MEAO_TERM(7, 1, 2);
        MEAO_TERM(8, 1, 3); MEAO_TERM(9, 1, 4); MEAO_TERM(10, 2, 3); MEAO_TERM(11, 2, 4);
    } else {
        MEAO_TERM(0, 2, 0); MEAO_TERM(1, 4, 0); MEAO_TERM(2, 1, 1); MEAO_TERM(3, 2, 2);
        MEAO_TERM(4, 3, 3); MEAO_TERM(5, 1, 3); MEAO_TERM(6, 2, 4);
    }
#undef MEAO_TERM
    return fma2(splat(L.intensity), ao - splat(1.0f), splat(1.0f));   // lerp(1, ao, gIntensity) REN:176
}

// ---- the same sum with the LDS reads pipelined by hand (checker set) --------------------------
// clang issues the ds_read's of a term right before their first use (s_waitcnt a few instructions
// later): every wave exposes the LDS latency 12+ times per texel pair.  Here the 18 sample pairs of
// the checker set are one flat sequence; the two 8-byte reads of pair k + DEPTH are issued before pair k
// is evaluated, as separate ds_read_b64 (the merged ds_read2_b64 form runs at half the LDS rate,
// tools/ubench_lds.hip).  The reads are inline asm, so the waits are too: LDS operations return in
// order, `s_waitcnt lgkmcnt(2 * DEPTH)` therefore means "pair k has arrived" whatever else is in flight
// behind it.  Arithmetic and its order are those of test_samples / accumulate_terms.
struct SamplePair { float2v s1, s2; };

constexpr int kCheckerTerms[7][2] = {{2, 0}, {4, 0}, {1, 1}, {2, 2}, {3, 3}, {1, 3}, {2, 4}};   // REN:162-168
constexpr int kCheckerPairs = 18;

constexpr int checker_pairs_in_term(int t) { return (kCheckerTerms[t][1] == 0 || kCheckerTerms[t][0] == kCheckerTerms[t][1]) ? 2 : 4; }
constexpr int checker_term_of_pair(int k)
{
    int t = 0;
    while (k >= checker_pairs_in_term(t)) { k -= checker_pairs_in_term(t); ++t; }
    return t;
}
constexpr int checker_index_in_term(int k)
{
    int t = 0;
    while (k >= checker_pairs_in_term(t)) { k -= checker_pairs_in_term(t); ++t; }
    return k;
}
// LDS offset (floats) of the first sample of pair i of term (X, Y); the second one is at minus that
constexpr int checker_pair_offset(int X, int Y, int P, int Q, int i)
{
    return i == 0 ? (Y == 0 ? X * Q : (X == Y ? X * P - X * Q : Y * P + X * Q))
         : i == 1 ? (Y == 0 ? X * P : (X == Y ? X * P + X * Q : Y * P - X * Q))
         : i == 2 ? X * P + Y * Q : X * P - Y * Q;
}

template <int BYTE_OFF>
__device__ __forceinline__ void lds_read_b64_async(uint32_t lds_addr, float2v &v)
{
    static_assert(BYTE_OFF >= 0 && BYTE_OFF < 65536 && BYTE_OFF % 8 == 0, "ds_read_b64 immediate offset");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(BYTE_OFF));
}

template <int K, int P, int Q>
__device__ __forceinline__ void issue_checker_pair(uint32_t base, SamplePair &into)
{
    if constexpr (K < kCheckerPairs) {
        constexpr int t = checker_term_of_pair(K);
        constexpr int off = checker_pair_offset(kCheckerTerms[t][0], kCheckerTerms[t][1], P, Q, checker_index_in_term(K));
        constexpr int centre_at = 4 * P + 4 * Q;                 // `base` is that many floats before the centre texel
        lds_read_b64_async<(centre_at + off) * 4>(base, into.s1);
        lds_read_b64_async<(centre_at - off) * 4>(base, into.s2);
    }
}

// Waits until at most PENDING LDS reads are outstanding.  The operands tie the wait into the data flow:
// the arrived pair is only readable after it, and the running sums (= the previous pair's arithmetic)
// are complete before it, so the schedule keeps one pair's arithmetic between two waits.
template <int PENDING>
__device__ __forceinline__ void wait_checker_pair(SamplePair &arrived, float2v &term_sum, float2v &ao)
{
    static_assert(PENDING == 0 || PENDING == 2 || PENDING == 4, "");
    if constexpr (PENDING == 0)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(arrived.s1), "+v"(arrived.s2), "+v"(term_sum), "+v"(ao));
    else if constexpr (PENDING == 2)
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(arrived.s1), "+v"(arrived.s2), "+v"(term_sum), "+v"(ao));
    else
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(arrived.s1), "+v"(arrived.s2), "+v"(term_sum), "+v"(ao));
}

template <int K, int P, int Q, int DEPTH>
__device__ __forceinline__ void pipelined_checker_step(const TermConstants<false> &L, uint32_t base, float2v inv_depth, float reject,
                                                       SamplePair (&ring)[DEPTH + 1], float2v &inv_range, float &neg_front,
                                                       float2v &term_sum, float2v &ao)
{
    if constexpr (K < kCheckerPairs) {
        constexpr int t = checker_term_of_pair(K), i = checker_index_in_term(K), n = checker_pairs_in_term(t);
        issue_checker_pair<K + DEPTH, P, Q>(base, ring[(K + DEPTH) % (DEPTH + 1)]);
        constexpr int behind = (K + DEPTH < kCheckerPairs ? DEPTH : kCheckerPairs - 1 - K);    // pairs issued after pair K
        SamplePair &s = ring[K % (DEPTH + 1)];
        wait_checker_pair<2 * behind>(s, term_sum, ao);
        if constexpr (i == 0) {
            inv_range = splat(L.inv_thickness[t]) * inv_depth;
            neg_front = -L.front_depth[t];
            asm volatile("" : "+v"(neg_front));      // VGPR operand: an SGPR source halves the VALU issue rate (tools/ubench_issue.hip)
        }
        const float2v d1 = float2v{mad(s.s1.x, inv_range.x, neg_front), mad(s.s1.y, inv_range.y, neg_front)};
        const float2v d2 = float2v{mad(s.s2.x, inv_range.x, neg_front), mad(s.s2.y, inv_range.y, neg_front)};
        const float2v r = float2v{pair_from_distances(d1.x, d2.x, reject), pair_from_distances(d1.y, d2.y, reject)};
        if constexpr (i == 0) term_sum = r;
        else term_sum = term_sum + r;                                       // (r0 + r1) (+ r2) (+ r3), REN:92-109
        if constexpr (i == n - 1) ao = fma2(splat(L.weight[t]), term_sum, ao);
        pipelined_checker_step<K + 1, P, Q, DEPTH>(L, base, inv_depth, reject, ring, inv_range, neg_front, term_sum, ao);
    }
}

template <int P, int Q, int DEPTH>
__device__ __forceinline__ float2v accumulate_terms_pipelined(const TermConstants<false> &L, const float *centre, float2v inv_depth)
{
    typedef __attribute__((address_space(3))) const float lds_float;
    const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_float *)(centre - (4 * P + 4 * Q))));
    SamplePair ring[DEPTH + 1];
    float2v ao = splat(0.0f), term_sum = splat(0.0f), inv_range = splat(0.0f);
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
        if (k == 0) issue_checker_pair<0, P, Q>(base, ring[0]);
        if (k == 1) issue_checker_pair<1, P, Q>(base, ring[1]);
    }
    float reject = L.reject_fadeoff, neg_front = 0.0f;
    asm volatile("" : "+v"(reject));
    pipelined_checker_step<0, P, Q, DEPTH>(L, base, inv_depth, reject, ring, inv_range, neg_front, term_sum, ao);
    return fma2(splat(L.intensity), ao - splat(1.0f), splat(1.0f));   // lerp(1, ao, gIntensity) REN:176
}

// Hook of the texel loop: begin(k) / end(k) are executed by every thread of the workgroup around iteration k
// (render_with_composite_kernel puts the loads of unrelated streaming work in flight under the arithmetic).
struct NoRenderHook {
    __device__ __forceinline__ void begin(int) {}
    __device__ __forceinline__ void end(int) {}
};

template <int AOFMT, bool RTNE, int DIV, bool EXH, typename Hook = NoRenderHook, int TILE_H = kRenTileH, int THREADS = ren_tile_w(EXH) * 4>
__device__ __forceinline__ void render_tile(const RenderArgs &a, float *tile, int frame, int block, Hook hook = Hook())
{
    typedef AoTexel<AOFMT> AO;
    constexpr int kRenTileW = ren_tile_w(EXH), kRenThreads = THREADS, kRenLdsW = kRenTileW + 2 * kRenApron;
    constexpr int kRenTileH = TILE_H, kRenLdsH = TILE_H + 2 * kRenApron;      // shadow the 32-row constants

    int b = block, lv = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < a.num_levels && b >= a.level[k].block_begin) lv = k;
    const RenderLevelArgs &L = a.level[lv];
    b -= L.block_begin;
    const int X0 = (b % L.tiles_x) * kRenTileW, Y0 = (b / L.tiles_x) * kRenTileH;
    const int lw = L.lw, lh = L.lh;
    const float *__restrict__ src = frame_ptr(L.src, a.frame_stride, frame);

    PhaseClock clk(24);      // 24: window loaded, converted, in LDS; 25: barrier; 26..29: texel-loop iterations
    __builtin_amdgcn_s_setprio(3);
    TermConstants<EXH> terms_storage;
    const TermConstants<EXH> &terms = terms_storage;
    typename AO::type *__restrict__ dst;
    // ---- stage the (64+32) x (32+32) window.  Window texel (vx,vy) (level coordinates, may
    // be outside the level) belongs to slice (vx&3, vy&3), slice texel (vx>>2, vy>>2); the
    // reference clamps the slice texel per slice (REN:118 Gather + clamp sampler) and finds
    // Linearize(out-of-range) / 0 in atlas texels beyond the level (DS1:39-46, DS2:35).
    {
        const float pad = through_f16<RTNE>(L.pad_value);
        const bool vec_ok = (lw & 3) == 0;
        constexpr int kQuadsX = kRenLdsW / 4, kQuads = kQuadsX * kRenLdsH, kRounds = (kQuads + kRenThreads - 1) / kRenThreads;
        constexpr bool kEven = kQuads % kRenThreads == 0;            // every thread fills the same number of quads (32-row tiles)
        // Phase 1: all 16-byte loads of this thread's quads are issued back to back (the plain loop
        // waited for each load before issuing the next: five dependent memory latencies per tile);
        // quads that touch the level's border take the scalar path in phase 2.
        float4v raw[kRounds];
        int row_at[kRounds];      // index of the first texel of the quad's row segment, or -1 = all padding
        bool whole[kRounds];
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int q = threadIdx.x + r * kRenThreads;
            const int qx = q % kQuadsX, qy = q / kQuadsX;
            const int px0 = clampi((X0 >> 2) - (kRenApron >> 2) + qx, 0, L.sw - 1) * 4;
            const int vy = Y0 - kRenApron + qy;
            const int py = clampi(vy >> 2, 0, L.sh - 1) * 4 + (vy & 3);
            const bool mine = kEven || q < kQuads;              // the last round of the 8-row tile is partly empty
            row_at[r] = (mine && py < lh) ? py * lw + px0 : -1;
            whole[r] = mine && py < lh && vec_ok && px0 + 3 < lw;
            if (whole[r]) raw[r] = *reinterpret_cast<const float4v *>(src + row_at[r]);
        }
        // (the texel loop's constants: fetched while the window loads are in flight, see TermConstants)
        terms_storage.load(L);
        dst = frame_ptr(static_cast<typename AO::type *>(L.dst), a.frame_stride, frame);
        asm volatile("" : "+s"(dst));
        // Phase 2: the f16 round trip the atlas store applies, then one 16-byte LDS store per quad
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int q = threadIdx.x + r * kRenThreads;
            const int qx = q % kQuadsX, qy = q / kQuadsX;
            float4v t = {pad, pad, pad, pad};
            if (whole[r]) {
                const float2v lo = through_f16_pair<RTNE>(raw[r].x, raw[r].y), hi = through_f16_pair<RTNE>(raw[r].z, raw[r].w);
                t = float4v{lo.x, lo.y, hi.x, hi.y};
            } else if (row_at[r] >= 0) {
                const float *row = src + row_at[r];
                const int px0 = row_at[r] % lw;
                if (px0 + 0 < lw) t.x = through_f16<RTNE>(row[0]);
                if (px0 + 1 < lw) t.y = through_f16<RTNE>(row[1]);
                if (px0 + 2 < lw) t.z = through_f16<RTNE>(row[2]);
                if (px0 + 3 < lw) t.w = through_f16<RTNE>(row[3]);
            }
            if (kEven || q < kQuads) *reinterpret_cast<float4v *>(&tile[qy * kRenLdsW + qx * 4]) = t;
        }
    }
    clk.mark(0);
    __syncthreads();
    clk.mark(1);
    __builtin_amdgcn_s_setprio(0);

    // ---- each lane: a texel pair (X, X+1) in each of the TILE_H / 8 iterations
    const bool pair_store = ((lw & 1) == 0);
    // a wave covers a compact 32 x 4 block (16 lanes x 4 rows) of the tile in each of the 4 iterations
    constexpr int kBlocksX = kRenTileW / 32, kWaves = kRenThreads / 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

    constexpr int kIterations = kBlocksX * (kRenTileH / 4) / kWaves;        // 32 x 4 blocks of the tile per wave
    static_assert(kIterations * kWaves == kBlocksX * (kRenTileH / 4), "the tile's blocks divide evenly among the waves");
#pragma unroll 1
    for (int k = 0; k < kIterations; ++k) {
        const int blk = k * kWaves + wave;
        const int txl = (blk % kBlocksX) * 16 + (lane & 15), ly = (blk / kBlocksX) * 4 + (lane >> 4);
        const int X = X0 + 2 * txl, Y = Y0 + ly;
        hook.begin(k);
        if (X < lw && Y < lh) {
            const float *centre = &tile[(ly + kRenApron) * kRenLdsW + 2 * txl + kRenApron];
            const float2v c = *reinterpret_cast<const float2v *>(centre);
            float2v inv_depth = float2v{rcp_strict<DIV>(c.x), rcp_strict<DIV>(c.y)};   // REN:140
            float2v out;     // one pair in flight ahead of the one evaluated; a second one changed nothing (r02 A/B)
            if constexpr (!EXH) out = accumulate_terms_pipelined<4 * kRenLdsW, 4, 1>(terms, centre, inv_depth);
            else out = accumulate_terms<EXH, 4 * kRenLdsW, 4>(terms, centre, inv_depth);

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
        hook.end(k);
        clk.mark(2 + (k & 3));
    }
}


}  // namespace
}  // namespace meao
