// meao_dev_blend.hpp -- one blend pass evaluated for an arbitrary window of its output level, into LDS (nested blend launches).
#pragma once

#include "meao_dev_upsample.hpp"

namespace meao {
namespace {

// One blend pass (Upsample.main_blendout) evaluated for an arbitrary window of its OUTPUT level, into LDS:
// out[r * out_pitch + c] = what a later pass would read back from Combined<k> at virtual texel
// (vx0 + c, vy0 + r) with clamp addressing (UPS:54-72), i.e. the stored-and-decoded result.
// Every output of the pass is a pure function of the global inputs (each blurred value only depends
// on its own 5-tap window of clamped taps, the bilateral taps on the texel's parity), so evaluating it
// here gives the bits the stand-alone pass writes.  Texels of the window that fall into the "own"
// rectangle are also stored to the pass's real target, so that the buffer exists for the debug views.
// Window at most 38 x 22: low-res D range <= 21 x 13, raw taps <= 25 x 17 (scratch: 1905 floats).
constexpr int kNestLowW = 21, kNestLowH = 13, kNestRawW = kNestLowW + 4, kNestRawH = kNestLowH + 4;
constexpr int kNestScratch = 3 * kNestRawW * kNestRawH + kNestLowW * kNestRawH + kNestLowW * kNestLowH;

// The low-res texels a window of the pass's output level touches (bilateral taps D = (X+1)>>1 and D-1, X clamped
// to the level) and, two further out on every side, the raw LoResAO1 / LoResDB taps of the blur (virtual
// coordinates, clamped on load).
struct NestExtent {
    int dx_lo, dy_lo, nlw, nlh, rx0, ry0, rw, rh;
    __device__ __forceinline__ NestExtent(const UpsampleArgs &in, int vx0, int vy0, int win_w, int win_h)
    {
        const int cx_min = clampi(vx0, 0, in.hw - 1), cx_max = clampi(vx0 + win_w - 1, 0, in.hw - 1);
        const int cy_min = clampi(vy0, 0, in.hh - 1), cy_max = clampi(vy0 + win_h - 1, 0, in.hh - 1);
        dx_lo = ((cx_min + 1) >> 1) - 1; nlw = ((cx_max + 1) >> 1) - dx_lo + 1;
        dy_lo = ((cy_min + 1) >> 1) - 1; nlh = ((cy_max + 1) >> 1) - dy_lo + 1;
        rx0 = dx_lo - 2; ry0 = dy_lo - 2; rw = nlw + 4; rh = nlh + 4;
    }
};

// TAPS_IN_LDS: the raw LoResAO1 taps (scratch[r * kNestRawW + c], r < rh, c < rw of NestExtent) were produced by
// another blend_window_into_lds call (the pass below, evaluated for exactly that window) instead of being
// read from Combined<k+1> in global memory.
template <int AOFMT, bool RTNE, int DIV, bool TAPS_IN_LDS = false>
__device__ __forceinline__ void blend_window_into_lds(const UpsampleArgs &in, float *out, int out_pitch, int vx0, int vy0,
                                                      int win_w, int win_h, float *scratch, int frame, int own_x0,
                                                      int own_y0, int own_w, int own_h)
{
    typedef AoTexel<AOFMT> AO;
    typedef typename AO::type ao_t;
    float *const r_ao = scratch;                                   // raw LoResAO1 taps
    float *const r_inv = r_ao + kNestRawW * kNestRawH;             // 1 / LoResDB
    float *const r_dep = r_inv + kNestRawW * kNestRawH;            // LoResDB
    float *const hb = r_dep + kNestRawW * kNestRawH;               // after BlurHorizontally
    float *const vb = hb + kNestLowW * kNestRawH;                  // after BlurVertically
    const int lw = in.lw, lh = in.lh, hw = in.hw, hh = in.hh;
    const float *__restrict__ lo_depth = frame_ptr(in.lo_depth, in.frame_stride, frame);
    const ao_t *__restrict__ lo_ao = frame_ptr(static_cast<const ao_t *>(in.lo_ao), in.frame_stride, frame);
    const float *__restrict__ hi_depth = frame_ptr(static_cast<const float *>(in.hi_depth), in.frame_stride, frame);
    const ao_t *__restrict__ hi_ao = frame_ptr(static_cast<const ao_t *>(in.hi_ao), in.frame_stride, frame);
    ao_t *__restrict__ dst = frame_ptr(static_cast<ao_t *>(in.dst[0]), in.frame_stride, frame);
    const BlurConsts bk = {in.step_size, in.blur_tolerance};
    const BilateralConsts bilateral_k(in.upsample_tolerance, in.noise_filter_strength);

    const NestExtent ext(in, vx0, vy0, win_w, win_h);
    const int dx_lo = ext.dx_lo, dy_lo = ext.dy_lo, nlw = ext.nlw, nlh = ext.nlh;
    const int rx0 = ext.rx0, ry0 = ext.ry0, rw = ext.rw, rh = ext.rh;           // raw taps (virtual, clamped on load)

    // The hi-res operands of the bilateral step depend on nothing computed here: loaded now, used three
    // barriers later (at most 38 x 22 window texels: four per lane).
    constexpr int kHoisted = (40 * 22 + kThreads - 1) / kThreads;          // items on the output pitch: at most 40 x 22
    float hoist_d[kHoisted];
    ao_t hoist_a[kHoisted];
    // (window items on the output array's pitch, a compile-time value at every call site; see the loops below)
#pragma unroll
    for (int j = 0; j < kHoisted; ++j) {
        const int i = min(static_cast<int>(threadIdx.x) + j * kThreads, out_pitch * win_h - 1);
        const int X = clampi(vx0 + min(i % out_pitch, win_w - 1), 0, hw - 1), Y = clampi(vy0 + i / out_pitch, 0, hh - 1);
        const uint32_t at = static_cast<uint32_t>(Y * hw + X);
        hoist_d[j] = *at_byte_offset(hi_depth, at * 4u);
        hoist_a[j] = *at_byte_offset(hi_ao, at * static_cast<uint32_t>(sizeof(ao_t)));
    }

    // Work items are laid out on the arrays' compile-time pitches (item i = row i / pitch, column i % pitch; columns past the
    // extent idle): the extents are run-time values, and a division by one costs ~25 VALU instructions where a division by
    // a constant costs three -- the four loops of this function did eight of them per lane (a third of the two-level
    // kernel's instructions were integer arithmetic, profiles/r03_pmc_summary.txt).
    for (int i = threadIdx.x; i < kNestRawW * rh; i += kThreads) {
        const int r = i / kNestRawW, c = i % kNestRawW;
        if (c >= rw) continue;
        const uint32_t idx = static_cast<uint32_t>(clampi(ry0 + r, 0, lh - 1) * lw + clampi(rx0 + c, 0, lw - 1));   // a level is < 2^30 texels
        const float d = *at_byte_offset(lo_depth, idx * 4u);
        r_dep[i] = d;
        r_inv[i] = rcp_strict<DIV>(d);                                          // UPS:67
        if constexpr (!TAPS_IN_LDS) r_ao[i] = AO::decode(*at_byte_offset(lo_ao, idx * static_cast<uint32_t>(sizeof(ao_t))));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kNestLowW * rh; i += kThreads) {               // BlurHorizontally, one output per lane
        const int r = i / kNestLowW, c = i % kNestLowW;
        if (c >= nlw) continue;
        float av[5], zv[5], o[1];
#pragma unroll
        for (int t = 0; t < 5; ++t) { av[t] = r_ao[r * kNestRawW + c + t]; zv[t] = r_inv[r * kNestRawW + c + t]; }
        blur_run<1>(bk, av, zv, o);
        hb[i] = o[0];
    }
    __syncthreads();
    // BlurVertically.  The interior extent (20 x 12 = 240 outputs) is one round of the workgroup on a pitch of 20
    auto blur_vertically = [&](auto pitch_c) __attribute__((always_inline)) {
        constexpr int kPitch = decltype(pitch_c)::value;
        for (int i = threadIdx.x; i < kPitch * nlh; i += kThreads) {
            const int r = i / kPitch, c = i % kPitch;
            if (kPitch != kNestLowW || c < nlw) {
                float av[5], zv[5], o[1];
#pragma unroll
                for (int t = 0; t < 5; ++t) { av[t] = hb[(r + t) * kNestLowW + c]; zv[t] = r_inv[(r + t) * kNestRawW + c + 2]; }
                blur_run<1>(bk, av, zv, o);
                vb[r * kNestLowW + c] = o[0];
            }
        }
    };
    if (nlw == kNestLowW - 1) blur_vertically(std::integral_constant<int, kNestLowW - 1>());
    else blur_vertically(std::integral_constant<int, kNestLowW>());
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kHoisted; ++j) {
        const int i = threadIdx.x + j * kThreads;
        if (i >= out_pitch * win_h) break;
        const int wr = i / out_pitch, wc = i % out_pitch;
        if (wc >= win_w) continue;
        const int X = clampi(vx0 + wc, 0, hw - 1), Y = clampi(vy0 + wr, 0, hh - 1);
        const int Dx = (X + 1) >> 1, Dy = (Y + 1) >> 1;
        // Tap k of the texel is Gather component g = (comp + k) & 3 of dispatch thread D: texel D + (gx[g], gy[g]), i.e. the four
        // texels {Dx - 1, Dx} x {Dy - 1, Dy} in an order that rotates with the texel's parity (comp, UPS:229-232).  comp is a per-lane
        // value here (the window is dealt to lanes linearly), so the taps' byte distances below D in either array -- {4, 0, pitch * 4,
        // pitch * 4 + 4} for g = 0..3 -- sit in one word that is rotated by comp bytes: five integer operations for four addresses.
        // (Indexing gx[] / gy[] with the run-time g made the compiler put the tables in memory: eight global loads per texel.)
        const uint32_t comp = ((static_cast<uint32_t>(Y) & 1u) << 1) | (((static_cast<uint32_t>(X ^ Y)) & 1u) ^ 1u);   // (X odd, Y odd): (1,0) 0, (0,0) 1, (0,1) 2, (1,1) 3
        constexpr uint32_t kBelowVb = 4u | (0u << 8) | (static_cast<uint32_t>(kNestLowW * 4) << 16) | (static_cast<uint32_t>(kNestLowW * 4 + 4) << 24);
        constexpr uint32_t kBelowDep = 4u | (0u << 8) | (static_cast<uint32_t>(kNestRawW * 4) << 16) | (static_cast<uint32_t>(kNestRawW * 4 + 4) << 24);
        static_assert(kNestLowW * 4 + 4 < 256 && kNestRawW * 4 + 4 < 256, "byte fields");
        const uint32_t below_vb = __builtin_amdgcn_alignbit(kBelowVb, kBelowVb, comp * 8u), below_dep = __builtin_amdgcn_alignbit(kBelowDep, kBelowDep, comp * 8u);
        const char *const vb_at_d = reinterpret_cast<const char *>(vb + ((Dy - dy_lo) * kNestLowW + (Dx - dx_lo)));
        const char *const dep_at_d = reinterpret_cast<const char *>(r_dep + ((Dy - ry0) * kNestRawW + (Dx - rx0)));
        float dk[4], ak[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ak[k] = *reinterpret_cast<const float *>(vb_at_d - ((below_vb >> (8 * k)) & 0xffu));
            dk[k] = *reinterpret_cast<const float *>(dep_at_d - ((below_dep >> (8 * k)) & 0xffu));
        }
        const uint32_t at = static_cast<uint32_t>(Y * hw + X) * static_cast<uint32_t>(sizeof(ao_t));      // byte offset in the level
        float v;
        if constexpr (!MEAO_X_UPS_EXACT_R8 && DIV == DIV_EXACT_RCP && AOFMT == MEAO_AO_R8) {
            const ao_t q = static_cast<ao_t>(bilateral_upsample_r8<true, false, MEAO_X_BIL_PAIR_RCP != 0>(hoist_d[j], AO::decode(hoist_a[j]), dk, ak, bilateral_k));
            out[i] = AO::decode(q);
            if (vx0 + wc == X && vy0 + wr == Y && X >= own_x0 && X < own_x0 + own_w && Y >= own_y0 && Y < own_y0 + own_h) *at_byte_offset(dst, at) = q;
            continue;
        }
        if constexpr (DIV == DIV_EXACT_RCP) {       // the four weight reciprocals back to back (see upsample_tile)
            const float ghd[1] = {hoist_d[j]}, gha[1] = {AO::decode(hoist_a[j])};
            const float gd[1][4] = {{dk[0], dk[1], dk[2], dk[3]}}, ga[1][4] = {{ak[0], ak[1], ak[2], ak[3]}};
            float gout[1];
            bilateral_upsample_grouped<1>(ghd, gha, gd, ga, bilateral_k, gout);
            v = gout[0];
        } else {
            v = bilateral_upsample<DIV>(hoist_d[j], AO::decode(hoist_a[j]), dk[0], dk[1], dk[2], dk[3], ak[0], ak[1], ak[2], ak[3],
                                        bilateral_k);
        }
        const ao_t q = AO::template encode<RTNE>(v);
        out[i] = AO::decode(q);
        if (vx0 + wc == X && vy0 + wr == Y && X >= own_x0 && X < own_x0 + own_w && Y >= own_y0 && Y < own_y0 + own_h) *at_byte_offset(dst, at) = q;
    }
    __syncthreads();
}


}  // namespace
}  // namespace meao
