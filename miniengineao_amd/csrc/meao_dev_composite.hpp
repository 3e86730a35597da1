// meao_dev_composite.hpp -- Blit.shader passes 1-3 on a texel pair (composite kernel and the composite carried by the render kernel).
#pragma once

#include "meao_dev.hpp"

namespace meao {
namespace {

// ------------------------------------------------------------------------------------------
// Composite (Blit.shader:66-134): pure streaming, 17 bytes per texel (RGBA16F read + write, AO).
// One lane = 4 texels = two 16-byte colour loads/stores + one 4-byte (R8) AO load.

__device__ __forceinline__ uint16_t f32_to_f16_rtne_bits(float x) { return f32_to_f16_bits<true>(x); }

// Texel pair q (texels 2q, 2q+1) of one frame: one 16-byte colour load / store per lane.
template <int AOFMT>
__device__ __forceinline__ void composite_pair(const void *ao_base, void *color_base, void *gbuffer0_base, int64_t pixels,
                                               int32_t mode, int64_t q)
{
    typedef AoTexel<AOFMT> AO;
    typedef typename AO::type ao_t;
    const int64_t p0 = q * 2;
    const bool full = p0 + 1 < pixels;
    const ao_t *ap = static_cast<const ao_t *>(ao_base) + p0;
    float aov[2] = {1.0f, 1.0f};
    if (full) {
        const typename AO::type2 a2 = *reinterpret_cast<const typename AO::type2 *>(ap);
        aov[0] = AO::decode(a2.x); aov[1] = AO::decode(a2.y);
    } else {
        aov[0] = AO::decode(ap[0]);
    }
    uint16_t c[8] = {};
    uint16_t *cp = static_cast<uint16_t *>(color_base) + p0 * 4;
    if (full) {
        const uint4v raw = *reinterpret_cast<const uint4v *>(cp);
        c[0] = raw.x & 0xffffu; c[1] = raw.x >> 16; c[2] = raw.y & 0xffffu; c[3] = raw.y >> 16;
        c[4] = raw.z & 0xffffu; c[5] = raw.z >> 16; c[6] = raw.w & 0xffffu; c[7] = raw.w >> 16;
    } else {
        for (int k = 0; k < 4; ++k) c[k] = cp[k];
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (p0 + e >= pixels) break;
        const float ao = aov[e];
        uint16_t *t = c + 4 * e;
        if (mode == MEAO_COMPOSITE_DEBUG) {                          // pass 3: frag returns ao in every channel
            t[0] = t[1] = t[2] = t[3] = f32_to_f16_rtne_bits(ao);
        } else if (mode == MEAO_COMPOSITE_MULTIPLY) {                // pass 2: dst * src.a
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = f32_to_f16_rtne_bits(f16_bits_to_f32(t[k]) * ao);
        } else {                                                     // pass 1: dst * (1 - src), src = 1 - ao
            const float occ = 1.0f - ao;                             // Blit.shader:84
            const float keep = 1.0f - occ;                           // OneMinusSrcColor / OneMinusSrcAlpha
#pragma unroll
            for (int k = 0; k < 3; ++k) t[k] = f32_to_f16_rtne_bits(f16_bits_to_f32(t[k]) * keep);
            uint8_t *g = static_cast<uint8_t *>(gbuffer0_base) + (p0 + e) * 4 + 3;   // GBuffer0.a = occlusion
            *g = static_cast<uint8_t>(f32_to_unorm8(unorm8_to_f32(*g) * keep));
        }
    }
    if (full) {
        uint4v outv;
        outv.x = c[0] | (static_cast<uint32_t>(c[1]) << 16); outv.y = c[2] | (static_cast<uint32_t>(c[3]) << 16);
        outv.z = c[4] | (static_cast<uint32_t>(c[5]) << 16); outv.w = c[6] | (static_cast<uint32_t>(c[7]) << 16);
        *reinterpret_cast<uint4v *>(cp) = outv;
    } else {
        for (int k = 0; k < 4; ++k) cp[k] = c[k];
    }
}


}  // namespace
}  // namespace meao
