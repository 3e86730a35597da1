/*
 * meao_hlsl_emul.c -- second, independently structured restatement of the
 * MiniEngineAO hot path: a literal emulation of the reference's compute
 * dispatches (thread groups of 8x8, groupshared arrays, barriers as phase
 * boundaries, SV_* system values, Texture.Load / RWTexture store / Gather
 * semantics, the dispatch grids of AmbientOcclusion.cs).
 *
 * TEST INFRASTRUCTURE ONLY (see meao_oracle.h).  Its only purpose is to be
 * compared bit-for-bit with the gather-form oracle (meao_oracle.c): the two
 * share nothing but the storage conversions and the host constant helpers.
 * Parity vs the reference on its own platform remains UNPINNED (see meao_oracle.h for the pins).
 *
 * Citations: DS1/DS2/REN/UPS = Assets/MiniEngineAO/Shaders/{Downsample1,
 * Downsample2,Render,Upsample}.compute, AO.cs = AmbientOcclusion.cs.
 */
#define _GNU_SOURCE
#include "meao_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* resource model                                                            */

typedef enum { FMT_F32, FMT_F16, FMT_UNORM8 } tex_fmt;

typedef struct {
    int w, h, slices;     /* slices = 1 for Texture2D, 16 for the tiled arrays */
    tex_fmt fmt;
    int f16_rounding;
    void *data;
} tex_t;

static size_t tex_elem(tex_fmt f) { return f == FMT_F32 ? 4 : (f == FMT_F16 ? 2 : 1); }

static float tex_fetch(const tex_t *t, int x, int y, int s)
{
    size_t i = ((size_t)s * t->h + y) * t->w + x;
    switch (t->fmt) {
    case FMT_F32: return ((const float *)t->data)[i];
    case FMT_F16: return meao_oracle_f16_to_f32(((const uint16_t *)t->data)[i]);
    default:      return meao_oracle_unorm8_to_f32(((const uint8_t *)t->data)[i]);
    }
}

/* Texture2D<float>[uint2]: out-of-range -> 0  (D3D resource load rule) */
static float tex_load(const tex_t *t, unsigned x, unsigned y)
{
    if (x >= (unsigned)t->w || y >= (unsigned)t->h) return 0.0f;
    return tex_fetch(t, (int)x, (int)y, 0);
}

/* RWTexture2D / RWTexture2DArray store: out-of-range (incl. negative) dropped,
 * value converted to the target's storage format. */
static void tex_store(tex_t *t, int x, int y, int s, float v)
{
    if (x < 0 || y < 0 || x >= t->w || y >= t->h || s < 0 || s >= t->slices) return;
    size_t i = ((size_t)s * t->h + y) * t->w + x;
    switch (t->fmt) {
    case FMT_F32: ((float *)t->data)[i] = v; break;
    case FMT_F16: ((uint16_t *)t->data)[i] = meao_oracle_f32_to_f16(v, t->f16_rounding); break;
    default:      ((uint8_t *)t->data)[i] = meao_oracle_f32_to_unorm8(v); break;
    }
}

typedef struct { float x, y, z, w; } f4;

/* Texture.Gather with a point/clamp sampler: the 2x2 footprint around the
 * sample position; .x=(i0,j1) .y=(i1,j1) .z=(i1,j0) .w=(i0,j0). */
static f4 tex_gather(const tex_t *t, float u, float v, int s)
{
    float px = u * (float)t->w - 0.5f, py = v * (float)t->h - 0.5f;
    int i0 = (int)floorf(px), j0 = (int)floorf(py);
    int i1 = i0 + 1, j1 = j0 + 1;
#define CL(a, n) ((a) < 0 ? 0 : ((a) > (n) - 1 ? (n) - 1 : (a)))
    i0 = CL(i0, t->w); i1 = CL(i1, t->w); j0 = CL(j0, t->h); j1 = CL(j1, t->h);
#undef CL
    f4 r;
    r.x = tex_fetch(t, i0, j1, s); r.y = tex_fetch(t, i1, j1, s);
    r.z = tex_fetch(t, i1, j0, s); r.w = tex_fetch(t, i0, j0, s);
    return r;
}

static float hl_saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
static float hl_clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* ------------------------------------------------------------------------ */
/* Downsample1.main  (DS1:52-81), grid = TiledDepth2 dims (AO.cs:643)        */

typedef struct { const tex_t *depth; tex_t *linear_z; float zp[4]; int reversed; } ds1_res;

static float ds1_linearize(const ds1_res *r, unsigned sx, unsigned sy)
{
    float raw = tex_load(r->depth, sx, sy);
    float dist = 1.0f / fmaf(r->zp[0], raw, r->zp[1]);
    if (r->reversed) { if (raw == 0.0f) dist = 1e5f; }
    else             { if (raw == 1.0f) dist = 1e5f; }
    tex_store(r->linear_z, (int)sx, (int)sy, 0, dist);
    return dist;
}

static void dispatch_downsample1(const ds1_res *r, tex_t *ds2x, tex_t *ds2x_atlas,
                                 tex_t *ds4x, tex_t *ds4x_atlas, int groups_x, int groups_y)
{
    for (int gy = 0; gy < groups_y; gy++)
    for (int gx = 0; gx < groups_x; gx++) {
        float cache[256];
        for (unsigned gi = 0; gi < 64; gi++) {           /* phase 1: fill the 16x16 tile */
            unsigned tx = gi & 7, ty = gi >> 3;
            unsigned bx = ((unsigned)gx << 4) | tx, by = ((unsigned)gy << 4) | ty;
            unsigned dst = (ty << 4) | tx;
            cache[dst]       = ds1_linearize(r, bx,     by);
            cache[dst + 8]   = ds1_linearize(r, bx | 8, by);
            cache[dst + 128] = ds1_linearize(r, bx,     by | 8);
            cache[dst + 136] = ds1_linearize(r, bx | 8, by | 8);
        }
        /* GroupMemoryBarrierWithGroupSync */
        for (unsigned gi = 0; gi < 64; gi++) {           /* phase 2: decimate + de-interleave */
            unsigned tx = gi & 7, ty = gi >> 3;
            float w1 = cache[(tx << 1) | (ty << 5)];
            unsigned px = (unsigned)gx * 8 + tx, py = (unsigned)gy * 8 + ty;     /* DTid */
            unsigned slice = ((px & 3) | (py << 2)) & 15;
            tex_store(ds2x, (int)px, (int)py, 0, w1);
            tex_store(ds2x_atlas, (int)(px >> 2), (int)(py >> 2), (int)slice, w1);
            if ((gi & 011) == 0) {                       /* octal: even column and even row */
                unsigned qx = px >> 1, qy = py >> 1;
                slice = ((qx & 3) | (qy << 2)) & 15;
                tex_store(ds4x, (int)qx, (int)qy, 0, w1);
                tex_store(ds4x_atlas, (int)(qx >> 2), (int)(qy >> 2), (int)slice, w1);
            }
        }
    }
}

/* Downsample2.main (DS2:32-51), grid = TiledDepth4 dims (AO.cs:657) */
static void dispatch_downsample2(const tex_t *ds4x, tex_t *ds8x, tex_t *ds8x_atlas,
                                 tex_t *ds16x, tex_t *ds16x_atlas, int groups_x, int groups_y)
{
    for (int gy = 0; gy < groups_y; gy++)
    for (int gx = 0; gx < groups_x; gx++)
        for (unsigned gi = 0; gi < 64; gi++) {
            unsigned px = (unsigned)gx * 8 + (gi & 7), py = (unsigned)gy * 8 + (gi >> 3);
            float m1 = tex_load(ds4x, px << 1, py << 1);
            unsigned slice = ((px & 3) | (py << 2)) & 15;
            tex_store(ds8x, (int)px, (int)py, 0, m1);
            tex_store(ds8x_atlas, (int)(px >> 2), (int)(py >> 2), (int)slice, m1);
            if ((gi & 011) == 0) {
                unsigned qx = px >> 1, qy = py >> 1;
                slice = ((qx & 3) | (qy << 2)) & 15;
                tex_store(ds16x, (int)qx, (int)qy, 0, m1);
                tex_store(ds16x_atlas, (int)(qx >> 2), (int)(qy >> 2), (int)slice, m1);
            }
        }
}

/* ------------------------------------------------------------------------ */
/* Render.main_interleaved (REN:112-177): TILE_DIM 16, 8x8 threads;           */
/* Render.main (WIDE_SAMPLING, REN:22,27-29,46-50): TILE_DIM 32, 16x16 threads */

typedef struct { const float *lds; const meao_oracle_render_consts *cb; int tile_dim, wide; } ren_thread;

static float ren_pair(const ren_thread *t, float front_depth, float inv_range, unsigned base, int offset)
{
    float o1 = fmaf(t->lds[(int)base + offset], inv_range, -front_depth);
    float o2 = fmaf(t->lds[(int)base - offset], inv_range, -front_depth);
    float q1 = hl_saturate(t->cb->reject_fadeoff * o1);
    float q2 = hl_saturate(t->cb->reject_fadeoff * o2);
    float acc = hl_clamp(o1, q2, 1.0f) + hl_clamp(o2, q1, 1.0f);
    return hl_saturate(fmaf(-q1, q2, acc));
}

static float ren_samples(const ren_thread *t, unsigned centre, unsigned x, unsigned y,
                         float inv_depth, float inv_thickness)
{
    if (t->wide) { x <<= 1; y <<= 1; }      /* REN:79-82 */
    float inv_range = inv_thickness * inv_depth;
    float front_depth = inv_thickness - 0.5f;
    int X = (int)x, Y = (int)y;
    const int REN_TILE = t->tile_dim;
    if (y == 0)
        return 0.5f * (ren_pair(t, front_depth, inv_range, centre, X) +
                       ren_pair(t, front_depth, inv_range, centre, X * REN_TILE));
    if (x == y)
        return 0.5f * (ren_pair(t, front_depth, inv_range, centre, X * REN_TILE - X) +
                       ren_pair(t, front_depth, inv_range, centre, X * REN_TILE + X));
    return 0.25f * (ren_pair(t, front_depth, inv_range, centre, Y * REN_TILE + X) +
                    ren_pair(t, front_depth, inv_range, centre, Y * REN_TILE - X) +
                    ren_pair(t, front_depth, inv_range, centre, X * REN_TILE + Y) +
                    ren_pair(t, front_depth, inv_range, centre, X * REN_TILE - Y));
}

/* wide = 0: main_interleaved on a 16-slice atlas; wide = 1: main on a 2D texture.
 * exhaustive: the SAMPLE_EXHAUSTIVELY term list (REN:146-157) instead of the checker set. */
static void dispatch_render(const tex_t *depth_tex, tex_t *occlusion,
                            const meao_oracle_render_consts *cb, int wide, int exhaustive,
                            int groups_x, int groups_y, int groups_z)
{
    /* (weight/thickness table slot, x, y) in accumulation order, REN:162-168 / REN:146-157 */
    static const unsigned checker[7][3] = {
        { 1, 2, 0 }, { 3, 4, 0 }, { 4, 1, 1 }, { 8, 2, 2 }, { 11, 3, 3 }, { 6, 1, 3 }, { 10, 2, 4 } };
    static const unsigned all68[12][3] = {
        { 0, 1, 0 }, { 1, 2, 0 }, { 2, 3, 0 }, { 3, 4, 0 }, { 4, 1, 1 }, { 8, 2, 2 }, { 11, 3, 3 },
        { 5, 1, 2 }, { 6, 1, 3 }, { 7, 1, 4 }, { 9, 2, 3 }, { 10, 2, 4 } };
    const unsigned (*plan)[3] = exhaustive ? all68 : checker;
    const int terms = exhaustive ? 12 : 7;
    const int tile = wide ? 32 : 16, tc = wide ? 16 : 8;      /* TILE_DIM, THREAD_COUNT_X/Y */
    const int apron = wide ? 7 : 3, centre_off = wide ? 8 : 4;
    for (int gz = 0; gz < groups_z; gz++)
    for (int gy = 0; gy < groups_y; gy++)
    for (int gx = 0; gx < groups_x; gx++) {
        float samples[32 * 32];
        for (int gi = 0; gi < tc * tc; gi++) {
            int tx = gi % tc, ty = gi / tc;
            int dx = gx * tc + tx, dy = gy * tc + ty;
            float u = (float)(dx + tx - apron) * cb->inv_slice_dim[0];
            float v = (float)(dy + ty - apron) * cb->inv_slice_dim[1];
            f4 g = tex_gather(depth_tex, u, v, gz);
            int dst = tx * 2 + ty * 2 * tile;
            samples[dst] = g.w; samples[dst + 1] = g.z;
            samples[dst + tile] = g.x; samples[dst + tile + 1] = g.y;
        }
        /* barrier */
        for (int gi = 0; gi < tc * tc; gi++) {
            int tx = gi % tc, ty = gi / tc;
            ren_thread th = { samples, cb, tile, wide };
            unsigned centre = (unsigned)(tx + ty * tile + centre_off * tile + centre_off);
            float inv_depth = 1.0f / samples[centre];
            float ao = 0.0f;
            for (int n = 0; n < terms; n++)
                ao = fmaf(cb->sample_weight[plan[n][0]],
                          ren_samples(&th, centre, plan[n][1], plan[n][2], inv_depth,
                                      cb->inv_thickness[plan[n][0]]), ao);
            int ox = gx * tc + tx, oy = gy * tc + ty;                        /* REN:174 */
            if (!wide) { ox = (ox << 2) | (gz & 3); oy = (oy << 2) | (gz >> 2); }   /* REN:172 */
            tex_store(occlusion, ox, oy, 0, fmaf(cb->intensity, ao - 1.0f, 1.0f));
        }
    }
}

/* ------------------------------------------------------------------------ */
/* Upsample.main / main_blendout (UPS:185-233)                               */

typedef struct {
    float depth_cache[256], ao_cache1[256], ao_cache2[256];
    const meao_oracle_upsample_consts *cb;
} ups_group;

static int ups_compare(const ups_group *g, float da, float db, float la, float lb)
{
    float t = fmaf(da, db, g->cb->step_size);
    return t * t > la * lb * g->cb->blur_tolerance;
}

static float ups_smart_blur(const float *p, int left, int middle, int right)
{   /* p[0..4] = a..e */
    float a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    b = (left | middle) ? b : c;
    a = left ? a : b;
    d = (right | middle) ? d : c;
    e = right ? e : d;
    return ((a + e) / 2.0f + b + c + d) / 4.0f;
}

/* shared body of BlurHorizontally (n=7 taps, stride 1, 3 outputs) and
 * BlurVertically (n=6 taps, stride 16, 2 outputs). */
static void ups_blur_run(ups_group *g, const float *src, const float *depth, int stride,
                         int ntaps, float *dst_base)
{
    float a[7], z[7], dz[6], ln[6]; int keep[5];
    for (int k = 0; k < ntaps; k++) { a[k] = src[k * stride]; z[k] = depth[k * stride]; }
    for (int k = 0; k + 1 < ntaps; k++) { dz[k] = z[k + 1] - z[k]; ln[k] = fmaf(dz[k], dz[k], g->cb->step_size); }
    for (int k = 0; k + 2 < ntaps; k++) keep[k] = ups_compare(g, dz[k], dz[k + 1], ln[k], ln[k + 1]);
    for (int k = 0; k + 4 < ntaps; k++)
        dst_base[k * stride] = ups_smart_blur(a + k, keep[k], keep[k + 1], keep[k + 2]);
}

static float ups_bilateral(const meao_oracle_upsample_consts *cb, float hi_depth, float hi_ao,
                           const float lo_depth[4], const float lo_ao[4])
{
    static const float nine_three_one_three[4] = { 9, 3, 1, 3 };
    float wt[4];
    for (int k = 0; k < 4; k++)
        wt[k] = nine_three_one_three[k] / (fabsf(hi_depth - lo_depth[k]) + cb->upsample_tolerance);
    float total = wt[0];                     /* dp4 against 1: mul by one, then mad chain */
    for (int k = 1; k < 4; k++) total = fmaf(wt[k], 1.0f, total);
    total += cb->noise_filter_strength;
    float wsum = lo_ao[0] * wt[0];
    for (int k = 1; k < 4; k++) wsum = fmaf(lo_ao[k], wt[k], wsum);
    wsum += cb->noise_filter_strength;
    return hi_ao * wsum / total;
}

static void dispatch_upsample(const tex_t *lo_db, const tex_t *hi_db, const tex_t *lo_ao,
                              const tex_t *lo_ao2 /* non-NULL: main_premin* */,
                              const tex_t *hi_ao /* NULL: kernel "main" / "main_premin" */, tex_t *result,
                              const meao_oracle_upsample_consts *cb, int groups_x, int groups_y)
{
    ups_group *g = (ups_group *)calloc(1, sizeof *g);
    if (!g) return;
    g->cb = cb;
    for (int gy = 0; gy < groups_y; gy++)
    for (int gx = 0; gx < groups_x; gx++) {
        /* uninitialised LDS is modelled as zeros (row 13 of ao_cache2 is read by
         * the last V-blur lane group but its product is never consumed). */
        memset(g->ao_cache2, 0, sizeof g->ao_cache2);
        for (int gi = 0; gi < 64; gi++) {                /* PrefetchData */
            int tx = gi & 7, ty = gi >> 3;
            int dx = gx * 8 + tx, dy = gy * 8 + ty;
            float u = (float)(dx + tx - 2) * cb->inv_low_res[0];
            float v = (float)(dy + ty - 2) * cb->inv_low_res[1];
            int idx = (tx << 1) | (ty << 5);
            f4 ao = tex_gather(lo_ao, u, v, 0);
            if (lo_ao2) {                                /* COMBINE_LOWER_RESOLUTIONS, UPS:58-60 */
                f4 b = tex_gather(lo_ao2, u, v, 0);
                ao.x = fminf(ao.x, b.x); ao.y = fminf(ao.y, b.y); ao.z = fminf(ao.z, b.z); ao.w = fminf(ao.w, b.w);
            }
            g->ao_cache1[idx] = ao.w; g->ao_cache1[idx + 1] = ao.z;
            g->ao_cache1[idx + 16] = ao.x; g->ao_cache1[idx + 17] = ao.y;
            f4 dp = tex_gather(lo_db, u, v, 0);
            g->depth_cache[idx] = 1.0f / dp.w; g->depth_cache[idx + 1] = 1.0f / dp.z;
            g->depth_cache[idx + 16] = 1.0f / dp.x; g->depth_cache[idx + 17] = 1.0f / dp.y;
        }
        /* barrier */
        for (int gi = 0; gi < 39; gi++) {                /* 13x13 -> 9x13 */
            int left = (gi / 3) * 16 + (gi % 3) * 3;
            ups_blur_run(g, g->ao_cache1 + left, g->depth_cache + left, 1, 7, g->ao_cache2 + left);
        }
        /* barrier */
        float vout[256]; memcpy(vout, g->ao_cache1, sizeof vout);
        for (int gi = 0; gi < 45; gi++) {                /* 9x13 -> 9x9 (+ one unused row) */
            int top = (gi / 9) * 32 + gi % 9;
            ups_blur_run(g, g->ao_cache2 + top, g->depth_cache + top + 2, 16, 6, vout + top);
        }
        memcpy(g->ao_cache1, vout, sizeof vout);
        /* barrier */
        for (int gi = 0; gi < 64; gi++) {
            int tx = gi & 7, ty = gi >> 3;
            int dx = gx * 8 + tx, dy = gy * 8 + ty;
            int i0 = tx + ty * 16;
            float lo_a[4] = { g->ao_cache1[i0 + 16], g->ao_cache1[i0 + 17], g->ao_cache1[i0 + 1], g->ao_cache1[i0] };
            float u0 = (float)dx * cb->inv_low_res[0], v0 = (float)dy * cb->inv_low_res[1];
            float u1 = (float)(dx * 2) * cb->inv_high_res[0], v1 = (float)(dy * 2) * cb->inv_high_res[1];
            f4 ha = { 1.0f, 1.0f, 1.0f, 1.0f };
            if (hi_ao) ha = tex_gather(hi_ao, u1, v1, 0);
            f4 ld = tex_gather(lo_db, u0, v0, 0);
            f4 hd = tex_gather(hi_db, u1, v1, 0);
            float lo_d[4] = { ld.x, ld.y, ld.z, ld.w };
            float hdv[4] = { hd.x, hd.y, hd.z, hd.w }, hav[4] = { ha.x, ha.y, ha.z, ha.w };
            static const int offx[4] = { -1, 0, 0, -1 }, offy[4] = { 0, 0, -1, -1 };
            for (int comp = 0; comp < 4; comp++) {       /* .xyzw, .yzwx, .zwxy, .wxyz */
                float rd[4], ra[4];
                for (int k = 0; k < 4; k++) { rd[k] = lo_d[(comp + k) & 3]; ra[k] = lo_a[(comp + k) & 3]; }
                tex_store(result, (dx << 1) + offx[comp], (dy << 1) + offy[comp], 0,
                          ups_bilateral(cb, hdv[comp], hav[comp], rd, ra));
            }
        }
    }
    free(g);
}

/* ------------------------------------------------------------------------ */
/* command-buffer order of AO.cs:496-531                                     */

int32_t meao_hlsl_emul_run(const meao_oracle_desc *d, const void *depth_in, meao_oracle_buffers *out)
{
    if (!d || !depth_in || !out) return -1;
    if (d->width < 1 || d->height < 1 || d->num_levels < 1 || d->num_levels > 4) return -1;
    int w[7], h[7];
    for (int k = 0; k < 7; k++) meao_oracle_level_dims(d->width, d->height, k, &w[k], &h[k]);
    tex_fmt aofmt = d->ao_format == MEAO_ORACLE_AO_R8 ? FMT_UNORM8 : FMT_F16;

    void *owned[32]; int nowned = 0, fail = 0;
#define MK(t, ptr, W, H, S, F) do { (t).w = (W); (t).h = (H); (t).slices = (S); (t).fmt = (F); \
        (t).f16_rounding = d->f16_rounding; (t).data = (void *)(ptr); \
        if (!(t).data) { (t).data = calloc((size_t)(W) * (H) * (S), tex_elem(F)); \
                         owned[nowned++] = (t).data; fail |= !(t).data; } } while (0)
    /* DepthCopy (AO.cs:608-614, Blit.shader pass 0): the depth texture sampled into an RFloat target */
    float *depth = (float *)malloc((size_t)w[0] * h[0] * sizeof(float));
    if (!depth) return -3;
    for (size_t i = 0; i < (size_t)w[0] * h[0]; i++) depth[i] = meao_oracle_decode_depth(depth_in, i, d->depth_format);
    tex_t depth_tex = { w[0], h[0], 1, FMT_F32, 0, (void *)depth };
    tex_t linear, low[4], tiled[4], occ[4], comb[3], result, hq[4];
    memset(hq, 0, sizeof hq);
    if (d->hq_levels < 0 || d->hq_levels > d->num_levels) return -1;
    MK(linear, out->linear_depth, w[0], h[0], 1, FMT_F16);
    for (int k = 1; k <= 4; k++) {
        MK(low[k - 1], out->low_depth[k - 1], w[k], h[k], 1, FMT_F32);
        MK(tiled[k - 1], out->tiled_depth[k - 1], w[k + 2], h[k + 2], 16, FMT_F16);
        MK(occ[k - 1], out->occlusion[k - 1], w[k], h[k], 1, aofmt);
        if (k <= 3) MK(comb[k - 1], out->combined[k - 1], w[k], h[k], 1, aofmt);
        if (k > d->num_levels - d->hq_levels && k <= d->num_levels)
            MK(hq[k - 1], out->occlusion_hq[k - 1], w[k], h[k], 1, aofmt);
    }
    MK(result, out->result, w[0], h[0], 1, aofmt);
#undef MK
    if (fail) { for (int i = 0; i < nowned; i++) free(owned[i]); free(depth); return -3; }

    ds1_res r1; r1.depth = &depth_tex; r1.linear_z = &linear; r1.reversed = d->reversed_z;
    meao_oracle_zbuffer_params(d, r1.zp);
    dispatch_downsample1(&r1, &low[0], &tiled[0], &low[1], &tiled[1], tiled[1].w, tiled[1].h);
    dispatch_downsample2(&low[1], &low[2], &tiled[2], &low[3], &tiled[3], tiled[3].w, tiled[3].h);

    for (int k = 1; k <= d->num_levels; k++) {
        meao_oracle_render_consts cb;
        meao_oracle_render_constants(d, k, &cb);
        const tex_t *src = &tiled[k - 1];                /* AO.cs:744-746, numthreads 8,8,1 */
        const int all = d->sample_set == MEAO_ORACLE_SAMPLES_EXHAUSTIVE;
        dispatch_render(src, &occ[k - 1], &cb, 0, all, (src->w + 7) / 8, (src->h + 7) / 8, src->slices);
        if (hq[k - 1].data) {                            /* Render.main on LowDepth<k>, numthreads 16,16,1 */
            meao_oracle_render_constants_hq(d, k, &cb);
            src = &low[k - 1];
            dispatch_render(src, &hq[k - 1], &cb, 1, all, (src->w + 15) / 16, (src->h + 15) / 16, 1);
        }
    }

    const tex_t *lo_ao = &occ[d->num_levels - 1];
    for (int hi = d->num_levels - 1; hi >= 0; hi--) {
        meao_oracle_upsample_consts cb;
        meao_oracle_upsample_constants(d, hi + 1, &cb);
        const tex_t *hi_db = hi ? &low[hi - 1] : &linear;
        const tex_t *hi_ao = hi ? &occ[hi - 1] : NULL;
        tex_t *dst = hi ? &comb[hi - 1] : &result;
        dispatch_upsample(&low[hi], hi_db, lo_ao, hq[hi].data ? &hq[hi] : NULL, hi_ao, dst, &cb,
                          (hi_db->w + 17) / 16, (hi_db->h + 17) / 16);   /* AO.cs:782-783 */
        lo_ao = dst;
    }
    for (int i = 0; i < nowned; i++) free(owned[i]);
    free(depth);
    return 0;
}
