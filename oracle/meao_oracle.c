/*
 * meao_oracle.c -- gather-form CPU restatement of the MiniEngineAO hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see meao_oracle.h).  The reference has no tests or
 * goldens and its platform (Unity/D3D) is unavailable: parity against the
 * reference running there is UNPINNED; pinned instead by fixtures produced by
 * interpreting the reference's own shader source (oracle/hlsl_interp.py),
 * agreement with meao_hlsl_emul.c, and analytical KATs.
 *
 * Structure: every output texel is written as a pure function of the input
 * depth image ("gather form") -- there are no thread groups, no LDS tiles and
 * no dispatch grids in this file.  The closed forms used are derived in
 * SURVEY.md section 8a and DESIGN.md; the reference lines each one restates
 * are cited as  AO.cs = Assets/MiniEngineAO/AmbientOcclusion.cs,
 * DS1/DS2/REN/UPS = Assets/MiniEngineAO/Shaders/{Downsample1,Downsample2,
 * Render,Upsample}.compute.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -mfma (see oracle/Makefile);
 * fmaf() is the only fused operation and is always spelled out.
 */
#define _GNU_SOURCE
#include "meao_oracle.h"

#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* scalar helpers (HLSL intrinsics, D3D NaN rules: min/max drop the NaN)     */

static inline float mad(float a, float b, float c) { return fmaf(a, b, c); }
static inline float sat(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------ */
/* storage conversions                                                       */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* f32 -> f16 store conversion of the HalfUAV / HalfTiledUAV targets
 * (AO.cs:454,461-464).  RTZ: round toward zero, finite overflow clamps to
 * 65504 (D3D functional-spec reading, canonical); RTNE: IEEE nearest-even,
 * overflow to inf (the other common hardware behaviour). */
uint16_t meao_oracle_f32_to_f16(float x, int32_t rounding)
{
    uint32_t u = f2u(x);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t absu = u & 0x7fffffffu;
    if (absu >= 0x7f800000u) {                 /* inf / nan */
        if (absu == 0x7f800000u) return (uint16_t)(sign | 0x7c00u);
        return (uint16_t)(sign | 0x7e00u | ((absu >> 13) & 0x1ffu));
    }
    int32_t e = (int32_t)(absu >> 23) - 127;   /* unbiased */
    uint32_t m = absu & 0x7fffffu;
    if (e > 15) {                              /* finite overflow */
        return (uint16_t)(sign | (rounding == MEAO_ORACLE_F16_RTZ ? 0x7bffu : 0x7c00u));
    }
    uint32_t h, rest, half;                    /* rest: discarded bits, half: 1/2 ulp */
    if (e >= -14) {                            /* normal f16 */
        h = ((uint32_t)(e + 15) << 10) | (m >> 13);
        rest = m & 0x1fffu; half = 0x1000u;
    } else if (e >= -25) {                     /* f16 subnormal */
        uint32_t full = m | 0x800000u;         /* 24-bit significand */
        int shift = -e - 1;                    /* 14..24: drop this many bits */
        h = full >> shift;
        rest = full & ((1u << shift) - 1u); half = 1u << (shift - 1);
    } else {
        h = 0; rest = absu ? 1u : 0u; half = 2u; /* below half the smallest subnormal */
    }
    if (rounding == MEAO_ORACLE_F16_RTNE) {
        if (rest > half || (rest == half && (h & 1u))) h += 1u; /* carry into exp is right */
        if (h >= 0x7c00u) h = 0x7c00u;
    }
    return (uint16_t)(sign | h);
}

float meao_oracle_f16_to_f32(uint16_t hv)
{
    uint32_t sign = ((uint32_t)hv & 0x8000u) << 16;
    uint32_t e = (hv >> 10) & 0x1fu, m = hv & 0x3ffu;
    if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
    if (e == 0) {
        float v = (float)m * 5.9604644775390625e-8f;     /* m * 2^-24, exact */
        return sign ? -v : v;
    }
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

/* f32 -> UNORM8 store of the FixedUAV targets (AO.cs:466-475): NaN -> 0,
 * clamp to [0,1], scale by 255, add 0.5, truncate. */
uint8_t meao_oracle_f32_to_unorm8(float x)
{
    if (!(x == x)) return 0;
    float c = sat(x);
    float s = c * 255.0f;
    s = s + 0.5f;
    return (uint8_t)s;
}

float meao_oracle_unorm8_to_f32(uint8_t v) { return (float)v / 255.0f; }

/* The depth-copy blit (Blit.shader pass 0): what sampling the depth texture returns as float. */
float meao_oracle_decode_depth(const void *depth, uint64_t index, int32_t depth_format)
{
    switch (depth_format) {
    case MEAO_ORACLE_DEPTH_UNORM16: return (float)((const uint16_t *)depth)[index] / 65535.0f;
    case MEAO_ORACLE_DEPTH_UNORM24: return (float)(((const uint32_t *)depth)[index] & 0xffffffu) / 16777215.0f;
    case MEAO_ORACLE_DEPTH_F16:     return meao_oracle_f16_to_f32(((const uint16_t *)depth)[index]);
    default:                        return ((const float *)depth)[index];
    }
}

/* AO buffer access in either storage mode */
static inline float ao_load(const void *buf, size_t idx, int fmt)
{
    if (fmt == MEAO_ORACLE_AO_R8) return meao_oracle_unorm8_to_f32(((const uint8_t *)buf)[idx]);
    return meao_oracle_f16_to_f32(((const uint16_t *)buf)[idx]);
}
static inline void ao_store(void *buf, size_t idx, float v, int fmt, int rounding)
{
    if (fmt == MEAO_ORACLE_AO_R8) ((uint8_t *)buf)[idx] = meao_oracle_f32_to_unorm8(v);
    else ((uint16_t *)buf)[idx] = meao_oracle_f32_to_f16(v, rounding);
}
static inline size_t ao_bytes(int fmt) { return fmt == MEAO_ORACLE_AO_R8 ? 1u : 2u; }

/* ------------------------------------------------------------------------ */
/* host-side constants                                                       */

void meao_oracle_level_dims(int32_t width, int32_t height, int32_t level, int32_t *w, int32_t *h)
{
    int32_t div = 1 << level;                  /* AO.cs:278-280 */
    *w = (width + (div - 1)) / div;
    *h = (height + (div - 1)) / div;
}

void meao_oracle_zbuffer_params(const meao_oracle_desc *d, float zp[4])
{
    float fpn = d->far_clip / d->near_clip;    /* AO.cs:563 */
    if (d->reversed_z) { zp[0] = fpn - 1.0f; zp[1] = 1.0f; }   /* AO.cs:565 */
    else               { zp[0] = 1.0f - fpn; zp[1] = fpn;  }   /* AO.cs:567 */
    zp[2] = 0.0f; zp[3] = 0.0f;
}

static float unity_sqrt(float v) { return (float)sqrt((double)v); }        /* Mathf.Sqrt */
static float unity_pow10(float e) { return (float)pow(10.0, (double)e); }  /* Mathf.Pow(10, e) */

void meao_oracle_sample_thickness(float t[12])
{
    /* AO.cs:577-590; all terms are float constant expressions */
    const float a = 0.2f * 0.2f, b = 0.4f * 0.4f, c = 0.6f * 0.6f, e = 0.8f * 0.8f;
    t[0]  = unity_sqrt(1.0f - a);
    t[1]  = unity_sqrt(1.0f - b);
    t[2]  = unity_sqrt(1.0f - c);
    t[3]  = unity_sqrt(1.0f - e);
    t[4]  = unity_sqrt(1.0f - a - a);
    t[5]  = unity_sqrt(1.0f - a - b);
    t[6]  = unity_sqrt(1.0f - a - c);
    t[7]  = unity_sqrt(1.0f - a - e);
    t[8]  = unity_sqrt(1.0f - b - b);
    t[9]  = unity_sqrt(1.0f - b - c);
    t[10] = unity_sqrt(1.0f - b - e);
    t[11] = unity_sqrt(1.0f - c - c);
}

static void render_constants(const meao_oracle_desc *d, int32_t source_level, int tiled,
                             meao_oracle_render_consts *out)
{
    int32_t sw, sh;
    meao_oracle_level_dims(d->width, d->height, source_level, &sw, &sh);
    float thick[12];
    meao_oracle_sample_thickness(thick);

    float tan_half_fov_h = 1.0f / d->proj00;                       /* AO.cs:572 */
    float thickness_multiplier = 2.0f * tan_half_fov_h;            /* AO.cs:678 */
    thickness_multiplier = thickness_multiplier * 10.0f;
    thickness_multiplier = thickness_multiplier / (float)sw;
    if (!tiled) thickness_multiplier = thickness_multiplier * 2.0f;          /* AO.cs:679 */
    if (d->single_pass_stereo) thickness_multiplier = thickness_multiplier * 2.0f;   /* AO.cs:680 */
    float inverse_range_factor = 1.0f / thickness_multiplier;      /* AO.cs:683 */
    for (int i = 0; i < 12; i++)
        out->inv_thickness[i] = inverse_range_factor / thick[i];   /* AO.cs:688 */

    static const float count[12] = { 4, 4, 4, 4, 4, 8, 8, 8, 4, 8, 8, 4 };  /* AO.cs:696-707 */
    for (int i = 0; i < 12; i++) out->sample_weight[i] = count[i] * thick[i];
    if (d->sample_set != MEAO_ORACLE_SAMPLES_EXHAUSTIVE) {         /* AO.cs:709-715 ("FIXME: should we
                                                                    * support SAMPLE_EXHAUSTIVELY mode?") */
        out->sample_weight[0] = 0; out->sample_weight[2] = 0; out->sample_weight[5] = 0;
        out->sample_weight[7] = 0; out->sample_weight[9] = 0;
    }
    float total = 0.0f;
    for (int i = 0; i < 12; i++) total += out->sample_weight[i];   /* AO.cs:718-721 */
    for (int i = 0; i < 12; i++) out->sample_weight[i] /= total;   /* AO.cs:723-724 */

    out->inv_slice_dim[0] = 1.0f / (float)sw;                      /* AO.cs:171,732 */
    out->inv_slice_dim[1] = 1.0f / (float)sh;
    out->reject_fadeoff = -1.0f / d->thickness_modifier;           /* AO.cs:733 */
    out->intensity = d->intensity;                                 /* AO.cs:734 */
}

void meao_oracle_render_constants(const meao_oracle_desc *d, int32_t level, meao_oracle_render_consts *out)
{
    render_constants(d, level + 2, 1, out);     /* source = TiledDepth<level>, dims of mip level+2 */
}

void meao_oracle_render_constants_hq(const meao_oracle_desc *d, int32_t level, meao_oracle_render_consts *out)
{
    render_constants(d, level, 0, out);         /* source = LowDepth<level> */
}

void meao_oracle_upsample_constants(const meao_oracle_desc *d, int32_t low_level,
                                    meao_oracle_upsample_consts *out)
{
    int32_t lw, lh, hw, hh;
    meao_oracle_level_dims(d->width, d->height, low_level, &lw, &lh);
    meao_oracle_level_dims(d->width, d->height, low_level - 1, &hw, &hh);
    float step_size = 1920.0f / (float)lw;                         /* AO.cs:760 */
    float bt = unity_pow10(d->blur_tolerance) * step_size;         /* AO.cs:761 */
    bt = 1.0f - bt;
    bt = bt * bt;                                                  /* AO.cs:762 */
    float ut = unity_pow10(d->upsample_tolerance);                 /* AO.cs:763 */
    float nf = unity_pow10(d->noise_filter_tolerance) + ut;        /* AO.cs:764 */
    nf = 1.0f / nf;
    out->inv_low_res[0] = 1.0f / (float)lw;  out->inv_low_res[1] = 1.0f / (float)lh;
    out->inv_high_res[0] = 1.0f / (float)hw; out->inv_high_res[1] = 1.0f / (float)hh;
    out->noise_filter_strength = nf;
    out->step_size = step_size;
    out->blur_tolerance = bt;
    out->upsample_tolerance = ut;
}

/* ------------------------------------------------------------------------ */
/* row-parallel driver                                                       */

typedef void (*row_fn)(void *arg, int y0, int y1);

/* A persistent pool of worker threads that draw chunks of rows from a shared counter.  (The first form spawned and joined
 * its threads in every one of the ~25 row passes of a frame: at 128-256 threads the spawns cost as much as the arithmetic, and
 * passes with fewer than two rows per thread -- every coarse level -- ran on one core.  That made the reported CPU baseline
 * 8-9x a single core on a 256-thread host.)  One call at a time uses the pool (pool_call); results do not depend on the
 * schedule: every output row is a pure function of the pass's inputs. */
enum { MAXT = 256 };
static struct {
    pthread_mutex_t call, mu;
    pthread_cond_t work;
    pthread_t tid[MAXT];
    int workers;                    /* threads created so far */
    int sleepers;                   /* workers blocked on `work` (under mu) */
    row_fn fn; void *arg; int rows, chunk;                    /* the job: published before `generation` moves */
    /* the three words the threads hammer, one cache line each */
    /* (job number << 9) | participants: ONE word, so that a worker decides whether it takes part in a job from the same
     * load that showed it the job.  (with the count in a separate field a worker that was not part of job G,
     * preempted between the two loads, could read job G+1's larger count while still remembering G, join G+1 twice and
     * release the caller while other workers were still writing rows.) */
    volatile unsigned generation __attribute__((aligned(64)));
    volatile int next __attribute__((aligned(64)));           /* first row nobody has drawn yet */
    volatile int running __attribute__((aligned(64)));        /* participants that have not finished the job */
} g_pool = { .call = PTHREAD_MUTEX_INITIALIZER, .mu = PTHREAD_MUTEX_INITIALIZER, .work = PTHREAD_COND_INITIALIZER, .chunk = 1 };

static inline void cpu_relax(void)
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

static void pool_draw_chunks(void)
{
    for (;;) {
        const int y0 = __atomic_fetch_add(&g_pool.next, g_pool.chunk, __ATOMIC_RELAXED);
        if (y0 >= g_pool.rows) break;
        const int y1 = y0 + g_pool.chunk < g_pool.rows ? y0 + g_pool.chunk : g_pool.rows;
        g_pool.fn(g_pool.arg, y0, y1);
    }
}

/* Between the passes of a frame (a millisecond apart) the workers spin on the job number; only after ~100 us without work do
 * they block on the condition variable.  (Waking 255 sleepers through one mutex for each of a frame's ~40 row passes took longer
 * than the passes themselves.) */
static void *pool_worker(void *p)
{
    const int id = (int)(intptr_t)p;
    unsigned seen = 0;
    for (;;) {
        int spins = 0;
        while (__atomic_load_n(&g_pool.generation, __ATOMIC_ACQUIRE) == seen) {
            if (++spins < 20000) { cpu_relax(); continue; }
            pthread_mutex_lock(&g_pool.mu);
            ++g_pool.sleepers;
            while (__atomic_load_n(&g_pool.generation, __ATOMIC_ACQUIRE) == seen) pthread_cond_wait(&g_pool.work, &g_pool.mu);
            --g_pool.sleepers;
            pthread_mutex_unlock(&g_pool.mu);
        }
        seen = __atomic_load_n(&g_pool.generation, __ATOMIC_ACQUIRE);
        if (id >= (int)(seen & 511u) - 1) continue;             /* not part of this job (the caller waits for its participants only) */
        pool_draw_chunks();
        __atomic_sub_fetch(&g_pool.running, 1, __ATOMIC_ACQ_REL);
    }
    return NULL;
}

static void par_rows(int nthreads, int rows, row_fn fn, void *arg)
{
    if (nthreads > MAXT) nthreads = MAXT;
    if (nthreads > rows) nthreads = rows;
    if (nthreads <= 1) { fn(arg, 0, rows); return; }
    pthread_mutex_lock(&g_pool.call);
    while (g_pool.workers < nthreads - 1) {                      /* grow the pool on demand; a failed spawn just leaves fewer workers */
        if (pthread_create(&g_pool.tid[g_pool.workers], NULL, pool_worker, (void *)(intptr_t)g_pool.workers) != 0) break;
        pthread_detach(g_pool.tid[g_pool.workers]);
        ++g_pool.workers;
    }
    const int participants = (g_pool.workers < nthreads - 1 ? g_pool.workers : nthreads - 1) + 1;
    g_pool.fn = fn; g_pool.arg = arg; g_pool.rows = rows;
    g_pool.chunk = rows / (4 * participants) > 0 ? rows / (4 * participants) : 1;
    __atomic_store_n(&g_pool.next, 0, __ATOMIC_RELAXED);
    __atomic_store_n(&g_pool.running, participants, __ATOMIC_RELAXED);
    const unsigned job = (g_pool.generation >> 9) + 1u;          /* MAXT = 256 participants fit the low 9 bits */
    __atomic_store_n(&g_pool.generation, (job << 9) | (unsigned)participants, __ATOMIC_RELEASE);      /* publishes the job */
    pthread_mutex_lock(&g_pool.mu);
    if (g_pool.sleepers > 0) pthread_cond_broadcast(&g_pool.work);
    pthread_mutex_unlock(&g_pool.mu);
    pool_draw_chunks();                                          /* the caller is participant number one */
    __atomic_sub_fetch(&g_pool.running, 1, __ATOMIC_ACQ_REL);
    for (int spins = 0; __atomic_load_n(&g_pool.running, __ATOMIC_ACQUIRE) > 0; ++spins)
        if (spins < 20000) cpu_relax(); else sched_yield();
    pthread_mutex_unlock(&g_pool.call);
}

/* ------------------------------------------------------------------------ */
/* pass 1+2: linearize, point-downsample, de-interleave  (DS1, DS2)          */

typedef struct {
    const meao_oracle_desc *d; const void *depth; float zp[4];
    int w[7], h[7];
    uint16_t *linear; float *low[4]; uint16_t *tiled[4];
} ds_ctx;

/* DS1:37-48.  Out-of-range texture loads return 0 (DS1:39). */
static inline float linearize(const ds_ctx *c, int x, int y)
{
    float dep = (x < c->w[0] && y < c->h[0])
                    ? meao_oracle_decode_depth(c->depth, (uint64_t)y * c->w[0] + x, c->d->depth_format) : 0.0f;
    float dist = 1.0f / mad(c->zp[0], dep, c->zp[1]);
    if (c->d->reversed_z ? (dep == 0.0f) : (dep == 1.0f)) dist = 1e5f;
    return dist;
}

static void ds_linear_rows(void *arg, int y0, int y1)
{   /* LinearZ[st] = dist  (DS1:46) */
    ds_ctx *c = (ds_ctx *)arg;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < c->w[0]; x++)
            c->linear[(size_t)y * c->w[0] + x] =
                meao_oracle_f32_to_f16(linearize(c, x, y), c->d->f16_rounding);
}

typedef struct { ds_ctx *c; int k; } ds_low_arg;
static void ds_low_rows(void *arg, int y0, int y1)
{   /* DSkx[i,j] = lin(k*i, k*j), k = 2,4,8,16: top-left point sample of each
     * block (DS1:64-78 keeps LDS[(2y)*16+2x]; DS2:35 reads DS4x[2*DTid]). */
    ds_low_arg *a = (ds_low_arg *)arg; ds_ctx *c = a->c; int k = a->k;
    int stride = 1 << k;                       /* level k = 1..4 -> stride 2,4,8,16 */
    for (int j = y0; j < y1; j++)
        for (int i = 0; i < c->w[k]; i++)
            c->low[k - 1][(size_t)j * c->w[k] + i] = linearize(c, stride * i, stride * j);
}

typedef struct { ds_ctx *c; int k; } ds_tile_arg;
static void ds_tile_rows(void *arg, int y0, int y1)
{   /* Atlas k (k=1..4) has the dims of mip k+2 and 16 slices; slice index
     * = (i&3) | ((j&3)<<2) of the level-k texel (i,j) (DS1:69,76; DS2:39,47).
     * Texels whose level-k source is outside level k are "padding": levels
     * 1,2 see Linearize of an out-of-range depth load (DS1:39-46), levels 3,4
     * see the 0 returned by the out-of-range DS4x load (DS2:35). */
    ds_tile_arg *a = (ds_tile_arg *)arg; ds_ctx *c = a->c; int k = a->k;
    int tw = c->w[k + 2], th = c->h[k + 2];
    float pad = (k <= 2) ? linearize(c, c->w[0], c->h[0]) : 0.0f;
    for (int row = y0; row < y1; row++) {       /* row over 16*th */
        int s = row / th, ty = row % th;
        for (int tx = 0; tx < tw; tx++) {
            int i = 4 * tx + (s & 3), j = 4 * ty + (s >> 2);
            float v = (i < c->w[k] && j < c->h[k]) ? c->low[k - 1][(size_t)j * c->w[k] + i] : pad;
            c->tiled[k - 1][((size_t)s * th + ty) * tw + tx] =
                meao_oracle_f32_to_f16(v, c->d->f16_rounding);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* pass 3: volumetric-obscurance render, one level  (REN main_interleaved)   */

typedef struct {
    const meao_oracle_desc *d; meao_oracle_render_consts k;
    const uint16_t *tiled; int sw, sh;      /* atlas slice dims (interleaved) / source dims (wide) */
    const float *flat;                      /* non-NULL: Render.main on a non-tiled f32 source    */
    void *out; int ow, oh;                  /* Occlusion<level> dims */
} ren_ctx;

/* slice texel with per-slice clamp addressing (REN:118-131 Gather + clamp); Render.main
 * (REN:116,121) gathers from the 2D source with the same clamp */
static inline float ren_tap(const ren_ctx *c, int s, int x, int y)
{
    x = clampi(x, 0, c->sw - 1); y = clampi(y, 0, c->sh - 1);
    if (c->flat) return c->flat[(size_t)y * c->sw + x];
    return meao_oracle_f16_to_f32(c->tiled[((size_t)s * c->sh + y) * c->sw + x]);
}

/* REN:60-75 */
static inline float test_sample_pair(const ren_ctx *c, int s, int cx, int cy, int dx, int dy,
                                     float front, float inv_range)
{
    float dis1 = mad(ren_tap(c, s, cx + dx, cy + dy), inv_range, -front);
    float dis2 = mad(ren_tap(c, s, cx - dx, cy - dy), inv_range, -front);
    float pse1 = sat(c->k.reject_fadeoff * dis1);
    float pse2 = sat(c->k.reject_fadeoff * dis2);
    float sum = clampf(dis1, pse2, 1.0f) + clampf(dis2, pse1, 1.0f);
    return sat(mad(-pse1, pse2, sum));
}

/* REN:77-110 with TILE_DIM offsets rewritten as (dx,dy) slice-texel offsets:
 * LDS offset o = dy*16 + dx. */
static inline float test_samples(const ren_ctx *c, int s, int cx, int cy, int x, int y,
                                 float inv_depth, float inv_thickness)
{
    if (c->flat) { x <<= 1; y <<= 1; }          /* WIDE_SAMPLING, REN:79-82 */
    float inv_range = inv_thickness * inv_depth;
    float front = inv_thickness - 0.5f;
    if (y == 0) {
        float p = test_sample_pair(c, s, cx, cy, x, 0, front, inv_range);
        float q = test_sample_pair(c, s, cx, cy, 0, x, front, inv_range);
        return 0.5f * (p + q);
    } else if (x == y) {
        float p = test_sample_pair(c, s, cx, cy, -x, x, front, inv_range);
        float q = test_sample_pair(c, s, cx, cy, x, x, front, inv_range);
        return 0.5f * (p + q);
    } else {
        float p = test_sample_pair(c, s, cx, cy, x, y, front, inv_range);
        float q = test_sample_pair(c, s, cx, cy, -x, y, front, inv_range);
        float r = test_sample_pair(c, s, cx, cy, y, x, front, inv_range);
        float t = test_sample_pair(c, s, cx, cy, -y, x, front, inv_range);
        return 0.25f * (((p + q) + r) + t);
    }
}

static void ren_rows(void *arg, int y0, int y1)
{
    ren_ctx *c = (ren_ctx *)arg;
    /* sample order and table indices of the 36-sample checker set, REN:162-168 */
    static const int sx36[7] = { 2, 4, 1, 2, 3, 1, 2 };
    static const int sy36[7] = { 0, 0, 1, 2, 3, 3, 4 };
    static const int ti36[7] = { 1, 3, 4, 8, 11, 6, 10 };
    /* SAMPLE_EXHAUSTIVELY, 68 samples, REN:146-157 */
    static const int sx68[12] = { 1, 2, 3, 4, 1, 2, 3, 1, 1, 1, 2, 2 };
    static const int sy68[12] = { 0, 0, 0, 0, 1, 2, 3, 2, 3, 4, 3, 4 };
    static const int ti68[12] = { 0, 1, 2, 3, 4, 8, 11, 5, 6, 7, 9, 10 };
    const int all = c->d->sample_set == MEAO_ORACLE_SAMPLES_EXHAUSTIVE;
    const int *sx = all ? sx68 : sx36, *sy = all ? sy68 : sy36, *ti = all ? ti68 : ti36;
    const int terms = all ? 12 : 7;
    for (int Y = y0; Y < y1; Y++) {
        for (int X = 0; X < c->ow; X++) {
            /* OutPixel = DTid.xy<<2 | (z&3, z>>2)  (REN:172)  inverted; Render.main: OutPixel = DTid.xy */
            int s = (X & 3) | ((Y & 3) << 2), cx = X >> 2, cy = Y >> 2;
            if (c->flat) { s = 0; cx = X; cy = Y; }
            float inv_depth = 1.0f / ren_tap(c, s, cx, cy);        /* REN:140 */
            float ao = 0.0f;
            for (int n = 0; n < terms; n++)
                ao = mad(c->k.sample_weight[ti[n]],
                         test_samples(c, s, cx, cy, sx[n], sy[n], inv_depth, c->k.inv_thickness[ti[n]]),
                         ao);
            float v = mad(c->k.intensity, ao - 1.0f, 1.0f);        /* lerp(1, ao, gIntensity) REN:176 */
            ao_store(c->out, (size_t)Y * c->ow + X, v, c->d->ao_format, c->d->f16_rounding);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* pass 4: depth-aware blur + bilateral upsample, one step  (UPS)            */

typedef struct {
    const meao_oracle_desc *d; meao_oracle_upsample_consts k;
    int lw, lh, hw, hh;
    const float *low_depth; const void *low_ao;
    const void *low_ao2;                     /* LoResAO2 of main_premin* (UPS:23,25), or NULL */
    const float *hi_depth32; const uint16_t *hi_depth16; const void *hi_ao;   /* hi_ao NULL: main */
    void *out;
    float *inv_depth;  /* [lh][lw]                 1/LoResDB           UPS:67 */
    float *ao;         /* [lh][lw]                 decoded LoResAO1           */
    float *hblur;      /* [lh][lw+2]   vx=-1..lw   AOCache2 contents  UPS:74  */
    float *vblur;      /* [lh+2][lw+2] v=-1..      AOCache1 after V   UPS:132 */
} ups_ctx;

/* UPS:83-87 */
static inline int compare_deltas(const ups_ctx *c, float d1, float d2, float l1, float l2)
{
    float t = mad(d1, d2, c->k.step_size);
    return t * t > (l1 * l2) * c->k.blur_tolerance;
}

/* One 5-tap output of BlurHorizontally / BlurVertically (UPS:89-170):
 * taps a[0..4] and inverse depths z[0..4] centred on index 2.  Each output of
 * the reference depends only on its own 5-wide window. */
static inline float smart_blur5(const ups_ctx *c, const float a[5], const float z[5])
{
    float d01 = z[1] - z[0], d12 = z[2] - z[1], d23 = z[3] - z[2], d34 = z[4] - z[3];
    float l01 = mad(d01, d01, c->k.step_size), l12 = mad(d12, d12, c->k.step_size);
    float l23 = mad(d23, d23, c->k.step_size), l34 = mad(d34, d34, c->k.step_size);
    int left   = compare_deltas(c, d01, d12, l01, l12);
    int middle = compare_deltas(c, d12, d23, l12, l23);
    int right  = compare_deltas(c, d23, d34, l23, l34);
    /* SmartBlur UPS:74-81 */
    float pc = a[2];
    float pb = (left | middle) ? a[1] : pc;
    float pa = left ? a[0] : pb;
    float pd = (right | middle) ? a[3] : pc;
    float pe = right ? a[4] : pd;
    return ((((pa + pe) * 0.5f + pb) + pc) + pd) * 0.25f;
}

static void ups_prefetch_rows(void *arg, int y0, int y1)
{   /* PrefetchData UPS:54-72, once per texel instead of once per tile slot */
    ups_ctx *c = (ups_ctx *)arg;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < c->lw; x++) {
            size_t i = (size_t)y * c->lw + x;
            c->ao[i] = ao_load(c->low_ao, i, c->d->ao_format);
            if (c->low_ao2)                                        /* COMBINE_LOWER_RESOLUTIONS UPS:58-60 */
                c->ao[i] = fminf(c->ao[i], ao_load(c->low_ao2, i, c->d->ao_format));
            c->inv_depth[i] = 1.0f / c->low_depth[i];
        }
}

static void ups_hblur_rows(void *arg, int y0, int y1)
{   /* Tile slot (ty,tx) holds low-res texel clamp(Gid*8-3+t) (UPS:191 + clamp);
     * AOCache2[row][col] is centred on tile col+2 = virtual x Gid*8-1+col. */
    ups_ctx *c = (ups_ctx *)arg;
    for (int y = y0; y < y1; y++)
        for (int vx = -1; vx <= c->lw; vx++) {
            float a[5], z[5];
            for (int t = 0; t < 5; t++) {
                int x = clampi(vx - 2 + t, 0, c->lw - 1);
                a[t] = c->ao[(size_t)y * c->lw + x];
                z[t] = c->inv_depth[(size_t)y * c->lw + x];
            }
            c->hblur[(size_t)y * (c->lw + 2) + (vx + 1)] = smart_blur5(c, a, z);
        }
}

static void ups_vblur_rows(void *arg, int r0, int r1)
{   /* rows r = vy+1, vy = -1..lh; taps are H-blurred rows clamp(vy-2..vy+2),
     * depths come from DepthCache column +2 (UPS:141-146) = same virtual x. */
    ups_ctx *c = (ups_ctx *)arg;
    for (int r = r0; r < r1; r++) {
        int vy = r - 1;
        for (int vx = -1; vx <= c->lw; vx++) {
            int xc = clampi(vx, 0, c->lw - 1);
            float a[5], z[5];
            for (int t = 0; t < 5; t++) {
                int y = clampi(vy - 2 + t, 0, c->lh - 1);
                a[t] = c->hblur[(size_t)y * (c->lw + 2) + (vx + 1)];
                z[t] = c->inv_depth[(size_t)y * c->lw + xc];
            }
            c->vblur[(size_t)r * (c->lw + 2) + (vx + 1)] = smart_blur5(c, a, z);
        }
    }
}

/* UPS:177-183 with the 4 low-res taps already in weight order 9,3,1,3 */
static inline float bilateral_upsample(const ups_ctx *c, float hi_depth, float hi_ao,
                                       const float lo_depth[4], const float lo_ao[4])
{
    static const float num[4] = { 9.0f, 3.0f, 1.0f, 3.0f };
    float w[4];
    for (int t = 0; t < 4; t++)
        w[t] = num[t] / (fabsf(hi_depth - lo_depth[t]) + c->k.upsample_tolerance);
    float total = ((w[0] + w[1]) + w[2]) + w[3];                 /* dot(weights, 1) */
    total = total + c->k.noise_filter_strength;
    float sum = lo_ao[0] * w[0];                                  /* dot(LowAO, weights): mul, mad x3 */
    sum = mad(lo_ao[1], w[1], sum);
    sum = mad(lo_ao[2], w[2], sum);
    sum = mad(lo_ao[3], w[3], sum);
    sum = sum + c->k.noise_filter_strength;
    return (hi_ao * sum) / total;
}

static void ups_bilateral_rows(void *arg, int y0, int y1)
{
    ups_ctx *c = (ups_ctx *)arg;
    /* Hi-res pixel (hx,hy) is written by dispatch thread D = ((hx+1)>>1,(hy+1)>>1)
     * through the Gather component selected by the parities (UPS:229-232):
     *   hx odd , hy even -> .x  taps (D.x-1,D.y) (D.x,D.y) (D.x,D.y-1) (D.x-1,D.y-1)
     *   hx even, hy even -> .y  rotated by one, .z by two, .w by three.
     * Gather order x=(c-1,c) y=(c,c) z=(c,c-1) w=(c-1,c-1) as (col,row). */
    static const int gx[4] = { -1, 0, 0, -1 };
    static const int gy[4] = { 0, 0, -1, -1 };
    for (int hy = y0; hy < y1; hy++)
        for (int hx = 0; hx < c->hw; hx++) {
            int Dx = (hx + 1) >> 1, Dy = (hy + 1) >> 1;
            int comp = (hx & 1) ? ((hy & 1) ? 3 : 0) : ((hy & 1) ? 2 : 1);
            float lo_depth[4], lo_ao[4];
            for (int t = 0; t < 4; t++) {
                int g = (comp + t) & 3;
                int vx = Dx + gx[g], vy = Dy + gy[g];
                /* LoResDB.Gather at corner D: clamp addressing (UPS:225) */
                lo_depth[t] = c->low_depth[(size_t)clampi(vy, 0, c->lh - 1) * c->lw + clampi(vx, 0, c->lw - 1)];
                /* AOCache1[Idx0 + ...]: blurred AO at the *virtual* texel (UPS:213-214) */
                lo_ao[t] = c->vblur[(size_t)(vy + 1) * (c->lw + 2) + (vx + 1)];
            }
            size_t hi = (size_t)hy * c->hw + hx;
            float hi_depth = c->hi_depth32 ? c->hi_depth32[hi] : meao_oracle_f16_to_f32(c->hi_depth16[hi]);
            float hi_ao = c->hi_ao ? ao_load(c->hi_ao, hi, c->d->ao_format) : 1.0f;   /* UPS:219-223 */
            ao_store(c->out, hi, bilateral_upsample(c, hi_depth, hi_ao, lo_depth, lo_ao),
                     c->d->ao_format, c->d->f16_rounding);
        }
}

static int upsample_pass(const meao_oracle_desc *d, int low_level, int nthreads,
                         const float *low_depth, const void *low_ao, const void *low_ao2,
                         const float *hi_depth32, const uint16_t *hi_depth16, const void *hi_ao,
                         void *out)
{
    ups_ctx c; memset(&c, 0, sizeof c);
    c.d = d;
    meao_oracle_upsample_constants(d, low_level, &c.k);
    meao_oracle_level_dims(d->width, d->height, low_level, &c.lw, &c.lh);
    meao_oracle_level_dims(d->width, d->height, low_level - 1, &c.hw, &c.hh);
    c.low_depth = low_depth; c.low_ao = low_ao; c.low_ao2 = low_ao2;
    c.hi_depth32 = hi_depth32; c.hi_depth16 = hi_depth16; c.hi_ao = hi_ao; c.out = out;
    size_t n = (size_t)c.lw * c.lh;
    c.inv_depth = (float *)malloc(n * 4);
    c.ao = (float *)malloc(n * 4);
    c.hblur = (float *)malloc((size_t)(c.lw + 2) * c.lh * 4);
    c.vblur = (float *)malloc((size_t)(c.lw + 2) * (c.lh + 2) * 4);
    int rc = -3;
    if (c.inv_depth && c.ao && c.hblur && c.vblur) {
        par_rows(nthreads, c.lh, ups_prefetch_rows, &c);
        par_rows(nthreads, c.lh, ups_hblur_rows, &c);
        par_rows(nthreads, c.lh + 2, ups_vblur_rows, &c);
        par_rows(nthreads, c.hh, ups_bilateral_rows, &c);
        rc = 0;
    }
    free(c.inv_depth); free(c.ao); free(c.hblur); free(c.vblur);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* composite: the raster blits that consume the AO texture (Blit.shader:66-134) */

int32_t meao_oracle_composite(int32_t width, int32_t height, int32_t ao_format, int32_t mode,
                              const void *ao, uint16_t *color, uint8_t *gbuffer0)
{
    if (!ao || !color || mode < 0 || mode > 2 || (mode == 1 && !gbuffer0)) return -1;
    size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; i++) {
        float a = ao_load(ao, i, ao_format);               /* tex2D(_AOTexture, uv).r, point sampled */
        uint16_t *c = color + 4 * i;
        if (mode == 2) {                                   /* pass 3: return ao in every channel */
            for (int k = 0; k < 4; k++) c[k] = meao_oracle_f32_to_f16(a, MEAO_ORACLE_F16_RTNE);
        } else if (mode == 0) {                            /* pass 2: Blend Zero SrcAlpha */
            for (int k = 0; k < 4; k++)
                c[k] = meao_oracle_f32_to_f16(meao_oracle_f16_to_f32(c[k]) * a, MEAO_ORACLE_F16_RTNE);
        } else {                                           /* pass 1: Blend Zero OneMinusSrc{Color,Alpha} */
            float occ = 1.0f - a;                          /* Blit.shader:84 */
            float keep = 1.0f - occ;
            for (int k = 0; k < 3; k++)                    /* gbuffer3 = (occ, occ, occ, 0) */
                c[k] = meao_oracle_f32_to_f16(meao_oracle_f16_to_f32(c[k]) * keep, MEAO_ORACLE_F16_RTNE);
            uint8_t *g = gbuffer0 + 4 * i + 3;             /* gbuffer0 = (0, 0, 0, occ) */
            *g = meao_oracle_f32_to_unorm8(meao_oracle_unorm8_to_f32(*g) * keep);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* whole pipeline in the order of RebuildCommandBuffers (AO.cs:496-531)      */

int32_t meao_oracle_run(const meao_oracle_desc *d, const void *depth,
                        meao_oracle_buffers *out, int32_t nthreads)
{
    if (!d || !depth || !out) return -1;
    if (d->depth_format < MEAO_ORACLE_DEPTH_F32 || d->depth_format > MEAO_ORACLE_DEPTH_F16) return -1;
    if (d->width < 1 || d->height < 1 || d->num_levels < 1 || d->num_levels > 4) return -1;
    if (d->ao_format != MEAO_ORACLE_AO_R8 && d->ao_format != MEAO_ORACLE_AO_F16) return -1;
    if (d->hq_levels < 0 || d->hq_levels > d->num_levels) return -1;
    if (d->sample_set != MEAO_ORACLE_SAMPLES_CHECKER && d->sample_set != MEAO_ORACLE_SAMPLES_EXHAUSTIVE) return -1;

    ds_ctx ds; memset(&ds, 0, sizeof ds);
    ds.d = d; ds.depth = depth;
    meao_oracle_zbuffer_params(d, ds.zp);
    for (int k = 0; k < 7; k++) meao_oracle_level_dims(d->width, d->height, k, &ds.w[k], &ds.h[k]);

    /* scratch for any buffer the caller did not ask for */
    void *owned[32]; int nowned = 0; int fail = 0;
#define GET(ptr, bytes) ((ptr) ? (void *)(ptr) : (owned[nowned] = malloc(bytes), fail |= !owned[nowned], owned[nowned++]))
    size_t ab = ao_bytes(d->ao_format);
    ds.linear = (uint16_t *)GET(out->linear_depth, (size_t)ds.w[0] * ds.h[0] * 2);
    void *occ[4], *comb[3], *res, *hq[4] = { NULL, NULL, NULL, NULL };
    for (int k = 1; k <= 4; k++) {
        ds.low[k - 1] = (float *)GET(out->low_depth[k - 1], (size_t)ds.w[k] * ds.h[k] * 4);
        ds.tiled[k - 1] = (uint16_t *)GET(out->tiled_depth[k - 1], (size_t)ds.w[k + 2] * ds.h[k + 2] * 16 * 2);
        occ[k - 1] = GET(out->occlusion[k - 1], (size_t)ds.w[k] * ds.h[k] * ab);
        if (k <= 3) comb[k - 1] = GET(out->combined[k - 1], (size_t)ds.w[k] * ds.h[k] * ab);
        if (k > d->num_levels - d->hq_levels && k <= d->num_levels)
            hq[k - 1] = GET(out->occlusion_hq[k - 1], (size_t)ds.w[k] * ds.h[k] * ab);
    }
    res = GET(out->result, (size_t)ds.w[0] * ds.h[0] * ab);
#undef GET
    int rc = fail ? -3 : 0;

    if (!rc) {
        par_rows(nthreads, ds.h[0], ds_linear_rows, &ds);
        for (int k = 1; k <= 4; k++) { ds_low_arg a = { &ds, k }; par_rows(nthreads, ds.h[k], ds_low_rows, &a); }
        for (int k = 1; k <= 4; k++) { ds_tile_arg a = { &ds, k }; par_rows(nthreads, 16 * ds.h[k + 2], ds_tile_rows, &a); }

        for (int k = 1; k <= d->num_levels; k++) {               /* AO.cs:519-522 */
            ren_ctx r; memset(&r, 0, sizeof r);
            r.d = d; meao_oracle_render_constants(d, k, &r.k);
            r.tiled = ds.tiled[k - 1]; r.sw = ds.w[k + 2]; r.sh = ds.h[k + 2];
            r.out = occ[k - 1]; r.ow = ds.w[k]; r.oh = ds.h[k];
            par_rows(nthreads, r.oh, ren_rows, &r);
            if (hq[k - 1]) {                                         /* Render.main on LowDepth<k> */
                memset(&r, 0, sizeof r);
                r.d = d; meao_oracle_render_constants_hq(d, k, &r.k);
                r.flat = ds.low[k - 1]; r.sw = ds.w[k]; r.sh = ds.h[k];
                r.out = hq[k - 1]; r.ow = ds.w[k]; r.oh = ds.h[k];
                par_rows(nthreads, r.oh, ren_rows, &r);
            }
        }

        /* AO.cs:528-531 generalised to num_levels: deepest rendered level is the
         * first low-res AO; each step blends with the next finer Occlusion. */
        const void *low_ao = occ[d->num_levels - 1];
        for (int hi = d->num_levels - 1; hi >= 1 && !rc; hi--) {
            rc = upsample_pass(d, hi + 1, nthreads, ds.low[hi], low_ao, hq[hi],
                               ds.low[hi - 1], NULL, occ[hi - 1], comb[hi - 1]);
            low_ao = comb[hi - 1];
        }
        if (!rc) rc = upsample_pass(d, 1, nthreads, ds.low[0], low_ao, hq[0], NULL, ds.linear, NULL, res);
    }
    for (int i = 0; i < nowned; i++) free(owned[i]);
    return rc;
}
