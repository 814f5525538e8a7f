/*
 * vr_oracle.c -- CPU restatement (plain C, scalar fp32, OpenMP over image rows) of
 * the reference's ray-cast hot path.  TEST INFRASTRUCTURE ONLY -- see vr_oracle.h
 * for who may use it and for the parity status ("PARITY UNPINNED" for the march
 * loop; RNG + loader pinned against the compiled reference).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  Build with -ffp-contract=off: every fp32 operation below is
 * individually rounded exactly as written; fused multiply-adds appear only where
 * written as fmaf().
 *
 * Arithmetic definitions for what OpenCL 1.2 leaves implementation-defined
 * (SURVEY.md App. B/C):
 *   native_divide(a,b) = a / b, fast_length/length = sqrtf(x*x + y*y + z*z),
 *   fast_normalize/normalize(v) = v * (1.0f / sqrtf(dot(v,v))), (0 for v == 0),
 *   dot = ((ax*bx) + (ay*by)) + (az*bz), min/max/clamp per the OpenCL spec text,
 *   native_powr = vro_powr() below,
 *   image reads per OpenCL 1.2 spec 8.2/8.3 (normalised coords, CLAMP_TO_EDGE +
 *   LINEAR or CLAMP + NEAREST): the linear filter is evaluated as nested lerps
 *   x -> y -> z with lerp(p,q,w) = fmaf(w, q - p, p) on the raw integer-valued
 *   texels, and the UNORM conversion is applied once to the filtered value as a
 *   multiplication by 1.0f/255.0f (1.0f/65535.0f) -- within the spec's 1.5-ulp
 *   allowance for normalised conversions (8.3.1.1) and its implementation-defined
 *   filter precision.
 */
#include "vr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ helpers */

typedef struct { float x, y, z; } f3;

static inline float vmin(float x, float y) { return y < x ? y : x; }   /* OpenCL min */
static inline float vmax(float x, float y) { return x < y ? y : x; }   /* OpenCL max */
static inline float vclamp(float x, float lo, float hi) { return vmin(vmax(x, lo), hi); }
static inline int iclamp(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline float dot3(f3 a, f3 b) { return ((a.x * b.x) + (a.y * b.y)) + (a.z * b.z); }
static inline float len3(f3 a) { return sqrtf(dot3(a, a)); }
static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
/*
 * LITERAL mode (vro_set_literal(1); CPU tests only): every place where the parity definitions
 * above re-associate, re-factor or approximate what the reference's source text says is
 * evaluated the way the text says instead -- gradients as six (27) full read_imagef at
 * pos -+ 1/res (volumeraycast.cl:159-178, :217-277), image reads with the per-texel
 * normalised conversion c / 255.0f (c / 65535.0f) BEFORE the filter and the filter as the
 * weighted sum of OpenCL 1.2 spec 8.2, native_powr / log / sin / cos / atan2 / acos by libm,
 * normalize (:289) as v / |v|.  tests/test_oracle_literal.py bounds |literal - parity| <= 1e-4 per
 * channel on every parity scene: the definitions the HIP kernel shares with this file are
 * then within north_star's tolerance of the literal reading, not merely equal to each other.
 */
static int g_literal = 0;
void vro_set_literal(int on) { g_literal = on ? 1 : 0; }
int vro_get_literal(void) { return g_literal; }

static inline f3 normalize3(f3 v)
{
    float d = dot3(v, v);
    if (d == 0.0f) return mk3(0.0f, 0.0f, 0.0f); /* SURVEY C5 */
    float inv = 1.0f / sqrtf(d);
    return scale3(v, inv);
}
/* `normalize` proper (only :289, the half vector): LITERAL mode divides by the length.  The
 * fast_normalize / fast_length / native_divide calls have no literal form -- OpenCL leaves
 * their precision to the implementation -- and keep the parity definition in both modes. */
static inline f3 normalize3_full(f3 v)
{
    if (!g_literal) return normalize3(v);
    float l = sqrtf(dot3(v, v));
    if (l == 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    return mk3(v.x / l, v.y / l, v.z / l);
}
static inline float lerpf(float p, float q, float w) { return fmaf(w, q - p, p); }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ---------------------------------------------------------------------- RNG */

/* random.cl:2-13 */
uint32_t vro_parallel_rng(uint32_t x)
{
    uint32_t value = x;
    value = (value ^ 61u) ^ (value >> 16);
    value *= 9u;
    value ^= value << 4;
    value *= 0x27d4eb2du;
    value ^= value >> 15;
    return value;
}

/* random.cl:22-28 */
uint32_t vro_parallel_rng3(uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t value = vro_parallel_rng(x);
    value = vro_parallel_rng(y ^ value);
    value = vro_parallel_rng(z ^ value);
    return value;
}

/* random.cl:44-47: (float)v / (float)UINT_MAX, (float)UINT_MAX == 2^32 */
float vro_map_uint_float(uint32_t v) { return (float)v / 4294967296.0f; }

/* -------------------------------------------------------------------- powr */

/*
 * Parity definition of native_powr(x, y) for x >= 0 (volumeraycast.cl:290,864):
 * exp(y * log(x)) with Cephes-style single-precision kernels written as explicit
 * fmaf chains, so the HIP kernel can execute the identical operation sequence.
 * powr(1, y) == 1 and powr(0, y > 0) == 0 exactly.
 */
float vro_powr(float x, float y)
{
    if (g_literal) return powf(x, y);
    if (!(x > 0.0f)) return (x == 0.0f) ? 0.0f : NAN;
    /* ---- log(x) = e*ln2 + log(m), m in [sqrt(1/2), sqrt(2)) */
    uint32_t ux = f2u(x);
    int e = 0;
    if (ux < 0x00800000u) { /* denormal: scale by 2^23 */
        x = x * 8388608.0f;
        ux = f2u(x);
        e = -23;
    }
    e += (int)(ux >> 23) - 126;
    float m = u2f((ux & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
    if (m < 0.70710678118654752440f) {
        e -= 1;
        m = m + m;
    }
    float f = m - 1.0f;
    float z = f * f;
    float p = 7.0376836292E-2f;
    p = fmaf(p, f, -1.1514610310E-1f);
    p = fmaf(p, f, 1.1676998740E-1f);
    p = fmaf(p, f, -1.2420140846E-1f);
    p = fmaf(p, f, 1.4249322787E-1f);
    p = fmaf(p, f, -1.6668057665E-1f);
    p = fmaf(p, f, 2.0000714765E-1f);
    p = fmaf(p, f, -2.4999993993E-1f);
    p = fmaf(p, f, 3.3333331174E-1f);
    p = (p * f) * z;
    float fe = (float)e;
    p = fmaf(fe, -2.12194440e-4f, p);
    p = fmaf(z, -0.5f, p);
    float lg = f + p;
    lg = fmaf(fe, 0.693359375f, lg);
    /* ---- exp(y * lg) */
    float t = y * lg;
    if (t > 88.0f) return INFINITY;
    if (t < -87.0f) return 0.0f;
    float n = floorf(fmaf(t, 1.44269504088896341f, 0.5f));
    t = fmaf(n, -0.693359375f, t);
    t = fmaf(n, 2.12194440e-4f, t);
    float q = 1.9875691500E-4f;
    q = fmaf(q, t, 1.3981999507E-3f);
    q = fmaf(q, t, 8.3334519073E-3f);
    q = fmaf(q, t, 4.1665795894E-2f);
    q = fmaf(q, t, 1.6666665459E-1f);
    q = fmaf(q, t, 5.0000001201E-1f);
    float r = fmaf(q, t * t, t) + 1.0f;
    int ni = (int)n; /* in [-126, 127] given the clamps above */
    return r * u2f((uint32_t)(ni + 127) << 23);
}

/* --------------------------------------------------------------- host side */

/* volumerendercl.cpp:39-54 */
uint32_t vro_round_pow2(uint32_t n)
{
    uint32_t val = n - 1u;
    val |= val >> 1;
    val |= val >> 2;
    val |= val >> 4;
    val |= val >> 8;
    val |= val >> 16;
    val++;
    uint32_t x = val >> 1;
    return (val - n) > (n - x) ? x : val;
}

/* volumerendercl.cpp:620-636 */
void vro_brick_layout(const uint32_t res[3], uint32_t edge[3], float brick_res_f[3],
                      uint32_t tex[3])
{
    for (int i = 0; i < 3; ++i) {
        uint32_t e = vro_round_pow2(res[i] / 64u);
        edge[i] = e > 1u ? e : 1u;
        brick_res_f[i] = (float)res[i] / (float)edge[i];
        tex[i] = (uint32_t)ceil((double)brick_res_f[i]);
    }
}

/* volumerendercl.cpp:347-362 (std::valarray<float> arithmetic) */
void vro_calc_scaling(const uint32_t res[3], const double thickness[3], float model_scale[3])
{
    float th[3] = {(float)thickness[0], (float)thickness[1], (float)thickness[2]};
    float inv0 = 1.f / th[0];
    float s[3], mx;
    for (int i = 0; i < 3; ++i) s[i] = (float)res[i] * (th[i] * inv0);
    mx = s[0];
    if (s[1] > mx) mx = s[1];
    if (s[2] > mx) mx = s[2];
    for (int i = 0; i < 3; ++i) model_scale[i] = mx / s[i];
}

/* volumerendercl.cpp:879-884 */
void vro_prefix_sum(const uint8_t *tff_rgba, uint32_t n, uint32_t *prefix)
{
    uint32_t acc = 0;
    for (uint32_t i = 0; i < n; ++i) {
        acc += tff_rgba[4 * (size_t)i + 3];
        prefix[i] = acc;
    }
}

/* volumerendercl.cpp:513-514: width + (LOCAL_SIZE - width % LOCAL_SIZE) */
uint32_t vro_padded(uint32_t n) { return n + (8u - n % 8u); }

int vro_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------ image reads */

typedef struct {
    const vro_scene *s;
    int w, h, d;            /* volume dims */
    float fw, fh, fd;
    float inv_max;          /* UNORM scale: 1/255, 1/65535 or 1 */
    size_t row, slice;
    int bw, bh, bd;         /* brick image dims */
    uint8_t *touched;       /* micro-brick bitmap or NULL */
    int mbx, mby;
    int nch;                /* channels per voxel: 1 (CL_R), 2 (CL_RG), 4 (CL_RGBA) */
} vol_t;

static inline float vox_raw_c(const vol_t *v, int x, int y, int z, int ch)
{
    size_t i = ((size_t)z * v->slice + (size_t)y * v->row + (size_t)x) * (size_t)v->nch + (size_t)ch;
    if (v->touched) {
        size_t b = ((size_t)(z >> 2) * (size_t)v->mby + (size_t)(y >> 2)) * (size_t)v->mbx +
                   (size_t)(x >> 2);
        uint8_t bit = (uint8_t)(1u << (b & 7));
        if (!(__atomic_load_n(&v->touched[b >> 3], __ATOMIC_RELAXED) & bit))
            __atomic_fetch_or(&v->touched[b >> 3], bit, __ATOMIC_RELAXED);
    }
    switch (v->s->format) {
    case VRO_UCHAR: return (float)((const uint8_t *)v->s->voxels)[i];
    case VRO_USHORT: return (float)((const uint16_t *)v->s->voxels)[i];
    default: return ((const float *)v->s->voxels)[i];
    }
}

/* the .x component every single-channel reader of the kernel looks at */
static inline float vox_raw(const vol_t *v, int x, int y, int z) { return vox_raw_c(v, x, y, z, 0); }

/* LITERAL mode: a texel as the image unit hands it to the filter -- CL_UNORM_INT8/16 converted
 * with c / 255.0f (c / 65535.0f), OpenCL 1.2 spec 8.3.1.1; CL_FLOAT as stored */
static inline float vox_norm_c(const vol_t *v, int x, int y, int z, int ch)
{
    float raw = vox_raw_c(v, x, y, z, ch);
    switch (v->s->format) {
    case VRO_UCHAR: return raw / 255.0f;
    case VRO_USHORT: return raw / 65535.0f;
    default: return raw;
    }
}

/* linearSmp (volumeraycast.cl:30-31) on the CL_R volume: OpenCL 1.2 spec 8.2,
 * normalised coords, CLAMP_TO_EDGE, LINEAR. */
static float vol_linear_c(const vol_t *v, float px, float py, float pz, int ch)
{
    float u = px * v->fw, vv = py * v->fh, ww = pz * v->fd;
    float ub = u - 0.5f, vb = vv - 0.5f, wb = ww - 0.5f;
    float fx = floorf(ub), fy = floorf(vb), fz = floorf(wb);
    float a = ub - fx, b = vb - fy, c = wb - fz;
    int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    int x0 = iclamp(ix, 0, v->w - 1), x1 = iclamp(ix + 1, 0, v->w - 1);
    int y0 = iclamp(iy, 0, v->h - 1), y1 = iclamp(iy + 1, 0, v->h - 1);
    int z0 = iclamp(iz, 0, v->d - 1), z1 = iclamp(iz + 1, 0, v->d - 1);
    if (g_literal) {
        /* spec 8.2: T = (1-a)(1-b)(1-c) T000 + a(1-b)(1-c) T100 + ... + a b c T111 */
        float a1 = 1.0f - a, b1 = 1.0f - b, c1 = 1.0f - c;
        return a1 * b1 * c1 * vox_norm_c(v, x0, y0, z0, ch) + a * b1 * c1 * vox_norm_c(v, x1, y0, z0, ch) +
               a1 * b * c1 * vox_norm_c(v, x0, y1, z0, ch) + a * b * c1 * vox_norm_c(v, x1, y1, z0, ch) +
               a1 * b1 * c * vox_norm_c(v, x0, y0, z1, ch) + a * b1 * c * vox_norm_c(v, x1, y0, z1, ch) +
               a1 * b * c * vox_norm_c(v, x0, y1, z1, ch) + a * b * c * vox_norm_c(v, x1, y1, z1, ch);
    }
    float c00 = lerpf(vox_raw_c(v, x0, y0, z0, ch), vox_raw_c(v, x1, y0, z0, ch), a);
    float c10 = lerpf(vox_raw_c(v, x0, y1, z0, ch), vox_raw_c(v, x1, y1, z0, ch), a);
    float c01 = lerpf(vox_raw_c(v, x0, y0, z1, ch), vox_raw_c(v, x1, y0, z1, ch), a);
    float c11 = lerpf(vox_raw_c(v, x0, y1, z1, ch), vox_raw_c(v, x1, y1, z1, ch), a);
    float c0 = lerpf(c00, c10, b);
    float c1 = lerpf(c01, c11, b);
    return lerpf(c0, c1, c) * v->inv_max;
}

static float vol_linear(const vol_t *v, float px, float py, float pz)
{
    return vol_linear_c(v, px, py, pz, 0);
}

/* nearestSmp (volumeraycast.cl:32-33): normalised, CLAMP (border 0 for CL_R), NEAREST */
static float vol_nearest_c(const vol_t *v, float px, float py, float pz, int ch)
{
    float fx = floorf(px * v->fw), fy = floorf(py * v->fh), fz = floorf(pz * v->fd);
    if (!(fx >= 0.0f && fx <= (float)(v->w - 1) && fy >= 0.0f && fy <= (float)(v->h - 1) &&
          fz >= 0.0f && fz <= (float)(v->d - 1)))
        return 0.0f;
    if (g_literal) return vox_norm_c(v, (int)fx, (int)fy, (int)fz, ch);
    return vox_raw_c(v, (int)fx, (int)fy, (int)fz, ch) * v->inv_max;
}

static float vol_nearest(const vol_t *v, float px, float py, float pz)
{
    return vol_nearest_c(v, px, py, pz, 0);
}

/* TF read: read_imagef(tffData, linearSmp, x) on the RGBA8 1-D image
 * (volumeraycast.cl:777,808); UNORM8 -> c / 255.0f. */
static void tff_linear(const vro_scene *s, float x, float out[4])
{
    int n = (int)s->tff_n;
    float ub = x * (float)n - 0.5f;
    float fl = floorf(ub);
    float a = ub - fl;
    int i = (int)fl;
    int i0 = iclamp(i, 0, n - 1), i1 = iclamp(i + 1, 0, n - 1);
    for (int c = 0; c < 4; ++c) {
        float t0 = (float)s->tff[4 * (size_t)i0 + c] / 255.0f;
        float t1 = (float)s->tff[4 * (size_t)i1 + c] / 255.0f;
        out[c] = g_literal ? (1.0f - a) * t0 + a * t1 : lerpf(t0, t1, a);
    }
}

/* read_imageui(tffPrefix, nearestSmp, x).x (volumeraycast.cl:780-781): normalised,
 * CLAMP, NEAREST -> border 0 outside [0, n-1]. */
static uint32_t prefix_nearest(const vro_scene *s, float x)
{
    float fi = floorf(x * (float)s->prefix_n);
    if (!(fi >= 0.0f && fi <= (float)(s->prefix_n - 1))) return 0u;
    return s->prefix[(int)fi];
}

/* read_imagef(volBrickData, (int4)(cell,0)).xy (volumeraycast.cl:765); out-of-range
 * coordinates are undefined in OpenCL: defined as (0,0). */
static void brick_minmax(const vol_t *v, int cx, int cy, int cz, float *mn, float *mx)
{
    if (cx < 0 || cy < 0 || cz < 0 || cx >= v->bw || cy >= v->bh || cz >= v->bd) {
        *mn = 0.0f;
        *mx = 0.0f;
        return;
    }
    size_t i = 2 * (((size_t)cz * (size_t)v->bh + (size_t)cy) * (size_t)v->bw + (size_t)cx);
    switch (v->s->format) {
    case VRO_UCHAR:
        *mn = (float)((const uint8_t *)v->s->bricks)[i] * v->inv_max;
        *mx = (float)((const uint8_t *)v->s->bricks)[i + 1] * v->inv_max;
        if (g_literal) {
            *mn = (float)((const uint8_t *)v->s->bricks)[i] / 255.0f;
            *mx = (float)((const uint8_t *)v->s->bricks)[i + 1] / 255.0f;
        }
        break;
    case VRO_USHORT:
        *mn = (float)((const uint16_t *)v->s->bricks)[i] * v->inv_max;
        *mx = (float)((const uint16_t *)v->s->bricks)[i + 1] * v->inv_max;
        if (g_literal) {
            *mn = (float)((const uint16_t *)v->s->bricks)[i] / 65535.0f;
            *mx = (float)((const uint16_t *)v->s->bricks)[i + 1] / 65535.0f;
        }
        break;
    default:
        *mn = ((const float *)v->s->bricks)[i];
        *mx = ((const float *)v->s->bricks)[i + 1];
    }
}

/* ------------------------------------------------------- device functions */

/* volumeraycast.cl:122-142 */
int vro_intersect_bbox(const float o[3], const float d[3], const float lower[3],
                       const float upper[3], float *tnear, float *tfar)
{
    float tmin[3], tmax[3];
    for (int i = 0; i < 3; ++i) {
        float inv = 1.0f / d[i];
        float tbot = inv * (lower[i] - o[i]);
        float ttop = inv * (upper[i] - o[i]);
        tmin[i] = vmin(ttop, tbot);
        tmax[i] = vmax(ttop, tbot);
    }
    float maxTmin = vmax(vmax(tmin[0], tmin[1]), vmax(tmin[0], tmin[2]));
    float minTmax = vmin(vmin(tmax[0], tmax[1]), vmin(tmax[0], tmax[2]));
    *tnear = maxTmin;
    *tfar = minTmax;
    return minTmax > maxTmin;
}

/* Trilinear blend of the 2x2x2 texels (xi, yi, zi) with weights (a, b, c): the filter of
 * vol_linear() on explicit texel indices. */
static float vol_tri(const vol_t *v, const int xi[2], const int yi[2], const int zi[2], float a,
                     float b, float c)
{
    float c00 = lerpf(vox_raw(v, xi[0], yi[0], zi[0]), vox_raw(v, xi[1], yi[0], zi[0]), a);
    float c10 = lerpf(vox_raw(v, xi[0], yi[1], zi[0]), vox_raw(v, xi[1], yi[1], zi[0]), a);
    float c01 = lerpf(vox_raw(v, xi[0], yi[0], zi[1]), vox_raw(v, xi[1], yi[0], zi[1]), a);
    float c11 = lerpf(vox_raw(v, xi[0], yi[1], zi[1]), vox_raw(v, xi[1], yi[1], zi[1]), a);
    float c0 = lerpf(c00, c10, b);
    float c1 = lerpf(c01, c11, b);
    return lerpf(c0, c1, c) * v->inv_max;
}

/* volumeraycast.cl:159-178; returns -gradient.xyz as used by the caller (:814).
 *
 * The six taps sit at pos -+ 1/volRes per axis (:162-171), i.e. exactly one texel away
 * from the centre sample.  They are evaluated in texel space: the centre sample's filter
 * weights (a, b, c) with the texel indices shifted by -+1 and clamped to the edge
 * (CLAMP_TO_EDGE).  This equals the literal form up to the fp32 rounding of the
 * coordinate add (<= 2^-12 texel at 2048^3, below the 8-bit weight precision of the
 * texture units the reference ran on) and is the parity definition shared with the HIP
 * kernel, which reuses the 4x4x4-neighbourhood loads between taps (32 instead of 56). */
static float vol_linear(const vol_t *v, float px, float py, float pz);

/* LITERAL mode: s1 / s2 of gradientCentralDiff exactly as written (:161-171): six full image
 * reads at pos -+ offset, offset = 1 / volRes */
static void central_diff_taps_literal(const vol_t *v, f3 pos, f3 *s1, f3 *s2)
{
    f3 off = mk3(1.0f / v->fw, 1.0f / v->fh, 1.0f / v->fd);
    s1->x = vol_linear(v, pos.x + (-off.x), pos.y + 0.0f, pos.z + 0.0f);
    s1->y = vol_linear(v, pos.x + 0.0f, pos.y + (-off.y), pos.z + 0.0f);
    s1->z = vol_linear(v, pos.x + 0.0f, pos.y + 0.0f, pos.z + (-off.z));
    s2->x = vol_linear(v, pos.x + off.x, pos.y + 0.0f, pos.z + 0.0f);
    s2->y = vol_linear(v, pos.x + 0.0f, pos.y + off.y, pos.z + 0.0f);
    s2->z = vol_linear(v, pos.x + 0.0f, pos.y + 0.0f, pos.z + off.z);
}

static f3 neg_gradient_central_diff(const vol_t *v, f3 pos)
{
    if (g_literal) {
        f3 s1, s2;
        central_diff_taps_literal(v, pos, &s1, &s2);
        f3 g = sub3(s2, s1);
        f3 n = normalize3(g);
        if (len3(n) == 0.0f) n = mk3(0.57735f, 0.57735f, 0.57735f);   /* :174-175 */
        return neg3(n);
    }
    float ub = pos.x * v->fw - 0.5f, vb = pos.y * v->fh - 0.5f, wb = pos.z * v->fd - 0.5f;
    float fx = floorf(ub), fy = floorf(vb), fz = floorf(wb);
    float a = ub - fx, b = vb - fy, c = wb - fz;
    int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    int X[4], Y[4], Z[4]; /* texel indices i-1, i, i+1, i+2, clamped to the edge */
    for (int k = 0; k < 4; ++k) {
        X[k] = iclamp(ix - 1 + k, 0, v->w - 1);
        Y[k] = iclamp(iy - 1 + k, 0, v->h - 1);
        Z[k] = iclamp(iz - 1 + k, 0, v->d - 1);
    }
    f3 s1, s2;
    s1.x = vol_tri(v, X + 0, Y + 1, Z + 1, a, b, c);
    s2.x = vol_tri(v, X + 2, Y + 1, Z + 1, a, b, c);
    s1.y = vol_tri(v, X + 1, Y + 0, Z + 1, a, b, c);
    s2.y = vol_tri(v, X + 1, Y + 2, Z + 1, a, b, c);
    s1.z = vol_tri(v, X + 1, Y + 1, Z + 0, a, b, c);
    s2.z = vol_tri(v, X + 1, Y + 1, Z + 2, a, b, c);
    f3 g = sub3(s2, s1);
    f3 n = normalize3(g);
    if (dot3(g, g) == 0.0f) /* length(normal) == 0 <=> g == 0, given normalize(0) = 0 */
        n = mk3(0.57735f, 0.57735f, 0.57735f);
    return neg3(n);
}

/* |s2 - s1| of gradientCentralDiff (:177, fast_length), same texel-space taps: the value
 * illumType 4 feeds to the transfer function (:796-799). */
static float gradient_central_diff_len(const vol_t *v, f3 pos)
{
    if (g_literal) {
        f3 s1, s2;
        central_diff_taps_literal(v, pos, &s1, &s2);
        return len3(sub3(s2, s1));
    }
    float ub = pos.x * v->fw - 0.5f, vb = pos.y * v->fh - 0.5f, wb = pos.z * v->fd - 0.5f;
    float fx = floorf(ub), fy = floorf(vb), fz = floorf(wb);
    float a = ub - fx, b = vb - fy, c = wb - fz;
    int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    int X[4], Y[4], Z[4];
    for (int k = 0; k < 4; ++k) {
        X[k] = iclamp(ix - 1 + k, 0, v->w - 1);
        Y[k] = iclamp(iy - 1 + k, 0, v->h - 1);
        Z[k] = iclamp(iz - 1 + k, 0, v->d - 1);
    }
    f3 s1, s2;
    s1.x = vol_tri(v, X + 0, Y + 1, Z + 1, a, b, c);
    s2.x = vol_tri(v, X + 2, Y + 1, Z + 1, a, b, c);
    s1.y = vol_tri(v, X + 1, Y + 0, Z + 1, a, b, c);
    s2.y = vol_tri(v, X + 1, Y + 2, Z + 1, a, b, c);
    s1.z = vol_tri(v, X + 1, Y + 1, Z + 0, a, b, c);
    s2.z = vol_tri(v, X + 1, Y + 1, Z + 2, a, b, c);
    return len3(sub3(s2, s1));
}

/* -gradientSobel(vol, pos).xyz (volumeraycast.cl:217-277, :824).  The 27 taps sit at
 * pos + offset * (i, j, k), i.e. whole texels from the centre sample: texel space again (the
 * centre's weights on indices shifted by i, j, k and clamped).  Weights of the reference's table:
 * d = (-1, 0, 1), s = (1, 2, 1); x: d[i] s[j] s[k], y: s[i] d[j] s[k], z: s[i] s[j] d[k].  Sums
 * run in the reference's loop order (i outer, k inner), every term included. */
static f3 neg_gradient_sobel(const vol_t *v, f3 pos)
{
    static const float dw[3] = {-1.f, 0.f, 1.f}, sw[3] = {1.f, 2.f, 1.f};
    float ub = pos.x * v->fw - 0.5f, vb = pos.y * v->fh - 0.5f, wb = pos.z * v->fd - 0.5f;
    float fx = floorf(ub), fy = floorf(vb), fz = floorf(wb);
    float a = ub - fx, b = vb - fy, c = wb - fz;
    int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (g_literal) { /* :254-269: 27 full image reads at pos + offset * (i, j, k) */
        f3 off = mk3(1.0f / v->fw, 1.0f / v->fh, 1.0f / v->fd);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k < 3; ++k) {
                    float smp = vol_linear(v, pos.x + off.x * (float)(i - 1), pos.y + off.y * (float)(j - 1),
                                           pos.z + off.z * (float)(k - 1));
                    gx = gx + ((dw[i] * sw[j]) * sw[k]) * smp;
                    gy = gy + ((sw[i] * dw[j]) * sw[k]) * smp;
                    gz = gz + ((sw[i] * sw[j]) * dw[k]) * smp;
                }
        f3 gl = mk3(gx / 27.f, gy / 27.f, gz / 27.f);
        if (len3(gl) == 0.f) gl = mk3(1.f, 1.f, 1.f);
        return neg3(normalize3(gl));
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) {
                int X[2] = {iclamp(ix - 1 + i, 0, v->w - 1), iclamp(ix + i, 0, v->w - 1)};
                int Y[2] = {iclamp(iy - 1 + j, 0, v->h - 1), iclamp(iy + j, 0, v->h - 1)};
                int Z[2] = {iclamp(iz - 1 + k, 0, v->d - 1), iclamp(iz + k, 0, v->d - 1)};
                float smp = vol_tri(v, X, Y, Z, a, b, c);
                gx = gx + ((dw[i] * sw[j]) * sw[k]) * smp;
                gy = gy + ((sw[i] * dw[j]) * sw[k]) * smp;
                gz = gz + ((sw[i] * sw[j]) * dw[k]) * smp;
            }
    f3 g = mk3(gx / 27.f, gy / 27.f, gz / 27.f); /* :270 */
    if (len3(g) == 0.f) g = mk3(1.f, 1.f, 1.f); /* :272-273 */
    return neg3(normalize3(g));
}

/* celShading, volumeraycast.cl:306-319 */
static f3 cel_shading(f3 color, f3 toLightDir, f3 n)
{
    f3 l = normalize3(toLightDir);
    float intensity = vmax(0.f, dot3(n, l));
    if (intensity > 0.95f) return color;
    if (intensity > 0.5f) return scale3(color, 0.6f);
    if (intensity > 0.25f) return scale3(color, 0.4f);
    return scale3(color, 0.2f);
}

/* volumeraycast.cl:280-291 with lightColor = materialColor = 1, exponent 40 */
static float specular_blinn_phong(f3 normal, f3 toLightDir, f3 toCameraDir)
{
    f3 h = add3(toCameraDir, toLightDir);
    if (dot3(h, h) < 1.e-6f) return 0.0f;
    h = normalize3_full(h);
    return (1.0f * 1.0f) * vro_powr(vmax(dot3(normal, h), 0.f), 40.f);
}

/* volumeraycast.cl:294-303 */
static f3 illumination(f3 color, f3 toLightDir, f3 n)
{
    f3 l = normalize3(toLightDir);
    f3 amb = scale3(color, 0.15f);
    f3 diff = scale3(scale3(color, vmax(0.f, dot3(n, l))), 0.7f);
    float sp = specular_blinn_phong(n, l, toLightDir) * 0.15f;
    return mk3((amb.x + diff.x) + sp, (amb.y + diff.y) + sp, (amb.z + diff.z) + sp);
}

/* generateBricks, volumeraycast.cl:932-961 */
int vro_generate_bricks(const void *voxels, const uint32_t res[3], int format,
                        const uint32_t tex[3], void *out)
{
    if (!voxels || !out || format < 0 || format > 2) return -1;
    int vpc[3];
    for (int i = 0; i < 3; ++i) vpc[i] = (int)ceilf((float)res[i] / (float)tex[i]);
    size_t row = res[0], slice = (size_t)res[0] * res[1];
#pragma omp parallel for collapse(2) schedule(static)
    for (int cz = 0; cz < (int)tex[2]; ++cz)
        for (int cy = 0; cy < (int)tex[1]; ++cy)
            for (int cx = 0; cx < (int)tex[0]; ++cx) {
                int lo[3] = {vpc[0] * cx, vpc[1] * cy, vpc[2] * cz};
                int hi[3];
                for (int i = 0; i < 3; ++i) hi[i] = iclamp(lo[i] + vpc[i], 0, (int)res[i] - 1);
                size_t o = 2 * (((size_t)cz * tex[1] + cy) * tex[0] + cx);
                if (format == VRO_FLOAT) {
                    float mn = 1.f, mx = 0.f; /* :945-946 */
                    for (int k = lo[2]; k < hi[2]; ++k)
                        for (int j = lo[1]; j < hi[1]; ++j)
                            for (int i = lo[0]; i < hi[0]; ++i) {
                                float val = ((const float *)voxels)[k * slice + j * row + i];
                                mn = vmin(mn, val);
                                mx = vmax(mx, val);
                            }
                    ((float *)out)[o] = mn;
                    ((float *)out)[o + 1] = mx;
                } else {
                    /* min = 1.0, max = 0.0 stored as UNORM: raw 255/65535 and 0 */
                    uint32_t top = format == VRO_UCHAR ? 255u : 65535u;
                    uint32_t mn = top, mx = 0u;
                    for (int k = lo[2]; k < hi[2]; ++k)
                        for (int j = lo[1]; j < hi[1]; ++j)
                            for (int i = lo[0]; i < hi[0]; ++i) {
                                size_t idx = k * slice + j * row + i;
                                uint32_t val = format == VRO_UCHAR
                                                   ? ((const uint8_t *)voxels)[idx]
                                                   : ((const uint16_t *)voxels)[idx];
                                if (val < mn) mn = val;
                                if (val > mx) mx = val;
                            }
                    if (format == VRO_UCHAR) {
                        ((uint8_t *)out)[o] = (uint8_t)mn;
                        ((uint8_t *)out)[o + 1] = (uint8_t)mx;
                    } else {
                        ((uint16_t *)out)[o] = (uint16_t)mn;
                        ((uint16_t *)out)[o + 1] = (uint16_t)mx;
                    }
                }
            }
    return 0;
}

/* downsampling, volumeraycast.cl:966-994 with the host's size rule (volumerendercl.cpp:245-251):
 * low-res size = ceil(res / factor); every low-res voxel is the sum of the normalised values of
 * its ceil(res / lowres)^3 box (clipped to the volume) divided by the FULL box volume, written
 * back in the volume's type (UNORM: saturate, round to nearest even). */
int vro_downsample(const void *voxels, const uint32_t res[3], int format, int factor, void *out,
                   uint32_t out_res[3])
{
    if (!voxels || !out || format < 0 || format > 2 || factor < 2) return -1;
    uint32_t lo[3];
    int vpc[3];
    for (int i = 0; i < 3; ++i) {
        lo[i] = (uint32_t)ceil((double)res[i] / (double)factor);
        vpc[i] = (int)ceilf((float)res[i] / (float)lo[i]);
        out_res[i] = lo[i];
    }
    const float inv_max = format == VRO_UCHAR ? 1.0f / 255.0f : format == VRO_USHORT ? 1.0f / 65535.0f : 1.0f;
    const size_t row = res[0], slice = (size_t)res[0] * res[1];
#pragma omp parallel for collapse(2) schedule(static)
    for (int cz = 0; cz < (int)lo[2]; ++cz)
        for (int cy = 0; cy < (int)lo[1]; ++cy)
            for (int cx = 0; cx < (int)lo[0]; ++cx) {
                int l[3] = {vpc[0] * cx, vpc[1] * cy, vpc[2] * cz}, u[3];
                for (int i = 0; i < 3; ++i) u[i] = iclamp(l[i] + vpc[i], 0, (int)res[i]);
                float value = 0.f;
                for (int k = l[2]; k < u[2]; ++k)
                    for (int j = l[1]; j < u[1]; ++j)
                        for (int i = l[0]; i < u[0]; ++i) {
                            size_t idx = k * slice + j * row + i;
                            float raw = format == VRO_UCHAR ? (float)((const uint8_t *)voxels)[idx]
                                        : format == VRO_USHORT ? (float)((const uint16_t *)voxels)[idx]
                                                               : ((const float *)voxels)[idx];
                            value += raw * inv_max;
                        }
                value /= (float)(vpc[0] * vpc[1] * vpc[2]);
                size_t o = ((size_t)cz * lo[1] + cy) * lo[0] + cx;
                if (format == VRO_FLOAT) ((float *)out)[o] = value;
                else if (format == VRO_UCHAR)
                    ((uint8_t *)out)[o] = (uint8_t)rintf(vclamp(value * 255.0f, 0.f, 255.0f));
                else
                    ((uint16_t *)out)[o] = (uint16_t)rintf(vclamp(value * 65535.0f, 0.f, 65535.0f));
            }
    return 0;
}

/* ---------------------------------------------------------- path tracer */

#define VRO_PI_F 3.14159274101257f /* M_PI_F */

static inline int in_volume(f3 p) /* volumeraycast.cl:93-96 */
{
    return vmax(fabsf(p.x), vmax(fabsf(p.y), fabsf(p.z))) < 1.f;
}

/* Parity definitions of log / sin / cos for the path tracer (volumeraycast.cl:424,
 * 456-459): fixed fp32 operation sequences shared verbatim with the HIP kernel. */
static float vro_logf(float x) /* x in (0, 1] here */
{
    if (g_literal) return logf(x);
    if (!(x > 0.0f)) return -INFINITY;
    uint32_t ux = f2u(x);
    int e = 0;
    if (ux < 0x00800000u) {
        x = x * 8388608.0f;
        ux = f2u(x);
        e = -23;
    }
    e += (int)(ux >> 23) - 126;
    float m = u2f((ux & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752440f) {
        e -= 1;
        m = m + m;
    }
    float f = m - 1.0f;
    float z = f * f;
    float p = 7.0376836292E-2f;
    p = fmaf(p, f, -1.1514610310E-1f);
    p = fmaf(p, f, 1.1676998740E-1f);
    p = fmaf(p, f, -1.2420140846E-1f);
    p = fmaf(p, f, 1.4249322787E-1f);
    p = fmaf(p, f, -1.6668057665E-1f);
    p = fmaf(p, f, 2.0000714765E-1f);
    p = fmaf(p, f, -2.4999993993E-1f);
    p = fmaf(p, f, 3.3333331174E-1f);
    p = (p * f) * z;
    float fe = (float)e;
    p = fmaf(fe, -2.12194440e-4f, p);
    p = fmaf(z, -0.5f, p);
    float lg = f + p;
    return fmaf(fe, 0.693359375f, lg);
}

/* sin and cos of x in [0, 2*pi]: quadrant reduction + Cephes sinf/cosf kernels */
static void vro_sincosf(float x, float *s, float *c)
{
    if (g_literal) { *s = sinf(x); *c = cosf(x); return; }
    float q = floorf(fmaf(x, 0.63661977236758134f, 0.5f)); /* nearest multiple of pi/2 */
    float r = fmaf(q, -1.5703125f, x);
    r = fmaf(q, -4.837512969970703125e-4f, r);
    r = fmaf(q, -7.54978995489188216e-8f, r);
    float z = r * r;
    float sp = -1.9515295891E-4f;
    sp = fmaf(sp, z, 8.3321608736E-3f);
    sp = fmaf(sp, z, -1.6666654611E-1f);
    float sv = fmaf(sp * z, r, r);
    float cp = 2.443315711809948E-005f;
    cp = fmaf(cp, z, -1.388731625493765E-003f);
    cp = fmaf(cp, z, 4.166664568298827E-002f);
    float cv = fmaf(cp * z, z, fmaf(z, -0.5f, 1.0f));
    int qi = ((int)q) & 3;
    switch (qi) {
    case 0: *s = sv; *c = cv; break;
    case 1: *s = cv; *c = -sv; break;
    case 2: *s = -sv; *c = -cv; break;
    default: *s = -cv; *c = sv; break;
    }
}

/* atan2 and acos of get_environment_coords (:506-510) as fixed fp32 sequences (Cephes atanf /
 * asinf kernels); vr_device_math.h executes the same ones. */
static float atan_pos(float x) /* x >= 0, +inf allowed */
{
    float y0 = 0.0f, t = x;
    if (x > 2.414213562373095f) { y0 = 1.5707963267948966f; t = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y0 = 0.7853981633974483f; t = (x - 1.0f) / (x + 1.0f); }
    float z = t * t;
    float p = 8.05374449538e-2f;
    p = fmaf(p, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    return y0 + fmaf(p * z, t, t);
}

float vro_atan2f(float y, float x)
{
    if (g_literal) return atan2f(y, x);
    float ax = fabsf(x), ay = fabsf(y);
    float a = (ax == 0.0f && ay == 0.0f) ? 0.0f : atan_pos(ay / ax);
    if (x < 0.0f) a = 3.14159265358979323846f - a;
    return y < 0.0f ? -a : a;
}

static float asin_kernel(float z, float s) /* asin(s) for z = s*s <= 0.25 */
{
    float p = 4.2163199048e-2f;
    p = fmaf(p, z, 2.4181311049e-2f);
    p = fmaf(p, z, 4.5470025998e-2f);
    p = fmaf(p, z, 7.4953002686e-2f);
    p = fmaf(p, z, 1.6666752422e-1f);
    return fmaf(p * z, s, s);
}

float vro_acosf(float x) /* x in [-1, 1] */
{
    if (g_literal) return acosf(x);
    float a = fabsf(x);
    if (a > 0.5f) {
        float z = 0.5f * (1.0f - a);
        float r = 2.0f * asin_kernel(z, sqrtf(z));
        return x > 0.0f ? r : 3.14159265358979323846f - r;
    }
    return 1.5707963267948966f - asin_kernel(x * x, x);
}

/* read_imagef(environment, linearSmp, get_environment_coords(rayDir)) (:506-510, :655-656):
 * float RGBA image, normalised coordinates, CLAMP_TO_EDGE, LINEAR (OpenCL 1.2 spec 8.2) */
static void env_lookup(const vro_frame_extras *ex, f3 dir, float out[4])
{
    float s = vro_atan2f(dir.z, dir.x) * (float)(0.5 / 3.14159265358979323846) + 0.5f;
    float t = vro_acosf(vmax(vmin(dir.y, 1.0f), -1.0f)) * (float)(1.0 / 3.14159265358979323846);
    int w = (int)ex->env_w, h = (int)ex->env_h;
    float ub = s * (float)w - 0.5f, vb = t * (float)h - 0.5f;
    float fx = floorf(ub), fy = floorf(vb);
    float a = ub - fx, b = vb - fy;
    int x0 = iclamp((int)fx, 0, w - 1), x1 = iclamp((int)fx + 1, 0, w - 1);
    int y0 = iclamp((int)fy, 0, h - 1), y1 = iclamp((int)fy + 1, 0, h - 1);
    const float *p = ex->env_rgba;
    for (int c = 0; c < 4; ++c) {
        float t0 = lerpf(p[4 * ((size_t)y0 * w + x0) + c], p[4 * ((size_t)y0 * w + x1) + c], a);
        float t1 = lerpf(p[4 * ((size_t)y1 * w + x0) + c], p[4 * ((size_t)y1 * w + x1) + c], a);
        out[c] = lerpf(t0, t1, b);
        if (g_literal) /* spec 8.2, 2-D: (1-a)(1-b) T00 + a(1-b) T10 + (1-a) b T01 + a b T11 */
            out[c] = (1.0f - a) * (1.0f - b) * p[4 * ((size_t)y0 * w + x0) + c] +
                     a * (1.0f - b) * p[4 * ((size_t)y0 * w + x1) + c] +
                     (1.0f - a) * b * p[4 * ((size_t)y1 * w + x0) + c] + a * b * p[4 * ((size_t)y1 * w + x1) + c];
    }
}

/* volumeraycast.cl:181-206: returns the un-negated float4 (normal, |s2-s1|) */
static void gradient_central_diff_tff(const vol_t *v, f3 pos, float out[4])
{
    f3 off = mk3(1.0f / v->fw, 1.0f / v->fh, 1.0f / v->fd);
    float c[4];
    f3 s1, s2;
    tff_linear(v->s, vol_linear(v, pos.x + (-off.x), pos.y + 0.0f, pos.z + 0.0f), c); s1.x = c[3];
    tff_linear(v->s, vol_linear(v, pos.x + 0.0f, pos.y + (-off.y), pos.z + 0.0f), c); s1.y = c[3];
    tff_linear(v->s, vol_linear(v, pos.x + 0.0f, pos.y + 0.0f, pos.z + (-off.z)), c); s1.z = c[3];
    tff_linear(v->s, vol_linear(v, pos.x + off.x, pos.y + 0.0f, pos.z + 0.0f), c); s2.x = c[3];
    tff_linear(v->s, vol_linear(v, pos.x + 0.0f, pos.y + off.y, pos.z + 0.0f), c); s2.y = c[3];
    tff_linear(v->s, vol_linear(v, pos.x + 0.0f, pos.y + 0.0f, pos.z + off.z), c); s2.z = c[3];
    f3 g = sub3(s2, s1);
    f3 n = normalize3(g);
    if (dot3(g, g) == 0.0f) n = mk3(0.57735f, 0.57735f, 0.57735f);
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
    out[3] = len3(g);
}

/* volumeraycast.cl:407-437.  `rand2`/the stride are loop-invariant. */
static int sample_interaction(const vol_t *v, uint32_t rnd, f3 *ray_pos, f3 ray_dir,
                              float max_ext, float color_io[4], vro_stats *st)
{
    float t = 0.f;
    f3 pos;
    float color[4] = {color_io[0], color_io[1], color_io[2], color_io[3]};
    uint32_t cnt = 0;
    uint32_t rand2 = vro_parallel_rng(rnd);
    float dt = vro_logf(1.f - vro_map_uint_float(rand2)) / max_ext;
    float thr = vro_map_uint_float(rnd);
    do {
        ++cnt;
        t -= dt;
        pos = add3(*ray_pos, scale3(ray_dir, t));
        if (!in_volume(pos)) return 0;
        float smp = vol_linear(v, pos.x * 0.5f + 0.5f, pos.y * 0.5f + 0.5f, pos.z * 0.5f + 0.5f);
        tff_linear(v->s, smp, color);
        st->samples_taken++;
        if (cnt > 512) return 0;
    } while (color[3] < thr);
    memcpy(color_io, color, sizeof color);
    *ray_pos = pos;
    return 1;
}

/* volumeraycast.cl:453-460 */
static f3 dir_phase_function(uint32_t rnd)
{
    uint32_t rand2 = vro_parallel_rng(rnd);
    float phi = (float)(2.0 * (double)VRO_PI_F) * vro_map_uint_float(rand2);
    float cos_theta = 1.0f - 2.0f * vro_map_uint_float(vro_parallel_rng(rand2));
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float s, c;
    vro_sincosf(phi, &s, &c);
    return mk3(c * sin_theta, s * sin_theta, cos_theta);
}

/* volumeraycast.cl:463-503 */
static f3 trace_volume(const vol_t *v, uint32_t rnd, f3 ray_pos, f3 ray_dir, float t0,
                       float max_ext, const float bg[4], vro_stats *st)
{
    float w = 1.f;
    ray_pos = add3(ray_pos, scale3(ray_dir, t0));
    float color[4] = {bg[0], bg[1], bg[2], bg[3]};
    int inter = sample_interaction(v, rnd, &ray_pos, ray_dir, max_ext, color, st);
    if (inter) {
        f3 light_dir = add3(neg3(ray_dir), mk3(0.5f, 0.5f, 0.f));
        f3 sp = mk3(ray_pos.x * 0.5f + 0.5f, ray_pos.y * 0.5f + 0.5f, ray_pos.z * 0.5f + 0.5f);
        float g[4];
        gradient_central_diff_tff(v, sp, g);
        for (int i = 0; i < 4; ++i) g[i] = -g[i];
        float glen = sqrtf(((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]) + g[3] * g[3]);
        if (glen > 0.5f) {
            f3 col = illumination(mk3(color[0], color[1], color[2]), light_dir,
                                  mk3(g[0], g[1], g[2]));
            color[0] = col.x; color[1] = col.y; color[2] = col.z;
        } else {
            f3 sdir = dir_phase_function(rnd);
            f3 spos = ray_pos;
            float sc[4] = {bg[0], bg[1], bg[2], bg[3]};
            sample_interaction(v, rnd, &spos, sdir, max_ext, sc, st);
            for (int i = 0; i < 4; ++i) color[i] = color[i] + (sc[i] - color[i]) * 0.5f; /* mix */
        }
        float cs[4] = {bg[0], bg[1], bg[2], bg[3]};
        inter = sample_interaction(v, rnd, &ray_pos, light_dir, max_ext, cs, st);
        if (inter) w = 0.6f;
    }
    return mk3(color[0] * w, color[1] * w, color[2] * w);
}

/* ------------------------------------------------- ambient occlusion (:38-80, :353-392) */

/* hybrid Tausworthe / LCG generator on the per-pixel uint4 state (volumeraycast.cl:50-80).  The
 * constant is a double literal: the product is taken in double and rounded to float on return
 * (OpenCL C with fp64 available). */
static uint32_t taus_step(uint32_t *z, int s1, int s2, int s3, uint32_t m)
{
    uint32_t b = (((*z << (uint32_t)s1) ^ *z) >> (uint32_t)s2);
    *z = (((*z & m) << (uint32_t)s3) ^ b);
    return *z;
}

static float hybrid_rand(uint32_t st[4])
{
    uint32_t a = taus_step(&st[0], 13, 19, 12, 4294967294u);
    uint32_t b = taus_step(&st[1], 2, 25, 4, 4294967288u);
    uint32_t c = taus_step(&st[2], 3, 11, 17, 4294967280u);
    uint32_t d = st[3]; /* lcgStep returns the old value (:66-71) */
    st[3] = 1664525u * st[3] + 1013904223u;
    return (float)(2.3283064365387e-10 * (double)(float)(a ^ b ^ c ^ d));
}

/* the three above and the box-edge test below as the reference exposes them, for the pin against
 * the compiled volumeraycast.cl (tests/test_oracle_golden.py, oracle/_ref/libref_kernel.so) */
uint32_t vro_ui_rand_step(uint32_t st[4], uint32_t p, int s1, int s2, int s3, uint32_t m)
{
    /* ui_randStep, :50-64: component p of the state (p > 2: locZ is never assigned in the
     * reference -- undefined, not called that way) */
    return p < 3 ? taus_step(&st[p], s1, s2, s3, m) : 0u;
}

uint32_t vro_lcg_step(uint32_t st[4], uint32_t a, uint32_t c)
{
    uint32_t old = st[3]; /* lcgStep, :67-72 */
    st[3] = a * st[3] + c;
    return old;
}

float vro_hybrid_rand(uint32_t st[4]) { return hybrid_rand(st); }

/* getUniformRandomSampleDirectionUpper, :353-364 */
static f3 ao_sample_dir(f3 n, uint32_t st[4])
{
    float z = (hybrid_rand(st) * 2.f) - 1.f;
    float phi = (hybrid_rand(st) * 2.f) * VRO_PI_F;
    float sn, cs;
    vro_sincosf(phi, &sn, &cs);
    float rad = sqrtf(1.f - z * z);
    f3 dir = mk3(rad * sn, rad * cs, z);
    if (dot3(n, dir) < 0) dir = mk3(dir.x * -1.f, dir.y * -1.f, dir.z * -1.f);
    return dir;
}

/* calcAO, :368-392 (pos in [0,1] volume coordinates) */
static float calc_ao(const vol_t *v, f3 n, uint32_t st[4], f3 pos, float stepSize, float r)
{
    float ao = 0.f;
    for (int i = 0; i < 16; ++i) {
        f3 dir = ao_sample_dir(n, st);
        float sample = 0.f;
        int cnt = 0;
        while ((float)cnt * stepSize < r) {
            ++cnt;
            f3 p = add3(pos, scale3(scale3(dir, (float)cnt), stepSize));
            float tfc[4];
            tff_linear(v->s, vol_linear(v, p.x, p.y, p.z), tfc);
            sample += tfc[3];
            if (sample > 0.98f) break;
        }
        sample /= (float)cnt;
        ao += sample;
    }
    return ao / 16.f;
}

/* ------------------------------------------------------------ the kernel */

typedef struct {
    const vro_camera_params *cam;
    const vro_rendering_params *rp;
    const vro_raycast_params *rc;
    const vro_pathtrace_params *pt;
    int use_ess;
    uint32_t gsx, gsy; /* padded launch size */
    const vro_frame_extras *ex;
    uint32_t hit_w, hit_h; /* hit image dims: W/8 + 1, H/8 + 1 (volumerendercl.cpp:482-488) */
} kargs_t;

/* what a work-item of volumeRender did, for the image-order ESS bookkeeping (:659-670, :912-925) */
enum { PX_SKIPPED = 0, PX_MISS = 1, PX_SILENT = 2, PX_END = 3, PX_DIFFERS = 4 };

/* volumeraycast.cl:323-343 */
static int check_bounding_box(f3 pos, f3 voxLen, float b0, float b1)
{
    return (pos.x < voxLen.x && pos.z < b0 + voxLen.z) ||
           (pos.x < voxLen.x && pos.y < voxLen.y) ||
           (pos.y < voxLen.y && pos.z < b0 + voxLen.z) ||
           (pos.x > 1.f - voxLen.x && pos.z < b0 + voxLen.z) ||
           (pos.y > 1.f - voxLen.y && pos.z < b0 + voxLen.z) ||
           (pos.x > 1.f - voxLen.x && pos.z > b1 - voxLen.z) ||
           (pos.y > 1.f - voxLen.y && pos.z > b1 - voxLen.z) ||
           (pos.x < voxLen.x && pos.z > b1 - voxLen.z) ||
           (pos.y < voxLen.y && pos.z > b1 - voxLen.z) ||
           (pos.x > 1.f - voxLen.x && pos.y < voxLen.y) ||
           (pos.x > 1.f - voxLen.x && pos.y > 1.f - voxLen.y) ||
           (pos.x < voxLen.x && pos.y > 1.f - voxLen.y);
}

int vro_check_bounding_box(const float pos[3], const float voxLen[3], float b0, float b1)
{
    return check_bounding_box(mk3(pos[0], pos[1], pos[2]), mk3(voxLen[0], voxLen[1], voxLen[2]), b0, b1);
}

/* getLastHit (:513-526): 3x3 sum around the work-group; reads outside the image (undefined for
 * a sampler-less read) are taken as 0 */
static uint32_t last_hit(const kargs_t *k, int gx8, int gy8)
{
    uint32_t sum = 0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            int x = gx8 + dx, y = gy8 + dy;
            if (x < 0 || y < 0 || x >= (int)k->hit_w || y >= (int)k->hit_h) continue;
            sum += k->ex->hit_in[(size_t)y * k->hit_w + x];
        }
    return sum;
}

/* volumeRender, volumeraycast.cl:589-926, one work-item.  `prev` is the pixel of
 * inAccumulate (or NULL).  Always writes out[4]. */
static int render_pixel(const vol_t *v, const kargs_t *k, uint32_t gx, uint32_t gy,
                        const float *prev, float out[4], vro_stats *st)
{
    const vro_camera_params *cam = k->cam;
    const vro_rendering_params *rp = k->rp;
    const float *V = cam->viewMat;
    const f3 ms = mk3(rp->modelScale[0], rp->modelScale[1], rp->modelScale[2]);

    /* :611-612 */
    float rnd = (float)vro_parallel_rng3(gx, gy, rp->seed) / 4294967296.0f;

    /* :614-628 */
    float gsx = (float)k->gsx, gsy = (float)k->gsy;
    float aspect = gsy / gsx;
    aspect = vmin(aspect, gsx / gsy);
    int maxImg = (int)(k->gsx > k->gsy ? k->gsx : k->gsy);
    float icx = ((float)(int)gx / (float)maxImg) * 2.f;
    float icy = ((float)(int)gy / (float)maxImg) * 2.f;
    if (k->gsx > k->gsy) { icx -= 1.0f; icy -= aspect; }
    else { icx -= aspect; icy -= 1.0f; }
    icy *= -1.f;
    float psx = 2.f / gsx, psy = 2.f / gsy;
    float rnd2 = (float)vro_parallel_rng3(gy, gx, 2u * rp->seed) / 4294967296.0f;
    icx += rnd2 * psx;
    icy += (-rnd) * psy;

    /* :633-650 */
    f3 npp = mk3(icx, icy, -1.0f);
    f3 rayDir = mk3(dot3(mk3(V[0], V[1], V[2]), npp), dot3(mk3(V[4], V[5], V[6]), npp),
                    dot3(mk3(V[8], V[9], V[10]), npp));
    f3 camPos = mul3(mk3(V[3], V[7], V[11]), ms);
    if (cam->ortho) {
        camPos = mk3(V[3], V[7], V[11]);
        f3 vpx = mk3(V[0], V[4], V[8]);
        f3 vpy = mk3(V[1], V[5], V[9]);
        f3 vpz = mk3(V[2], V[6], V[10]);
        rayDir = neg3(vpz);
        npp = add3(add3(camPos, scale3(vpx, icx)), scale3(vpy, icy));
        npp = scale3(npp, len3(camPos));
        camPos = mul3(npp, ms);
    }
    rayDir = normalize3(mul3(rayDir, ms));

    /* :653-656; the environment map replaces the background when it is wider than one texel */
    float bgf = rp->useGradient ? (0.7f + 0.5f * rayDir.y) : 1.f;
    float env[4];
    for (int i = 0; i < 4; ++i) env[i] = rp->backgroundColor[i] * bgf;
    if (k->ex && k->ex->env_rgba && k->ex->env_w > 1) env_lookup(k->ex, rayDir, env);

    /* :658-670 image-order ESS: nothing was hit in or around this work-group last frame */
    if (rp->imgEss) {
        if (!last_hit(k, (int)(gx / 8u), (int)(gy / 8u))) {
            for (int i = 0; i < 4; ++i) out[i] = rp->showEss ? 1.f - env[i] : env[i];
            return PX_SKIPPED;
        }
    }

    /* :672-683 */
    float tnear, tfar;
    float o[3] = {camPos.x, camPos.y, camPos.z}, d[3] = {rayDir.x, rayDir.y, rayDir.z};
    int hit = vro_intersect_bbox(o, d, cam->bbox_bl, cam->bbox_tr, &tnear, &tfar);
    if (!hit || tfar < 0) {
        memcpy(out, env, sizeof env);
        return PX_MISS;
    }
    st->rays_hit++;

    /* :686-706 path tracing */
    if (rp->technique == 1) {
        uint32_t random = vro_parallel_rng3(gx, gy, rp->seed);
        f3 col = trace_volume(v, random, camPos, rayDir, tnear, k->pt->max_extinction, env, st);
        if (rp->iteration != 0 && prev) {
            float it1 = (float)(rp->iteration + 1u);
            col.x = prev[0] + (col.x - prev[0]) / it1;
            col.y = prev[1] + (col.y - prev[1]) / it1;
            col.z = prev[2] + (col.z - prev[2]) / it1;
        }
        out[0] = col.x; out[1] = col.y; out[2] = col.z; out[3] = 1.f;
        return PX_SILENT;
    }

    /* :709-733 */
    float sampleDist = tfar - tnear;
    if (sampleDist <= 0.f) { /* unreachable after the hit test; write bg */
        memcpy(out, env, sizeof env);
        return PX_SILENT;
    }
    f3 resf = mk3(v->fw, v->fh, v->fd);
    float stepSize = vmin(sampleDist,
                          sampleDist / (k->rc->samplingRate *
                                        len3(mul3(scale3(rayDir, sampleDist), resf))));
    float samples = ceilf(sampleDist / stepSize);
    stepSize = sampleDist / samples;
    st->samples_nominal += (uint64_t)samples;

    tnear = vmax(0.f, tnear);
    float result[4] = {env[0], env[1], env[2], env[3]};
    float alpha = 0.f;
    float t = tnear;
    f3 voxLen = mk3(1.f / v->fw, 1.f / v->fh, 1.f / v->fd);
    float refInterval = 1.f / k->rc->samplingRate;
    float t_exit = tfar;
    float offset = (len3(voxLen) * rnd) * 2.0f;

    /* :737-760 DDA initialisation (ESS build only) */
    int bricksRes[3] = {v->bw, v->bh, v->bd};
    int stepv[3] = {0, 0, 0}, cell[3] = {0, 0, 0}, exitc[3] = {0, 0, 0};
    float tv[3] = {0, 0, 0}, deltaT[3] = {0, 0, 0}, brickDia = 0.f;
    if (k->use_ess) {
        float brickLen[3], invRay[3], roc[3];
        float dirv[3] = {rayDir.x, rayDir.y, rayDir.z};
        float camv[3] = {camPos.x, camPos.y, camPos.z};
        for (int i = 0; i < 3; ++i) {
            brickLen[i] = 1.f / k->rc->brickRes[i];
            invRay[i] = 1.f / dirv[i];
            /* sign(): the select() at :743-744 is a no-op */
            stepv[i] = dirv[i] > 0.f ? 1 : (dirv[i] < 0.f ? -1 : 0);
            deltaT[i] = (float)stepv[i] * ((brickLen[i] * 2.f) * invRay[i]);
            roc[i] = (camv[i] + dirv[i] * tnear) - (-1.f);
            cell[i] = iclamp((int)floorf(roc[i] / (2.f * brickLen[i])), 0, bricksRes[i] - 1);
            /* cell - isgreaterequal(dir, 0): vector relational yields -1 for true */
            int cadj = cell[i] - (dirv[i] >= 0.f ? -1 : 0);
            tv[i] = tnear + ((float)cadj * (2.f * brickLen[i]) - roc[i]) * invRay[i];
            exitc[i] = stepv[i] * bricksRes[i];
            if (exitc[i] < 0) exitc[i] = -1;
        }
        brickDia = sqrtf(((brickLen[0] * brickLen[0]) + (brickLen[1] * brickLen[1])) +
                         (brickLen[2] * brickLen[2])) * 2.f;
    }

    const f3 toLight = neg3(rayDir);
    f3 pos = mk3(0.f, 0.f, 0.f); /* :722: lives outside the loops, showEss looks at the last one */
    int outer_first = 1;
    /* :763 / (non-ESS: a single pass of the inner loop with t_exit = tfar) */
    while (k->use_ess ? (t < tfar) : outer_first) {
        outer_first = 0;
        if (k->use_ess) {
            float mn, mx;
            brick_minmax(v, cell[0], cell[1], cell[2], &mn, &mx);
            st->bricks_visited++;
            float inc[3];
            inc[0] = (tv[0] <= tv[1]) && (tv[0] <= tv[2]) ? 1.f : 0.f;
            inc[1] = (tv[1] <= tv[0]) && (tv[1] <= tv[2]) ? 1.f : 0.f;
            inc[2] = (tv[2] <= tv[0]) && (tv[2] <= tv[1]) ? 1.f : 0.f;
            for (int i = 0; i < 3; ++i) cell[i] += (int)inc[i] * stepv[i];
            t_exit = ((1.f * (tv[0] * inc[0])) + (1.f * (tv[1] * inc[1]))) +
                     (1.f * (tv[2] * inc[2]));
            t_exit = vclamp(t_exit, t + stepSize, t + brickDia);
            for (int i = 0; i < 3; ++i) tv[i] += inc[i] * deltaT[i];

            float tfc[4];
            tff_linear(v->s, mx, tfc);
            if (tfc[3] < 1e-6f) {
                uint32_t pmin = prefix_nearest(v->s, mn);
                uint32_t pmax = prefix_nearest(v->s, mx);
                if (pmin == pmax) {
                    st->bricks_skipped++;
                    t = t_exit;
                    continue;
                }
            }
        }
        /* :790-880 */
        while (t < t_exit) {
            st->samples_taken++;
            pos = add3(camPos, scale3(rayDir, t - offset));
            pos = mk3(pos.x * 0.5f + 0.5f, pos.y * 0.5f + 0.5f, pos.z * 0.5f + 0.5f);
            float tfc[4];
            f3 grad = mk3(0.f, 0.f, 0.f);
            if (rp->illumType == 4) { /* :796-799 gradient magnitude through the TF */
                tff_linear(v->s, gradient_central_diff_len(v, pos), tfc);
            } else if (v->nch == 4) { /* :840-845 CL_RGBA: the voxel is the colour, no shading */
                for (int c = 0; c < 4; ++c)
                    tfc[c] = rp->useLinear ? vol_linear_c(v, pos.x, pos.y, pos.z, c)
                                           : vol_nearest_c(v, pos.x, pos.y, pos.z, c);
            } else if (v->nch == 2) { /* :846-855 CL_RG: (r, g, 0, 1) -> colour (r, 0, 0), opacity TF(|g|) */
                float r = rp->useLinear ? vol_linear_c(v, pos.x, pos.y, pos.z, 0)
                                        : vol_nearest_c(v, pos.x, pos.y, pos.z, 0);
                float g = rp->useLinear ? vol_linear_c(v, pos.x, pos.y, pos.z, 1)
                                        : vol_nearest_c(v, pos.x, pos.y, pos.z, 1);
                float t4[4];
                tff_linear(v->s, fabsf(g / 1.f), t4);
                tfc[0] = r; tfc[1] = 0.f; tfc[2] = 0.f; tfc[3] = t4[3];
            } else {
                float density = rp->useLinear ? vol_linear(v, pos.x, pos.y, pos.z)
                                              : vol_nearest(v, pos.x, pos.y, pos.z);
                tff_linear(v->s, density, tfc);
                if (tfc[3] > 0.1f && rp->illumType) { /* :809-830 */
                    st->samples_shaded++;
                    if (rp->illumType == 1 || rp->illumType == 5) {
                        grad = neg_gradient_central_diff(v, pos);
                    } else if (rp->illumType == 2) { /* :816-818 */
                        float g[4];
                        gradient_central_diff_tff(v, pos, g);
                        grad = mk3(-g[0], -g[1], -g[2]);
                    } else if (rp->illumType == 3) { /* :819-821 */
                        grad = neg_gradient_sobel(v, pos);
                    }
                    f3 c = rp->illumType == 5 ? cel_shading(mk3(tfc[0], tfc[1], tfc[2]), toLight, grad)
                                              : illumination(mk3(tfc[0], tfc[1], tfc[2]), toLight, grad);
                    tfc[0] = c.x; tfc[1] = c.y; tfc[2] = c.z;
                }
                if (tfc[3] > 0.1f && k->rc->contours) { /* :832-837 */
                    if (!rp->illumType) grad = neg_gradient_central_diff(v, pos);
                    float e = fabsf(dot3(rayDir, grad));
                    tfc[0] *= e; tfc[1] *= e; tfc[2] *= e;
                }
            }
            tfc[0] = env[0] - tfc[0]; /* :856 */
            tfc[1] = env[1] - tfc[1];
            tfc[2] = env[2] - tfc[2];
            if (k->rc->aerial) { /* :857-861 */
                float depthCue = 1.f - (t - tnear) / sampleDist;
                tfc[3] *= depthCue;
            }
            float opacity = 1.f - vro_powr(1.f - tfc[3], refInterval); /* :864 */
            float oma = 1.f - alpha;
            result[0] = result[0] - (tfc[0] * opacity) * oma;
            result[1] = result[1] - (tfc[1] * opacity) * oma;
            result[2] = result[2] - (tfc[2] * opacity) * oma;
            alpha = alpha + opacity * oma;
            if (t >= tfar) break;
            if ((double)alpha > 0.98) { /* ERT_THRESHOLD is a double literal (:28) */
                if (k->rc->useAO) { /* :870-876 ambient occlusion only on solid surfaces */
                    uint32_t st4[4];
                    st4[0] = st4[1] = st4[2] = st4[3] = vro_parallel_rng3(gx, gy, rp->seed); /* :611 */
                    f3 n = neg_gradient_central_diff(v, pos);
                    float vl = len3(voxLen);
                    float ao = calc_ao(v, n, st4, pos, vl * 0.9f, vl * 5.f);
                    float f = 1.f - 0.5f * ao;
                    result[0] *= f; result[1] *= f; result[2] *= f;
                }
                break;
            }
            t += stepSize;
        }
        if (!k->use_ess) break;
        if (t >= tfar || (double)alpha > 0.98) break; /* :882 */
        if (cell[0] == exitc[0] || cell[1] == exitc[1] || cell[2] == exitc[2]) break; /* :883 */
        t = t_exit; /* :884 */
    }

    /* :888-896 visualise the skipping: rays that never sampled (pos still 0) and rays that end
     * next to an edge of the box get the inverted background colour */
    if (rp->showEss && check_bounding_box(pos, voxLen, 0.f, 1.f)) {
        for (int i = 0; i < 3; ++i) result[i] = fabsf(1.f - rp->backgroundColor[i]);
        alpha = 1.f;
    }

    /* :898-909 (float accumulation) */
    result[3] = alpha;
    if (rp->iteration != 0 && prev) {
        float it1 = (float)(rp->iteration + 1u);
        for (int i = 0; i < 3; ++i) result[i] = prev[i] + (result[i] - prev[i]) / it1;
    }
    memcpy(out, result, sizeof result);
    /* :912-925: did this work-item change its pixel? */
    return (result[0] != env[0] || result[1] != env[1] || result[2] != env[2]) ? (PX_END | PX_DIFFERS)
                                                                             : PX_END;
}

int vro_render_tile(const vro_scene *scene, const vro_camera_params *cam,
                    const vro_rendering_params *render, const vro_raycast_params *raycast,
                    const vro_pathtrace_params *pathtrace, int use_ess, uint32_t W, uint32_t H,
                    uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, const float *in_accum,
                    float *out, vro_stats *stats, uint8_t *touched, int threads)
{
    return vro_render_tile_ex(scene, cam, render, raycast, pathtrace, use_ess, W, H, x0, y0, w, h,
                              in_accum, out, stats, touched, threads, NULL);
}

void vro_hit_image_init(uint32_t W, uint32_t H, uint8_t *hit_in, uint8_t *hit_out)
{
    /* volumerendercl.cpp:482-488: the input image is created from a vector of 32-bit ones but read
     * as one byte per texel, i.e. from the bytes 1,0,0,0,1,...; the output image starts
     * uninitialised (taken as 0 here) */
    const size_t n = (size_t)(W / 8u + 1u) * (H / 8u + 1u);
    for (size_t i = 0; i < n; ++i) {
        hit_in[i] = (i % 4u) == 0u ? 1u : 0u;
        hit_out[i] = 0u;
    }
}

int vro_render_tile_ex(const vro_scene *scene, const vro_camera_params *cam,
                    const vro_rendering_params *render, const vro_raycast_params *raycast,
                    const vro_pathtrace_params *pathtrace, int use_ess, uint32_t W, uint32_t H,
                    uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, const float *in_accum,
                    float *out, vro_stats *stats, uint8_t *touched, int threads,
                    const vro_frame_extras *ex)
{
    if (!scene || !scene->voxels || !scene->tff || !cam || !render || !raycast || !out)
        return -1;
    if (scene->format < 0 || scene->format > 2 || scene->tff_n == 0) return -1;
    if (x0 + w > W || y0 + h > H) return -1;
    if (use_ess && (!scene->bricks || !scene->prefix || scene->prefix_n == 0)) return -1;
    if (render->illumType > 5) return -2;
    if (render->imgEss) {
        /* the hit image is kept per 8x8 work-group: the tile must consist of whole groups */
        if (!ex || !ex->hit_in || !ex->hit_out) return -1;
        if (x0 % 8u || y0 % 8u || (w % 8u && x0 + w != W) || (h % 8u && y0 + h != H)) return -1;
        if (render->technique == 1) return -2;
    }
    if (ex && ex->env_rgba && (ex->env_w == 0 || ex->env_h == 0)) return -1;

    vol_t v;
    memset(&v, 0, sizeof v);
    v.s = scene;
    v.w = (int)scene->res[0]; v.h = (int)scene->res[1]; v.d = (int)scene->res[2];
    v.fw = (float)v.w; v.fh = (float)v.h; v.fd = (float)v.d;
    v.inv_max = scene->format == VRO_UCHAR ? 1.0f / 255.0f
                : scene->format == VRO_USHORT ? 1.0f / 65535.0f : 1.0f;
    v.row = (size_t)v.w;
    v.slice = (size_t)v.w * (size_t)v.h;
    v.bw = (int)scene->bricks_res[0]; v.bh = (int)scene->bricks_res[1]; v.bd = (int)scene->bricks_res[2];
    v.touched = touched;
    v.mbx = (v.w + 3) / 4;
    v.mby = (v.h + 3) / 4;
    v.nch = scene->channels > 1 ? (int)scene->channels : 1;
    if (v.nch != 1 && v.nch != 2 && v.nch != 4) return -1;

    vro_pathtrace_params pt_default = {100.f};
    kargs_t k = {cam, render, raycast, pathtrace ? pathtrace : &pt_default, use_ess,
                 vro_padded(W), vro_padded(H), ex, W / 8u + 1u, H / 8u + 1u};
    uint8_t *status = NULL;
    if (render->imgEss) {
        status = (uint8_t *)malloc((size_t)w * h);
        if (!status) return -3;
    }

    vro_stats total;
    memset(&total, 0, sizeof total);
#ifdef _OPENMP
    int nt = threads > 0 ? threads : omp_get_max_threads();
#else
    int nt = 1;
    (void)threads;
#endif
#pragma omp parallel num_threads(nt)
    {
        vro_stats st;
        memset(&st, 0, sizeof st);
        /* work unit = 8 consecutive pixels of a row, handed out dynamically */
        const int64_t cw = ((int64_t)w + 7) / 8;
#pragma omp for schedule(dynamic, 4)
        for (int64_t u = 0; u < (int64_t)h * cw; ++u) {
            const int64_t ly = u / cw;
            const uint32_t xb = (uint32_t)(u % cw) * 8u;
            for (uint32_t lx = xb; lx < xb + 8u && lx < w; ++lx) {
                size_t o = ((size_t)ly * w + lx) * 4;
                int px = render_pixel(&v, &k, x0 + lx, y0 + (uint32_t)ly,
                                      in_accum ? in_accum + o : NULL, out + o, &st);
                if (status) status[(size_t)ly * w + lx] = (uint8_t)px;
            }
        }
#pragma omp critical
        {
            total.samples_taken += st.samples_taken;
            total.samples_nominal += st.samples_nominal;
            total.samples_shaded += st.samples_shaded;
            total.bricks_visited += st.bricks_visited;
            total.bricks_skipped += st.bricks_skipped;
            total.rays_hit += st.rays_hit;
        }
    }
    if (stats) *stats = total;
    if (status) {
        /* :664-667, :680-681, :912-925 on a device that runs the 64 work-items of a group in
         * lock-step: a skipped group writes 0; work-items that miss the box write 0 early; the
         * group's first work-item, if it reaches the end, has the last word (1 if any work-item
         * that reached the end changed its pixel); otherwise the texel keeps its old value */
        for (uint32_t gy0 = 0; gy0 < h; gy0 += 8u)
            for (uint32_t gx0 = 0; gx0 < w; gx0 += 8u) {
                int any_miss = 0, any_diff = 0;
                for (uint32_t yy = gy0; yy < gy0 + 8u && yy < h; ++yy)
                    for (uint32_t xx = gx0; xx < gx0 + 8u && xx < w; ++xx) {
                        const uint8_t sx = status[(size_t)yy * w + xx];
                        any_miss |= sx == PX_MISS;
                        any_diff |= (sx & PX_DIFFERS) != 0;
                    }
                const uint8_t first = status[(size_t)gy0 * w + gx0];
                uint8_t *dst = &ex->hit_out[(size_t)((y0 + gy0) / 8u) * k.hit_w + (x0 + gx0) / 8u];
                if (first == PX_SKIPPED) *dst = 0;
                else if ((first & 3) == PX_END) *dst = any_diff ? 1 : 0;
                else if (any_miss) *dst = 0;
            }
        free(status);
    }
    return 0;
}

/* ------------------------------------------------------ synthetic volumes */

/* voxel centre p = 2(i+0.5)/N - 1; sphere d = max(0, 1 - |p|/0.9);
 * shells = d*(0.5+0.5*cos(24*pi*|p|)), values below 0.35 set to 0. */
int vro_synth_volume(int kind, const uint32_t res[3], int format, void *out)
{
    if (!out || format < 0 || format > 2 || kind < 0 || kind > 1) return -1;
    size_t row = res[0], slice = (size_t)res[0] * res[1];
#pragma omp parallel for schedule(static)
    for (int64_t z = 0; z < (int64_t)res[2]; ++z)
        for (uint32_t y = 0; y < res[1]; ++y)
            for (uint32_t x = 0; x < res[0]; ++x) {
                double px = 2.0 * (x + 0.5) / res[0] - 1.0;
                double py = 2.0 * (y + 0.5) / res[1] - 1.0;
                double pz = 2.0 * ((double)z + 0.5) / res[2] - 1.0;
                double r = sqrt(px * px + py * py + pz * pz);
                double dv = 1.0 - r / 0.9;
                if (dv < 0.0) dv = 0.0;
                if (kind == 1) {
                    dv = dv * (0.5 + 0.5 * cos(24.0 * 3.14159265358979323846 * r));
                    if (dv < 0.35) dv = 0.0;
                }
                size_t i = (size_t)z * slice + (size_t)y * row + x;
                if (format == VRO_UCHAR) ((uint8_t *)out)[i] = (uint8_t)lround(255.0 * dv);
                else if (format == VRO_USHORT) ((uint16_t *)out)[i] = (uint16_t)lround(65535.0 * dv);
                else ((float *)out)[i] = (float)dv;
            }
    return 0;
}
