// vr_device_math.h -- fp32 building blocks of the gfx950 ray-cast kernels.
//
// Parity contract (DESIGN.md "Numerics"): every function here is a fixed sequence of
// individually rounded IEEE fp32 operations (the translation unit is compiled with
// -ffp-contract=off and correctly rounded divide/sqrt); fused multiply-adds occur only
// where __builtin_fmaf is written.  The CPU oracle executes the same sequences, so the
// images agree bit for bit and no ERT / shading branch can flip between the two.
//
// What OpenCL 1.2 leaves implementation-defined in the reference kernel is fixed here
// (SURVEY.md App. B/C): native_divide = '/', fast_length/length = sqrt(x*x+y*y+z*z),
// fast_normalize/normalize = v * (1/sqrt(dot)), min/max/clamp per the spec text,
// native_powr = vr_powr.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct f3 { float x, y, z; };

#define VR_DEV __device__ __forceinline__

VR_DEV float vmin(float x, float y) { return y < x ? y : x; }   // OpenCL min(x, y)
VR_DEV float vmax(float x, float y) { return x < y ? y : x; }   // OpenCL max(x, y)
VR_DEV float vclamp(float x, float lo, float hi) { return vmin(vmax(x, lo), hi); }
VR_DEV int iclamp(int x, int lo, int hi) { return min(max(x, lo), hi); }
VR_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
VR_DEV float dot3(f3 a, f3 b) { return ((a.x * b.x) + (a.y * b.y)) + (a.z * b.z); }
VR_DEV float len3(f3 a) { return sqrtf(dot3(a, a)); }
VR_DEV f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
VR_DEV f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
VR_DEV f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
VR_DEV f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
VR_DEV f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
VR_DEV f3 normalize3(f3 v)
{
    float d = dot3(v, v);
    if (d == 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    float inv = 1.0f / sqrtf(d);
    return scale3(v, inv);
}
VR_DEV float lerpf(float p, float q, float w) { return __builtin_fmaf(w, q - p, p); }

// random.cl:2-13 (Wang hash), :22-28
VR_DEV uint32_t parallel_rng(uint32_t x)
{
    uint32_t value = x;
    value = (value ^ 61u) ^ (value >> 16);
    value *= 9u;
    value ^= value << 4;
    value *= 0x27d4eb2du;
    value ^= value >> 15;
    return value;
}
VR_DEV uint32_t parallel_rng3(uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t value = parallel_rng(x);
    value = parallel_rng(y ^ value);
    value = parallel_rng(z ^ value);
    return value;
}
// random.cl:44-47; (float)UINT_MAX == 2^32
VR_DEV float map_uint_float(uint32_t v) { return (float)v / 4294967296.0f; }

// log(x) for x > 0 as e*ln2 + log(m), m in [sqrt(1/2), sqrt(2)): Cephes logf kernel as an
// explicit fmaf chain.
VR_DEV float vr_logf_pos(float x)
{
    uint32_t ux = __float_as_uint(x);
    int e = 0;
    if (ux < 0x00800000u) {
        x = x * 8388608.0f;
        ux = __float_as_uint(x);
        e = -23;
    }
    e += (int)(ux >> 23) - 126;
    float m = __uint_as_float((ux & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752440f) {
        e -= 1;
        m = m + m;
    }
    float f = m - 1.0f;
    float z = f * f;
    float p = 7.0376836292E-2f;
    p = __builtin_fmaf(p, f, -1.1514610310E-1f);
    p = __builtin_fmaf(p, f, 1.1676998740E-1f);
    p = __builtin_fmaf(p, f, -1.2420140846E-1f);
    p = __builtin_fmaf(p, f, 1.4249322787E-1f);
    p = __builtin_fmaf(p, f, -1.6668057665E-1f);
    p = __builtin_fmaf(p, f, 2.0000714765E-1f);
    p = __builtin_fmaf(p, f, -2.4999993993E-1f);
    p = __builtin_fmaf(p, f, 3.3333331174E-1f);
    p = (p * f) * z;
    float fe = (float)e;
    p = __builtin_fmaf(fe, -2.12194440e-4f, p);
    p = __builtin_fmaf(z, -0.5f, p);
    float lg = f + p;
    return __builtin_fmaf(fe, 0.693359375f, lg);
}

VR_DEV float vr_logf(float x)
{
    if (!(x > 0.0f)) return -__builtin_inff();
    return vr_logf_pos(x);
}

// Parity definition of native_powr(x, y), x >= 0 (volumeraycast.cl:290,864):
// exp(y*log(x)); powr(1,y) == 1 and powr(0,y>0) == 0 exactly.
VR_DEV float vr_powr(float x, float y)
{
    if (!(x > 0.0f)) return (x == 0.0f) ? 0.0f : __builtin_nanf("");
    float t = y * vr_logf_pos(x);
    if (t > 88.0f) return __builtin_inff();
    if (t < -87.0f) return 0.0f;
    float n = floorf(__builtin_fmaf(t, 1.44269504088896341f, 0.5f));
    t = __builtin_fmaf(n, -0.693359375f, t);
    t = __builtin_fmaf(n, 2.12194440e-4f, t);
    float q = 1.9875691500E-4f;
    q = __builtin_fmaf(q, t, 1.3981999507E-3f);
    q = __builtin_fmaf(q, t, 8.3334519073E-3f);
    q = __builtin_fmaf(q, t, 4.1665795894E-2f);
    q = __builtin_fmaf(q, t, 1.6666665459E-1f);
    q = __builtin_fmaf(q, t, 5.0000001201E-1f);
    float r = __builtin_fmaf(q, t * t, t) + 1.0f;
    int ni = (int)n;
    return r * __uint_as_float((uint32_t)(ni + 127) << 23);
}

// atan2 / acos of get_environment_coords (volumeraycast.cl:506-510): the oracle's fixed fp32
// sequences (vro_atan2f / vro_acosf, Cephes atanf / asinf kernels)
VR_DEV float vr_atan_pos(float x)   // x >= 0, +inf allowed
{
    float y0 = 0.0f, t = x;
    if (x > 2.414213562373095f) { y0 = 1.5707963267948966f; t = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y0 = 0.7853981633974483f; t = (x - 1.0f) / (x + 1.0f); }
    float z = t * t;
    float p = 8.05374449538e-2f;
    p = __builtin_fmaf(p, z, -1.38776856032e-1f);
    p = __builtin_fmaf(p, z, 1.99777106478e-1f);
    p = __builtin_fmaf(p, z, -3.33329491539e-1f);
    return y0 + __builtin_fmaf(p * z, t, t);
}

VR_DEV float vr_atan2f(float y, float x)
{
    float ax = fabsf(x), ay = fabsf(y);
    float a = (ax == 0.0f && ay == 0.0f) ? 0.0f : vr_atan_pos(ay / ax);
    if (x < 0.0f) a = 3.14159265358979323846f - a;
    return y < 0.0f ? -a : a;
}

VR_DEV float vr_asin_kernel(float z, float s)   // asin(s) for z = s*s <= 0.25
{
    float p = 4.2163199048e-2f;
    p = __builtin_fmaf(p, z, 2.4181311049e-2f);
    p = __builtin_fmaf(p, z, 4.5470025998e-2f);
    p = __builtin_fmaf(p, z, 7.4953002686e-2f);
    p = __builtin_fmaf(p, z, 1.6666752422e-1f);
    return __builtin_fmaf(p * z, s, s);
}

VR_DEV float vr_acosf(float x)   // x in [-1, 1]
{
    float a = fabsf(x);
    if (a > 0.5f) {
        float z = 0.5f * (1.0f - a);
        float r = 2.0f * vr_asin_kernel(z, sqrtf(z));
        return x > 0.0f ? r : 3.14159265358979323846f - r;
    }
    return 1.5707963267948966f - vr_asin_kernel(x * x, x);
}

// sin/cos of x in [0, 2*pi] (path tracer phase function, volumeraycast.cl:456-459)
VR_DEV void vr_sincosf(float x, float *s, float *c)
{
    float q = floorf(__builtin_fmaf(x, 0.63661977236758134f, 0.5f));
    float r = __builtin_fmaf(q, -1.5703125f, x);
    r = __builtin_fmaf(q, -4.837512969970703125e-4f, r);
    r = __builtin_fmaf(q, -7.54978995489188216e-8f, r);
    float z = r * r;
    float sp = -1.9515295891E-4f;
    sp = __builtin_fmaf(sp, z, 8.3321608736E-3f);
    sp = __builtin_fmaf(sp, z, -1.6666654611E-1f);
    float sv = __builtin_fmaf(sp * z, r, r);
    float cp = 2.443315711809948E-005f;
    cp = __builtin_fmaf(cp, z, -1.388731625493765E-003f);
    cp = __builtin_fmaf(cp, z, 4.166664568298827E-002f);
    float cv = __builtin_fmaf(cp * z, z, __builtin_fmaf(z, -0.5f, 1.0f));
    int qi = ((int)q) & 3;
    if (qi == 0) { *s = sv; *c = cv; }
    else if (qi == 1) { *s = cv; *c = -sv; }
    else if (qi == 2) { *s = -sv; *c = -cv; }
    else { *s = -cv; *c = sv; }
}
