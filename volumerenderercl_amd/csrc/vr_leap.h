// vr_leap.h -- k steps of the reference's `t += stepSize` chain (volumeraycast.cl:879) in O(1).
//
// The reference advances the ray parameter by repeated fp32 additions, each rounded to nearest
// even; the image depends on that exact sequence (it decides which sample is the last one of a
// brick segment, :790, and of the ray, :868).  Stepping over a run of k samples that are known to
// composite to nothing would cost k dependent additions.  Inside one binade [2^e, 2^(e+1)) all
// values are multiples of q = ulp = 2^(e-23), so with step = s q + r (0 <= r < q) every addition
// moves t by the SAME whole number of ulps: s if r < q/2, s + 1 if r > q/2 -- whatever t is.  In
// the tie case r = q/2 round-half-even makes the mantissa even with the first addition and keeps it
// even afterwards, so from then on the increment is s (s even) or s + 1 (s odd).  Hence
//      t_k = t_0 + k * inc * q        exactly, as long as t_k stays inside the binade,
// which is integer arithmetic on the mantissa.  vr_leap does one literal addition (it crosses
// binade boundaries and absorbs the cases t < step), then as many steps as fit into the binade at
// once, and repeats: one or two iterations for the rays of a frame (t in [1, 8)), about log2(k)
// when a ray starts at t = 0 inside the volume.  Plain C / C++ / HIP: tests/test_leap.py runs it on
// the CPU against the literal loop over adversarial and random inputs.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define VR_LEAP_FN __host__ __device__ inline
#else
#define VR_LEAP_FN static inline
#endif

VR_LEAP_FN uint32_t vr_leap_bits(float f)
{
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);   // a register move on every target
    return u;
}
VR_LEAP_FN float vr_leap_float(uint32_t u)
{
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

// t after k additions `t = t + step` (each rounded to nearest even).  t >= 0, step > 0, finite.
VR_LEAP_FN float vr_leap(float t, float step, uint32_t k)
{
    const uint32_t sb = vr_leap_bits(step);
    const int es = (int)(sb >> 23);
    const uint32_t ms = (sb & 0x7fffffu) | 0x800000u;
    while (k) {
        t = t + step;                                    // literal: any binade crossing happens here
        --k;
        if (!k) break;
        uint32_t tb = vr_leap_bits(t);
        const int e = (int)(tb >> 23);
        const int sh = e - es;                           // step = ms * 2^-sh ulps of t
        if (es == 0 || e == 0 || e >= 254 || sh < 0 || sh > 23) continue;   // odd ranges: literal steps
        const uint32_t s = ms >> sh, r = ms & ((1u << sh) - 1u), half = sh ? (1u << (sh - 1)) : 0u;
        uint32_t inc;
        if (sh == 0 || r < half) inc = s;
        else if (r > half) inc = s + 1u;
        else {
            // tie: one more literal addition makes the mantissa even (round half to even)
            const float t2 = t + step;
            --k;
            t = t2;
            tb = vr_leap_bits(t);
            if ((int)(tb >> 23) != e) continue;          // it left the binade: start over there
            if (!k) break;
            inc = (s & 1u) ? s + 1u : s;
        }
        if (inc == 0u) return t;                         // step below half an ulp: t no longer moves
        uint32_t M = (tb & 0x7fffffu) | 0x800000u;
        const uint32_t room = 0xffffffu - M;             // ulps left in this binade
        uint32_t m = (uint32_t)((float)room / (float)inc);   // floor(room / inc), or one more
        m = m ? m - 1u : 0u;                             // conservative: never beyond the binade
        if (m > k) m = k;
        M += m * inc;
        k -= m;
        t = vr_leap_float((tb & 0xff800000u) | (M & 0x7fffffu));
    }
    return t;
}
