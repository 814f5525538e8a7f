/* CPU harness for vr_leap.h (tests/test_leap.py): the literal chain and the leap, side by side. */
#include "../../volumerenderercl_amd/csrc/vr_leap.h"

float leap_fast(float t, float step, uint32_t k) { return vr_leap(t, step, k); }

float leap_literal(float t, float step, uint32_t k)
{
    volatile float v = t;
    for (uint32_t i = 0; i < k; ++i) v = v + step;
    return v;
}

/* number of mismatches over n triples; first mismatch index in *first (or -1) */
long leap_check(const float *t, const float *step, const uint32_t *k, long n, long *first)
{
    long bad = 0;
    *first = -1;
    for (long i = 0; i < n; ++i) {
        float a = leap_fast(t[i], step[i], k[i]), b = leap_literal(t[i], step[i], k[i]);
        if (vr_leap_bits(a) != vr_leap_bits(b)) {
            if (*first < 0) *first = i;
            ++bad;
        }
    }
    return bad;
}
