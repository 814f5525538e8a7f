"""The arithmetic claim behind the path tracer's leaps (volumerenderercl_amd/csrc/vr_pathtrace.hip), checked on
the CPU with numpy's float32: a tracking walk's parameter advances by t <- t + s, rounded to nearest-even every step; while
t stays in one binade every step adds the same number of ulps to its bit pattern -- q = floor(s / ulp(t)) when the
remainder is below half an ulp, q + 1 when it is above -- so n steps are one integer multiply-add.  The kernel uses exactly
this (and takes no leap on a tie, where the rounding depends on the parity of the sum)."""
import numpy as np


def _inc(t_bits, s_bits):
    """ulps one step adds to a positive normal t (bit pattern) for a positive normal s, or None where the kernel takes
    no closed-form stretch (s not below t's binade by 1..24 exponents, or a tie)."""
    et, es = t_bits >> 23, s_bits >> 23
    d = et - es
    if not (1 <= d <= 24) or et >= 255 or es <= 0:
        return None
    ms = (s_bits & 0x7FFFFF) | 0x800000
    q, rem, half = ms >> d, ms & ((1 << d) - 1), (1 << d) >> 1
    if rem == half:
        return None
    return q + (1 if rem > half else 0)


def test_steps_within_a_binade_are_an_integer_multiply_add():
    rng = np.random.default_rng(7)
    checked = 0
    for _ in range(4000):
        # strides like the walks' (-log(1 - u) / max_extinction) and parameters a few thousand strides further on
        s = np.float32(-np.log(1.0 - rng.random()) / rng.choice([1.0, 10.0, 100.0, 1000.0]))
        t = np.float32(s * np.float32(rng.integers(2, 5000)) * np.float32(0.5 + rng.random()))
        if not (np.isfinite(s) and s > 0 and np.isfinite(t) and t > 0):
            continue
        tb, sb = int(t.view(np.uint32)), int(s.view(np.uint32))
        inc = _inc(tb, sb)
        if inc is None:
            continue
        room = (tb | 0x7FFFFF) - tb
        n_max = room // inc if inc else 600
        n = int(min(n_max, 600))
        # the reference's way: n rounded additions
        x = t
        for k in range(1, n + 1):
            x = np.float32(x + s)
            assert int(x.view(np.uint32)) == tb + k * inc, (float(t), float(s), k)
        checked += n
        # ... and the step after the last one that fits is the one that leaves the binade (or would overshoot its top)
        if inc and n == n_max:
            nxt = np.float32(x + s)
            assert int(nxt.view(np.uint32)) > (tb | 0x7FFFFF) or (int(nxt.view(np.uint32)) - int(x.view(np.uint32))) in (inc, inc + 1, inc - 1)
    assert checked > 200000


def test_ties_have_no_constant_increment():
    # s = 1.5 ulps of t: the sum lies exactly between two floats and goes to the even one -- from an odd mantissa the
    # first step adds 1 ulp, the next (now from an even mantissa) 2: the increment depends on the parity of t
    t = np.nextafter(np.float32(1.0), np.float32(2.0))      # mantissa ...001 (ulp 2^-23)
    s = np.float32(2.0 ** -24 * 3)
    tb, sb = int(t.view(np.uint32)), int(s.view(np.uint32))
    assert _inc(tb, sb) is None
    x1 = np.float32(t + s)
    x2 = np.float32(x1 + s)
    d1 = int(x1.view(np.uint32)) - tb
    d2 = int(x2.view(np.uint32)) - int(x1.view(np.uint32))
    assert (d1, d2) == (1, 2)


def test_cell_coordinate_and_position_are_monotone_in_t():
    """What lets a leap check only its two ends: fl(fma(b, t, a)) clamped and truncated, and fl(org + fl(dir * t)), do not
    decrease (increase) in t for b, dir >= 0 (<= 0)."""
    rng = np.random.default_rng(11)
    for _ in range(200):
        a, b = np.float32(rng.uniform(-5, 70)), np.float32(rng.uniform(-40, 40))
        org, dr = np.float32(rng.uniform(-1, 1)), np.float32(rng.uniform(-1, 1))
        ts = np.sort(rng.uniform(0, 2.5, 4000).astype(np.float32))
        u = (b.astype(np.float64) * ts.astype(np.float64) + a.astype(np.float64)).astype(np.float32)   # one rounding: an fma
        cell = np.clip(u, 0, 127).astype(np.int32)
        pos = (org + (dr * ts).astype(np.float32)).astype(np.float32)
        dc, dp = np.diff(cell), np.diff(pos)
        assert (dc >= 0).all() if b >= 0 else (dc <= 0).all()
        assert (dp >= 0).all() if dr >= 0 else (dp <= 0).all()
