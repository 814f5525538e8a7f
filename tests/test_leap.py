"""vr_leap (csrc/vr_leap.h): k steps of the reference's `t += stepSize` chain at once must give the
bits of the literal chain of k rounded fp32 additions (volumeraycast.cl:879) -- the march kernel
steps over runs of empty samples with it.  Compiled for the CPU here (-ffp-contract=off, no
excess precision) and compared with the literal loop on adversarial and random inputs."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("leap") / "libleap.so")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
                           "-msse2", "-mfpmath=sse", os.path.join(ROOT, "tests", "cxx", "leap_harness.c"),
                           "-o", so])
    L = C.CDLL(so)
    L.leap_check.restype = C.c_long
    L.leap_check.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.POINTER(C.c_long)]
    L.leap_fast.restype = C.c_float
    L.leap_fast.argtypes = [C.c_float, C.c_float, C.c_uint32]
    L.leap_literal.restype = C.c_float
    L.leap_literal.argtypes = [C.c_float, C.c_float, C.c_uint32]
    return L


def _check(L, t, step, k):
    t = np.ascontiguousarray(t, dtype=np.float32)
    step = np.ascontiguousarray(step, dtype=np.float32)
    k = np.ascontiguousarray(k, dtype=np.uint32)
    first = C.c_long(-1)
    bad = L.leap_check(t.ctypes.data, step.ctypes.data, k.ctypes.data, len(t), C.byref(first))
    if bad:
        i = first.value
        raise AssertionError("%d of %d differ; first: t=%r step=%r k=%d leap=%r literal=%r" % (
            bad, len(t), float(t[i]), float(step[i]), int(k[i]),
            L.leap_fast(float(t[i]), float(step[i]), int(k[i])),
            L.leap_literal(float(t[i]), float(step[i]), int(k[i]))))


def test_ray_like_inputs(lib):
    """t in [0, 8), step the size the kernel computes (1e-4 .. 1e-1), runs of 0 .. 600 samples."""
    rng = np.random.default_rng(1)
    n = 400000
    t = (rng.random(n) * 8).astype(np.float32)
    t[: n // 20] = 0.0                                   # rays starting inside the volume
    step = (10.0 ** rng.uniform(-4, -1, n)).astype(np.float32)
    k = rng.integers(0, 600, n).astype(np.uint32)
    _check(lib, t, step, k)


def test_long_runs_across_binades(lib):
    rng = np.random.default_rng(2)
    n = 20000
    t = (10.0 ** rng.uniform(-6, 1, n)).astype(np.float32)
    step = (10.0 ** rng.uniform(-5, 0, n)).astype(np.float32)
    k = rng.integers(0, 20000, n).astype(np.uint32)
    _check(lib, t, step, k)


def test_ties_round_half_to_even(lib):
    """step = (s + 1/2) ulp(t): every addition is a tie; s odd and even, odd and even mantissas."""
    rng = np.random.default_rng(3)
    ts, steps, ks = [], [], []
    for e in range(-3, 4):
        q = np.float32(2.0 ** (e - 23))
        for _ in range(3000):
            m = int(rng.integers(1 << 23, 1 << 24))
            s = int(rng.integers(1, 1 << 12))
            ts.append(np.float32(m) * q)
            steps.append(np.float32((s + 0.5)) * q)      # exact: s + 1/2 has few bits
            ks.append(int(rng.integers(0, 3000)))
    _check(lib, np.array(ts, np.float32), np.array(steps, np.float32), np.array(ks, np.uint32))


def test_raw_bit_patterns(lib):
    """Uniform over bit patterns of positive normal floats in a wide range, step <= 4 t."""
    rng = np.random.default_rng(4)
    n = 300000
    tb = rng.integers(0x30000000, 0x48000000, n).astype(np.uint32)
    t = tb.view(np.float32)
    sb = (tb.astype(np.int64) - rng.integers(-(2 << 23), 26 << 23, n)).clip(0x00800000, 0x7f000000).astype(np.uint32)
    step = sb.view(np.float32)
    k = rng.integers(0, 2000, n).astype(np.uint32)
    _check(lib, t, step, k)


def test_degenerate(lib):
    _check(lib, [0.0, 1.0, 1.0, 3.0, 1e-30, 5.0], [1.0, 1e-9, 1.0, 2.0 ** -24, 1e-30, 1e-3],
           [5, 100, 1 << 20, 1000, 50, 0])
