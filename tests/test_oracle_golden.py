"""Pins the CPU oracle: (1) against the reference itself where it builds with no stand-ins
(oracle/_ref: random.cl RNG) -- only where /root/reference was available to build it;
(2) against the known-answer values the compiled reference produced (SURVEY.md App. E,
tests/golden/survey_kats.json); (3) host-side sizing rules."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import vro

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "survey_kats.json")))
REF_RNG = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libref_rng.so")


def test_rng_kats():
    L = vro.lib()
    for x, out in KATS["ParallelRNG"].items():
        assert L.vro_parallel_rng(int(x)) == out
    for k in KATS["ParallelRNG3"]:
        assert L.vro_parallel_rng3(k["x"], k["y"], k["z"]) == k["out"]
    for v, out in KATS["mapUintFloat"].items():
        assert L.vro_map_uint_float(int(v)) == out
    from volumerenderercl_amd import frontend
    mt = frontend.Mt19937()
    assert [mt() for _ in range(3)] == KATS["mt19937_default_first3"]


@pytest.mark.skipif(not os.path.exists(REF_RNG), reason="oracle/_ref not built (no /root/reference)")
def test_rng_bit_exact_vs_compiled_reference():
    """random.cl compiled from the reference's own source (no stand-ins, RTLD_LAZY)."""
    ref = C.CDLL(REF_RNG, mode=1)
    ref.ParallelRNG.restype = C.c_uint32
    ref.ParallelRNG.argtypes = [C.c_uint32]
    ref.ParallelRNG3.restype = C.c_uint32
    ref.ParallelRNG3.argtypes = [C.c_uint32] * 3
    ref.mapUintFloat.restype = C.c_float
    ref.mapUintFloat.argtypes = [C.c_uint32]
    L = vro.lib()
    rng = np.random.default_rng(1)
    xs = np.concatenate([np.arange(0, 2000), rng.integers(0, 2 ** 32, 20000),
                         [2 ** 32 - 1, 2 ** 31, 2 ** 31 - 1]]).astype(np.uint64)
    for x in xs:
        x = int(x)
        assert L.vro_parallel_rng(x) == ref.ParallelRNG(x)
        assert L.vro_map_uint_float(x) == ref.mapUintFloat(x)
    tr = rng.integers(0, 2 ** 32, (5000, 3))
    for a, b, c in tr:
        assert L.vro_parallel_rng3(int(a), int(b), int(c)) == ref.ParallelRNG3(int(a), int(b), int(c))
    for gx in range(0, 1032, 37):
        for gy in range(0, 1032, 41):
            assert L.vro_parallel_rng3(gx, gy, 3499211612) == ref.ParallelRNG3(gx, gy, 3499211612)


def test_intersect_bbox_kat():
    k = KATS["intersectBBox"]
    f3 = lambda v: (C.c_float * 3)(*v)
    tn, tf = C.c_float(), C.c_float()
    hit = vro.lib().vro_intersect_bbox(f3(k["orig"]), f3(k["dir"]), f3(k["lower"]), f3(k["upper"]),
                                       C.byref(tn), C.byref(tf))
    assert hit == k["hit"]
    assert abs(tn.value - k["tnear"]) < 2e-7 and abs(tf.value - k["tfar"]) < 3e-7


def test_generate_bricks_kat():
    """8^3 volume, 2^3 brick image: last voxel plane excluded."""
    k = KATS["generateBricks_8cube"]
    z, y, x = np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij")
    vol = ((x + 8 * y + 64 * z) % 256).astype(np.uint8)
    out = np.zeros((2, 2, 2, 2), dtype=np.uint8)
    u3 = lambda v: (C.c_uint32 * 3)(*v)
    rc = vro.lib().vro_generate_bricks(vol.ctypes.data_as(C.c_void_p), u3([8, 8, 8]), vro.UCHAR,
                                       u3(k["brick_image"]), out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    np.testing.assert_array_equal(out[..., 0], np.array(k["min"]))
    np.testing.assert_array_equal(out[..., 1], np.array(k["max"]))


def test_background_pixels_kat():
    """volumeRender entry (volumeraycast.cl:589-683): RNG jitter, padded-grid NDC, view
    transform, modelScale, gradient background -- all rays miss the moved clip box."""
    k = KATS["volumeRender_background"]
    cam = vro.CameraParams()
    cam.viewMat[:] = k["viewMat"]
    cam.bbox_bl[:] = k["bbox_bl"] + [0]
    cam.bbox_tr[:] = k["bbox_tr"] + [0]
    cam.ortho = k["ortho"]
    rp = vro.RenderingParams()
    rp.backgroundColor[:] = k["background"]
    rp.modelScale[:] = k["modelScale"] + [0]
    rp.illumType, rp.useLinear, rp.useGradient, rp.seed = 1, 1, k["useGradient"], k["seed"]
    rc = vro.RaycastParams()
    rc.samplingRate = 1.5
    rc.brickRes[:] = [8, 8, 8, 0]
    vol = np.zeros((8, 8, 8), np.uint8)
    tff = np.zeros((1024, 4), np.uint8)
    img, st, _ = vro.render_tile(vol, vro.UCHAR, tff, cam, rp, rc, use_ess=True, W=k["W"], H=k["H"])
    assert st["rays_hit"] == 0
    np.testing.assert_allclose(img[0, 0], k["pixel_0_0"], atol=1.5e-7, rtol=0)
    np.testing.assert_allclose(img[11, 19], k["pixel_19_11"], atol=1.5e-7, rtol=0)


def test_round_pow2_and_brick_layout():
    L = vro.lib()
    # volumerendercl.cpp:39-54: nearest of the two surrounding powers of two
    expect = {0: 0, 1: 1, 2: 2, 3: 4, 4: 4, 5: 4, 6: 8, 7: 8, 8: 8, 11: 8, 12: 16, 16: 16, 32: 32}
    for n, e in expect.items():
        assert L.vro_round_pow2(n) == e, n
    assert vro.brick_layout([256, 256, 256]) == ([4, 4, 4], [64.0, 64.0, 64.0], [64, 64, 64])
    assert vro.brick_layout([1024] * 3) == ([16] * 3, [64.0] * 3, [64] * 3)
    assert vro.brick_layout([2048] * 3) == ([32] * 3, [64.0] * 3, [64] * 3)
    assert vro.brick_layout([250, 129, 30]) == ([4, 2, 1], [62.5, 64.5, 30.0], [63, 65, 30])
    assert L.vro_padded(512) == 520 and L.vro_padded(1024) == 1032 and L.vro_padded(20) == 24


def test_calc_scaling():
    assert vro.calc_scaling([256, 256, 128], [1.0, 1.0, 1.0]) == [1.0, 1.0, 2.0]
    assert vro.calc_scaling([100, 100, 50], [1.0, 1.0, 2.0]) == [1.0, 1.0, 1.0]
    ms = vro.calc_scaling([64, 32, 16], [0.5, 1.0, 3.0])
    np.testing.assert_allclose(ms, [96 / 64, 96 / 64, 1.0], rtol=1e-6)


def test_powr_definition():
    """Parity definition of native_powr: exact at the ends, accurate in between."""
    L = vro.lib()
    for y in (1 / 1.5, 40.0, 0.2, 3.0):
        assert L.vro_powr(1.0, y) == 1.0
        assert L.vro_powr(0.0, y) == 0.0
    xs = np.linspace(1e-4, 1.0, 4001, dtype=np.float32)
    for y, tol in ((np.float32(1 / 1.5), 6e-7), (np.float32(40.0), 2e-5)):
        got = np.array([L.vro_powr(float(x), float(y)) for x in xs], dtype=np.float64)
        ref = np.power(xs.astype(np.float64), float(y))
        m = ref > 1e-30
        assert np.max(np.abs(got[m] - ref[m]) / ref[m]) < tol


def test_prefix_sum():
    tff = np.zeros((1024, 4), np.uint8)
    tff[:, 3] = np.arange(1024) % 251
    np.testing.assert_array_equal(vro.prefix_sum(tff), np.cumsum(tff[:, 3].astype(np.uint32)))


# ---- the builtin-free functions of the reference's kernel file itself -----------------
REF_KERNEL = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libref_kernel.so")
KFREE = json.load(open(os.path.join(HERE, "golden", "kernel_free.json")))


def _u4(st):
    return (C.c_uint32 * 4)(*[int(v) for v in st])


def test_hybrid_rng_and_box_edges_match_the_committed_reference_outputs():
    """tests/golden/kernel_free.json: what volumeraycast.cl's own ui_randStep / lcgStep /
    hybridui_rand (:48-80, calcAO's generator) and checkBoundingBox (:323-343, showEss) returned when
    the compiled reference was run here (oracle/gen_kernel_golden.py)."""
    L = vro.lib()
    for ch in KFREE["hybridui_rand_chains"]:
        st = _u4(ch["state"])
        for want in ch["draws_hex"]:
            got = L.vro_hybrid_rand(st)
            assert float(np.float32(got)).hex() == want
        assert list(st) == ch["final_state"]
    for k in KFREE["steps"]:
        st = _u4(k["state"])
        r = L.vro_ui_rand_step(st, *k["args"]) if k["fn"] == "ui_randStep" else L.vro_lcg_step(st, *k["args"])
        assert r == k["ret"] and list(st) == k["after"], k
    f3 = lambda v: (C.c_float * 3)(*v)
    for k in KFREE["checkBoundingBox"]:
        assert L.vro_check_bounding_box(f3(k["pos"]), f3(k["voxLen"]), k["bound"][0], k["bound"][1]) == k["ret"], k


@pytest.mark.skipif(not os.path.exists(REF_KERNEL), reason="oracle/_ref not built (no /root/reference)")
def test_hybrid_rng_bit_exact_vs_compiled_reference_kernel_file():
    """volumeraycast.cl compiled from the reference's own source (66 OpenCL builtins left undefined,
    RTLD_LAZY, nothing stands in for them): the functions that reach none of them, over 10^5 states."""
    ref = C.CDLL(REF_KERNEL, mode=1)
    U4 = C.POINTER(C.c_uint32)
    ref.refk_hybrid_rand.restype = C.c_float
    ref.refk_hybrid_rand.argtypes = [U4]
    ref.refk_ui_rand_step.restype = C.c_uint32
    ref.refk_ui_rand_step.argtypes = [U4, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint32]
    ref.refk_lcg_step.restype = C.c_uint32
    ref.refk_lcg_step.argtypes = [U4, C.c_uint32, C.c_uint32]
    ref.refk_check_bounding_box.restype = C.c_int
    ref.refk_check_bounding_box.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float]
    for name, n in (("ParallelRNG", 1), ("ParallelRNG3", 3)):   # the #include of random.cl, once more
        getattr(ref, name).restype = C.c_uint32
        getattr(ref, name).argtypes = [C.c_uint32] * n
    L = vro.lib()
    rng = np.random.default_rng(7)
    states = rng.integers(0, 2 ** 32, (100000, 4), dtype=np.uint64)
    states[:64, :] = np.arange(64, dtype=np.uint64)[:, None]           # small states
    states[64:128, 1:] = 0xFFFFFFFF
    for i, st in enumerate(states):
        a, b = _u4(st), _u4(st)
        ra, rb = L.vro_hybrid_rand(a), ref.refk_hybrid_rand(b)
        assert np.float32(ra).tobytes() == np.float32(rb).tobytes() and list(a) == list(b), (i, list(st))
        if i % 8 == 0:     # a second draw from the advanced state, and the single steps
            assert np.float32(L.vro_hybrid_rand(a)).tobytes() == np.float32(ref.refk_hybrid_rand(b)).tobytes()
            for args in ((0, 13, 19, 12, 4294967294), (1, 2, 25, 4, 4294967288), (2, 3, 11, 17, 4294967280)):
                a, b = _u4(st), _u4(st)
                assert L.vro_ui_rand_step(a, *args) == ref.refk_ui_rand_step(b, *args) and list(a) == list(b)
            a, b = _u4(st), _u4(st)
            assert L.vro_lcg_step(a, 1664525, 1013904223) == ref.refk_lcg_step(b, 1664525, 1013904223)
            assert list(a) == list(b)
            x, y, z = int(st[0]), int(st[1]), int(st[2])
            assert L.vro_parallel_rng(x) == ref.ParallelRNG(x)
            assert L.vro_parallel_rng3(x, y, z) == ref.ParallelRNG3(x, y, z)
    f3 = lambda v: (C.c_float * 3)(*v)
    for _ in range(20000):
        vox = rng.choice([1 / 32, 1 / 48, 1 / 256, 1 / 100], 3).astype(np.float32)
        pos = (rng.choice([0.0, 1.0, 0.5], 3) + rng.normal(0, 1.5, 3) * vox).astype(np.float32)
        b0, b1 = float(np.float32(rng.random() * 0.3)), float(np.float32(0.7 + rng.random() * 0.3))
        assert L.vro_check_bounding_box(f3(pos), f3(vox), b0, b1) == ref.refk_check_bounding_box(f3(pos), f3(vox), b0, b1)


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(REF_KERNEL), "ref_kernel.o")),
                    reason="oracle/_ref not built (no /root/reference)")
def test_list_of_builtin_free_reference_functions():
    """What can be pinned is exactly what oracle/ref_kernel_free.py finds in the compiled object: nothing
    that needs an OpenCL builtin is executed, and nothing executable is left out."""
    from oracle import ref_kernel_free
    res = ref_kernel_free.analyse(os.path.join(os.path.dirname(REF_KERNEL), "ref_kernel.o"))
    free = sorted(f for f, undefined in res.items() if not undefined)
    assert free == sorted(["ParallelRNG", "ParallelRNG2", "ParallelRNG3", "map256", "mapUintFloat", "ui_randStep",
                           "lcgStep", "hybridui_rand", "getf4", "checkBoundingBox"])
    assert len(res["volumeRender"]) >= 50 and res["generateBricks"] and res["intersectBBox"]
