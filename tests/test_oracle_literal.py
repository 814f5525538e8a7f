"""How far are the oracle's PARITY DEFINITIONS from a LITERAL reading of the reference's source?

The HIP kernels are bit-equal to oracle/vr_oracle.c, which evaluates a few things in a form
chosen so that the kernel can match bit for bit: gradients in texel space (centre weights on
shifted texel indices), the UNORM scale applied once after the filter, the filter as nested
fmaf lerps, native_powr / log / sin / cos / atan2 / acos as fixed fp32 polynomial sequences,
normalize as v * (1 / |v|).  `literal=True` (vro_set_literal) evaluates all of these the way
/root/reference/src/kernel/volumeraycast.cl reads instead: six (27) full image reads at
pos -+ 1/res (:159-178, :217-277), c / 255.0f per texel before the OpenCL 1.2 spec 8.2 weighted
sum, libm for the builtins, v / |v|.

On every parity scene of the GPU suite the two must agree within north_star's tolerance, 1e-4
per channel: "kernel == oracle" then also means "kernel within 1e-4 of the literal reading".
The integer work counters may differ where a sample sits on a threshold (ERT 0.98, shading 0.1,
brick boundary); the test reports them and bounds the image."""
import numpy as np
import pytest

from oracle import vro
from tests import common, scenes

TOL = 1e-4   # per channel, float RGBA (BASELINE.json north_star)
WORST = {}


def _both(vol, fmt, tff, params, W, H, ess, **kw):
    cam, rp, rc, pt = params
    a, sa, _ = vro.render_tile(vol, fmt, tff, cam, rp, rc, pt, use_ess=ess, W=W, H=H, **kw)
    b, sb, _ = vro.render_tile(vol, fmt, tff, cam, rp, rc, pt, use_ess=ess, W=W, H=H, literal=True, **kw)
    assert np.isfinite(b).all()
    return a, b, sa, sb


def _check(name, a, b, sa, sb):
    d = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())
    WORST[name] = d
    print("literal vs parity definition: %-40s max|diff| = %.3g   samples %d / %d, shaded %d / %d" % (
        name, d, sa["samples_taken"], sb["samples_taken"], sa["samples_shaded"], sb["samples_shaded"]))
    assert d <= TOL, "%s: literal evaluation differs from the parity definitions by %.3g" % (name, d)
    # the same rays do (almost exactly) the same work
    assert sa["rays_hit"] == sb["rays_hit"] and sa["samples_nominal"] == sb["samples_nominal"]
    assert abs(sa["samples_taken"] - sb["samples_taken"]) <= max(4, sa["samples_taken"] // 10000)


@pytest.mark.parametrize("i", range(len(scenes.CASES)))
def test_raycast_scene(i):
    fmt, res, size, view, tff, kw = scenes.CASES[i]
    vol = common.noise_volume(res, fmt, seed=7, smooth=False)
    a, b, sa, sb = _both(vol, fmt, common.tffs()[tff], scenes.oracle_params(res, view, kw), size[0], size[1],
                         kw.get("ess", True))
    _check("CASES[%d]" % i, a, b, sa, sb)


@pytest.mark.parametrize("i", range(len(scenes.PT_CASES)))
def test_pathtrace_scene(i):
    fmt, res, size, view, tff, smooth, kw = scenes.PT_CASES[i]
    vol = common.noise_volume(res, fmt, seed=11, smooth=smooth)
    a, b, sa, sb = _both(vol, fmt, common.tffs()[tff], scenes.oracle_params(res, view, dict(kw, technique=1)),
                         size[0], size[1], True)
    _check("PT_CASES[%d]" % i, a, b, sa, sb)


@pytest.mark.parametrize("i", range(len(scenes.MC_CASES)))
def test_multichannel_scene(i):
    fmt, nch, res, view, kw = scenes.MC_CASES[i]
    vol = scenes.multichannel_volume(fmt, nch, res)
    a, b, sa, sb = _both(vol, fmt, common.tffs()["default"], scenes.oracle_params(res, view, kw), 80, 64,
                         kw.get("ess", True))
    _check("MC_CASES[%d]" % i, a, b, sa, sb)


@pytest.mark.parametrize("i", range(len(scenes.FP_CASES)))
def test_footprint_scene(i):
    fmt, res, size, view, tff, kw = scenes.FP_CASES[i]
    vol = common.noise_volume(res, fmt, seed=9, smooth=False)
    a, b, sa, sb = _both(vol, fmt, common.tffs()[tff], scenes.oracle_params(res, view, kw), size[0], size[1],
                         kw.get("ess", True))
    _check("FP_CASES[%d]" % i, a, b, sa, sb)


@pytest.mark.parametrize("kind,fmt", [("shells", 0), ("sphere", 1)])
def test_synthetic_benchmark_field(kind, fmt):
    """The bench workloads' fields at 128^3, default TF, shading, ESS, rot30."""
    vol = vro.synth_volume(kind, [128, 128, 128], fmt)
    a, b, sa, sb = _both(vol, fmt, common.tffs()["default"], scenes.oracle_params((128, 128, 128), "rot30", {}),
                         160, 128, True)
    _check("synthetic %s" % kind, a, b, sa, sb)


def test_environment_map_scene():
    rng = np.random.default_rng(5)
    env = rng.random((12, 32, 4)).astype(np.float32)
    vol = common.noise_volume((40, 40, 40), 0, seed=7, smooth=False)
    a, b, sa, sb = _both(vol, 0, common.tffs()["default"], scenes.oracle_params((40, 40, 40), "rot30", {}),
                         72, 56, True, env=env)
    _check("environment map", a, b, sa, sb)


def test_powr_definition_against_libm():
    """vro_powr (the parity definition of native_powr) against libm powf over the arguments the
    path uses: opacity correction powr(1 - a, 1 / rate) and the specular term powr(x, 40)."""
    L = vro.lib()
    worst = 0.0
    for y in (1.0 / 1.5, 1.0 / 0.7, 40.0):
        for x in np.linspace(0.0, 1.0, 20001, dtype=np.float32):
            ref = float(np.float32(np.power(np.float64(x), np.float64(np.float32(y)))))
            worst = max(worst, abs(float(L.vro_powr(float(x), float(y))) - ref))
    print("max |vro_powr - pow| on [0, 1]: %.3g" % worst)
    assert worst <= 2e-7
