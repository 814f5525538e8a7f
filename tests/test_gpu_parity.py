"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  The north_star's bar is 1e-4 per channel; the kernels execute the oracle's
fp32 operation sequences, so the tests ask for a difference of exactly 0 (VRHIP_TEST_TOL=1e-4
relaxes them to the bar) and the integer work counters must match bit for bit.
"""
import ctypes
import os

import numpy as np
import pytest

from oracle import vro
from tests import common
from tests import scenes
from tests.scenes import CASES, FP_CASES, MC_CASES, PT_CASES
from volumerenderercl_amd import FLOAT, UCHAR, USHORT, VolumeRenderCL, frontend

pytestmark = pytest.mark.gpu

TOL = float(os.environ.get("VRHIP_TEST_TOL", "0"))   # per channel, float RGBA: bit-exact (north_star allows 1e-4)
SEED = 3499211612


@pytest.fixture(scope="module")
def vr():
    r = VolumeRenderCL()
    r.initialize()
    yield r
    r.close()


def _setup(vr, vol, fmt, tff, view, **kw):
    vr.loadVolumeArrays([vol], fmt, thickness=kw.get("thickness", (1.0, 1.0, 1.0)))
    vr.setTransferFunction(tff)
    vr.setSeed(kw.get("seed", SEED))
    vr.setIllumination(kw.get("illum", 1))
    vr.setLinearInterpolation(kw.get("linear", True))
    vr.setCamOrtho(kw.get("ortho", False))
    vr.setUseGradient(kw.get("gradient_bg", False))
    vr.setContours(kw.get("contours", False))
    vr.setAerial(kw.get("aerial", False))
    vr.setObjEss(kw.get("ess", True))
    vr.updateSamplingRate(kw.get("rate", 1.5))
    vr.setAmbientOcclusion(kw.get("ao", False))
    vr.setTechnique(kw.get("technique", 0))
    vr.setExtinction(kw.get("ext", 100.0))
    bb = kw.get("bbox", (-1, -1, -1, 1, 1, 1))
    vr.setBBox(*bb)
    vr.setShowESS(kw.get("show_ess", False))
    vr.setImgEss(kw.get("img_ess", False))
    if "background" in kw:
        vr.setBackground(kw["background"])          # (alpha becomes 0 like in the reference)
    else:
        vr.params()[1].backgroundColor[:] = [1.0, 1.0, 1.0, 1.0]
    vr.updateView(view)
    vr.setIteration(0)


def _compare(vr, vol, fmt, tff, W, H, ess=True, pathtrace=False):
    """One frame through the INSTRUMENTED kernels (stats on: images + the six work counters) and
    the same frame through the PRODUCTION kernels (stats off: DDA pre-pass, ray list, two-phase
    march, footprint volume where it applies) -- both against the oracle."""
    it = vr.params()[1].iteration
    vr.setStatsEnabled(True)
    got = vr.runRaycastNoGL(W, H)
    gstats = vr.getStats()
    seed = vr.params()[1].seed    # the jitter seed this frame was rendered with
    vr.setIteration(it)
    ref, rstats, _ = common.oracle_frame(vr, vol, fmt, tff, W, H, use_ess=ess)
    diff = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    assert np.isfinite(got).all()
    assert diff.max() <= TOL, "max abs diff %.3g at %s" % (
        diff.max(), np.unravel_index(diff.argmax(), diff.shape))
    if pathtrace:
        # technique 1 reuses the two brick counters for its majorant-grid culling (steps whose
        # bound was consulted / whose voxel fetch was skipped); the oracle has no such grid
        # (and samples_nominal for the steps among them that were taken in leaps over a macro cell)
        assert gstats["samples_nominal"] <= gstats["bricks_skipped"] <= gstats["bricks_visited"] <= gstats["samples_taken"]
        gstats = dict(gstats, bricks_visited=0, bricks_skipped=0, samples_nominal=0)
    assert gstats == rstats
    # the production (uninstrumented) instantiation of the same frame
    vr.setStatsEnabled(False)
    pinned = vr._fixed_seed
    vr.setSeed(seed)              # (the frame above may have drawn from the mt19937 sequence)
    prod = vr.runRaycastNoGL(W, H)
    vr.setSeed(pinned)
    vr.setIteration(it)
    pdiff = np.abs(prod.astype(np.float64) - ref.astype(np.float64))
    assert np.isfinite(prod).all()
    assert pdiff.max() <= TOL, "production kernels: max abs diff %.3g at %s" % (
        pdiff.max(), np.unravel_index(pdiff.argmax(), pdiff.shape))
    return got, ref, gstats




def _same_params(vr, res, view, kw):
    """The renderer-free parameter builder of the CPU tests (tests/scenes.py) produces the
    product's own kernel-argument structs, byte for byte."""
    mine = scenes.oracle_params(res, view, kw, seed=vr.params()[1].seed)
    mine[1].iteration = vr.params()[1].iteration
    for a, b in zip(mine, vr.params()):
        assert bytes(a) == bytes(b), type(a).__name__


@pytest.mark.parametrize("fmt,res,size,view,tff,kw", CASES)
def test_frame_matches_oracle(vr, fmt, res, size, view, tff, kw):
    vol = common.noise_volume(res, fmt, seed=7, smooth=False)
    table = common.tffs()[tff]
    _setup(vr, vol, fmt, table, common.views()[view], **kw)
    got, ref, stats = _compare(vr, vol, fmt, table, size[0], size[1], ess=kw.get("ess", True))
    _same_params(vr, res, view, kw)
    assert stats["rays_hit"] > 0 and stats["samples_taken"] > 0




@pytest.mark.parametrize("fmt,res,size,view,tff,smooth,kw", PT_CASES)
def test_pathtrace_frame_matches_oracle(vr, fmt, res, size, view, tff, smooth, kw):
    """technique 1 (volumeraycast.cl:686-706, trace_volume :463-503): one sample per pixel."""
    vol = common.noise_volume(res, fmt, seed=11, smooth=smooth)
    table = common.tffs()[tff]
    _setup(vr, vol, fmt, table, common.views()[view], technique=1, **kw)
    got, ref, stats = _compare(vr, vol, fmt, table, size[0], size[1], pathtrace=True)
    _same_params(vr, res, view, dict(kw, technique=1))
    assert stats["rays_hit"] > 0 and stats["samples_taken"] > stats["rays_hit"]
    assert np.ptp(ref[..., :3]) > 0.05   # something was traced


def test_pathtrace_accumulates_like_oracle(vr):
    """Progressive rendering: running mean over iterations with the std::mt19937 seed stream
    (volumerendercl.cpp:212, volumeraycast.cl:689-704)."""
    vol = common.noise_volume((40, 40, 40), FLOAT, seed=4, smooth=True)
    table = common.tffs()["default"]
    _setup(vr, vol, FLOAT, table, common.views()["rot30"], technique=1)
    mt = frontend.Mt19937()
    W, H = 64, 48
    ref = None
    for it in range(4):
        seed = mt()
        vr.setSeed(seed)
        vr.setIteration(it)
        got = vr.runRaycastNoGL(W, H)
        vr.setIteration(it)     # runRaycastNoGL advanced it; the oracle needs the same value
        ref, _, _ = common.oracle_frame(vr, vol, FLOAT, table, W, H, in_accum=ref)
        np.testing.assert_array_equal(got, ref)
    vr.setIteration(0)


@pytest.mark.parametrize("view", ["far", "rot30", "close", "inside", "offaxis"])
def test_uninstrumented_frame_matches_oracle(vr, view):
    """The production kernels (no statistics): DDA pre-pass + live patch list, phase 1, sorted
    phase 2.  Most other parity cases run the instrumented variants, which skip the pre-pass."""
    vol = common.noise_volume((64, 64, 64), UCHAR, seed=31, smooth=True)
    vol[vol < 60] = 0
    table = common.tffs()["default"]
    views = dict(common.views())
    views["far"] = frontend.view_matrix(translation=(0.0, 0.0, 6.0))
    views["offaxis"] = frontend.view_matrix(frontend.quat_from_axis_angle((0.3, 1, 0.2), 70.0),
                                            translation=(0.9, -0.6, 3.5))
    W, H = 200, 136
    _setup(vr, vol, UCHAR, table, views[view])
    vr.setStatsEnabled(False)
    ref = None
    for it in range(2):     # second frame: accumulation + phase-2 order from the first frame's costs
        vr.setIteration(it)
        got = vr.runRaycastNoGL(W, H)
        vr.setIteration(it)
        ref, _, _ = common.oracle_frame(vr, vol, UCHAR, table, W, H, in_accum=ref)
        np.testing.assert_array_equal(got, ref)
    vr.setIteration(0)


def test_empty_skipping_is_exact(vr, monkeypatch):
    """Ray caster: stepping over runs of samples in empty cells (opacity exactly 0) leaves the
    image and every work counter unchanged."""
    vol = common.noise_volume((96, 80, 72), UCHAR, seed=22, smooth=True)
    vol[vol < 90] = 0          # large exactly-empty regions next to structure
    table = common.tffs()["default"]
    W, H = 128, 96
    outs = []
    monkeypatch.setenv("VRHIP_EMPTY_SKIP", "1")      # (by default only where the ESS bricks are >= 32 voxels)
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("VRHIP_NO_EMPTY_SKIP", env)
        r2 = VolumeRenderCL()
        r2.initialize()
        try:
            for ess in (True, False):
                _setup(r2, vol, UCHAR, table, common.views()["rot30"], ess=ess)
                r2.setStatsEnabled(True)
                outs.append((r2.runRaycastNoGL(W, H), r2.getStats()))
            if not env:
                r2.setIteration(0)
                ref, rstats, _ = common.oracle_frame(r2, vol, UCHAR, table, W, H, use_ess=False)
        finally:
            r2.close()
    for i in range(2):
        np.testing.assert_array_equal(outs[i][0], outs[i + 2][0])
        assert outs[i][1] == outs[i + 2][1]
    np.testing.assert_array_equal(outs[1][0], ref)
    assert outs[1][1] == rstats


@pytest.mark.parametrize("env", [{"VRHIP_CULL_RADIUS": "0"}, {"VRHIP_CULL_RADIUS": "12"},
                                 {"VRHIP_EMPTY_SKIP": "1"}, {"VRHIP_EMPTY_SKIP": "1", "VRHIP_CELL_SHIFT": "3"}])
def test_schedules_and_culling_do_not_change_pixels(vr, monkeypatch, env):
    """Scheduling devices of the product library -- patch culling in the DDA pre-pass (off / wide radius),
    empty-run skipping forced on where the heuristic leaves it off (empty bits on cells of 4 voxels, and
    of 8: one grid for bounds and bits) -- on a 256^3 field with large empty regions, bricks of 4 voxels,
    three views, two seeds: bit-identical to the default schedule's frames and equal to the oracle's.
    (The experiment kernels of round 2 live in A/B builds only: test_experiment_kernels_*.)"""
    res = (256, 256, 256)
    vol = vro.synth_volume("shells", list(res), UCHAR)
    vol[:, :, :96] = 0            # a large empty slab the culling can work with
    tff = common.tffs()["default"]
    W, H = 200, 152
    frames = {}
    for name, e in (("default", {"VRHIP_EMPTY_SKIP": "1"}), ("variant", dict(env, VRHIP_EMPTY_SKIP="1"))):
        for k, v in e.items():
            monkeypatch.setenv(k, v)
        r2 = VolumeRenderCL()
        r2.initialize()
        try:
            r2.loadVolumeArrays([vol], UCHAR)
            r2.setTransferFunction(tff)
            for view in ("rot30", "close", "inside"):
                r2.updateView(common.views()[view])
                for seed in (SEED, 581869302):
                    r2.setSeed(seed)
                    r2.setIteration(0)
                    frames[(name, view, seed)] = r2.runRaycastNoGL(W, H)
            if name == "default":
                r2.updateView(common.views()["rot30"])
                r2.setSeed(SEED)
                r2.setIteration(0)
                cam, rp, rc, pt = common.to_oracle_params(*r2.params())
                rp.seed, rp.iteration = SEED, 0
                ref, _, _ = vro.render_tile(vol, UCHAR, tff, cam, rp, rc, pt, W=W, H=H)
        finally:
            r2.close()
        for k in e:
            monkeypatch.delenv(k, raising=False)
    np.testing.assert_array_equal(frames[("default", "rot30", SEED)], ref)
    for (name, view, seed), img in frames.items():
        if name == "variant":
            np.testing.assert_array_equal(img, frames[("default", view, seed)], err_msg="%s %s" % (view, seed))


@pytest.mark.parametrize("fmt,res", [(UCHAR, (96, 80, 72)), (USHORT, (70, 33, 50)), (FLOAT, (64, 64, 40))])
def test_cell_grid_bounds(fmt, res, monkeypatch):
    """The cell grids (min, max of the voxels a fetch in or next to a cell of E^3 voxels can read; E = 8
    for the opacity bounds, 4 for the ray caster's empty bits): the one-wave-per-cell kernel and the
    separable streaming build (x and y ranges per voxel slice from whole micro-brick lines, then the z
    range; the coarse grid reduced from the fine one) all give exactly the extrema of the voxels
    [E c - 1, E c + E + 1]^3 clipped to the volume."""
    vol = common.noise_volume(res, fmt, seed=31, smooth=False)
    got = {}
    for name, env in (("wave", "1"), ("stream", None)):
        if env:
            monkeypatch.setenv("VRHIP_CELLS_PER_WAVE", env)
        else:
            monkeypatch.delenv("VRHIP_CELLS_PER_WAVE", raising=False)
        r2 = VolumeRenderCL()
        r2.initialize()
        try:
            r2.loadVolumeArrays([vol], fmt)
            r2.setTransferFunction(common.tffs()["default"])
            got[name, 8], shift = r2.downloadCells()           # built directly (no fine grid yet)
            assert shift == 3
            got[name, 4], shift = r2.downloadCells(fine=True)
            assert shift == 2
        finally:
            r2.close()
        r3 = VolumeRenderCL()
        r3.initialize()
        try:
            r3.loadVolumeArrays([vol], fmt)
            r3.setTransferFunction(common.tffs()["default"])
            fine3, _ = r3.downloadCells(fine=True)
            coarse3, _ = r3.downloadCells()                    # reduced from the fine grid
            np.testing.assert_array_equal(fine3, got[name, 4])
            np.testing.assert_array_equal(coarse3, got[name, 8])
        finally:
            r3.close()
    v = vol.astype(np.float32)
    for E in (8, 4):
        cz, cy, cx = got["wave", E].shape[:3]
        assert (cz, cy, cx) == tuple(-(-n // E) for n in v.shape)
        want = np.empty_like(got["wave", E])
        for k in range(cz):
            for j in range(cy):
                for i in range(cx):
                    box = v[max(E * k - 1, 0):E * k + E + 2, max(E * j - 1, 0):E * j + E + 2,
                            max(E * i - 1, 0):E * i + E + 2]
                    want[k, j, i] = (box.min(), box.max())
        np.testing.assert_array_equal(got["wave", E], want)
        np.testing.assert_array_equal(got["stream", E], want)


def test_pathtrace_culling_is_exact(vr, monkeypatch):
    """The majorant grid only skips fetches that cannot change the walk: the image with and
    without it is identical, and it does skip a large share of the fetches."""
    vol = common.noise_volume((96, 80, 72), FLOAT, seed=21, smooth=True)
    table = common.tffs()["default"]
    W, H = 128, 96
    _setup(vr, vol, FLOAT, table, common.views()["rot30"], technique=1)
    vr.setStatsEnabled(True)
    got = vr.runRaycastNoGL(W, H)
    st = vr.getStats()
    vr.setStatsEnabled(False)
    assert st["bricks_visited"] == st["samples_taken"] and st["bricks_skipped"] > 0
    monkeypatch.setenv("VRHIP_PT_NO_CULL", "1")
    r2 = VolumeRenderCL()
    r2.initialize()
    try:
        _setup(r2, vol, FLOAT, table, common.views()["rot30"], technique=1)
        r2.setStatsEnabled(True)
        plain = r2.runRaycastNoGL(W, H)
        st2 = r2.getStats()
    finally:
        r2.close()
    np.testing.assert_array_equal(got, plain)
    assert st2["samples_taken"] == st["samples_taken"] and st2["bricks_skipped"] == 0


def test_pathtrace_leaps_are_exact(vr, monkeypatch):
    """A walk takes all its steps inside a macro cell (4^3 cells of the bound grid) whose bound is below its threshold
    at once -- or inside the cube of macro cells around it that are free as well (vr_pathtrace.hip): image, step count and
    culled steps are those of the kernel that takes every step on its own (VRHIP_PT_NO_LEAP) and of the one whose leaps
    stay inside one macro cell (VRHIP_PT_NO_FAR_LEAP), on a field with large empty regions and on noise, three
    accumulating iterations each -- and most steps of the first are taken in leaps."""
    table = common.tffs()["default"]
    zz, yy, xx = np.meshgrid(np.linspace(-1, 1, 128), np.linspace(-1, 1, 144), np.linspace(-1, 1, 160), indexing="ij")
    ball = np.clip(1.0 - np.sqrt(xx * xx + yy * yy + zz * zz) / 0.5, 0, 1).astype(np.float32)   # empty beyond r = 0.5
    for name, vol in (("sphere", ball), ("noise", common.noise_volume((96, 80, 72), FLOAT, seed=21, smooth=True))):
        W, H = 160, 120
        out = {}
        for mode in ("far", "near", "none"):
            monkeypatch.delenv("VRHIP_PT_NO_LEAP", raising=False)
            monkeypatch.delenv("VRHIP_PT_NO_FAR_LEAP", raising=False)
            if mode == "none":
                monkeypatch.setenv("VRHIP_PT_NO_LEAP", "1")
            elif mode == "near":
                monkeypatch.setenv("VRHIP_PT_NO_FAR_LEAP", "1")
            r2 = VolumeRenderCL()
            r2.initialize()
            try:
                _setup(r2, vol, FLOAT, table, common.views()["rot30"], technique=1)
                r2.setStatsEnabled(True)
                frames, stats = [], []
                for it in range(3):
                    frames.append(r2.runRaycastNoGL(W, H).copy())
                    stats.append(r2.getStats())
                r2.setStatsEnabled(False)
                r2.setIteration(0)
                plain = r2.runRaycastNoGL(W, H).copy()       # the production kernel
            finally:
                r2.close()
            out[mode] = (frames, stats, plain)
        for mode in ("far", "near"):
            for a, b in zip(out[mode][0], out["none"][0]):
                np.testing.assert_array_equal(a, b)
            np.testing.assert_array_equal(out[mode][2], out["none"][2])
            np.testing.assert_array_equal(out[mode][2], out[mode][0][0])
            for sa, sb in zip(out[mode][1], out["none"][1]):
                assert sb["samples_nominal"] == 0
                assert dict(sa, samples_nominal=0) == sb, (name, mode)
        if name == "sphere":
            far, near = out["far"][1][0], out["near"][1][0]
            assert near["samples_nominal"] > 0.5 * near["samples_taken"], near
            assert far["samples_nominal"] > near["samples_nominal"], (far, near)   # the cube reaches further


def test_pathtrace_tiles_equal_full_frame(vr):
    import torch
    vol = common.noise_volume((40, 40, 40), FLOAT, seed=6, smooth=True)
    table = common.tffs()["default"]
    W, H, TW, TH = 120, 70, 32, 32
    _setup(vr, vol, FLOAT, table, common.views()["rot30"], technique=1)
    full = vr.runRaycastNoGL(W, H)
    vr.setIteration(0)
    tiles_x, tiles_y = (W + TW - 1) // TW, (H + TH - 1) // TH
    ids = np.arange(tiles_x * tiles_y, dtype=np.uint32)[::2].copy()
    out = torch.zeros((len(ids), TH, TW, 4), dtype=torch.float32, device="cuda")
    vr.render_tiles(W, H, TW, TH, ids, out.data_ptr())
    torch.cuda.synchronize()
    vr.setIteration(0)
    o = out.cpu().numpy()
    for k, t in enumerate(ids):
        tx, ty = int(t) % tiles_x, int(t) // tiles_x
        x0, y0 = tx * TW, ty * TH
        w, h = min(TW, W - x0), min(TH, H - y0)
        np.testing.assert_array_equal(o[k, :h, :w], full[y0:y0 + h, x0:x0 + w])


def test_empty_volume_is_background(vr):
    """Analytic case: all-zero volume + default TF -> every pixel is the background."""
    vol = np.zeros((32, 32, 32), dtype=np.uint8)
    _setup(vr, vol, UCHAR, frontend.tff_from_stops(), common.views()["default"])
    got = vr.runRaycastNoGL(64, 64)
    np.testing.assert_array_equal(got[..., :3], np.ones((64, 64, 3), np.float32))
    np.testing.assert_array_equal(got[..., 3], np.zeros((64, 64), np.float32))


@pytest.mark.parametrize("fmt,res", [(UCHAR, (64, 64, 64)), (UCHAR, (100, 50, 30)),
                                     (UCHAR, (129, 65, 250)), (USHORT, (96, 96, 96)),
                                     (USHORT, (70, 130, 33)), (FLOAT, (64, 64, 64)),
                                     (FLOAT, (130, 40, 77)), (UCHAR, (256, 256, 256))])
def test_bricks_match_oracle(vr, fmt, res):
    """generateBricks (volumeraycast.cl:932-961): bit-exact min/max grid, last plane excluded."""
    vol = common.noise_volume(res, fmt, seed=3, smooth=False)
    vr.loadVolumeArrays([vol], fmt)
    vr.setTransferFunction(frontend.tff_from_stops())
    tex, brf, edge = vr.brickInfo()
    e_edge, e_brf, e_tex = vro.brick_layout(res)
    assert tex == e_tex and edge == e_edge and brf == e_brf
    got = vr.downloadBricks()
    ref = vro.generate_bricks(vol, fmt)
    np.testing.assert_array_equal(got, ref)


def test_bricks_known_answer_survey_app_e(vr):
    """8^3 known-answer values recorded from the compiled reference.
    The reference probe used a 2^3 brick image; here the grid comes from the host sizing
    rule (8/64 -> edge 1 -> 8^3 bricks), so the KAT is checked through the oracle in
    tests/test_oracle_golden.py; this test checks GPU == oracle on that same volume."""
    z, y, x = np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij")
    vol = ((x + 8 * y + 64 * z) % 256).astype(np.uint8)
    vr.loadVolumeArrays([vol], UCHAR)
    vr.setTransferFunction(frontend.tff_from_stops())
    np.testing.assert_array_equal(vr.downloadBricks(), vro.generate_bricks(vol, UCHAR))


def test_tiles_equal_full_frame(vr):
    """Image-tile decomposition: any tile rendered alone equals the same pixels
    of the full frame (the camera uses the padded full-frame size, App. A.2)."""
    import torch
    vol = common.noise_volume((48, 48, 48), UCHAR, seed=5)
    tff = frontend.tff_from_stops()
    W, H, TW, TH = 150, 100, 32, 48
    _setup(vr, vol, UCHAR, tff, common.views()["rot30"])
    full = vr.runRaycastNoGL(W, H)
    vr.setIteration(0)
    tiles_x, tiles_y = (W + TW - 1) // TW, (H + TH - 1) // TH
    ids = np.array([t for t in range(tiles_x * tiles_y) if t % 3 != 1], dtype=np.uint32)
    out = torch.zeros((len(ids), TH, TW, 4), dtype=torch.float32, device="cuda")
    vr.render_tiles(W, H, TW, TH, ids, out.data_ptr())
    torch.cuda.synchronize()
    vr.getLastExecTime()
    o = out.cpu().numpy()
    for k, t in enumerate(ids):
        tx, ty = int(t) % tiles_x, int(t) // tiles_x
        x0, y0 = tx * TW, ty * TH
        w, h = min(TW, W - x0), min(TH, H - y0)
        np.testing.assert_array_equal(o[k, :h, :w], full[y0:y0 + h, x0:x0 + w])
        # oracle agrees on the tile as well
    ref, _, _ = common.oracle_frame(vr, vol, UCHAR, tff, W, H, tile=(32, 48, 32, 48))
    assert np.abs(ref - full[48:96, 32:64]).max() <= TOL


def test_accumulation_running_mean(vr):
    """volumeraycast.cl:898-909 with fp32 accumulation: frame k is the
    running mean of frames 0..k, each with its own jitter seed."""
    vol = common.noise_volume((40, 40, 40), UCHAR, seed=9)
    tff = frontend.tff_from_stops()
    W, H = 64, 48
    _setup(vr, vol, UCHAR, tff, common.views()["rot30"])
    acc = None
    for it, seed in enumerate(frontend.MT19937_FIRST_SEEDS):
        vr.setSeed(seed)
        vr.setIteration(it)
        got = vr.runRaycastNoGL(W, H)
        vr.setIteration(it)
        ref, _, _ = common.oracle_frame(vr, vol, UCHAR, tff, W, H, in_accum=acc)
        assert np.abs(got - ref).max() <= TOL
        acc = ref


def test_default_seed_sequence(vr):
    """Without a pinned seed the frames use std::mt19937()'s outputs."""
    r = VolumeRenderCL()
    r.initialize()
    vol = common.noise_volume((32, 32, 32), UCHAR, seed=1)
    r.loadVolumeArrays([vol], UCHAR)
    r.setTransferFunction(frontend.tff_from_stops())
    r.updateView(frontend.view_matrix())
    seeds = []
    for _ in range(3):
        r.runRaycastNoGL(32, 32)
        seeds.append(int(r.params()[1].seed))
    r.close()
    assert tuple(seeds) == frontend.MT19937_FIRST_SEEDS


def test_touched_microbricks_match_oracle(vr):
    """Compulsory-traffic instrumentation (SURVEY 8d B_frame): the set of 4^3 micro-bricks
    touched by any voxel fetch is identical on GPU and in the oracle."""
    vol = common.noise_volume((64, 64, 64), UCHAR, seed=11)
    tff = frontend.tff_from_stops()
    W, H = 96, 96
    _setup(vr, vol, UCHAR, tff, common.views()["rot30"])
    n, bm = vr.countTouched(W, H, want_bitmap=True)
    _, _, ref_bm = common.oracle_frame(vr, vol, UCHAR, tff, W, H, want_touched=True)
    np.testing.assert_array_equal(bm, ref_bm)
    assert n == int(np.unpackbits(ref_bm).sum()) and n > 0


def test_pathtrace_touched_microbricks_match_oracle(vr):
    vol = common.noise_volume((64, 64, 64), FLOAT, seed=12)
    tff = frontend.tff_from_stops()
    W, H = 80, 72
    _setup(vr, vol, FLOAT, tff, common.views()["rot30"], technique=1)
    n, bm = vr.countTouched(W, H, want_bitmap=True)
    _, _, ref_bm = common.oracle_frame(vr, vol, FLOAT, tff, W, H, want_touched=True)
    np.testing.assert_array_equal(bm, ref_bm)
    assert n == int(np.unpackbits(ref_bm).sum()) and n > 0


def test_error_behaviour(vr):
    """Errors surface as exceptions with the reference's meaning."""
    r = VolumeRenderCL()
    r.initialize()
    assert r.runRaycastNoGL(32, 32) is None          # silent no-op without data
    r.setTransferFunction(np.zeros(4096, np.uint8))  # silent no-op without data
    with pytest.raises(RuntimeError):
        r.loadVolumeArrays([np.zeros(10, np.uint8)], UCHAR, res=(4, 4, 4))
    vol = common.noise_volume((16, 16, 16), UCHAR)
    r.loadVolumeArrays([vol], UCHAR)
    with pytest.raises(RuntimeError):                # no TF yet -> bricks missing
        r.runRaycastNoGL(32, 32)
    r.setTransferFunction(frontend.tff_from_stops())
    r.updateView(frontend.view_matrix())
    r.setIllumination(6)
    with pytest.raises((RuntimeError, ValueError)):  # unknown shading mode, loud
        r.runRaycastNoGL(32, 32)
    r.setIllumination(1)
    r.setImgEss(True)
    r.setTechnique(1)
    with pytest.raises(RuntimeError):                # unsupported combination, loud
        r.runRaycastNoGL(32, 32)
    r.close()


def test_dat_file_through_loader_matches_oracle(vr):
    """'Same .dat/.raw input': a USHORT .dat (values stretched to 65535 by the loader, SURVEY
    C15) goes through loadVolumeData and renders like the oracle fed with the reference
    loader's bytes (tests/golden/loader)."""
    import base64
    import json
    import os
    from volumerenderercl_amd import datraw
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loader")
    case = [c for c in json.load(open(os.path.join(gold, "expected.json")))["cases"]
            if c["case"] == "c3"][0]
    n = vr.loadVolumeData(datraw.Properties(os.path.join(gold, "c3.dat")))
    assert n == 1 and vr.getResolution() == case["res"]
    tff = frontend.opaque_ramp_tff()
    vr.setTransferFunction(tff)
    vr.setSeed(SEED)
    vr.setIllumination(1)
    vr.setObjEss(True)
    vr.setBBox(-1, -1, -1, 1, 1, 1)
    vr.updateView(common.views()["rot30"])
    vr.setIteration(0)
    got = vr.runRaycastNoGL(64, 48)
    vr.setIteration(0)
    ref_vol = np.frombuffer(base64.b64decode(case["data"][0]), dtype=np.uint16).reshape(5, 4, 3)
    ref, _, _ = common.oracle_frame(vr, ref_vol, USHORT, tff, 64, 48)
    assert np.abs(got - ref).max() <= TOL


def test_cpp_host_cli_matches_oracle(tmp_path):
    """The C++ host (VolumeRenderCL + DatRawReader + vrhip_render CLI) end to end: render a
    golden .dat headless, read the float RGBA file back, compare with the oracle."""
    import os
    import subprocess
    from volumerenderercl_amd import datraw
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "volumerenderercl_amd", "vrhip_render")
    dat = os.path.join(root, "tests", "golden", "loader", "c1.dat")
    out = str(tmp_path / "frame")
    W, H = 72, 40
    cmd = [exe, "--dat", dat, "--size", str(W), str(H), "--rotate", "1", "1", "0", "30",
           "--seed", str(SEED), "--out", out]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    got = np.fromfile(out + ".rgba.f32", dtype=np.float32).reshape(H, W, 4)
    assert os.path.getsize(out + ".ppm") > W * H * 3
    # same inputs for the oracle
    rd = datraw.DatRawReader()
    rd.read_files(datraw.Properties(dat))
    p = rd.properties()
    vol = rd.data()[0].reshape(p.volume_res[2], p.volume_res[1], p.volume_res[0])
    tff = frontend.tff_from_stops()
    cam = vro.CameraParams()
    cam.viewMat[:] = frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30.0))
    cam.bbox_bl[:] = [-1, -1, -1, 0]
    cam.bbox_tr[:] = [1, 1, 1, 0]
    rp = vro.RenderingParams()
    rp.backgroundColor[:] = [1, 1, 1, 0]   # setBackground zeroes alpha (volumerendercl.cpp:1027)
    rp.modelScale[:] = vro.calc_scaling(p.volume_res[:3], p.slice_thickness) + [0]
    rp.illumType, rp.useLinear, rp.seed = 1, 1, SEED
    rc = vro.RaycastParams()
    rc.samplingRate = 1.5
    _, brf, _ = vro.brick_layout(p.volume_res[:3])
    rc.brickRes[:] = brf + [0]
    ref, _, _ = vro.render_tile(vol, vro.UCHAR, tff, cam, rp, rc, W=W, H=H)
    assert np.abs(got - ref).max() <= TOL


@pytest.mark.parametrize("args,frames", [(["--ranks", "1"], 1), (["--ranks", "3", "--loopback", "--tile", "32"], 1),
                                         # RCCL on one GPU: ncclCommInitAll over one device, the root's
                                         # grouped ncclSend / ncclRecv to itself, assembly behind the receive
                                         (["--ranks", "1", "--force-gather", "--tile", "32"], 2),
                                         (["--ranks", "1", "--force-gather", "--tile", "48", "--pathtrace"], 2),
                                         (["--ranks", "2", "--loopback", "--tile", "16"], 3),
                                         (["--ranks", "4", "--loopback", "--tile", "48", "--pathtrace"], 2)])
def test_cpp_host_tile_ranks_equal_single_renderer(tmp_path, args, frames):
    """vrhip_render --ranks N (csrc/host/tilegather.cpp): one renderer per rank with the whole
    volume, interleaved tiles, gather to rank 0 (RCCL send/recv between GPUs; `--loopback` keeps
    every rank on this box's one GPU and moves the tiles by device copies), one assembly kernel.
    The written frame -- also the running mean over several frames with the mt19937 seed stream --
    equals the single renderer's bit for bit."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "volumerenderercl_amd", "vrhip_render")
    W, H = 200, 136        # not a multiple of the tile sizes
    base = [exe, "--synth", "shells", "96", "USHORT", "--size", str(W), str(H), "--rotate", "1", "1", "0", "30",
            "--frames", str(frames)]
    extra = [a for a in args if a == "--pathtrace"]
    outs = []
    for name, more in (("single", extra), ("ranks", args)):
        out = str(tmp_path / name)
        res = subprocess.run(base + more + ["--out", out], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr
        outs.append(np.fromfile(out + ".rgba.f32", dtype=np.float32).reshape(H, W, 4))
        if name == "ranks":
            assert '"ranks": %s' % args[1] in res.stdout
            want = "loopback" if "--loopback" in args else "RCCL send/recv (root included)" if "--force-gather" in args \
                else "RCCL send/recv"
            assert '"transport": "%s' % want in res.stdout, res.stdout
    assert np.isfinite(outs[0]).all() and outs[0].std() > 0
    np.testing.assert_array_equal(outs[0], outs[1])


def test_cpp_host_throughput_path_equals_frames_one_by_one(tmp_path):
    """The C++ host's throughput path (VolumeRenderCL::shareVolumes / renderFrames, TileGather::submitFrames /
    collectFrames; `vrhip_render --frames-per-launch K`): N independent frames -- jitter seeds = the first N outputs
    of the default-seeded std::mt19937, iteration 0 -- rendered in launch sets by one or two renderers over a shared
    volume, and by 1-4 tile ranks with batched, sparse, pipelined exchanges (device copies with --loopback; RCCL's
    grouped send / recv from the root to itself with --ranks 1 --force-gather), equal `--independent` (one
    runRaycastNoGL per frame) bit for bit -- and the oracle's frames of those seeds."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "volumerenderercl_amd", "vrhip_render")
    W, H, N, NF = 200, 136, 96, 7        # (the frame is not a multiple of the tile sizes; 7 frames: ragged sets)
    base = [exe, "--synth", "shells", str(N), "USHORT", "--size", str(W), str(H), "--rotate", "1", "1", "0", "30",
            "--frames", str(NF)]

    def run(name, more):
        out = str(tmp_path / name)
        res = subprocess.run(base + more + ["--out", out], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr
        frames = np.fromfile(out + ".frames.rgba.f32", dtype=np.float32).reshape(NF, H, W, 4)
        last = np.fromfile(out + ".rgba.f32", dtype=np.float32).reshape(H, W, 4)
        np.testing.assert_array_equal(last, frames[-1])
        return frames, res.stdout

    one_by_one, _ = run("single", ["--independent"])
    assert np.isfinite(one_by_one).all() and one_by_one.std() > 0 and np.abs(one_by_one[0] - one_by_one[1]).max() > 0
    # the oracle's frames of the same seeds
    vol = vro.synth_volume("shells", [N, N, N], vro.USHORT)
    tff = frontend.tff_from_stops()
    cam, rp, rc, pt = scenes.oracle_params((N, N, N), "rot30", {"background": (1.0, 1.0, 1.0)}, seed=0)   # (setBackground: alpha 0)
    mt = frontend.Mt19937()
    for f in range(NF):
        rp.seed, rp.iteration = mt(), 0
        ref, _, _ = vro.render_tile(vol, vro.USHORT, tff, cam, rp, rc, pt, W=W, H=H)
        assert np.abs(one_by_one[f].astype(np.float64) - ref).max() <= TOL, f
    for name, more in (("b3x2", ["--frames-per-launch", "3", "--frames-in-flight", "2"]),
                       ("b4x1", ["--frames-per-launch", "4", "--frames-in-flight", "1", "--round-budget", "5"]),
                       ("b16x3", ["--frames-per-launch", "16", "--frames-in-flight", "3"]),
                       ("r1", ["--ranks", "1", "--loopback", "--tile", "32", "--frames-per-launch", "3"]),
                       ("r2", ["--ranks", "2", "--loopback", "--tile", "16", "--frames-per-launch", "2"]),
                       ("r3", ["--ranks", "3", "--loopback", "--tile", "48", "--frames-per-launch", "7", "--root-share", "0.5"]),
                       ("r4", ["--ranks", "4", "--loopback", "--tile", "32", "--frames-per-launch", "1"]),
                       ("rccl", ["--ranks", "1", "--force-gather", "--tile", "32", "--frames-per-launch", "3"]),
                       ("dense2", ["--ranks", "2", "--loopback", "--tile", "32", "--independent"])):
        got, stdout = run(name, more)
        np.testing.assert_array_equal(got, one_by_one, err_msg=name)
        if name == "rccl":
            assert '"transport": "RCCL send/recv (root included)"' in stdout, stdout
        if name.startswith("r") and "--frames-per-launch" in more:
            import json
            line = json.loads(stdout.strip().splitlines()[-1])
            assert line["sent_bytes_per_frame"] > 0 or name == "r1"    # (one rank without --force-gather sends nothing)
            if name in ("r2", "r3", "r4"):      # the background tiles travelled as one pixel each
                assert line["sent_bytes_per_frame"] < line["dense_bytes_per_frame"], line


def test_cpp_host_bench_runs_the_benchmarks_schedule(tmp_path):
    """`vrhip_render --bench` on a 256^3 volume: two renderers over one shared volume, launch sets of 8 frames, round
    budget 48, HIP-event time per frame -- and the launch info says which kernels ran (the 12-wave instantiations:
    sets of >= 4 frames)."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "volumerenderercl_amd", "vrhip_render")
    out = str(tmp_path / "b")
    res = subprocess.run([exe, "--synth", "shells", "256", "UCHAR", "--size", "512", "512", "--rotate", "1", "1", "0", "30",
                          "--frames", "32", "--frames-per-launch", "8", "--bench", "--out", out],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["bench"] is True and line["renderers"] == 2 and line["launch_sets"] == 4 and line["frames_per_launch_set"] == 8
    assert line["round_budget"] == 48 and line["phase1_waves"] == 12 and line["phase2_waves"] == 12
    assert 0.0 < line["ms_per_frame"] < 50.0
    last = np.fromfile(out + ".rgba.f32", dtype=np.float32).reshape(512, 512, 4)
    assert np.isfinite(last).all() and last.std() > 0


def test_pack_tiles_and_message_positions(vr):
    """vrhip_pack_tiles / vrhip_message_positions (the C++ host's sparse gather message, packed and read on the GPU)
    against numpy: slot list, one pixel per slot, the whole tiles in slot order, the count, and the positions the
    root's assembly reads."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(3)
    for S, P in ((1, 256), (37, 1024), (1500, 256), (5000, 256)):
        tiles = np.zeros((S, P, 4), dtype=np.float32)
        whole = rng.random(S) < 0.4
        for s_ in range(S):
            tiles[s_] = rng.random(4).astype(np.float32)
            if whole[s_]:
                tiles[s_, rng.integers(0, P)] += 1.0          # one pixel differs (also: the last, the first)
        if S > 2:
            tiles[2] = tiles[2, 0]; tiles[2, P - 1, 3] = 7.0; whole[2] = True
            tiles[1] = tiles[1, 0]; tiles[1, 0, 0] = -0.0; tiles[1, 1:, 0] = 0.0; whole[1] = True   # (-0.0 != 0.0 bit for bit)
        spad = (S + 3) // 4 * 4
        t = torch.from_numpy(tiles).cuda()
        msg = torch.full((spad + 4 * S + 4 * S * P,), -1.0, dtype=torch.float32, device="cuda")
        scratch = torch.zeros(S, dtype=torch.int32, device="cuda")
        count = torch.zeros(1, dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        rc = vr.lib.vrhip_pack_tiles(vr.handle, C.c_void_p(st), C.c_void_p(t.data_ptr()), S, P,
                                     C.c_void_p(scratch.data_ptr()), C.c_void_p(msg.data_ptr()), C.c_void_p(count.data_ptr()))
        assert rc == 0
        torch.cuda.synchronize()
        c = int(count.item())
        assert c == int(whole.sum())
        m = msg.cpu().numpy()
        slots = m[:c].view(np.int32)
        np.testing.assert_array_equal(slots, np.nonzero(whole)[0])
        np.testing.assert_array_equal(m[spad:spad + 4 * S].reshape(S, 4).view(np.uint32), tiles[:, 0, :].view(np.uint32))
        np.testing.assert_array_equal(m[spad + 4 * S: spad + 4 * S + 4 * c * P].reshape(c, P, 4).view(np.uint32),
                                      tiles[whole].view(np.uint32))
        pos = torch.zeros(2 * S, dtype=torch.int32, device="cuda")
        ptrs = (C.c_void_p * 2)(msg.data_ptr(), msg.data_ptr())
        counts = (C.c_uint32 * 2)(c, 0)
        rc = vr.lib.vrhip_message_positions(vr.handle, C.c_void_p(st), ptrs, counts, 2, S, C.c_void_p(pos.data_ptr()))
        assert rc == 0
        torch.cuda.synchronize()
        want = np.full(S, -1, dtype=np.int32)
        want[np.nonzero(whole)[0]] = np.arange(c)
        got = pos.cpu().numpy()
        np.testing.assert_array_equal(got[:S], want)
        assert (got[S:] == -1).all()


def test_cpp_host_cli_pathtrace_matches_oracle(tmp_path):
    """Headless path tracing through the C++ host: 3 samples per pixel accumulated with the
    std::mt19937 seed stream of the reference's VolumeRenderCL member."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "volumerenderercl_amd", "vrhip_render")
    out = str(tmp_path / "pt")
    W, H, N = 64, 48, 32
    cmd = [exe, "--synth", "sphere", str(N), "FLOAT", "--size", str(W), str(H), "--rotate", "1", "1",
           "0", "30", "--pathtrace", "--extinction", "60", "--frames", "3", "--out", out]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    got = np.fromfile(out + ".rgba.f32", dtype=np.float32).reshape(H, W, 4)
    vol = vro.synth_volume("sphere", [N, N, N], vro.FLOAT)
    tff = frontend.tff_from_stops()
    cam = vro.CameraParams()
    cam.viewMat[:] = frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30.0))
    cam.bbox_bl[:] = [-1, -1, -1, 0]
    cam.bbox_tr[:] = [1, 1, 1, 0]
    rp = vro.RenderingParams()
    rp.backgroundColor[:] = [1, 1, 1, 0]
    rp.modelScale[:] = [1, 1, 1, 0]
    rp.illumType, rp.useLinear, rp.technique = 1, 1, 1
    rc = vro.RaycastParams()
    rc.samplingRate = 1.5
    _, brf, _ = vro.brick_layout([N, N, N])
    rc.brickRes[:] = brf + [0]
    mt = frontend.Mt19937()
    ref = None
    for it in range(3):
        rp.seed, rp.iteration = mt(), it
        ref, _, _ = vro.render_tile(vol, vro.FLOAT, tff, cam, rp, rc, pt=vro.PathtraceParams(60.0),
                                    W=W, H=H, in_accum=ref)
    assert np.abs(got - ref).max() <= TOL


def test_large_tf_and_odd_viewport(vr):
    """4096-entry transfer function (64 KiB of LDS for the table alone) on a viewport that is not
    a multiple of the patch size, FLOAT volume with a non-cubic shape."""
    vol = common.noise_volume((50, 38, 61), FLOAT, seed=17, smooth=True)
    table = frontend.tff_from_stops(n=4096)
    W, H = 203, 117
    _setup(vr, vol, FLOAT, table, common.views()["rot30"])
    got, ref, stats = _compare(vr, vol, FLOAT, table, W, H)
    assert stats["rays_hit"] > 0
    vr.setStatsEnabled(False)
    got2 = vr.runRaycastNoGL(W, H)      # production kernels on the same frame
    vr.setIteration(0)
    np.testing.assert_array_equal(got2, ref)


@pytest.mark.parametrize("fmt,res,factor", [(UCHAR, (130, 70, 66), 2), (USHORT, (200, 64, 40), 3),
                                            (FLOAT, (128, 48, 50), 2), (UCHAR, (260, 33, 47), 4)])
def test_downsample_matches_oracle(vr, fmt, res, factor):
    """downsampling kernel (volumeraycast.cl:966-994) with the host's size rule: bit-exact."""
    vol = common.noise_volume(res, fmt, seed=8, smooth=False)
    vr.loadVolumeArrays([vol], fmt)
    got = vr.downsampleVolume(0, factor)
    ref = vro.downsample(vol, fmt, factor)
    np.testing.assert_array_equal(got, ref)
    with pytest.raises((ValueError, RuntimeError)):
        vr.downsampleVolume(0, 1)                       # "Factor must be greater or equal 2."
    with pytest.raises((ValueError, RuntimeError)):
        vr.downsampleVolume(0, 64)                      # below the 64-voxel minimum


def test_cpp_host_downsampling_writes_dat_raw(tmp_path):
    """volumeDownsampling through the C++ host: <name>_<N>.raw/.dat next to the .dat, raw bytes
    equal to the oracle's, and the pair loads again (UCHAR: the reference writes the format as the
    enum's integer, which its own reader then infers from the file size)."""
    import os
    import subprocess
    from volumerenderercl_amd import datraw
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "volumerenderercl_amd", "vrhip_render")
    vol = common.noise_volume((128, 40, 36), UCHAR, seed=2, smooth=False)
    (tmp_path / "v.raw").write_bytes(vol.tobytes())
    (tmp_path / "v.dat").write_text("ObjectFileName: v.raw\nResolution: 128 40 36\nFormat: UCHAR\n"
                                    "SliceThickness: 1 1 2\n")
    res = subprocess.run([exe, "--dat", str(tmp_path / "v.dat"), "--downsample", "2"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    base = str(tmp_path / "v_64")
    raw = np.fromfile(base + ".raw", dtype=np.uint8)
    ref = vro.downsample(vol, vro.UCHAR, 2)
    np.testing.assert_array_equal(raw, ref.reshape(-1))
    text = open(base + ".dat").read()
    assert "Resolution: \t\t64 20 18" in text and "SliceThickness: \t1 1 2" in text
    assert "Format: \t\t\t0" in text
    rd = datraw.DatRawReader()
    rd.read_files(datraw.Properties(base + ".dat"))
    np.testing.assert_array_equal(rd.data()[0].reshape(-1), ref.reshape(-1))
    # the Python twin of the class writes the same two files, byte for byte
    cpp_dat, cpp_raw = open(base + ".dat", "rb").read(), open(base + ".raw", "rb").read()
    os.remove(base + ".dat")
    os.remove(base + ".raw")
    r = VolumeRenderCL()
    r.initialize()
    try:
        r.loadVolumeData(datraw.Properties(str(tmp_path / "v.dat")))
        assert r.volumeDownsampling(0, 2) == base
        with pytest.raises(ValueError):
            r.volumeDownsampling(0, 1)
        assert r.getPlatformNames() == ["AMD HIP (ROCm)"]
        assert r.getDeviceNames(0, "GPU") == [r.getCurrentDeviceName()] and r.getDeviceNames(0, "CPU") == []
    finally:
        r.close()
    assert open(base + ".dat", "rb").read() == cpp_dat and open(base + ".raw", "rb").read() == cpp_raw


def test_cpp_host_cli_reads_gui_state_and_tff(tmp_path):
    """The headless host consumes the reference GUI's own saved files: JSON camera
    state and a gradient-stop .tff."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "volumerenderercl_amd", "vrhip_render")
    q = frontend.quat_from_axis_angle((0.2, 1, 0.1), 40.0)
    stops = [(0.0, (0, 0, 0, 0)), (0.3, (220, 60, 20, 10)), (0.8, (20, 90, 200, 120)), (1.0, (255, 255, 255, 255))]
    frontend.write_cam_state(str(tmp_path / "s.json"), q, (0.1, -0.2, 2.5), rayStepSize=1.0,
                             useAerial=True, showContours=True)
    frontend.write_tff_stops(str(tmp_path / "t.tff"), stops)
    W, H, N = 72, 56, 40
    out = str(tmp_path / "f")
    cmd = [exe, "--synth", "sphere", str(N), "UCHAR", "--size", str(W), str(H), "--state",
           str(tmp_path / "s.json"), "--tf-stops", str(tmp_path / "t.tff"), "--seed", str(SEED), "--out", out]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    got = np.fromfile(out + ".rgba.f32", dtype=np.float32).reshape(H, W, 4)
    st = frontend.read_cam_state(str(tmp_path / "s.json"))
    vol = vro.synth_volume("sphere", [N, N, N], vro.UCHAR)
    tff = frontend.tff_from_stops(frontend.read_tff_stops(str(tmp_path / "t.tff")))
    cam = vro.CameraParams()
    cam.viewMat[:] = frontend.view_matrix(st["rotation"], st["translation"])
    cam.bbox_bl[:] = [-1, -1, -1, 0]
    cam.bbox_tr[:] = [1, 1, 1, 0]
    rp = vro.RenderingParams()
    rp.backgroundColor[:] = [1, 1, 1, 0]
    rp.modelScale[:] = [1, 1, 1, 0]
    rp.illumType, rp.useLinear, rp.seed = 1, 1, SEED
    rc = vro.RaycastParams()
    rc.samplingRate = st["rayStepSize"]
    rc.contours, rc.aerial = 1, 1
    _, brf, _ = vro.brick_layout([N, N, N])
    rc.brickRes[:] = brf + [0]
    ref, _, _ = vro.render_tile(vol, vro.UCHAR, tff, cam, rp, rc, W=W, H=H)
    assert np.abs(got - ref).max() <= TOL


def test_ambient_occlusion_changes_terminated_rays(vr):
    """AO darkens exactly the rays that end by early ray termination (alpha channel >= 0.98) and
    leaves the others untouched."""
    vol = common.noise_volume((48, 48, 48), UCHAR, seed=7, smooth=False)
    table = common.tffs()["opaque"]
    _setup(vr, vol, UCHAR, table, common.views()["rot30"])
    vr.setStatsEnabled(False)
    plain = vr.runRaycastNoGL(96, 80)
    vr.setIteration(0)
    vr.setAmbientOcclusion(True)
    ao = vr.runRaycastNoGL(96, 80)
    vr.setIteration(0)
    vr.setAmbientOcclusion(False)
    ert = plain[..., 3] >= 0.98
    assert ert.sum() > 100
    np.testing.assert_array_equal(ao[~ert], plain[~ert])
    np.testing.assert_array_equal(ao[..., 3], plain[..., 3])
    changed = np.any(ao[ert][:, :3] != plain[ert][:, :3], axis=1)
    assert changed.mean() > 0.1 and np.all(ao[ert][:, :3] <= plain[ert][:, :3] + 1e-7)


def _sparse_volume(res=(48, 48, 48)):
    vol = np.zeros(res[::-1], np.uint8)
    c = common.noise_volume((16, 16, 16), UCHAR, seed=3, smooth=False)
    vol[14:30, 20:36, 10:26] = np.maximum(c, 60)
    return vol


@pytest.mark.parametrize("kw", [{}, {"show_ess": True}, {"ess": False, "illum": 0}])
def test_image_order_ess_sequence_matches_oracle(vr, kw):
    """imgEss (:659-670, :912-925): four frames from the reference's initial hit images; every
    frame and both hit images equal the oracle's, with the ping-pong swap in between."""
    vol = _sparse_volume()
    tff = frontend.tff_from_stops()
    W, H = 136, 104
    _setup(vr, vol, UCHAR, tff, common.views()["rot30"], img_ess=True, **kw)
    vr.updateOutputImg(W, H)
    hin, hout = vro.hit_image_init(W, H)
    g_in, g_out = vr.getImageEss(W, H)
    assert np.array_equal(g_in, hin) and np.array_equal(g_out, hout)
    cam, rp, rc, pt = common.to_oracle_params(*vr.params())
    skipped = []
    for frame in range(4):
        got = vr.runRaycastNoGL(W, H)
        vr.setIteration(0)
        ref, _, _ = vro.render_tile(vol, UCHAR, tff, cam, rp, rc, pt, use_ess=kw.get("ess", True),
                                    W=W, H=H, hit_in=hin, hit_out=hout)
        assert np.abs(got - ref).max() <= TOL
        hin, hout = hout, hin
        g_in, g_out = vr.getImageEss(W, H)
        assert np.array_equal(g_in, hin), "frame %d: hit image differs" % frame
        assert np.array_equal(g_out, hout)
        skipped.append(int((hin == 0).sum()))
    assert 0 < skipped[-1] and hin.sum() > 0        # something is skipped, something is hit
    # updateOutputImg starts the hit images over (volumerendercl.cpp:482-488)
    vr.updateOutputImg(W, H)
    g_in, _ = vr.getImageEss(W, H)
    assert np.array_equal(g_in, vro.hit_image_init(W, H)[0])
    vr.setImgEss(False)


def test_image_order_ess_tiles(vr):
    """A tile render reads the whole hit image and updates the texels of its own groups only."""
    import torch
    vol = _sparse_volume()
    tff = frontend.tff_from_stops()
    W, H, TW, TH = 136, 104, 32, 48
    _setup(vr, vol, UCHAR, tff, common.views()["rot30"], img_ess=True)
    vr.updateOutputImg(W, H)
    full = vr.runRaycastNoGL(W, H)
    vr.setIteration(0)
    hit_full, _ = vr.getImageEss(W, H)              # after the swap: what the frame wrote
    vr.updateOutputImg(W, H)
    tiles_x, tiles_y = (W + TW - 1) // TW, (H + TH - 1) // TH
    ids = np.array([t for t in range(tiles_x * tiles_y) if t % 2 == 0], dtype=np.uint32)
    out = torch.zeros((len(ids), TH, TW, 4), dtype=torch.float32, device="cuda")
    vr.render_tiles(W, H, TW, TH, ids, out.data_ptr())
    torch.cuda.synchronize()
    hit_tiles, _ = vr.getImageEss(W, H)
    o = out.cpu().numpy()
    own = np.zeros_like(hit_full, dtype=bool)
    for k, t in enumerate(ids):
        tx, ty = int(t) % tiles_x, int(t) // tiles_x
        x0, y0 = tx * TW, ty * TH
        w, h = min(TW, W - x0), min(TH, H - y0)
        np.testing.assert_array_equal(o[k, :h, :w], full[y0:y0 + h, x0:x0 + w])
        own[y0 // 8:(y0 + h + 7) // 8, x0 // 8:(x0 + w + 7) // 8] = True
    assert np.array_equal(hit_tiles[own], hit_full[own])
    assert not hit_tiles[~own].any()                # untouched texels of the zeroed output image
    vr.setImgEss(False)


def test_cpp_host_cli_image_order_ess(tmp_path):
    """vrhip_render --img-ess --show-ess --frames 3: the C++ host carries the hit images (and
    the accumulate image) from frame to frame like runRaycast does."""
    import os
    import subprocess
    from volumerenderercl_amd import datraw
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "volumerenderercl_amd", "vrhip_render")
    dat = os.path.join(root, "tests", "golden", "loader", "c1.dat")
    out = str(tmp_path / "frame")
    W, H = 104, 72
    cmd = [exe, "--dat", dat, "--size", str(W), str(H), "--translate", "0.3", "-0.2", "3.5",
           "--seed", str(SEED), "--img-ess", "--show-ess", "--frames", "3", "--out", out]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    got = np.fromfile(out + ".rgba.f32", dtype=np.float32).reshape(H, W, 4)
    rd = datraw.DatRawReader()
    rd.read_files(datraw.Properties(dat))
    p = rd.properties()
    vol = rd.data()[0].reshape(p.volume_res[2], p.volume_res[1], p.volume_res[0])
    tff = frontend.tff_from_stops()
    cam = vro.CameraParams()
    cam.viewMat[:] = frontend.view_matrix(translation=(0.3, -0.2, 3.5))
    cam.bbox_bl[:] = [-1, -1, -1, 0]
    cam.bbox_tr[:] = [1, 1, 1, 0]
    rp = vro.RenderingParams()
    rp.backgroundColor[:] = [1, 1, 1, 0]
    rp.modelScale[:] = vro.calc_scaling(p.volume_res[:3], p.slice_thickness) + [0]
    rp.illumType, rp.useLinear, rp.seed = 1, 1, SEED
    rp.imgEss, rp.showEss = 1, 1
    rc = vro.RaycastParams()
    rc.samplingRate = 1.5
    _, brf, _ = vro.brick_layout(p.volume_res[:3])
    rc.brickRes[:] = brf + [0]
    hin, hout = vro.hit_image_init(W, H)
    acc = None
    for it in range(3):
        rp.iteration = it
        acc, _, _ = vro.render_tile(vol, vro.UCHAR, tff, cam, rp, rc, W=W, H=H, in_accum=acc,
                                    hit_in=hin, hit_out=hout)
        hin, hout = hout, hin
    assert (hin == 0).sum() > 0
    assert np.abs(got - acc).max() <= TOL


def _env_map(w=96, h=48, seed=11):
    rng = np.random.default_rng(seed)
    env = rng.random((h, w, 4), dtype=np.float32) * 1.5
    env[..., 3] = 0.0                  # what the .hdr loader produces
    return env


@pytest.mark.parametrize("kw", [
    {"view": "rot30"},
    {"view": "inside", "ess": False, "illum": 0},
    {"view": "close", "technique": 1, "ext": 40.0},
    {"view": "rot30", "img_ess": True, "show_ess": True},
    {"view": "default", "ortho": True, "contours": True},
])
def test_environment_map_matches_oracle(vr, kw):
    """createEnvironmentMap (:506-510, :655-656): the map is sampled along every ray direction
    (atan2/acos coordinates, linear filter, clamp to edge) and stands in for the background."""
    kw = dict(kw)
    view = kw.pop("view")
    vol = common.noise_volume((40, 40, 40), FLOAT, seed=4, smooth=True)
    tff = common.tffs()["default"]
    W, H = 88, 64
    env = _env_map()
    _setup(vr, vol, FLOAT, tff, common.views()[view], **kw)
    vr.setEnvironmentMap(env)
    try:
        vr.updateOutputImg(W, H)
        vr.setStatsEnabled(False)
        got = vr.runRaycastNoGL(W, H)
        vr.setIteration(0)
        extra = {}
        if kw.get("img_ess"):
            extra["hit_in"], extra["hit_out"] = vro.hit_image_init(W, H)
        cam, rp, rc, pt = common.to_oracle_params(*vr.params())
        ref, _, _ = vro.render_tile(vol, FLOAT, tff, cam, rp, rc, pt, use_ess=kw.get("ess", True),
                                    W=W, H=H, env=env, **extra)
        assert np.abs(got - ref).max() <= TOL
        # the map is really in the picture: without it the frame differs
        vr.setEnvironmentMap(None)
        vr.updateOutputImg(W, H)
        plain = vr.runRaycastNoGL(W, H)
        vr.setIteration(0)
        assert np.abs(plain - got).max() > 1e-3
    finally:
        vr.setEnvironmentMap(None)
        vr.setImgEss(False)


def test_cpp_host_cli_environment_map(tmp_path):
    """vrhip_render --env FILE.hdr: the C++ host decodes the Radiance file (golden fixture) and
    the frame equals the oracle's with the same texels."""
    import os
    import subprocess
    from volumerenderercl_amd import datraw
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "volumerenderercl_amd", "vrhip_render")
    dat = os.path.join(root, "tests", "golden", "loader", "c1.dat")
    hdr = os.path.join(root, "tests", "golden", "hdr", "rle.hdr")
    out = str(tmp_path / "frame")
    W, H = 72, 40
    cmd = [exe, "--dat", dat, "--size", str(W), str(H), "--rotate", "1", "1", "0", "30",
           "--seed", str(SEED), "--env", hdr, "--out", out]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    got = np.fromfile(out + ".rgba.f32", dtype=np.float32).reshape(H, W, 4)
    rd = datraw.DatRawReader()
    rd.read_files(datraw.Properties(dat))
    p = rd.properties()
    vol = rd.data()[0].reshape(p.volume_res[2], p.volume_res[1], p.volume_res[0])
    tff = frontend.tff_from_stops()
    cam = vro.CameraParams()
    cam.viewMat[:] = frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30.0))
    cam.bbox_bl[:] = [-1, -1, -1, 0]
    cam.bbox_tr[:] = [1, 1, 1, 0]
    rp = vro.RenderingParams()
    rp.backgroundColor[:] = [1, 1, 1, 0]
    rp.modelScale[:] = vro.calc_scaling(p.volume_res[:3], p.slice_thickness) + [0]
    rp.illumType, rp.useLinear, rp.seed = 1, 1, SEED
    rc = vro.RaycastParams()
    rc.samplingRate = 1.5
    _, brf, _ = vro.brick_layout(p.volume_res[:3])
    rc.brickRes[:] = brf + [0]
    env = np.fromfile(hdr + ".f32", dtype="<f4").reshape(6, 24, 4)
    ref, _, _ = vro.render_tile(vol, vro.UCHAR, tff, cam, rp, rc, W=W, H=H, env=env)
    assert np.abs(got - ref).max() <= TOL
    bad = subprocess.run(cmd[:-4] + ["--env", hdr + ".missing", "--out", out], capture_output=True,
                         text=True, timeout=300)
    assert bad.returncode != 0 and "Error loading environment map file." in bad.stderr




@pytest.mark.parametrize("fmt,nch,res,view,kw", MC_CASES)
def test_multichannel_volume_matches_oracle(vr, fmt, nch, res, view, kw):
    vol = scenes.multichannel_volume(fmt, nch, res)
    tff = common.tffs()["default"]
    W, H = 80, 64
    _setup(vr, vol, fmt, tff, common.views()[view], **kw)
    got, ref, stats = _compare(vr, vol, fmt, tff, W, H, ess=kw.get("ess", True))
    _same_params(vr, res, view, kw)
    assert stats["samples_taken"] > 0
    # the extra channels are really used: dropping them changes the frame
    _setup(vr, np.ascontiguousarray(vol[..., 0]), fmt, tff, common.views()[view], **kw)
    vr.setStatsEnabled(False)
    single = vr.runRaycastNoGL(W, H)
    vr.setIteration(0)
    if kw.get("illum", 1) != 4:
        assert np.abs(single - got).max() > 1e-3
    else:
        assert np.array_equal(single, got)      # illumType 4 only ever reads .x


def test_cpp_host_cli_rgba_dat(tmp_path):
    """A .dat with `ChannelOrder: RGBA` through DatRawReader + VolumeRenderCL + the CLI: the
    interleaved raw file is uploaded as a CL_RGBA volume (volumerendercl.cpp:697-705)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "volumerenderercl_amd", "vrhip_render")
    res = (20, 24, 16)
    planes = [common.noise_volume(res, UCHAR, seed=30 + c, smooth=False) for c in range(4)]
    vol = np.stack(planes, axis=-1)
    vol[..., 3] //= 6
    vol.tofile(str(tmp_path / "rgba.raw"))
    (tmp_path / "rgba.dat").write_text("ObjectFileName: rgba.raw\nResolution: 20 24 16\n"
                                       "Format: UCHAR\nChannelOrder: RGBA\n")
    out = str(tmp_path / "frame")
    W, H = 64, 48
    cmd = [exe, "--dat", str(tmp_path / "rgba.dat"), "--size", str(W), str(H), "--rotate", "1", "1",
           "0", "30", "--seed", str(SEED), "--out", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out + ".rgba.f32", dtype=np.float32).reshape(H, W, 4)
    tff = frontend.tff_from_stops()
    cam = vro.CameraParams()
    cam.viewMat[:] = frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30.0))
    cam.bbox_bl[:] = [-1, -1, -1, 0]
    cam.bbox_tr[:] = [1, 1, 1, 0]
    rp = vro.RenderingParams()
    rp.backgroundColor[:] = [1, 1, 1, 0]
    rp.modelScale[:] = vro.calc_scaling(list(res), [1.0, 1.0, 1.0]) + [0]
    rp.illumType, rp.useLinear, rp.seed = 1, 1, SEED
    rc = vro.RaycastParams()
    rc.samplingRate = 1.5
    _, brf, _ = vro.brick_layout(list(res))
    rc.brickRes[:] = brf + [0]
    ref, _, _ = vro.render_tile(vol, vro.UCHAR, tff, cam, rp, rc, W=W, H=H)
    assert np.abs(got - ref).max() <= TOL
    # the orders the reference uploads but never renders are refused
    (tmp_path / "bgra.dat").write_text("ObjectFileName: rgba.raw\nResolution: 20 24 16\n"
                                       "Format: UCHAR\nChannelOrder: BGRA\n")
    bad = subprocess.run([exe, "--dat", str(tmp_path / "bgra.dat"), "--out", out],
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "not supported" in bad.stderr




@pytest.mark.parametrize("fmt,res,size,view,tff,kw", FP_CASES)
def test_footprint_volume_frames_match_oracle(vr, monkeypatch, fmt, res, size, view, tff, kw):
    """The footprint volume (one load per trilinear fetch) only changes where the
    eight voxels come from: the un-instrumented frame equals the oracle's, and the frame of a
    renderer with the footprint volume disabled, bit for bit."""
    vol = common.noise_volume(res, fmt, seed=12, smooth=False)
    table = common.tffs()[tff]
    W, H = size
    _setup(vr, vol, fmt, table, common.views()[view], **kw)
    vr.setStatsEnabled(False)
    got = vr.runRaycastNoGL(W, H)
    vr.setIteration(0)
    ref, _, _ = common.oracle_frame(vr, vol, fmt, table, W, H, use_ess=kw.get("ess", True))
    assert np.abs(got - ref).max() <= TOL
    monkeypatch.setenv("VRHIP_NO_FOOTPRINT", "1")
    r2 = VolumeRenderCL()
    r2.initialize()
    try:
        _setup(r2, vol, fmt, table, common.views()[view], **kw)
        r2.setStatsEnabled(False)
        plain = r2.runRaycastNoGL(W, H)
    finally:
        r2.close()
    assert np.array_equal(got, plain)
    # another viewport (other rays, other fetches): same agreement
    small = vr.runRaycastNoGL(24, 16)
    vr.setIteration(0)
    ref_small, _, _ = common.oracle_frame(vr, vol, fmt, table, 24, 16, use_ess=kw.get("ess", True))
    assert np.abs(small - ref_small).max() <= TOL


def test_time_series_steps_match_oracle(vr):
    """One volume per time step (volumerendercl.cpp:736-752), setTimestep (:1167-1174): bricks,
    skip bitmap, cell grid and footprint volume follow the selected step; an out-of-range step is
    ignored like in the reference."""
    steps = [common.noise_volume((40, 44, 36), UCHAR, seed=40 + t, smooth=False) for t in range(3)]
    tff = common.tffs()["default"]
    W, H = 72, 56
    vr.loadVolumeArrays(steps, UCHAR)
    vr.setTransferFunction(tff)
    vr.setSeed(SEED)
    vr.setIllumination(1)
    vr.setObjEss(True)
    vr.setTechnique(0)
    vr.setAmbientOcclusion(False)
    vr.setShowESS(False)
    vr.setImgEss(False)
    vr.updateView(common.views()["rot30"])
    assert vr.getResolution()[3] == 3
    frames = {}
    for t in (0, 2, 1, 2):
        vr.setTimestep(t)
        for stats in (True, False):      # instrumented (plain layout) and default kernels
            vr.setStatsEnabled(stats)
            got = vr.runRaycastNoGL(W, H)
            vr.setIteration(0)
            ref, _, _ = common.oracle_frame(vr, steps[t], UCHAR, tff, W, H)
            assert np.abs(got - ref).max() <= TOL, "time step %d" % t
        frames.setdefault(t, got)
        assert np.array_equal(frames[t], got)
    assert np.abs(frames[0] - frames[1]).max() > 1e-3
    vr.setTimestep(7)                    # ignored
    vr.setStatsEnabled(False)
    again = vr.runRaycastNoGL(W, H)
    vr.setIteration(0)
    assert np.array_equal(again, frames[2])
    # a step that stays is rendered from its footprint volume from the third frame on (a series that
    # moves on with every frame never builds one): the same pixels either way
    for _ in range(4):
        again = vr.runRaycastNoGL(W, H)
        vr.setIteration(0)
        assert np.array_equal(again, frames[2])


@pytest.mark.parametrize("kind,fmt,ess,illum", [("shells", UCHAR, True, 1), ("sphere", USHORT, True, 1),
                                                ("sphere", UCHAR, False, 0)])
def test_midsize_synthetic_volume_matches_oracle(vr, kind, fmt, ess, illum):
    """256^3 synthetic fields (SURVEY 8d formulas, generated on the GPU and by the oracle): brick
    edge 4, many micro-brick rows, the cell grid at 32^3 cells, the footprint volume with 65^3
    micro-bricks of entries -- instrumented and default kernels against the oracle.  The third case is
    BASELINE config 1's exact conjunction: 256^3 UCHAR sphere, no shading, no ESS."""
    res = (256, 256, 256)
    vol = vro.synth_volume(kind, list(res), fmt)
    tff = common.tffs()["default"]
    W, H = 176, 144
    vr.synthVolume(kind, res, fmt)
    np.testing.assert_array_equal(vr.downloadVolume(0), vol)
    vr.setTransferFunction(tff)
    vr.setSeed(SEED)
    for name, val in (("setIllumination", illum), ("setLinearInterpolation", True), ("setCamOrtho", False),
                      ("setUseGradient", False), ("setContours", False), ("setAerial", False),
                      ("setObjEss", ess), ("setAmbientOcclusion", False), ("setTechnique", 0),
                      ("setShowESS", False), ("setImgEss", False)):
        getattr(vr, name)(val)
    vr.updateSamplingRate(1.5)
    vr.setBBox(-1, -1, -1, 1, 1, 1)
    vr.params()[1].backgroundColor[:] = [1.0, 1.0, 1.0, 1.0]
    vr.updateView(common.views()["rot30"])
    vr.setIteration(0)
    got, ref, stats = _compare(vr, vol, fmt, tff, W, H, ess=ess)
    if ess:
        assert stats["bricks_skipped"] > 0 and stats["samples_shaded"] > 0
    else:
        assert stats["bricks_visited"] == 0 and stats["samples_shaded"] == 0 and stats["samples_taken"] > 0
    vr.setStatsEnabled(False)
    plain = vr.runRaycastNoGL(W, H)
    li = vr.lastLaunchInfo()
    assert li["instrumented"] == 0 and li["extras"] == 0 and li["footprint"] == 1 and li["ray_list"] == (1 if ess else 0)
    vr.setIteration(0)
    assert np.array_equal(plain, got)
    vr.setObjEss(True)
    vr.setIllumination(1)


def _timed_schedule_batches_match_oracle(r, vol, fmt, tff, ref_bricks, W, H, n_seeds=6):
    """The benchmark's throughput schedule -- two renderers over one shared volume, each rendering its frames as
    ONE launch set (vrhip_render_batch) with a phase-1 round budget of 48 -- at a volume size where that means the
    12-wave marching kernels (three waves per SIMD) WITH the empty-run lookahead: the instantiation bench.py's
    timed region runs, and the one no single-frame test reaches (a set of >= 4 frames, ESS bricks >= 16 voxels).
    Every frame of both sets against the oracle's frame of its jitter seed."""
    import torch

    dev = torch.device("cuda")
    mt = frontend.Mt19937(77)
    seeds = [mt() for _ in range(2 * n_seeds)]
    cam, rp, rc, pt = common.to_oracle_params(*r.params())
    refs = []
    for sd in seeds:
        rp.seed, rp.iteration = sd, 0
        refs.append(vro.render_tile(vol, fmt, tff, cam, rp, rc, pt, W=W, H=H, bricks=ref_bricks)[0])
    r.setStatsEnabled(False)
    r.setRoundBudget(48)
    twin = r.shareVolumes()
    try:
        outs = []
        for k, x in enumerate((r, twin)):
            out = torch.zeros((n_seeds, H, W, 4), dtype=torch.float32, device=dev)
            x.setFrameTiming(False)
            x.render_batch(W, H, seeds[k * n_seeds:(k + 1) * n_seeds], out.data_ptr())   # both sets in flight
            outs.append(out)
        torch.cuda.synchronize()
        for k, x in enumerate((r, twin)):
            li = x.lastLaunchInfo()
            assert li["frames"] == n_seeds and li["round_budget"] == 48 and li["prepass"] == 1 and li["ray_list"] == 1, li
            assert li["phase1_waves"] == 12 and li["phase2_waves"] == 12 and li["empty_skip"] == 1, li
            assert li["footprint"] == 1 and li["instrumented"] == 0 and li["extras"] == 0 and li["skip_in_lds"] == 1, li
            got = outs[k].cpu().numpy()
            for i in range(n_seeds):
                d = np.abs(got[i].astype(np.float64) - refs[k * n_seeds + i])
                assert d.max() <= TOL, "renderer %d, frame %d of its launch set: %g at %s" % (
                    k, i, d.max(), np.unravel_index(d.argmax(), d.shape))
        # the same kernels, second sets on both renderers (control blocks alternate, cost map primed)
        for k, x in enumerate((twin, r)):
            x.render_batch(W, H, seeds[k * n_seeds:(k + 1) * n_seeds], outs[k].data_ptr())
        torch.cuda.synchronize()
        for k in range(2):
            got = outs[k].cpu().numpy()
            for i in range(n_seeds):
                assert np.abs(got[i].astype(np.float64) - refs[k * n_seeds + i]).max() <= TOL, (k, i)
    finally:
        twin.close()
        r.setFrameTiming(True)
        r.setRoundBudget(10)


def test_headline_size_volume_matches_oracle():
    """The benchmark's own size: 2048^3 UCHAR "shells" (2^33 voxels: 64-bit element offsets, 32-bit
    brick-slice stride at its limit, ESS brick edge 32, 512^3 micro-bricks, a 256^3 cell grid).
    Field generated in HBM, downloaded, and a small frame -- instrumented and production kernels --
    compared with the oracle on the downloaded voxels: image, the six work counters, the ESS
    bricks and the touched-micro-brick bitmap."""
    N = 2048
    res = (N, N, N)
    tff = common.tffs()["default"]
    W, H = 96, 80
    r = VolumeRenderCL()
    r.initialize()
    try:
        r.synthVolume("shells", res, UCHAR)
        vol = r.downloadVolume(0)
        assert vol.shape == (N, N, N)
        # the field itself, on three slices (SURVEY 8d formula, the oracle's restatement of it)
        c = 2.0 * (np.arange(N) + 0.5) / N - 1.0
        for z in (0, 777, N - 1):
            rr = np.sqrt(c[None, :] ** 2 + c[:, None] ** 2 + c[z] ** 2)
            dv = np.maximum(1.0 - rr / 0.9, 0.0) * (0.5 + 0.5 * np.cos(24.0 * np.pi * rr))
            dv[dv < 0.35] = 0.0
            want = np.floor(255.0 * dv + 0.5).astype(np.uint8)
            bad = int((want != vol[z]).sum())
            # (cos() of the device's libm against numpy's: a value within rounding of a .5 may differ)
            assert bad <= 8 and np.abs(want.astype(int) - vol[z].astype(int)).max() <= 1, (z, bad)
        r.setTransferFunction(tff)
        ref_bricks = vro.generate_bricks(vol, UCHAR)
        assert ref_bricks.shape == (64, 64, 64, 2)
        np.testing.assert_array_equal(r.downloadBricks(0), ref_bricks)
        r.setSeed(SEED)
        r.updateView(common.views()["rot30"])
        r.setIteration(0)
        cam, rp, rc, pt = common.to_oracle_params(*r.params())
        rp.seed, rp.iteration = SEED, 0
        ref, rstats, ref_bm = vro.render_tile(vol, UCHAR, tff, cam, rp, rc, pt, W=W, H=H,
                                               bricks=ref_bricks, want_touched=True)
        assert rstats["bricks_skipped"] > 0 and rstats["samples_shaded"] > 0
        for stats in (True, False):
            r.setStatsEnabled(stats)
            r.setIteration(0)
            got = r.runRaycastNoGL(W, H)
            assert np.abs(got.astype(np.float64) - ref).max() <= TOL, "stats=%s" % stats
            if stats:
                assert r.getStats() == rstats
        r.setIteration(0)
        n, bm = r.countTouched(W, H, want_bitmap=True)
        np.testing.assert_array_equal(bm, ref_bm)
        assert n == int(np.unpackbits(ref_bm).sum()) and n > 0
        r.setSeed(SEED)
        r.setIteration(0)
        _timed_schedule_batches_match_oracle(r, vol, UCHAR, tff, ref_bricks, 256, 192)
        _config4_tile_split_at_size(r, vol, tff, ref_bricks)
    finally:
        r.close()


def _config4_tile_split_at_size(r, vol, tff, ref_bricks):
    """BASELINE config 4 as far as one GPU goes: the 2048^2 frame of the 2048^3 volume cut into 64 x 64
    tiles for EIGHT ranks (dealt by distance from the centre), the eight tile shares rendered one after the other on
    this GPU through eight TileDrivers -- instrumented kernels (summed work counters) and production
    kernels -- gathered by a stand-in collective that hands rank 0 the peers' blocks, assembled by rank
    0's driver, and compared with the oracle's frame of the whole viewport."""
    import torch
    from volumerenderercl_amd import tiles

    V, T, WORLD = 2048, 64, 8
    dev = torch.device("cuda")
    r.set_stream(torch.cuda.current_stream().cuda_stream)

    class SharedGather:          # one process plays all ranks: peers deposit, the root collects
        def __init__(self):
            self.blocks = {}

        class _Done:
            def wait(self):
                pass

        def for_rank(self, rank):
            outer = self

            class _D:
                def gather(self, tensor, gather_list, dst=0, async_op=False):
                    if rank != 0:
                        outer.blocks[rank] = tensor.clone()
                    else:
                        gather_list[0].copy_(tensor)
                        for k, t in outer.blocks.items():
                            gather_list[k].copy_(t)
                        outer.blocks = {}
                    return outer._Done()
            return _D()

    hub = SharedGather()
    splits = [tiles.TileSplit(V, V, T, T, WORLD, k) for k in range(WORLD)]
    assert sum(len(s_.my_tiles) for s_ in splits) == (V // T) ** 2 and splits[0].cap == 128
    drivers = [tiles.TileDriver(r, splits[k], dev, dist=hub.for_rank(k)) for k in range(WORLD)]
    r.setSeed(SEED)
    r.setIteration(0)
    cam, rp, rc, pt = common.to_oracle_params(*r.params())
    rp.seed, rp.iteration = SEED, 0
    ref, rstats, _ = vro.render_tile(vol, UCHAR, tff, cam, rp, rc, pt, W=V, H=V, bricks=ref_bricks)
    assert rstats["rays_hit"] > 3000000 and rstats["samples_shaded"] > 0
    frame = torch.zeros((V, V, 4), dtype=torch.float32, device=dev)
    try:
        for stats in (True, False):
            r.setStatsEnabled(stats)
            total = dict.fromkeys(rstats, 0)
            frame.zero_()
            for k in list(range(1, WORLD)) + [0]:          # the peers first, the root last
                r.setSeed(SEED)
                r.setIteration(0)
                drivers[k].submit()
                if stats:
                    for name, v in r.getStats().items():
                        total[name] += v
                if k:
                    drivers[k].pending.pop(0)              # (a peer has nothing to collect)
            drivers[0].collect(frame)
            torch.cuda.synchronize()
            got = frame.cpu().numpy()
            d = np.abs(got.astype(np.float64) - ref)
            assert d.max() <= TOL, "config 4 tile split, stats=%s: %g at %s" % (
                stats, d.max(), np.unravel_index(d.argmax(), d.shape))
            if stats:
                assert total == rstats
    finally:
        r.setStatsEnabled(False)
        r.set_stream(None, use_own=True)


def test_config3_size_volume_matches_oracle():
    """BASELINE config 3's size: 1024^3 USHORT "shells" -- ESS bricks of 16 voxels, the smallest for which
    the ray caster steps over empty cells by default (four cells of 4 voxels per brick edge), 128-byte
    micro-bricks, a 17 GB footprint volume.  Field generated in HBM, downloaded; two seeds of a small
    frame through the instrumented and the production kernels against the oracle: image, work counters,
    ESS bricks."""
    N = 1024
    tff = common.tffs()["default"]
    W, H = 128, 96
    r = VolumeRenderCL()
    r.initialize()
    try:
        r.synthVolume("shells", (N, N, N), USHORT)
        vol = r.downloadVolume(0)
        assert vol.shape == (N, N, N) and vol.dtype == np.uint16
        r.setTransferFunction(tff)
        ref_bricks = vro.generate_bricks(vol, USHORT)
        assert ref_bricks.shape == (64, 64, 64, 2)
        np.testing.assert_array_equal(r.downloadBricks(0), ref_bricks)
        r.updateView(common.views()["rot30"])
        for seed in (SEED, 581869302):
            r.setSeed(seed)
            r.setIteration(0)
            cam, rp, rc, pt = common.to_oracle_params(*r.params())
            rp.seed, rp.iteration = seed, 0
            ref, rstats, _ = vro.render_tile(vol, USHORT, tff, cam, rp, rc, pt, W=W, H=H, bricks=ref_bricks)
            assert rstats["bricks_skipped"] > 0 and rstats["samples_shaded"] > 0
            for stats in (True, False):
                r.setStatsEnabled(stats)
                r.setIteration(0)
                got = r.runRaycastNoGL(W, H)
                assert np.abs(got.astype(np.float64) - ref).max() <= TOL, "seed=%d stats=%s" % (seed, stats)
                if stats:
                    assert r.getStats() == rstats
        r.setIteration(0)
        _timed_schedule_batches_match_oracle(r, vol, USHORT, tff, ref_bricks, 224, 160)
    finally:
        r.close()


def test_midsize_volume_with_the_forced_lookahead_at_three_waves(monkeypatch):
    """256^3 (ESS bricks of 4 voxels: the lookahead is off by default there) with VRHIP_EMPTY_SKIP=1 and
    VRHIP_OCC=3: the 12-wave kernels with the empty-run lookahead on a SINGLE frame and on a launch set --
    a conjunction the defaults never pick at this size -- against the oracle."""
    import torch

    monkeypatch.setenv("VRHIP_EMPTY_SKIP", "1")
    monkeypatch.setenv("VRHIP_OCC", "3")
    res = (256, 256, 256)
    vol = vro.synth_volume("shells", list(res), UCHAR)
    tff = common.tffs()["default"]
    W, H = 200, 152
    r = VolumeRenderCL()
    r.initialize()
    try:
        r.synthVolume("shells", res, UCHAR)
        r.setTransferFunction(tff)
        r.updateView(common.views()["rot30"])
        r.setSeed(SEED)
        r.setIteration(0)
        got, ref, stats = _compare(r, vol, UCHAR, tff, W, H)
        li = r.lastLaunchInfo()     # (the production frame of _compare)
        assert li["phase1_waves"] == 12 and li["phase2_waves"] == 12 and li["empty_skip"] == 1 and li["frames"] == 1, li
        mt = frontend.Mt19937(5)
        seeds = [mt() for _ in range(5)]
        out = torch.zeros((len(seeds), H, W, 4), dtype=torch.float32, device="cuda")
        r.setRoundBudget(48)
        r.render_batch(W, H, seeds, out.data_ptr())
        torch.cuda.synchronize()
        li = r.lastLaunchInfo()
        assert li["phase1_waves"] == 12 and li["empty_skip"] == 1 and li["frames"] == 5 and li["round_budget"] == 48, li
        bat = out.cpu().numpy()
        for i, sd in enumerate(seeds):
            r.setSeed(sd)
            r.setIteration(0)
            ref_i, _, _ = common.oracle_frame(r, vol, UCHAR, tff, W, H)
            assert np.abs(bat[i].astype(np.float64) - ref_i).max() <= TOL, i
    finally:
        r.close()


@pytest.mark.parametrize("size", [(640, 160), (160, 640), (456, 152)])
def test_patch_classes_on_frames_that_are_not_square(size):
    """The jitter moves a ray by up to max(gsx, gsy) / gs pixels along an axis (volumeraycast.cl:625-628): on a 4:1
    or 1:4 frame by four pixels along the short axis.  The pre-pass's patch classes (background patches written
    without a ray) must hold for every such ray: launch sets of frames with different jitter seeds, and single
    frames, against the oracle -- along the silhouette of the box and next to the outermost shell."""
    import torch

    W, H = size
    res = (256, 256, 256)
    vol = vro.synth_volume("shells", list(res), UCHAR)
    tff = common.tffs()["default"]
    r = VolumeRenderCL()
    r.initialize()
    try:
        r.synthVolume("shells", res, UCHAR)
        r.setTransferFunction(tff)
        # background alpha 1: a ray that misses the box keeps it (:655), a class-1 patch's rays end with alpha 0
        r.params()[1].backgroundColor[:] = [0.25, 0.5, 0.75, 1.0]
        for view in ("rot30", "default"):
            r.updateView(common.views()[view])
            mt = frontend.Mt19937(99)
            seeds = [mt() for _ in range(6)]
            out = torch.zeros((len(seeds), H, W, 4), dtype=torch.float32, device="cuda")
            r.render_batch(W, H, seeds, out.data_ptr())
            torch.cuda.synchronize()
            assert r.lastLaunchInfo()["patch_classes"] == 1
            bat = out.cpu().numpy()
            for i, sd in enumerate(seeds):
                r.setSeed(sd)
                r.setIteration(0)
                ref_i, _, _ = common.oracle_frame(r, vol, UCHAR, tff, W, H)
                d = np.abs(bat[i].astype(np.float64) - ref_i)
                assert d.max() <= TOL, "%s view, frame %d: %g at %s" % (view, i, d.max(), np.unravel_index(d.argmax(), d.shape))
                if i < 2:
                    r.setStatsEnabled(False)
                    one = r.runRaycastNoGL(W, H)
                    r.setIteration(0)
                    assert np.array_equal(one, bat[i])
    finally:
        r.close()


def test_config5_size_pathtrace_matches_oracle():
    """BASELINE config 5's size and field: 1024^3 FLOAT sphere, technique 1 (Woodcock-tracking
    path tracer, max_extinction 100).  Field generated in HBM, downloaded (4 GiB); iterations 0-2 of the
    progressive render (running mean, std::mt19937 seed stream) of a small frame through the instrumented
    and the production kernels against the oracle; the micro-bricks the product's own fetches touch are a
    subset of the reference's fetch set."""
    N = 1024
    tff = common.tffs()["default"]
    W, H = 96, 80
    r = VolumeRenderCL()
    r.initialize()
    try:
        r.synthVolume("sphere", (N, N, N), FLOAT)
        vol = r.downloadVolume(0)
        assert vol.shape == (N, N, N) and vol.dtype == np.float32
        c = 2.0 * (np.arange(N) + 0.5) / N - 1.0          # the field on two slices (SURVEY 8d formula)
        for z in (300, N // 2):
            rr = np.sqrt(c[None, :] ** 2 + c[:, None] ** 2 + c[z] ** 2)
            np.testing.assert_allclose(vol[z], np.maximum(1.0 - rr / 0.9, 0.0).astype(np.float32), atol=2e-7)
        r.setTransferFunction(tff)
        r.setTechnique(1)
        r.setExtinction(100.0)
        r.updateView(common.views()["rot30"])
        for stats in (True, False):
            r.setStatsEnabled(stats)
            mt = frontend.Mt19937()
            ref = None
            for it in range(3):
                seed = mt()
                r.setSeed(seed)
                r.setIteration(it)
                got = r.runRaycastNoGL(W, H)
                cam, rp, rc, pt = common.to_oracle_params(*r.params())
                rp.seed, rp.iteration = seed, it
                ref, rstats, _ = vro.render_tile(vol, FLOAT, tff, cam, rp, rc, pt, W=W, H=H, in_accum=ref)
                assert np.abs(got.astype(np.float64) - ref).max() <= TOL, "stats=%s iteration %d" % (stats, it)
                if stats:
                    g = r.getStats()
                    assert g["samples_nominal"] <= g["bricks_skipped"] <= g["bricks_visited"] <= g["samples_taken"]
                    assert dict(g, bricks_visited=0, bricks_skipped=0, samples_nominal=0) == rstats
                    assert g["samples_nominal"] > 0.5 * g["samples_taken"]   # most steps are taken in leaps
                    assert g["bricks_skipped"] > 0.9 * g["samples_taken"]     # the sphere culls too
        r.setStatsEnabled(False)
        r.setSeed(SEED)
        r.setIteration(0)
        n_ref, _ = r.countTouched(W, H)
        n_fetched = r.countFetched(W, H)
        assert 0 < n_fetched < n_ref
    finally:
        r.close()


def test_shared_volume_twin_renders_the_same_frames(vr):
    """vrhip_share_volumes / VolumeRenderCL.shareVolumes: a second renderer on its own stream
    renders from the first one's voxels and bricks; frames of both equal the oracle's, also when
    they are in flight together, and the owner is untouched when the twin goes away."""
    import torch
    vol = common.noise_volume((56, 48, 40), USHORT, seed=17, smooth=False)
    tff = common.tffs()["default"]
    W, H = 96, 72
    _setup(vr, vol, USHORT, tff, common.views()["rot30"])
    vr.setStatsEnabled(False)
    twin = vr.shareVolumes()
    s2 = torch.cuda.Stream()
    twin.set_stream(s2.cuda_stream)
    try:
        seeds = [SEED, 581869302, 3890346734, 3586334585]
        outs = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in seeds]
        for k, seed in enumerate(seeds):            # alternate: two frames in flight at a time
            r = (vr, twin)[k % 2]
            r.setSeed(seed)
            r.setIteration(0)
            r.runRaycast(W, H, out_dev_ptr=outs[k].data_ptr())
        torch.cuda.synchronize()
        cam, rp, rc, pt = common.to_oracle_params(*vr.params())
        for k, seed in enumerate(seeds):
            rp.seed, rp.iteration = seed, 0
            ref, _, _ = vro.render_tile(vol, USHORT, tff, cam, rp, rc, pt, W=W, H=H)
            assert np.abs(outs[k].cpu().numpy() - ref).max() <= TOL, "frame %d" % k
        # the twin has its own transfer function and parameters
        twin.setTransferFunction(common.tffs()["opaque"])
        twin.setIllumination(0)
        twin.setSeed(SEED)
        twin.setIteration(0)
        got = twin.runRaycastNoGL(W, H)
        cam, rp, rc, pt = common.to_oracle_params(*twin.params())   # (after the render: its seed)
        rp.iteration = 0
        ref, _, _ = vro.render_tile(vol, USHORT, common.tffs()["opaque"], cam, rp, rc, pt, W=W, H=H)
        assert np.abs(got - ref).max() <= TOL
        with pytest.raises(ValueError):
            twin.shareVolumes()                     # only an owner can share
    finally:
        twin.close()
    vr.setSeed(SEED)
    vr.setIteration(0)
    again = vr.runRaycastNoGL(W, H)
    vr.setIteration(0)
    ref, _, _ = common.oracle_frame(vr, vol, USHORT, tff, W, H)
    assert np.abs(again - ref).max() <= TOL


def test_transfer_function_edits_keep_shared_bricks_alive(vr):
    """the owner's setTransferFunction (the most common interactive action) re-runs
    generateBricks; the twin keeps a pointer to the owner's bricks, so they must neither move nor
    be freed.  Owner TF edit -> twin TF edit (its skip bitmap is rebuilt from the shared bricks)
    -> both frames against the oracle."""
    vol = common.noise_volume((56, 48, 40), UCHAR, seed=23, smooth=False)
    W, H = 96, 72
    _setup(vr, vol, UCHAR, common.tffs()["default"], common.views()["rot30"])
    vr.setStatsEnabled(False)
    twin = vr.shareVolumes()
    try:
        ballast = []
        for name_owner, name_twin in (("opaque", "haze"), ("haze", "default"), ("default", "opaque")):
            vr.setTransferFunction(common.tffs()[name_owner])      # would have freed + re-allocated
            # something else takes whatever allocation a free would have released
            ballast.append(np.zeros(1, dtype=np.uint8))
            other = VolumeRenderCL()
            other.initialize()
            other.loadVolumeArrays([common.noise_volume((64, 64, 64), UCHAR, seed=len(ballast))], UCHAR)
            other.setTransferFunction(common.tffs()["opaque"])
            twin.setTransferFunction(common.tffs()[name_twin])     # skip bitmap from the shared bricks
            for r, name in ((vr, name_owner), (twin, name_twin)):
                r.setSeed(SEED)
                r.setIteration(0)
                got = r.runRaycastNoGL(W, H)
                cam, rp, rc, pt = common.to_oracle_params(*r.params())
                rp.iteration = 0
                ref, _, _ = vro.render_tile(vol, UCHAR, common.tffs()[name], cam, rp, rc, pt, W=W, H=H)
                assert np.abs(got - ref).max() <= TOL, (name_owner, name_twin, name)
            other.close()
    finally:
        twin.close()


def test_tile_driver_lanes_and_batches_on_one_gpu(vr):
    """TileDriver with two renderers per rank (frames in flight) and several frames per gather:
    rank 0 of a 2-rank split with a stand-in collective -- its tiles of every frame of the batch
    must equal the same pixels of the full frame with that frame's seed."""
    import torch
    from volumerenderercl_amd import tiles

    class OneRankDist:           # gather: rank 0's own block arrives, the others stay zero
        class _Done:
            def wait(self):
                pass

        def gather(self, tensor, gather_list, dst=0, async_op=False):
            gather_list[0].copy_(tensor)
            return self._Done()

    vol = common.noise_volume((48, 48, 48), UCHAR, seed=21, smooth=False)
    tff = common.tffs()["default"]
    W, H, T = 160, 96, 32
    _setup(vr, vol, UCHAR, tff, common.views()["rot30"])
    vr.setStatsEnabled(False)
    dev = torch.device("cuda")
    twin = vr.shareVolumes()
    s2 = torch.cuda.Stream()
    twin.set_stream(s2.cuda_stream)
    vr.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        split = tiles.TileSplit(W, H, T, T, 2, 0)
        drv = tiles.TileDriver(vr, split, dev, dist=OneRankDist(), batch=3,
                               lanes=[(vr, torch.cuda.current_stream()), (twin, s2)])
        seeds = [SEED, 581869302, 3890346734]
        frames = torch.zeros((3, H, W, 4), dtype=torch.float32, device=dev)

        def before(i, r):
            r.setSeed(seeds[i])
            r.setIteration(0)

        drv.submit_batch(3, before)
        drv.collect_batch(frames)
        torch.cuda.synchronize()
        got = frames.cpu().numpy()
        for i, seed in enumerate(seeds):
            vr.setSeed(seed)
            vr.setIteration(0)
            full = vr.runRaycastNoGL(W, H)
            for t in split.my_tiles:
                x0, y0, w, h = split.tile_rect(t)
                np.testing.assert_array_equal(got[i, y0:y0 + h, x0:x0 + w], full[y0:y0 + h, x0:x0 + w])
    finally:
        twin.close()
        vr.set_stream(None, use_own=True)


def test_tile_driver_orders_the_renderers_own_stream(vr):
    """a renderer left on its OWN (non-blocking) stream -- nobody called set_stream --
    driven through TileDriver: the collective (here a stand-in that copies on torch's current
    stream, like RCCL orders itself behind it) must see finished tiles, frame after frame into the
    same two buffers."""
    import torch
    from volumerenderercl_amd import tiles

    class OneRankDist:
        class _Done:
            def wait(self):
                pass

        def gather(self, tensor, gather_list, dst=0, async_op=False):
            gather_list[0].copy_(tensor)      # on the current stream
            return self._Done()

    vol = common.noise_volume((96, 96, 96), UCHAR, seed=5, smooth=False)
    tff = common.tffs()["haze"]               # dense: long frames, so a missing wait shows
    W, H, T = 512, 384, 64
    _setup(vr, vol, UCHAR, tff, common.views()["close"])
    vr.setStatsEnabled(False)
    vr.set_stream(None, use_own=True)
    assert vr.get_stream() != torch.cuda.current_stream().cuda_stream
    dev = torch.device("cuda")
    split = tiles.TileSplit(W, H, T, T, 2, 0)
    drv = tiles.TileDriver(vr, split, dev, dist=OneRankDist(), batch=1)
    seeds = [SEED, 581869302, 3890346734, 3586334585, 545404204]
    frames = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in seeds]
    for i, seed in enumerate(seeds):
        vr.setSeed(seed)
        vr.setIteration(0)
        drv.submit()
        drv.collect(frames[i])
    torch.cuda.synchronize()
    for i, seed in enumerate(seeds):
        vr.setSeed(seed)
        vr.setIteration(0)
        full = vr.runRaycastNoGL(W, H)
        got = frames[i].cpu().numpy()
        for t in split.my_tiles:
            x0, y0, w, h = split.tile_rect(t)
            np.testing.assert_array_equal(got[y0:y0 + h, x0:x0 + w], full[y0:y0 + h, x0:x0 + w])
    # world == 1: the caller's frame tensor is consumed on the current stream
    one = tiles.TileDriver(vr, tiles.TileSplit(W, H, T, T, 1, 0), dev)
    vr.setSeed(SEED)
    vr.setIteration(0)
    out = one.render_frame(torch.zeros((H, W, 4), dtype=torch.float32, device=dev)).clone()
    vr.setSeed(SEED)
    vr.setIteration(0)
    np.testing.assert_array_equal(out.cpu().numpy(), vr.runRaycastNoGL(W, H))


def test_shared_twin_follows_its_own_time_step(vr):
    """A twin shares every time step of the owner and selects its own: bricks come shared, skip
    bitmap, cell grid and footprint volume are the twin's."""
    steps = [common.noise_volume((40, 40, 40), UCHAR, seed=60 + t, smooth=False) for t in range(2)]
    tff = common.tffs()["default"]
    W, H = 64, 48
    vr.loadVolumeArrays(steps, UCHAR)
    vr.setTransferFunction(tff)
    for name, val in (("setIllumination", 1), ("setObjEss", True), ("setTechnique", 0),
                      ("setAmbientOcclusion", False), ("setShowESS", False), ("setImgEss", False),
                      ("setContours", False), ("setAerial", False), ("setCamOrtho", False)):
        getattr(vr, name)(val)
    vr.updateView(common.views()["rot30"])
    vr.setStatsEnabled(False)
    vr.setTimestep(0)
    twin = vr.shareVolumes()
    try:
        twin.setTimestep(1)
        for r, t in ((vr, 0), (twin, 1), (vr, 0)):
            r.setSeed(SEED)
            r.setIteration(0)
            got = r.runRaycastNoGL(W, H)
            cam, rp, rc, pt = common.to_oracle_params(*r.params())
            rp.iteration = 0
            ref, _, _ = vro.render_tile(steps[t], UCHAR, tff, cam, rp, rc, pt, W=W, H=H)
            assert np.abs(got - ref).max() <= TOL, "time step %d" % t
    finally:
        twin.close()


@pytest.mark.parametrize("refill,budget", [("1", 1), ("5", 3), ("16", 32), ("3", 0)])
def test_phase2_schedules_do_not_change_pixels(vr, monkeypatch, refill, budget):
    """The scheduling knobs of the two-phase march -- rounds in phase 1 before a ray is suspended
    (vrhip_set_round_budget; 0 = single phase) and idle ray slots per wave before phase 2 refills
    them from the sorted list (VRHIP_REFILL_MIN; partial refills rank the idle slots) -- move work
    between kernels and lanes, never a pixel."""
    vol = vro.synth_volume("shells", [96, 96, 96], UCHAR)
    tff = common.tffs()["default"]
    W, H = 136, 120
    _setup(vr, vol, UCHAR, tff, common.views()["rot30"])
    vr.setStatsEnabled(False)
    want = vr.runRaycastNoGL(W, H)
    vr.setIteration(0)
    ref, _, _ = common.oracle_frame(vr, vol, UCHAR, tff, W, H)
    assert np.abs(want - ref).max() <= TOL
    monkeypatch.setenv("VRHIP_REFILL_MIN", refill)
    r2 = VolumeRenderCL()
    r2.initialize()
    try:
        _setup(r2, vol, UCHAR, tff, common.views()["rot30"])
        r2.setRoundBudget(budget)
        r2.setStatsEnabled(False)
        for _ in range(2):          # second frame: sorted by the first one's per-pixel cost
            got = r2.runRaycastNoGL(W, H)
            r2.setIteration(0)
            assert np.array_equal(got, want)
    finally:
        r2.close()


@pytest.mark.parametrize("fmt,res,kw", [(UCHAR, (48, 48, 48), {}), (USHORT, (40, 56, 36), {"ess": False}),
                                         (FLOAT, (44, 44, 44), {"illum": 0, "contours": True})])
def test_batch_of_frames_in_one_launch_set(vr, fmt, res, kw):
    """vrhip_render_batch: several independent frames (own jitter seeds) share one work queue and
    one set of launches; every frame equals its stand-alone rendering (and the oracle), whole
    frames and tile subsets, default and instrumented kernels."""
    import torch
    vol = common.noise_volume(res, fmt, seed=33, smooth=False)
    tff = common.tffs()["default"]
    W, H, T = 112, 80, 32
    _setup(vr, vol, fmt, tff, common.views()["rot30"], **kw)
    seeds = [SEED, 581869302, 3890346734, 3586334585, 545404204]
    singles = []
    vr.setStatsEnabled(False)
    for s in seeds:
        vr.setSeed(s)
        vr.setIteration(0)
        singles.append(vr.runRaycastNoGL(W, H))
    cam, rp, rc, pt = common.to_oracle_params(*vr.params())
    rp.seed, rp.iteration = seeds[3], 0
    ref, _, _ = vro.render_tile(vol, fmt, tff, cam, rp, rc, pt, use_ess=kw.get("ess", True), W=W, H=H)
    assert np.abs(singles[3] - ref).max() <= TOL
    for stats in (False, True):
        vr.setStatsEnabled(stats)
        out = torch.zeros((len(seeds), H, W, 4), dtype=torch.float32, device="cuda")
        vr.render_batch(W, H, seeds, out.data_ptr())
        torch.cuda.synchronize()
        vr.getLastExecTime()
        got = out.cpu().numpy()
        for f in range(len(seeds)):
            assert np.array_equal(got[f], singles[f]), "frame %d (stats %s)" % (f, stats)
    # tile subset, compact output per frame
    vr.setStatsEnabled(False)
    tiles_x = (W + T - 1) // T
    ids = np.array([0, 2, 5, 7, 9], dtype=np.uint32)
    out = torch.zeros((3, len(ids), T, T, 4), dtype=torch.float32, device="cuda")
    vr.render_batch(W, H, seeds[:3], out.data_ptr(), tile_w=T, tile_h=T, tile_ids=ids)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for f in range(3):
        for k, t in enumerate(ids):
            x0, y0 = (int(t) % tiles_x) * T, (int(t) // tiles_x) * T
            w, h = min(T, W - x0), min(T, H - y0)
            assert np.array_equal(got[f, k, :h, :w], singles[f][y0:y0 + h, x0:x0 + w])
    # what chains frames is refused
    vr.setAmbientOcclusion(True)
    with pytest.raises(RuntimeError):
        vr.render_batch(W, H, seeds[:2], out.data_ptr())
    vr.setAmbientOcclusion(False)


def test_tile_driver_submit_frames_on_one_gpu(vr):
    """TileDriver.submit_frames: the rank's tile share of several frames rendered as one launch
    set per renderer (render_batch with the gather buffer's frame stride, which is larger than
    the rank's own tile count here), gathered with a stand-in collective."""
    import torch
    from volumerenderercl_amd import tiles

    class OneRankDist:
        class _Done:
            def wait(self):
                pass

        def __init__(self, rank):
            self.rank = rank

        def gather(self, tensor, gather_list, dst=0, async_op=False):
            gather_list[self.rank].copy_(tensor)
            return self._Done()

    vol = common.noise_volume((48, 48, 48), UCHAR, seed=23, smooth=False)
    tff = common.tffs()["default"]
    W, H, T = 160, 96, 32
    _setup(vr, vol, UCHAR, tff, common.views()["rot30"])
    vr.setStatsEnabled(False)
    dev = torch.device("cuda")
    twin = vr.shareVolumes()
    s2 = torch.cuda.Stream()
    twin.set_stream(s2.cuda_stream)
    vr.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        # the share of a rank of 4 that has fewer tiles than the slot count of the gather
        probe = tiles.TileSplit(W, H, T, T, 4, 0)
        short = min(range(4), key=lambda k: len(probe.tiles_of[k]))
        split = tiles.TileSplit(W, H, T, T, 4, short)
        assert len(split.my_tiles) < split.cap
        split.rank = 0          # assemble locally; the stand-in puts the block where that rank's goes
        drv = tiles.TileDriver(vr, split, dev, dist=OneRankDist(short), batch=6,
                               lanes=[(vr, torch.cuda.current_stream()), (twin, s2)])
        seeds = [SEED, 581869302, 3890346734, 3586334585, 545404204]
        frames = torch.zeros((6, H, W, 4), dtype=torch.float32, device=dev)
        drv.submit_frames(seeds)
        drv.collect_batch(frames)
        torch.cuda.synchronize()
        got = frames.cpu().numpy()
        for i, seed in enumerate(seeds):
            vr.setSeed(seed)
            vr.setIteration(0)
            full = vr.runRaycastNoGL(W, H)
            for t in split.my_tiles:
                x0, y0, w, h = split.tile_rect(t)
                np.testing.assert_array_equal(got[i, y0:y0 + h, x0:x0 + w], full[y0:y0 + h, x0:x0 + w])
    finally:
        twin.close()
        vr.set_stream(None, use_own=True)


def test_batch_with_the_maximum_number_of_frames(vr):
    """256 frames per batch: the frame index uses all ten spare bits of the work items (five of the
    patch row, five of the patch column)."""
    import torch
    vol = common.noise_volume((40, 40, 40), UCHAR, seed=35, smooth=False)
    tff = common.tffs()["default"]
    W, H = 72, 56
    _setup(vr, vol, UCHAR, tff, common.views()["rot30"])
    vr.setStatsEnabled(False)
    mt = frontend.Mt19937()
    N = 256
    seeds = [mt() for _ in range(N)]
    out = torch.zeros((N, H, W, 4), dtype=torch.float32, device="cuda")
    vr.render_batch(W, H, seeds, out.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for f in (0, 1, 15, 16, 31, 32, 33, 63, 64, 127, 128, 200, 254, 255):
        vr.setSeed(seeds[f])
        vr.setIteration(0)
        assert np.array_equal(got[f], vr.runRaycastNoGL(W, H)), "frame %d" % f
    with pytest.raises(ValueError):
        vr.render_batch(W, H, seeds + [1], out.data_ptr())
    # a tile share of many frames, frames apart by a stride (what a rank of a multi-GPU split renders)
    from volumerenderercl_amd import tiles
    split = tiles.TileSplit(W, H, 16, 16, 4, 1)
    n = len(split.my_tiles)
    stride = (n + 1) * 16 * 16
    tout = torch.zeros((N, stride, 4), dtype=torch.float32, device="cuda")
    vr.render_batch(W, H, seeds, tout.data_ptr(), 16, 16, split.my_tiles, frame_stride=stride)
    torch.cuda.synchronize()
    tg = tout.cpu().numpy()
    for f in (0, 37, 255):
        tl = tg[f, :n * 256].reshape(n, 16, 16, 4)
        for k, t in enumerate(split.my_tiles):
            x0, y0, w, h = split.tile_rect(t)
            np.testing.assert_array_equal(tl[k, :h, :w], got[f, y0:y0 + h, x0:x0 + w])


def test_anisotropic_grid_samples_before_the_entry_face(monkeypatch):
    """real samples near tnear lie up to 2 |voxLen| BEFORE the entry face (t - offset,
    volumeraycast.cl:733 / :791) and are fetched clamp-to-edge.  On a 256 x 256 x 32 grid that is up
    to ~8 texels in x: two cells of four voxels below 0.  The empty-cell lookahead has to clamp such a
    position to the NEAR border cell (data touching the x = 0 face), not to the far one (empty here)."""
    X, Y, Z = 256, 256, 32
    zz, yy, xx = np.meshgrid(np.linspace(-1, 1, Z), np.linspace(-1, 1, Y), np.linspace(-1, 1, X), indexing="ij")
    f = 0.55 + 0.4 * np.cos(9 * xx) * np.cos(7 * yy + 1) * np.cos(5 * zz + 2)
    f[:, :, X // 2:] = 0.0                       # the far half (and its border cells) is exactly empty
    vol = np.round(np.clip(f, 0, 1) * 255).astype(np.uint8)
    assert vol[:, :, 0].min() > 0                # data on the x = 0 face
    tff = common.tffs()["default"]
    # camera on the -x side: the rays enter through the x = 0 face
    view = frontend.view_matrix(frontend.quat_from_axis_angle((0, 1, 0), -88.0), (0.0, 0.0, 2.0))
    W, H = 160, 120
    monkeypatch.setenv("VRHIP_EMPTY_SKIP", "1")
    r2 = VolumeRenderCL()
    r2.initialize()
    try:
        for ess in (True, False):
            for seed in (SEED, 581869302):
                _setup(r2, vol, UCHAR, tff, view, ess=ess, seed=seed)
                _compare(r2, vol, UCHAR, tff, W, H, ess=ess)
    finally:
        r2.close()


def test_same_signature_kernels_get_their_own_launch_cache_entries():
    """vr_prepare_kernel caches blocks/CU (and raises the dynamic-LDS limit) per kernel
    ADDRESS; instantiations that share a function-pointer type (phase 1 / phase 2, INSTR / FP / XS
    variants) must not share an entry.  VRHIP_DEBUG prints one line per new entry."""
    import os
    import re
    import subprocess
    import sys
    code = r"""
import numpy as np
from tests import common
from volumerenderercl_amd import UCHAR, VolumeRenderCL
vr = VolumeRenderCL(); vr.initialize()
vol = common.noise_volume((48, 48, 48), UCHAR, seed=3, smooth=False)
vr.loadVolumeArrays([vol], UCHAR)
big = np.repeat(common.tffs()["default"].reshape(-1, 4), 4, axis=0).reshape(-1)   # 4096 entries: 64 KiB of LDS
for tff in (common.tffs()["default"], big):
    vr.setTransferFunction(tff)
    vr.updateView(common.views()["rot30"])
    for stats in (False, True):
        for contours in (False, True):
            vr.setStatsEnabled(stats); vr.setContours(contours); vr.setSeed(7); vr.setIteration(0)
            vr.runRaycastNoGL(64, 64)
vr.close()
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, VRHIP_DEBUG="1"),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    entries = re.findall(r"\[vrhip\] (.+?) @(0x[0-9a-f]+): device (\d+), lds=(\d+) B", p.stderr)
    assert entries, p.stderr[-2000:]
    keys = [(a, d, l) for _, a, d, l in entries]
    assert len(keys) == len(set(keys))           # one line per (kernel, device, LDS size)
    by_sig = {}
    for what, addr, dev, lds in entries:
        by_sig.setdefault((what, dev, lds), set()).add(addr)
    # phase 1 and phase 2 (and the instrumented / XS instantiations) are different kernels behind
    # one description and LDS size: every one of them has to show up
    assert max(len(v) for v in by_sig.values()) >= 2, entries
    assert len({a for _, a, _, _ in entries}) >= 4, entries


def test_owner_going_away_detaches_its_sharers(vr):
    """a sharing renderer dereferenced its owner on every frame.  Destroying / clearing /
    re-sizing the owner now detaches the sharers first: they answer "No volume data is loaded." instead
    of reading freed host and device memory; an in-place upload of the same size reaches them."""
    vol = common.noise_volume((40, 36, 32), UCHAR, seed=31, smooth=False)
    tff = common.tffs()["default"]
    W, H = 64, 48
    owner = VolumeRenderCL()
    owner.initialize()
    _setup(owner, vol, UCHAR, tff, common.views()["rot30"])
    twin = owner.shareVolumes()
    try:
        twin.setSeed(SEED)
        twin.setIteration(0)
        a = twin.runRaycastNoGL(W, H)
        twin.setIteration(0)
        ref, _, _ = common.oracle_frame(twin, vol, UCHAR, tff, W, H)
        np.testing.assert_array_equal(a, ref)
        # in-place upload of other voxels of the same size: the twin renders the new ones
        vol2 = common.noise_volume((40, 36, 32), UCHAR, seed=32, smooth=True)
        owner._check(owner._lib.vrhip_upload_volume(owner._h, vol2.ctypes.data, (ctypes.c_uint32 * 3)(40, 36, 32),
                                                    UCHAR, 0))
        owner.setTransferFunction(tff)           # rebuilds the owner's bricks
        twin.setTransferFunction(tff)            # the twin takes them and rebuilds what it derived
        twin.setSeed(SEED)
        twin.setIteration(0)
        b = twin.runRaycastNoGL(W, H)
        twin.setIteration(0)
        ref2, _, _ = common.oracle_frame(twin, vol2, UCHAR, tff, W, H)
        np.testing.assert_array_equal(b, ref2)
        assert np.abs(ref2 - ref).max() > 0.01
        owner.close()                            # the twin is detached, not dangling
        with pytest.raises(RuntimeError, match="No volume data"):
            twin.runRaycastNoGL(W, H)
    finally:
        twin.close()
        owner.close()


def _run_py(code, env=None, timeout=600):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, **(env or {})),
                          capture_output=True, text=True, timeout=timeout)


def test_tile_driver_over_a_world_size_1_rccl_group():
    """RCCL executes: TileDriver(force_gather=True) puts a world of ONE rank through the world > 1 code --
    tile buffers, dist.gather on a world-size-1 `nccl` process group (asynchronous, one in flight), the
    assembly on rank 0 -- with whole frames one by one (submit / collect), batches of frames rendered by
    two renderers in flight (submit_frames) and the synchronous render_frame; every frame equals the
    full-frame render with the same seed.  In a process of its own: the process group is global state."""
    code = r"""
import numpy as np, torch, torch.distributed as dist
from tests import common
from volumerenderercl_amd import UCHAR, VolumeRenderCL, tiles
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29547", rank=0, world_size=1, device_id=dev)
vr = VolumeRenderCL(); vr.initialize()
vol = common.noise_volume((56, 48, 40), UCHAR, seed=5, smooth=False)
vr.loadVolumeArrays([vol], UCHAR)
vr.setTransferFunction(common.tffs()["default"])
vr.updateView(common.views()["rot30"])
W, H, T = 200, 136, 32                     # ragged right / bottom tiles
seeds = [3499211612, 581869302, 3890346734, 3586334585, 545404204]
full = []
for sd in seeds:
    vr.setSeed(sd); vr.setIteration(0); full.append(vr.runRaycastNoGL(W, H))
split = tiles.TileSplit(W, H, T, T, 1, 0)
# (1) synchronous frames and the pipelined pair submit / collect on the renderer's own stream
drv = tiles.TileDriver(vr, split, dev, force_gather=True)
frame = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
vr.setSeed(seeds[0]); vr.setIteration(0)
assert np.array_equal(drv.render_frame(frame).cpu().numpy(), full[0])
vr.setSeed(seeds[1]); vr.setIteration(0); drv.submit()
vr.setSeed(seeds[2]); vr.setIteration(0); drv.submit()          # two gathers in flight
for k in (1, 2):
    drv.collect(frame); torch.cuda.synchronize()
    assert np.array_equal(frame.cpu().numpy(), full[k]), k
# (2) batches of frames, two renderers in flight, one gather per batch
twin = vr.shareVolumes()
s1, s2 = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)
vr.set_stream(s1.cuda_stream); twin.set_stream(s2.cuda_stream)
drv2 = tiles.TileDriver(vr, split, dev, batch=4, lanes=[(vr, s1), (twin, s2)], force_gather=True)
frames = torch.zeros((4, H, W, 4), dtype=torch.float32, device=dev)
drv2.submit_frames(seeds[:4]); drv2.submit_frames(seeds[1:5])
for lo in (0, 1):
    drv2.collect_batch(frames); torch.cuda.synchronize()
    got = frames.cpu().numpy()
    for i in range(4):
        assert np.array_equal(got[i], full[lo + i]), (lo, i)
# (2b) the same with uniform tiles travelling as one pixel (sparse): all-gather of the counts + payload gather
drv3 = tiles.TileDriver(vr, split, dev, batch=4, lanes=[(vr, s1), (twin, s2)], force_gather=True, sparse=True)
drv3.submit_frames(seeds[:4]); drv3.submit_frames(seeds[1:5])
for lo in (0, 1):
    drv3.collect_batch(frames); torch.cuda.synchronize()
    got = frames.cpu().numpy()
    for i in range(4):
        assert np.array_equal(got[i], full[lo + i]), ("sparse", lo, i)
assert 0 < drv3.gather_stats["sent_bytes"] < drv3.gather_stats["dense_bytes"], drv3.gather_stats
drv4 = tiles.TileDriver(vr, split, dev, force_gather=True, sparse=True)
vr.setSeed(seeds[3]); vr.setIteration(0)
assert np.array_equal(drv4.render_frame(frame).cpu().numpy(), full[3])
# (3) the collectives bench.py issues around the timed region
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
c = torch.arange(6, dtype=torch.int64, device=dev); dist.all_reduce(c, op=dist.ReduceOp.SUM)
dist.barrier(); assert float(t.item()) == 1.5 and c.tolist() == list(range(6))
twin.close(); vr.close()
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
"""
    p = _run_py(code, env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert p.returncode == 0 and "RCCL_WORLD1_OK" in p.stdout, (p.stdout[-1000:], p.stderr[-3000:])


def test_bench_multi_gpu_path_rehearsed_over_rccl_on_one_gpu(tmp_path):
    """bench.py's distributed code path end to end on one GPU (VRHIP_BENCH_FORCE_GATHER=1): RCCL process
    group, batched gathers with one in flight, assembly, the barrier / all-reduce bracket, the JSON line --
    with the bench's own parity check against the oracle (exit code 3 on a mismatch)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "line.json"
    env = dict(os.environ, VRHIP_BENCH_FORCE_GATHER="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29549")
    for extra in ([], ["--frames-in-flight", "1", "--frames-per-launch", "1"]):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "sphere64", "--viewport", "192",
                            "--steps", "12", "--warmup", "2", "--frames-per-gather", "5", "--cpu-seconds", "0.5",
                            "--out-json", str(out)] + extra, cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
        d = json.loads(out.read_text())
        assert d["n_gpus"] == 1 and "RCCL gather per 5 frames" in d["config"]["parallelism"]
        assert d["parity"]["max_abs_diff"] <= TOL and d["parity"]["counters_equal"]
        assert d["value"] > 0 and d["work_per_frame"]["samples_taken"] > 0
        assert d["gather"]["mode"].startswith("sparse") and d["gather"]["sent_bytes_per_frame_to_root"] > 0


def test_bench_multi_gpu_path_rehearsed_with_two_ranks_over_gloo(tmp_path):
    """bench.py as torch.distributed.run launches it for N = 2 -- two ranks, every collective, tile shares, the sparse
    batched gather, rank 0's assembly and its parity check of the timed region's own frames -- with both ranks on this
    box's one GPU over gloo (VRHIP_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device): nothing of the N > 1 path
    runs for the first time on the 8-GPU node except the transport."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VRHIP_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for i, extra in enumerate((["--steps", "20", "--warmup", "5"],            # the round driver's arguments: one batch
                               ["--steps", "12", "--warmup", "2", "--frames-per-gather", "5"],
                               ["--steps", "6", "--warmup", "1", "--frames-in-flight", "1", "--frames-per-launch", "1"])):
        out = tmp_path / ("line%d.json" % i)
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(29561 + i), os.path.join(root, "bench.py"),
                            "--gpus", "2", "--workload", "sphere64", "--viewport", "192", "--tile", "32",
                            "--out-json", str(out)] + extra, cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
        d = json.loads(out.read_text())
        assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
        assert d["parity"]["max_abs_diff"] <= TOL and d["parity"]["counters_equal"]
        assert len(d["parity"]["timed_frames"]) >= 1 and all(c["max_abs_diff"] <= TOL for c in d["parity"]["timed_frames"])
        assert d["cpu_baseline"] is None          # (rank 0 at N = 1 only)
        assert len(d["gather"]["tiles_per_rank"]) == 2


def test_control_blocks_alternate_cleanly(vr):
    """The work-queue control words live in two blocks; the first kernel of a set of launches zeroes the
    block of the next set (no memset launch per frame).  Whatever is interleaved -- instrumented frames,
    the traffic pass (which uses a block on its own), tiles, batches, the path tracer, phase timing --
    every frame must equal the one a fresh renderer produces."""
    import torch
    vol = common.noise_volume((48, 40, 44), UCHAR, seed=41, smooth=False)
    tff = common.tffs()["default"]
    W, H = 96, 72
    _setup(vr, vol, UCHAR, tff, common.views()["rot30"])
    vr.setStatsEnabled(False)
    fresh = VolumeRenderCL()
    fresh.initialize()
    try:
        _setup(fresh, vol, UCHAR, tff, common.views()["rot30"])

        def ref(seed, technique=0):
            fresh.setTechnique(technique)
            fresh.setSeed(seed)
            fresh.setIteration(0)
            return fresh.runRaycastNoGL(W, H)

        def got(seed, technique=0):
            vr.setTechnique(technique)
            vr.setSeed(seed)
            vr.setIteration(0)
            return vr.runRaycastNoGL(W, H)

        seeds = [SEED, 581869302, 3890346734, 3586334585, 545404204, 4161255391]
        want = {s_: ref(s_) for s_ in seeds}
        want_pt = ref(seeds[0], 1)
        np.testing.assert_array_equal(got(seeds[0]), want[seeds[0]])
        np.testing.assert_array_equal(got(seeds[1]), want[seeds[1]])
        vr.setIteration(0)
        vr.countTouched(W, H)                                   # a pass of its own on the current block
        np.testing.assert_array_equal(got(seeds[2]), want[seeds[2]])
        vr.setStatsEnabled(True)
        np.testing.assert_array_equal(got(seeds[3]), want[seeds[3]])
        vr.setStatsEnabled(False)
        np.testing.assert_array_equal(got(seeds[0], 1), want_pt)   # path tracer: another first kernel
        vr.setPhaseTiming(True)
        np.testing.assert_array_equal(got(seeds[4]), want[seeds[4]])
        p1, p2 = vr.getLastPhaseTimes()
        assert p1 > 0 and p2 >= 0 and abs((p1 + p2) - vr.getLastExecTime()) < 0.2 * vr.getLastExecTime() + 2e-5
        vr.setPhaseTiming(False)
        np.testing.assert_array_equal(got(seeds[5]), want[seeds[5]])
        assert vr.getLastPhaseTimes() == (0.0, 0.0)
        # the events around a frame are a switch too (vrhip_set_frame_timing): same pixels, no kernel time
        assert vr.getLastExecTime() > 0
        vr.setFrameTiming(False)
        vr.setPhaseTiming(True)                                 # (has nothing to measure against: no data either)
        np.testing.assert_array_equal(got(seeds[4]), want[seeds[4]])
        assert vr.getLastExecTime() == 0.0 and vr.getLastPhaseTimes() == (0.0, 0.0)
        vr.setPhaseTiming(False)
        vr.setFrameTiming(True)
        np.testing.assert_array_equal(got(seeds[5]), want[seeds[5]])
        assert vr.getLastExecTime() > 0
        out = torch.zeros((3, H, W, 4), dtype=torch.float32, device="cuda")
        for _ in range(3):                                      # batches: odd number of sets in a row
            vr.setTechnique(0)
            vr.render_batch(W, H, seeds[:3], out.data_ptr())
        torch.cuda.synchronize()
        for f in range(3):
            np.testing.assert_array_equal(out[f].cpu().numpy(), want[seeds[f]])
        np.testing.assert_array_equal(got(seeds[1]), want[seeds[1]])
    finally:
        fresh.close()
        vr.setTechnique(0)


def test_product_library_refuses_the_experiment_kernels(monkeypatch):
    """vr_march_kernel, vr_raycast_staged_kernel and leap stepping are compiled into A/B builds only
    (-DVR_EXPERIMENTS); the product library says so instead of rendering with other kernels."""
    for var in ("VRHIP_MARCH", "VRHIP_LDS_STAGE", "VRHIP_MARCH_MICRO"):
        monkeypatch.setenv(var, "1")
        r = VolumeRenderCL()
        with pytest.raises(RuntimeError, match="VR_EXPERIMENTS"):
            r.initialize()
        monkeypatch.delenv(var)


def test_experiment_kernels_do_not_change_pixels():
    """The A/B build with the experiment kernels (tools/mkvariant.sh experiments -DVR_EXPERIMENTS
    -DVR_LEAP_STEPPING; skipped where it has not been built): the decoupled march kernel, leap stepping in the
    two-phase kernels and the LDS brick staging kernel (with and without the staged boxes) render frames
    bit-identical to that library's default schedule, which equals the oracle's."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "volumerenderercl_amd", "_variants", "libvrhip_experiments.so")
    if not os.path.exists(lib):
        pytest.skip("A/B build with the experiment kernels not present")
    import ctypes
    from volumerenderercl_amd import _srchash
    try:
        vlib = ctypes.CDLL(lib)
        vlib.vrhip_build_source_hash.restype = ctypes.c_char_p
        built = vlib.vrhip_build_source_hash().decode()
    except (OSError, AttributeError):
        built = None
    if built != _srchash.source_hash():
        pytest.skip("A/B build with the experiment kernels is older than the sources (tools/mkvariant.sh rebuilds it)")
    code = r"""
import os, sys
import numpy as np
from oracle import vro
from tests import common
from volumerenderercl_amd import UCHAR, VolumeRenderCL
res = (256, 256, 256)
vol = vro.synth_volume("shells", list(res), UCHAR)
vol[:, :, :96] = 0
tff = common.tffs()["default"]
W, H = 200, 152
frames = {}
envs = [("default", {}), ("march", {"VRHIP_MARCH": "1"}), ("march1", {"VRHIP_MARCH": "1", "VRHIP_MARCH_MICRO": "1"}),
        ("leap", {"VRHIP_MARCH_MICRO": "3"}), ("lds1", {"VRHIP_LDS_STAGE": "1"}), ("lds2", {"VRHIP_LDS_STAGE": "2"})]
for name, e in envs:
    os.environ["VRHIP_EMPTY_SKIP"] = "1"
    for k, v in e.items():
        os.environ[k] = v
    r2 = VolumeRenderCL(); r2.initialize()
    r2.loadVolumeArrays([vol], UCHAR); r2.setTransferFunction(tff)
    for view in ("rot30", "close", "inside"):
        r2.updateView(common.views()[view])
        for seed in (3499211612, 581869302):
            r2.setSeed(seed); r2.setIteration(0)
            frames[(name, view, seed)] = r2.runRaycastNoGL(W, H)
    if name == "default":
        r2.updateView(common.views()["rot30"]); r2.setSeed(3499211612); r2.setIteration(0)
        cam, rp, rc, pt = common.to_oracle_params(*r2.params())
        rp.seed, rp.iteration = 3499211612, 0
        ref, _, _ = vro.render_tile(vol, UCHAR, tff, cam, rp, rc, pt, W=W, H=H)
        assert np.array_equal(frames[("default", "rot30", 3499211612)], ref)
    r2.close()
    for k in e:
        del os.environ[k]
for (name, view, seed), img in frames.items():
    assert np.array_equal(img, frames[("default", view, seed)]), (name, view, seed)
print("EXPERIMENTS_OK", len(frames))
"""
    p = _run_py(code, env={"VRHIP_LIB_PATH": lib})
    assert p.returncode == 0 and "EXPERIMENTS_OK 36" in p.stdout, (p.stdout[-500:], p.stderr[-3000:])


def _random_scene(rng):
    """A scene drawn from the parameter space the reference's GUI can reach: voxel type, odd non-cubic resolution,
    anisotropic slice thickness, empty halves, transfer-function stops, camera inside / outside / orthographic,
    clip box, sampling rate, shading, ESS, filtering, jitter seed, odd viewports."""
    fmt = [UCHAR, USHORT, FLOAT][int(rng.integers(3))]
    res = tuple(int(v) for v in rng.integers(17, 70, size=3))
    vol = common.noise_volume(res, fmt, seed=int(rng.integers(1 << 30)), smooth=bool(rng.integers(2)))
    if rng.random() < 0.5:   # a slab of nothing: bricks to skip, cells to step over
        ax, cut = int(rng.integers(3)), float(rng.uniform(0.2, 0.6))
        sl = [slice(None)] * 3
        n = vol.shape[ax]
        sl[ax] = slice(0, int(n * cut)) if rng.random() < 0.5 else slice(int(n * (1 - cut)), n)
        vol[tuple(sl)] = 0
    pos = np.sort(rng.uniform(0.02, 0.98, size=int(rng.integers(1, 4))))
    stops = [(0.0, (0, 0, 0, 0))]
    for p in pos:
        stops.append((float(p), tuple(int(v) for v in rng.integers(0, 256, size=3)) +
                      (int(rng.integers(0, 256)) if rng.random() < 0.7 else 0,)))
    stops.append((1.0, tuple(int(v) for v in rng.integers(0, 256, size=3)) + (int(rng.integers(0, 256)),)))
    tff = frontend.tff_from_stops(stops)
    q = frontend.quat_from_axis_angle(tuple(rng.normal(size=3)), float(rng.uniform(0, 360)))
    z = float(rng.choice([rng.uniform(0.2, 0.9), rng.uniform(1.2, 4.0)]))
    view = frontend.view_matrix(q, (float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)), z))
    kw = {"illum": int(rng.integers(2)), "ess": bool(rng.random() < 0.75), "linear": bool(rng.random() < 0.85),
          "ortho": bool(rng.random() < 0.2), "rate": float(rng.choice([0.5, 1.0, 1.5, 2.0, 3.1])),
          "gradient_bg": bool(rng.random() < 0.3), "seed": int(rng.integers(1, 1 << 32))}
    if rng.random() < 0.3:
        lo = rng.uniform(-1.0, -0.2, size=3)
        hi = rng.uniform(0.2, 1.0, size=3)
        kw["bbox"] = tuple(float(v) for v in lo) + tuple(float(v) for v in hi)
    if rng.random() < 0.3:
        kw["thickness"] = tuple(float(v) for v in rng.choice([0.5, 1.0, 1.0, 2.0, 3.3], size=3))
    if rng.random() < 0.25:
        kw["background"] = tuple(float(v) for v in rng.uniform(0, 1, size=3)) + (1.0,)
    W, H = int(rng.integers(33, 150)), int(rng.integers(33, 120))
    return vol, fmt, tff, view, kw, W, H


@pytest.mark.parametrize("case", range(48))
def test_randomised_scenes_match_oracle(vr, case):
    """Scenes nobody wrote by hand (seeded: the same 48 every run): instrumented and production kernels against the
    oracle, image and work counters (_compare)."""
    rng = np.random.default_rng(20261004 + case)
    vol, fmt, tff, view, kw, W, H = _random_scene(rng)
    _setup(vr, vol, fmt, tff, view, **kw)
    try:
        vr.updateOutputImg(W, H)
        _compare(vr, vol, fmt, tff, W, H, ess=kw["ess"])
    finally:
        vr.setBBox(-1, -1, -1, 1, 1, 1)
        vr.setUseGradient(False)
        vr.setCamOrtho(False)
        vr.setLinearInterpolation(True)


@pytest.mark.parametrize("case", range(32))
def test_randomised_scenes_with_the_rarer_modes_match_oracle(vr, case):
    """The same with the modes that live in kernel variants of their own: illumination 2-5, contours, the depth
    cue, ambient occlusion, showEss, nearest filtering, progressive accumulation over two iterations."""
    rng = np.random.default_rng(77261004 + case)
    vol, fmt, tff, view, kw, W, H = _random_scene(rng)
    kw.update(illum=int(rng.integers(0, 6)), contours=bool(rng.random() < 0.3), aerial=bool(rng.random() < 0.3),
              ao=bool(rng.random() < 0.25), show_ess=bool(rng.random() < 0.25), linear=bool(rng.random() < 0.6))
    _setup(vr, vol, fmt, tff, view, **kw)
    try:
        vr.updateOutputImg(W, H)
        _compare(vr, vol, fmt, tff, W, H, ess=kw["ess"])
    finally:
        _setup(vr, vol, fmt, tff, view)   # every switch back to its default


@pytest.mark.parametrize("case", range(24))
def test_randomised_scenes_path_traced_match_oracle(vr, case):
    """The same through the path tracer (technique 1), with a random extinction."""
    rng = np.random.default_rng(55261004 + case)
    vol, fmt, tff, view, kw, W, H = _random_scene(rng)
    kw = {k: v for k, v in kw.items() if k in ("seed", "bbox", "thickness", "background", "ortho", "gradient_bg")}
    kw["ext"] = float(rng.choice([10.0, 40.0, 100.0, 250.0]))
    _setup(vr, vol, fmt, tff, view, technique=1, **kw)
    try:
        vr.updateOutputImg(W, H)
        _compare(vr, vol, fmt, tff, W, H, pathtrace=True)
    finally:
        _setup(vr, vol, fmt, tff, view)


@pytest.mark.parametrize("case", range(16))
def test_randomised_batches_of_tile_subsets_equal_their_frames(vr, case):
    """vrhip_render_batch on random scenes: a random number of frames (own seeds), a random tile size and a random
    subset of the tiles in one set of launches (two or three waves per SIMD by the set's size) -- every tile of
    every frame equals the same region of the frame rendered alone, which _compare checks against the oracle."""
    import torch
    rng = np.random.default_rng(99261004 + case)
    vol, fmt, tff, view, kw, W, H = _random_scene(rng)
    kw["linear"] = True if rng.random() < 0.8 else kw["linear"]
    _setup(vr, vol, fmt, tff, view, **kw)
    try:
        vr.updateOutputImg(W, H)
        _compare(vr, vol, fmt, tff, W, H, ess=kw["ess"])
        n_frames = int(rng.integers(1, 10))
        seeds = [int(v) for v in rng.integers(1, 1 << 32, size=n_frames)]
        vr.setStatsEnabled(False)
        singles = []
        for s in seeds:
            vr.setSeed(s)
            vr.setIteration(0)
            singles.append(vr.runRaycastNoGL(W, H))
        T = int(rng.choice([16, 32, 64]))
        tx, ty = (W + T - 1) // T, (H + T - 1) // T
        n_tiles = tx * ty
        ids = np.sort(rng.choice(n_tiles, size=int(rng.integers(1, n_tiles + 1)), replace=False)).astype(np.uint32)
        out = torch.zeros((n_frames, len(ids), T, T, 4), dtype=torch.float32, device="cuda")
        vr.render_batch(W, H, seeds, out.data_ptr(), tile_w=T, tile_h=T, tile_ids=ids)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for f in range(n_frames):
            for k, t in enumerate(ids):
                x0, y0 = (int(t) % tx) * T, (int(t) // tx) * T
                w, h = min(T, W - x0), min(T, H - y0)
                assert np.array_equal(got[f, k, :h, :w], singles[f][y0:y0 + h, x0:x0 + w]), (f, int(t))
    finally:
        _setup(vr, vol, fmt, tff, view)


@pytest.mark.parametrize("case", range(12))
def test_randomised_frame_sequences_match_oracle(vr, case):
    """Random scenes through the modes that chain frames: image-order ESS (the hit-image ping-pong over three
    frames, with whatever shading / filtering the scene drew) or progressive accumulation (the running mean over
    three iterations with their own jitter seeds), ray caster or path tracer."""
    rng = np.random.default_rng(33261004 + case)
    vol, fmt, tff, view, kw, W, H = _random_scene(rng)
    img_ess = case % 2 == 0
    technique = 1 if (not img_ess and rng.random() < 0.4) else 0
    if technique:
        kw = {k: v for k, v in kw.items() if k in ("seed", "bbox", "thickness", "background", "ortho", "gradient_bg")}
    else:
        kw["illum"] = int(rng.integers(0, 6))
    _setup(vr, vol, fmt, tff, view, technique=technique, img_ess=img_ess, **kw)
    try:
        vr.updateOutputImg(W, H)
        vr.setStatsEnabled(False)
        if img_ess:
            hin, hout = vro.hit_image_init(W, H)
            for frame in range(3):
                got = vr.runRaycastNoGL(W, H)
                vr.setIteration(0)
                cam, rp, rc, pt = common.to_oracle_params(*vr.params())   # (the frame's jitter seed: set when it renders)
                ref, _, _ = vro.render_tile(vol, fmt, tff, cam, rp, rc, pt, use_ess=kw.get("ess", True), W=W, H=H,
                                            hit_in=hin, hit_out=hout)
                assert np.abs(got - ref).max() <= TOL, frame
                hin, hout = hout, hin
                g_in, g_out = vr.getImageEss(W, H)
                assert np.array_equal(g_in, hin) and np.array_equal(g_out, hout), frame
        else:
            acc = None
            for it in range(3):
                vr.setSeed(int(rng.integers(1, 1 << 32)))
                vr.setIteration(it)
                got = vr.runRaycastNoGL(W, H)
                vr.setIteration(it)
                ref, _, _ = common.oracle_frame(vr, vol, fmt, tff, W, H, use_ess=kw.get("ess", True), in_accum=acc)
                assert np.abs(got - ref).max() <= TOL, it
                acc = ref
    finally:
        _setup(vr, vol, fmt, tff, view)
