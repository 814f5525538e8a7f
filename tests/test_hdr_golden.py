"""The host layer's Radiance .hdr decoder (createEnvironmentMap's loader) against golden vectors
produced by the reference's own loader (oracle/gen_hdr_golden.py, reference inc/hdr_loader.h
compiled into oracle/_ref/libref_hdr.so in the build container)."""
import json
import os

import numpy as np
import pytest

from oracle import vro
from volumerenderercl_amd import datraw

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hdr")
with open(os.path.join(GOLD, "index.json")) as f:
    CASES = json.load(f)


@pytest.mark.parametrize("case", CASES, ids=[c["file"] for c in CASES])
def test_hdr_loader_matches_reference(case):
    path = os.path.join(GOLD, case["file"])
    if not case["ok"]:
        with pytest.raises(RuntimeError, match="Error loading environment map file."):
            datraw.load_hdr(path)
        return
    img = datraw.load_hdr(path)
    assert img.shape == (case["height"], case["width"], 4) and img.dtype == np.float32
    want = np.fromfile(path + ".f32", dtype="<f4").reshape(img.shape)
    assert np.array_equal(img, want)
    assert np.all(img[..., 3] == 0.0)


def test_missing_file_raises():
    with pytest.raises(RuntimeError):
        datraw.load_hdr(os.path.join(GOLD, "does_not_exist.hdr"))


def test_environment_coordinates_functions():
    """atan2 / acos behind get_environment_coords (volumeraycast.cl:506-510): the oracle's fixed
    fp32 sequences stay within 2 ulp of the correctly rounded values."""
    L = vro.lib()
    rng = np.random.default_rng(5)
    xy = rng.normal(size=(4000, 2)).astype(np.float32)
    got = np.array([L.vro_atan2f(float(y), float(x)) for x, y in xy], dtype=np.float64)
    want = np.arctan2(xy[:, 1].astype(np.float64), xy[:, 0].astype(np.float64))
    assert np.max(np.abs(got - want)) < 5e-7
    c = np.linspace(-1, 1, 4001).astype(np.float32)
    got = np.array([L.vro_acosf(float(v)) for v in c], dtype=np.float64)
    assert np.max(np.abs(got - np.arccos(c.astype(np.float64)))) < 5e-7
    assert L.vro_atan2f(0.0, 0.0) == 0.0 and L.vro_acosf(1.0) == 0.0
