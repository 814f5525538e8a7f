"""The C++ host loader (libvrhost.so) against the outputs of the REFERENCE's own DatRawReader on
the same .dat/.raw files (tests/golden/loader, produced by oracle/gen_loader_golden.py from the
compiled reference): properties, normalised bytes (USHORT stretch, FLOAT / max, endianness),
histograms, time-series name expansion and error behaviour."""
import base64
import json
import os

import numpy as np
import pytest

from volumerenderercl_amd import datraw

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "loader")
CASES = json.load(open(os.path.join(GOLD, "expected.json")))["cases"]


@pytest.mark.parametrize("case", CASES, ids=[c["case"] for c in CASES])
def test_loader_matches_reference(case):
    r = datraw.DatRawReader()
    dat = os.path.join(GOLD, case["case"] + ".dat")
    if case["rc"] != 0:
        with pytest.raises(RuntimeError) as e:
            r.read_files(datraw.Properties(dat))
        assert str(e.value).replace(GOLD + "/", "") == case["error"]
        return
    r.read_files(datraw.Properties(dat))
    p = r.properties()
    if case["case"] == "c12":
        # SURVEY C13 (deliberate fix): with no Format key the reference assumes UCHAR but never
        # reads the bytes (all-zero volume, max_value left at FLT_MIN); we read them as UCHAR.
        assert p.format == datraw.UCHAR and p.volume_res == case["res"]
        raw = np.fromfile(os.path.join(GOLD, "c12.raw"), dtype=np.uint8)
        np.testing.assert_array_equal(r.data()[0], raw)
        assert all(b == 0 for b in base64.b64decode(case["data"][0]))
        return
    assert p.volume_res == case["res"]
    assert p.slice_thickness == case["thickness"]
    assert p.format == case["format"] and p.endianness == case["endianness"]
    assert p.image_channel_order == case["channel_order"]
    assert p.min_value == case["min_value"] and p.max_value == case["max_value"]
    assert len(r.data()) == case["n_timesteps"]
    for t in range(case["n_timesteps"]):
        assert r.data()[t].tobytes() == base64.b64decode(case["data"][t])   # bit-exact
        hist = {str(i): v for i, v in enumerate(r.histograms()[t]) if v}
        assert hist == case["histogram"][t]


def test_empty_name_is_invalid_argument():
    with pytest.raises(ValueError):
        datraw.DatRawReader().read_files(datraw.Properties(""))


def test_renderer_loads_through_the_loader():
    """loadVolumeData wiring (no GPU needed up to the upload): model scale from slice thickness
    as calcScaling does (volumerendercl.cpp:347-362)."""
    from oracle import vro
    r = datraw.DatRawReader()
    r.read_files(datraw.Properties(os.path.join(GOLD, "c4.dat")))
    p = r.properties()
    assert vro.calc_scaling(p.volume_res[:3], p.slice_thickness) == [3.0, 3.0, 1.0]
