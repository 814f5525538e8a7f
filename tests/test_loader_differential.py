"""The C++ host loader against the REFERENCE's own DatRawReader, live: random .dat/.raw files read by both
(oracle/_ref/libref_datraw.so, compiled from /root/reference/src/io/datrawreader.cpp by `make -C oracle ref`).
Runs where that library exists (the build container); the committed fixtures of test_loader_golden.py cover the
same reader everywhere else."""
import ctypes as C
import os

import numpy as np
import pytest

from volumerenderercl_amd import datraw

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libref_datraw.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="compiled reference loader not present")


class Result(C.Structure):
    _fields_ = [("res", C.c_uint * 4), ("thickness", C.c_double * 3), ("format", C.c_int),
                ("endianness", C.c_int), ("min_value", C.c_float), ("max_value", C.c_float),
                ("n_timesteps", C.c_ulonglong), ("bytes_per_timestep", C.c_ulonglong),
                ("channel_order", C.c_char * 16)]


def _random_dataset(rng, d):
    """Writes a .dat with its .raw file(s) into d; returns the .dat's path."""
    fmt = ["UCHAR", "USHORT", "FLOAT"][int(rng.integers(3))]
    res = [int(v) for v in rng.integers(1, 9, size=3)]
    n = res[0] * res[1] * res[2]
    big = fmt != "UCHAR" and rng.random() < 0.4
    steps = int(rng.choice([1, 1, 1, 2, 3]))
    sep = str(rng.choice([" ", "\t", "  ", " \t"]))
    names = []
    for t in range(steps):
        name = "v%d_%d.raw" % (int(rng.integers(1000)), t)
        if fmt == "UCHAR":
            a = rng.integers(0, int(rng.choice([2, 17, 256])), n).astype(np.uint8)
        elif fmt == "USHORT":
            a = rng.integers(0, int(rng.choice([2, 300, 4096, 65536])), n).astype(">u2" if big else "<u2")
        else:
            a = (rng.random(n) * float(rng.choice([0.5, 1.0, 7.5, 1000.0]))).astype(">f4" if big else "<f4")
            if rng.random() < 0.2:
                a[:] = 0      # an all-zero field: what does the normalisation divide by?
        a.tofile(os.path.join(d, name))
        names.append(name)
    lines = ["ObjectFileName:%s%s" % (sep, " ".join(names))]
    cube = res[0] == res[1] == res[2] and rng.random() < 0.5 and steps == 1
    if not cube:
        lines.append("Resolution:%s%d %d %d" % (sep, res[0], res[1], res[2]))
    lines.append("Format:%s%s" % (sep, fmt))
    if rng.random() < 0.6:
        th = [float(rng.choice([0.5, 1.0, 1.0, 2.5, 3.0])) for _ in range(3)]
        lines.append("SliceThickness:%s%g %g %g" % (sep, th[0], th[1], th[2]))
    if big or rng.random() < 0.2:
        lines.append("Endianness:%s%s" % (sep, "BIG" if big else "LITTLE"))
    if rng.random() < 0.3:
        lines.append("ObjectModel:%sI" % sep)
    if rng.random() < 0.3:
        lines.append("Unknown:%swhatever 1 2 3" % sep)
    order = rng.permutation(len(lines) - 1) + 1 if rng.random() < 0.5 else np.arange(1, len(lines))
    lines = [lines[0]] + [lines[i] for i in order]
    dat = os.path.join(d, "f.dat")
    with open(dat, "w") as f:
        f.write("\n".join(lines) + ("\n" if rng.random() < 0.8 else ""))
    return dat


@pytest.mark.parametrize("case", range(64))
def test_random_files_read_like_the_reference(case, tmp_path):
    ref = C.CDLL(REF)
    ref.refdr_error.restype = C.c_char_p
    rng = np.random.default_rng(44261004 + case)
    dat = _random_dataset(rng, str(tmp_path))
    want = Result()
    rc = ref.refdr_load(dat.encode(), C.byref(want))
    r = datraw.DatRawReader()
    if rc != 0:
        with pytest.raises(RuntimeError) as e:
            r.read_files(datraw.Properties(dat))
        assert str(e.value) == ref.refdr_error().decode()
        return
    r.read_files(datraw.Properties(dat))
    p = r.properties()
    assert p.volume_res == list(want.res)
    assert p.slice_thickness == list(want.thickness)
    assert p.format == want.format and p.endianness == want.endianness
    assert p.image_channel_order == want.channel_order.decode()
    assert len(r.data()) == want.n_timesteps
    # NaN == NaN here: an all-zero FLOAT field divides 0 by 0 in both readers
    np.testing.assert_array_equal(np.float32([p.min_value, p.max_value]), np.float32([want.min_value, want.max_value]))
    for t in range(want.n_timesteps):
        buf = (C.c_char * want.bytes_per_timestep)()
        assert ref.refdr_copy_data(t, buf, want.bytes_per_timestep) == 0
        assert r.data()[t].tobytes() == bytes(buf), "time step %d" % t   # bit-exact
        h = (C.c_double * 256)()
        assert ref.refdr_copy_histogram(t, h) == 0
        np.testing.assert_array_equal(np.asarray(r.histograms()[t], dtype=np.float64), np.asarray(list(h)))
