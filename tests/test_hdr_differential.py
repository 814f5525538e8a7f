"""The host layer's Radiance .hdr decoder against the REFERENCE's own loader, live: random files -- flat and
run-length coded scanlines, header variants, truncated bodies -- decoded by both (oracle/_ref/libref_hdr.so =
/root/reference/inc/hdr_loader.h behind a harness, `make -C oracle ref`).  Runs where that library exists (the
build container); the committed fixtures of test_hdr_golden.py cover the same decoder everywhere else."""
import ctypes as C
import os

import numpy as np
import pytest

from volumerenderercl_amd import datraw

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libref_hdr.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="compiled reference .hdr loader not present")


def _rle_row(row, rng):
    """One run-length coded scanline (new RLE): marker 2 2 hi lo, then the 4 component planes as run / literal
    packets whose lengths a coin decides (any valid packet sequence decodes to the same row)."""
    w = row.shape[0]
    out = bytearray([2, 2, w >> 8, w & 255])
    for c in range(4):
        plane = row[:, c]
        i = 0
        while i < w:
            run = 1
            while i + run < w and run < 127 and plane[i + run] == plane[i]:
                run += 1
            if run >= 2 and rng.random() < 0.7:
                run = int(rng.integers(2, run + 1)) if run > 2 else run
                out += bytes([128 + run, int(plane[i])])
                i += run
            else:
                n = int(rng.integers(1, min(128, w - i) + 1))
                out += bytes([n]) + bytes(int(v) for v in plane[i:i + n])
                i += n
    return bytes(out)


@pytest.mark.parametrize("case", range(48))
def test_random_hdr_files_decode_like_the_reference(case, tmp_path):
    ref = C.CDLL(REF)
    ref.refhdr_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    ref.refhdr_free.argtypes = [C.POINTER(C.c_float)]
    rng = np.random.default_rng(66261004 + case)
    w, h = int(rng.integers(1, 70)), int(rng.integers(1, 12))
    px = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    px[..., 3] = rng.integers(100, 150, (h, w))
    if rng.random() < 0.6:   # long runs, so that run packets appear
        rep = int(rng.integers(2, 9))
        px = np.repeat(px[:, ::rep], rep, axis=1)[:, :w]
    if rng.random() < 0.5:
        px[int(rng.integers(h)), int(rng.integers(w)), 3] = 0   # exponent 0 = black
    rle = 8 <= w < 32768 and rng.random() < 0.6
    if not rle and w >= 8:
        px[:, 0, 0] = 7        # a flat scanline must not start like the run-length marker
    body = b"".join(_rle_row(r, rng) for r in px) if rle else px.tobytes()
    magic = str(rng.choice(["#?RADIANCE", "#?RGBE"]))
    header = magic + "\n"
    if rng.random() < 0.5:
        header += "# a comment\n"
    header += "FORMAT=32-bit_rle_%s\n" % str(rng.choice(["rgbe", "xyze"]))
    if rng.random() < 0.4:
        header += "EXPOSURE=%g\n" % float(rng.choice([0.5, 1.0, 2.0]))
    header += "\n-Y %d +X %d\n" % (h, w)
    if rng.random() < 0.15:
        body = body[:int(len(body) * rng.uniform(0.2, 0.9))]   # truncated file
    path = str(tmp_path / "r.hdr")
    with open(path, "wb") as f:
        f.write(header.encode() + body)
    pix = C.POINTER(C.c_float)()
    rw, rh = C.c_uint(), C.c_uint()
    ok = ref.refhdr_load(path.encode(), C.byref(pix), C.byref(rw), C.byref(rh))
    if not ok:
        with pytest.raises(RuntimeError, match="Error loading environment map file."):
            datraw.load_hdr(path)
        return
    want = np.ctypeslib.as_array(pix, shape=(rh.value, rw.value, 4)).copy()
    ref.refhdr_free(pix)
    got = datraw.load_hdr(path)
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.array_equal(got, want)
