#!/usr/bin/env python3
"""Generate tests/golden/hdr/: small Radiance .hdr inputs (written here, flat and run-length
coded scanlines, the header variants the loader distinguishes) plus what the REFERENCE's own
loader (oracle/_ref/libref_hdr.so = /root/reference/inc/hdr_loader.h behind a harness, built by
`make -C oracle ref`) returns for them.

Runs only where /root/reference exists (this container); the fixtures it writes are data (inputs
+ expected outputs) and are committed.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "hdr")


def rle_row(row):
    """One run-length coded scanline (new RLE): marker 2 2 hi lo, then the 4 component planes."""
    w = row.shape[0]
    out = bytearray([2, 2, w >> 8, w & 255])
    for c in range(4):
        plane = row[:, c]
        i = 0
        while i < w:
            run = 1
            while i + run < w and run < 127 and plane[i + run] == plane[i]:
                run += 1
            if run >= 3:
                out += bytes([128 + run, int(plane[i])])
                i += run
            else:
                j = i
                lit = bytearray()
                while j < w and len(lit) < 128:
                    r = 1
                    while j + r < w and r < 3 and plane[j + r] == plane[j]:
                        r += 1
                    if r >= 3:
                        break
                    lit.append(int(plane[j]))
                    j += 1
                out += bytes([len(lit)]) + bytes(lit)
                i = j
    return bytes(out)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = C.CDLL(os.path.join(HERE, "_ref", "libref_hdr.so"))
    ref.refhdr_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)),
                                C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    rng = np.random.default_rng(20261003)
    cases = []

    def image(w, h, smooth):
        px = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        px[..., 3] = rng.integers(118, 140, (h, w))
        if smooth:   # long runs, so that the coder emits run packets
            px = np.repeat(px[:, ::8], 8, axis=1)[:, :w]
        px[0, 0, 3] = 0          # exponent 0 = black
        return px

    def emit(name, header, body, expect_ok=True):
        with open(os.path.join(OUT, name), "wb") as f:
            f.write(header.encode() + body)
        cases.append((name, expect_ok))

    # 1 run-length coded, usual header
    px = image(24, 6, True)
    emit("rle.hdr", "#?RADIANCE\n# comment\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=2.0\n\n-Y 6 +X 24\n",
         b"".join(rle_row(r) for r in px))
    # 2 flat scanlines although wide enough for RLE (first byte != 2)
    px = image(16, 5, False)
    px[:, 0, 0] = 7
    emit("flat.hdr", "#?RGBE\nFORMAT=32-bit_rle_rgbe\nGAMMA=1.0\n\n-Y 5 +X 16\n", px.tobytes())
    # 3 narrow image (< 8 pixels): never RLE; XYZE format flag is accepted
    px = image(5, 4, False)
    emit("narrow.hdr", "#?RADIANCE\nFORMAT=32-bit_rle_xyze\n\n-Y 4 +X 5\n", px.tobytes())
    # 4 flat scanline that starts with a 2 but is not an RLE marker (second byte != 2)
    px = image(12, 3, False)
    px[:, 0, 0] = 2
    px[:, 0, 1] = 9
    emit("flat2.hdr", "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 3 +X 12\n", px.tobytes())
    # 5 mixed: RLE rows and flat rows in one file; resolution line with the axes swapped in order
    px = image(32, 4, True)
    rows = [rle_row(px[0]), None, rle_row(px[2]), None]
    px[1, 0, 0] = 5
    px[3, 0, 0] = 200
    rows[1], rows[3] = px[1].tobytes(), px[3].tobytes()
    emit("mixed.hdr", "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n+X 32 -Y 4\n", b"".join(rows))
    # 6-8 rejected files: unknown format, truncated pixels, RLE length mismatch
    emit("badformat.hdr", "#?RADIANCE\nFORMAT=32-bit_float\n\n-Y 2 +X 8\n", bytes(64), False)
    emit("truncated.hdr", "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 4 +X 8\n", bytes([9] * 40), False)
    emit("badlen.hdr", "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 16\n",
         bytes([2, 2, 0, 17]) + bytes([128 + 17, 3] * 4), False)

    index = []
    for name, expect_ok in cases:
        p = C.POINTER(C.c_float)()
        w, h = C.c_uint(0), C.c_uint(0)
        ok = ref.refhdr_load(os.path.join(OUT, name).encode(), C.byref(p), C.byref(w), C.byref(h))
        assert bool(ok) == expect_ok, (name, ok)
        entry = {"file": name, "ok": bool(ok)}
        if ok:
            arr = np.ctypeslib.as_array(p, shape=(h.value * w.value * 4,)).copy()
            ref.refhdr_free(p)
            arr.astype("<f4").tofile(os.path.join(OUT, name + ".f32"))
            entry.update(width=w.value, height=h.value)
        index.append(entry)
    with open(os.path.join(OUT, "index.json"), "w") as f:
        json.dump(index, f, indent=1)
    print("wrote", len(index), "cases to", OUT)


if __name__ == "__main__":
    main()
