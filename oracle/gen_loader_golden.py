#!/usr/bin/env python3
"""Generate tests/golden/loader/: small .dat/.raw inputs plus the outputs of the REFERENCE's
own DatRawReader on them (oracle/_ref/libref_datraw.so, compiled from
/root/reference/src/io/datrawreader.cpp by `make -C oracle ref`).

Runs only where /root/reference exists (this container); the fixtures it writes are data
(inputs + expected outputs) and are committed.  TEST INFRASTRUCTURE ONLY."""
import base64
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "loader")


class Result(C.Structure):
    _fields_ = [("res", C.c_uint * 4), ("thickness", C.c_double * 3), ("format", C.c_int),
                ("endianness", C.c_int), ("min_value", C.c_float), ("max_value", C.c_float),
                ("n_timesteps", C.c_ulonglong), ("bytes_per_timestep", C.c_ulonglong),
                ("channel_order", C.c_char * 16)]


def write(name, text):
    with open(os.path.join(OUT, name), "w") as f:
        f.write(text)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = C.CDLL(os.path.join(HERE, "_ref", "libref_datraw.so"))
    ref.refdr_error.restype = C.c_char_p
    rng = np.random.default_rng(20240917)
    cases = []

    def raw(name, arr):
        arr.tofile(os.path.join(OUT, name))

    # 1 UCHAR with slice thickness
    raw("c1.raw", rng.integers(0, 256, 5 * 4 * 3, dtype=np.uint8))
    write("c1.dat", "ObjectFileName: c1.raw\nResolution: 5 4 3\nSliceThickness: 1.0 1.0 2.5\nFormat: UCHAR\n")
    # 2 USHORT little endian, maximum below 65535 -> stretch
    raw("c2.raw", rng.integers(0, 4096, 4 * 4 * 4).astype("<u2"))
    write("c2.dat", "ObjectFileName:\tc2.raw\nResolution:\t4 4 4\nFormat:\tUSHORT\nSliceThickness: 1 1 1\n")
    # 3 USHORT flagged big endian
    raw("c3.raw", rng.integers(0, 30000, 3 * 4 * 5).astype(">u2"))
    write("c3.dat", "ObjectFileName: c3.raw\nResolution: 3 4 5\nFormat: USHORT\nEndianness: BIG\n")
    # 4 FLOAT little endian
    raw("c4.raw", (rng.random(3 * 3 * 3) * 7.5).astype("<f4"))
    write("c4.dat", "ObjectFileName: c4.raw\nResolution: 3 3 3\nFormat: FLOAT\nSliceThickness: 0.5 0.5 1.5\n")
    # 5 FLOAT big endian
    raw("c5.raw", (rng.random(2 * 3 * 4) * 100.0).astype(">f4"))
    write("c5.dat", "ObjectFileName: c5.raw\nResolution: 2 3 4\nFormat: FLOAT\nEndianness: BIG\n")
    # 6 resolution inferred from the file size (cube)
    raw("c6.raw", rng.integers(0, 256, 64, dtype=np.uint8))
    write("c6.dat", "ObjectFileName: c6.raw\nFormat: UCHAR\n")
    # 7 time series, names listed
    raw("c7_a.raw", rng.integers(0, 256, 27, dtype=np.uint8))
    raw("c7_b.raw", rng.integers(0, 256, 27, dtype=np.uint8))
    write("c7.dat", "ObjectFileName: c7_a.raw c7_b.raw\nResolution: 3 3 3\nFormat: UCHAR\n")
    # 8 time series by count: numbered names generated from the first one
    for i in (1, 2, 3):
        raw("c8.%03d" % i, rng.integers(0, 256, 8, dtype=np.uint8))
    write("c8.dat", "ObjectFileName: c8.001\nResolution: 2 2 2\nFormat: UCHAR\nTimeSeries: 3\n")
    # 9 channel order / object model key, extra unknown keys
    raw("c9.raw", rng.integers(0, 256, 8, dtype=np.uint8))
    write("c9.dat", "ObjectFileName: c9.raw\nResolution: 2 2 2\nFormat: UCHAR\nObjectModel: I\nNodes: n.txt\nFoo: bar\n")
    # 10 missing raw file, 11 missing ObjectFileName
    write("c10.dat", "ObjectFileName: does_not_exist.raw\nResolution: 2 2 2\nFormat: UCHAR\n")
    write("c11.dat", "Resolution: 2 2 2\nFormat: UCHAR\n")
    # 12 format missing (reference assumes UCHAR but never reads the bytes: SURVEY C13)
    raw("c12.raw", rng.integers(1, 256, 8, dtype=np.uint8))
    write("c12.dat", "ObjectFileName: c12.raw\nResolution: 2 2 2\n")

    # 13 time series by count with the digits at the end of the name (the working form; c8's
    #    name has a digit in front, which the reference's expansion trips over)
    for i in (1, 2, 3):
        raw("ts.%03d" % i, rng.integers(0, 256, 8, dtype=np.uint8))
    write("c13.dat", "ObjectFileName: ts.001\nResolution: 2 2 2\nFormat: UCHAR\nTimeSteps: 3\n")

    for name in ["c%d" % i for i in range(1, 14)]:
        res = Result()
        rc = ref.refdr_load(os.path.join(OUT, name + ".dat").encode(), C.byref(res))
        entry = {"case": name, "rc": rc}
        if rc != 0:
            entry["error"] = ref.refdr_error().decode().replace(OUT + "/", "")
        else:
            entry.update(res=list(res.res), thickness=list(res.thickness), format=res.format,
                         endianness=res.endianness, min_value=float(res.min_value),
                         max_value=float(res.max_value), n_timesteps=int(res.n_timesteps),
                         bytes_per_timestep=int(res.bytes_per_timestep),
                         channel_order=res.channel_order.decode(), data=[], histogram=[])
            for t in range(res.n_timesteps):
                buf = (C.c_char * res.bytes_per_timestep)()
                assert ref.refdr_copy_data(t, buf, res.bytes_per_timestep) == 0
                entry["data"].append(base64.b64encode(bytes(buf)).decode())
                h = (C.c_double * 256)()
                assert ref.refdr_copy_histogram(t, h) == 0
                entry["histogram"].append({str(i): v for i, v in enumerate(h) if v})
        cases.append(entry)
    with open(os.path.join(OUT, "expected.json"), "w") as f:
        json.dump({"_source": "outputs of the reference DatRawReader (compiled from "
                              "/root/reference/src/io/datrawreader.cpp) on the inputs in this "
                              "directory; generated by oracle/gen_loader_golden.py",
                   "cases": cases}, f, indent=1)
    print("wrote", len(cases), "cases to", OUT)


if __name__ == "__main__":
    main()
