#!/usr/bin/env python3
"""tools/stamps.py WORKLOAD -- where do the ray-cast waves spend their cycles?

Runs the diagnostic build of libvrhip (compiled with -DVR_STAMPS, selected through
VRHIP_LIB_PATH) on a bench workload and prints the per-phase share of summed wave time.
Stamps drain the memory queues, so read the SHARES, not the frame time.  Covers phase 1 only
(run with VRHIP_ROUND_BUDGET=0 to see the single-phase march)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VRHIP_LIB_PATH",
                      os.path.join(ROOT, "volumerenderercl_amd", "_variants", "libvrhip_stamps.so"))

import bench  # noqa: E402
from volumerenderercl_amd import VolumeRenderCL, frontend  # noqa: E402

PHASES = ["queue pop + tile load", "ray set-up", "DDA (brick steps)",
          "batch evaluation (fetch, TF, shading, powr)", "-", "-",
          "composite + transitions", "suspend / frame write", "LDS staging (block start)", "-",
          "rounds (count)", "tiles (count)", "wave lifetime"]


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "shells2048"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    kind, res, fmt_name, illum, tff_name, ess = bench.WORKLOADS[wl]
    vr = VolumeRenderCL()
    vr.initialize()
    vr.synthVolume(kind, (res, res, res), bench.FMT[fmt_name])
    tff = {"default": frontend.tff_from_stops, "haze": frontend.haze_tff,
           "opaque": frontend.opaque_ramp_tff}[tff_name]()
    vr.setTransferFunction(tff)
    vr.setIllumination(illum)
    vr.setObjEss(ess)
    vr.updateView(frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30.0)))
    mt = frontend.Mt19937()
    dbg = vr.lib.vrhip_debug_stamps
    dbg.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    out = (C.c_uint64 * 16)()
    vr.setSeed(mt())
    vr.runRaycast(1024, 1024)
    dbg(out, 1)
    ms = []
    for _ in range(frames):
        vr.setSeed(mt())
        vr.setIteration(0)
        vr.runRaycast(1024, 1024)
        ms.append(vr.getLastExecTime() * 1e3)
    dbg(out, 0)
    total = float(out[12])
    print("%s: %d frames, kernel %.3f ms/frame (stamped build), summed wave lifetime %.3g cycles"
          % (wl, frames, sum(ms) / len(ms), total))
    acc = 0.0
    for i in range(9):
        acc += out[i]
        if PHASES[i] != "-":
            print("  %-44s %6.2f %%" % (PHASES[i], 100.0 * out[i] / total))
    print("  %-34s %6.2f %%" % ("(unattributed)", 100.0 * (total - acc) / total))
    print("  DDA loop iterations/frame %.0f -> cycles per DDA iteration %.0f" % (
        out[9] / frames, out[2] / max(out[9], 1)))
    rounds, tiles = out[10] / frames, out[11] / frames
    print("  rounds/frame %.0f  tiles/frame %.0f  cycles/round %.0f  cycles/tile %.0f" % (
        rounds, tiles, total / frames / max(rounds, 1), total / frames / max(tiles, 1)))
    vr.close()


if __name__ == "__main__":
    main()
