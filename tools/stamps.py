#!/usr/bin/env python3
"""tools/stamps.py WORKLOAD -- where do the ray-cast waves spend their cycles?

Runs the diagnostic build of libvrhip (compiled with -DVR_STAMPS, selected through
VRHIP_LIB_PATH) on a bench workload and prints the per-phase share of summed wave time.
Stamps drain the memory queues, so read the SHARES, not the frame time.  Both phases are reported
(VRHIP_ROUND_BUDGET=0 gives the single-phase march)."""
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
          "batch evaluation (fetch, TF, shading, powr)", "empty-run lookahead + skip", "-",
          "composite + transitions", "suspend / frame write", "LDS staging (block start)", "-",
          "rounds (count)", "tiles (count)", "wave lifetime"]


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "shells2048"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    kind, res, fmt_name, illum, tff_name, ess = bench.WORKLOADS[wl][:6]
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
    out = (C.c_uint64 * 32)()
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
    print("%s: %d frames, kernel %.3f ms/frame (stamped build)" % (wl, frames, sum(ms) / len(ms)))
    for name, base in (("phase 1", 0), ("phase 2", 16)):
        o = [int(out[base + i]) for i in range(16)]
        total = float(o[12])
        if total <= 0:
            continue
        print(" %s: summed wave lifetime %.3g cycles" % (name, total))
        acc = 0.0
        for i in range(9):
            acc += o[i]
            if PHASES[i] != "-" and o[i]:
                print("  %-44s %6.2f %%" % (PHASES[i], 100.0 * o[i] / total))
        print("  %-44s %6.2f %%" % ("(unattributed)", 100.0 * (total - acc) / total))
        rounds, tiles = o[10] / frames, o[11] / frames
        print("  DDA iterations/frame %.0f (%.0f cycles each)  rounds/frame %.0f (%.0f cycles each)  "
              "tiles|groups/frame %.0f" % (o[9] / frames, o[2] / max(o[9], 1), rounds,
                                           total / frames / max(rounds, 1), tiles))
    # timeline: when do the waves of each phase end (last frame)?
    import numpy as np
    spans = (C.c_uint64 * (2 * 2 * 8192))()
    f = vr.lib.vrhip_debug_wave_spans
    f.argtypes = [C.POINTER(C.c_uint64)]
    if f(spans) == 0:
        a = np.frombuffer(spans, dtype=np.uint64).reshape(2, 2, 8192).astype(np.int64)
        for ph in (0, 1):
            st, en = a[ph, 0], a[ph, 1]
            ok = en > 0
            if not ok.any():
                continue
            # persistent grid: every wave starts with the launch, so its lifetime is its end time
            life = np.sort((en[ok] - st[ok]) / 1e3)
            print(" phase %d: %d waves; lifetime in kcycles: p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f "
                  "mean %.0f" % (ph + 1, ok.sum(), *(np.percentile(life, q) for q in (10, 50, 90, 99, 100)),
                                 life.mean()))
    vr.close()


if __name__ == "__main__":
    main()
