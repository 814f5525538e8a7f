#!/usr/bin/env python3
"""tools/pt_stamps.py [WORKLOAD] [PASSES] -- where do the path tracer's waves spend their time?  Runs the diagnostic build
(-DVR_STAMPS: tools/mkvariant.sh stamps -DVR_STAMPS, selected through VRHIP_LIB_PATH) and prints the stages' shares of
the summed wave lifetime, the calls per wave and the drain (wave time after the queue was empty)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VRHIP_LIB_PATH", os.path.join(ROOT, "volumerenderercl_amd", "_variants", "libvrhip_stamps.so"))
import bench  # noqa: E402
from volumerenderercl_amd import VolumeRenderCL, frontend  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "pt1024f_sphere"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kind, res, fmt_name = bench.WORKLOADS[wl][:3]
vr = VolumeRenderCL()
vr.initialize()
vr.synthVolume(kind, (res, res, res), bench.FMT[fmt_name])
vr.setTransferFunction(frontend.tff_from_stops())
vr.setTechnique(1)
vr.setExtinction(100.0)
vr.updateView(frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30.0)))
dbg = vr.lib.vrhip_debug_pt_stamps
dbg.argtypes = [C.POINTER(C.c_uint64), C.c_int]
out = (C.c_uint64 * 16)()
mt = frontend.Mt19937()
vr.setSeed(mt()); vr.setIteration(0); vr.runRaycast(1024, 1024)
dbg(out, 1)
ms = []
for i in range(passes):
    vr.setSeed(mt()); vr.setIteration(0); vr.runRaycast(1024, 1024)
    ms.append(vr.getLastExecTime() * 1e3)
dbg(out, 0)
o = [int(v) for v in out]
waves = o[14]
life = float(o[11])
print("%s: %d passes, %.3f ms per pass (stamped build), %d waves per pass" % (wl, passes, sum(ms) / len(ms), waves // passes))
names = ["stage 1: pixels to idle lanes, ray set-up", "stage 2: positions, cells, bound loads", "stage 2: fetch + transfer function",
         "stage 2: exit conditions", "stage 2: leap", "stage 3: walk ends (shading, next walk, write)", "loop control"]
for i, n in enumerate(names):
    print("  %-48s %5.1f %%" % (n, 100.0 * o[i] / life))
print("  per wave: %.1f refills, %.1f rounds, %.1f walk-end stages; lifetime mean %.0f kcycles, longest %.0f kcycles; "
      "after the queue was empty: %.1f %% of the wave time" % (o[8] / waves, o[9] / waves, o[10] / waves, life / waves / 1e3,
                                                               o[13] / 1e3, 100.0 * o[12] / life))
print("  cycles per round %.0f, per refill %.0f, per walk-end stage %.0f" % (
    (o[1] + o[2] + o[3] + o[4]) / max(o[9], 1), o[0] / max(o[8], 1), o[5] / max(o[10], 1)))
vr.close()
