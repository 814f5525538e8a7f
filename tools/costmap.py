#!/usr/bin/env python3
"""tools/costmap.py [WORKLOAD] [VIEWPORT] -- one frame at a time: who goes to the 4-lane kernel, and for how long?
Renders a few frames with the single-frame schedule and prints the histogram of the per-pixel cost map (4-lane rounds
of 16 samples per ray), the samples per ray where the VR_RAYLEN variant is selected, and the phase times."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from volumerenderercl_amd import VolumeRenderCL, frontend  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "shells2048"
V = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
kind, res, fmt_name, illum, tff_name, ess = bench.WORKLOADS[wl][:6]
vr = VolumeRenderCL()
vr.initialize()
vr.synthVolume(kind, (res, res, res), bench.FMT[fmt_name])
tff = {"default": frontend.tff_from_stops, "haze": frontend.haze_tff, "opaque": frontend.opaque_ramp_tff}[tff_name]()
vr.setTransferFunction(tff)
vr.setIllumination(illum)
vr.setObjEss(ess)
vr.updateView(frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30.0)))
mt = frontend.Mt19937()
vr.setPhaseTiming(True)
ms, p1, p2 = [], [], []
for k in range(8):
    vr.setSeed(mt())
    vr.setIteration(0)
    vr.runRaycast(V, V)
    if k >= 3:
        ms.append(vr.getLastExecTime() * 1e3)
        a, b = vr.getLastPhaseTimes()
        p1.append(a * 1e3)
        p2.append(b * 1e3)
print("%s %dx%d: frame %.3f ms (pre-pass + phase 1 %.3f, sort + phase 2 %.3f), launch info %s" % (
    wl, V, V, np.mean(ms), np.mean(p1), np.mean(p2), vr.lastLaunchInfo()))
cost = np.zeros(V * V, dtype=np.uint16)
rc = vr.lib.vrhip_download_cost_map(vr.handle, cost.ctypes.data_as(C.c_void_p), cost.size)
assert rc == 0, rc
n = cost.astype(np.int64)
print("pixels with a cost: %d of %d; sum of rounds %d; max %d" % ((n > 0).sum(), n.size, n.sum(), n.max()))
edges = [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 1000]
for i in range(len(edges) - 1):
    sel = (n >= edges[i]) & (n < edges[i + 1])
    print("  rounds [%3d,%4d): rays %7d  ray-rounds %9d (%.1f %%)" % (edges[i], edges[i + 1], sel.sum(), n[sel].sum(),
                                                                 100.0 * n[sel].sum() / max(1, n.sum())))
srt = np.sort(n)[::-1]
for k in (1, 16, 256, 1024, 4096, 16384, 32768, 65536):
    if k <= srt.size:
        print("  the %6d-th longest ray: %d rounds" % (k, srt[k - 1]))
vr.close()
