#!/usr/bin/env python3
"""tools/raylen.py WORKLOAD -- distribution of samples taken per ray (diagnostic build
-DVR_RAYLEN writes the count into the alpha channel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VRHIP_LIB_PATH", os.path.join(ROOT, "volumerenderercl_amd", "_variants", "libvrhip_raylen.so"))
import numpy as np
import bench
from volumerenderercl_amd import VolumeRenderCL, frontend

wl = sys.argv[1] if len(sys.argv) > 1 else "shells2048"
kind, res, fmt_name, illum, tff_name, ess = bench.WORKLOADS[wl][:6]
vr = VolumeRenderCL(); vr.initialize()
vr.synthVolume(kind, (res, res, res), bench.FMT[fmt_name])
tff = {"default": frontend.tff_from_stops, "haze": frontend.haze_tff, "opaque": frontend.opaque_ramp_tff}[tff_name]()
vr.setTransferFunction(tff); vr.setIllumination(illum); vr.setObjEss(ess)
vr.updateView(frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30.0)))
vr.setSeed(frontend.Mt19937()())
img = vr.runRaycastNoGL(1024, 1024)
n = img[..., 3].astype(np.int64).ravel()
print(wl, "rays", n.size, "total samples", n.sum(), "max", n.max(), "mean", n.mean())
for q in (50, 90, 99, 99.9, 99.99):
    print("  p%-6s %d" % (q, np.percentile(n, q)))
edges = [0, 1, 16, 64, 128, 256, 512, 1024, 2048, 4096, 100000]
h, _ = np.histogram(n, bins=edges)
for i in range(len(h)):
    sel = (n >= edges[i]) & (n < edges[i + 1])
    print("  [%5d,%6d): rays %7d  samples %10d (%.1f %%)" % (edges[i], edges[i + 1], h[i], n[sel].sum(), 100.0 * n[sel].sum() / max(n.sum(), 1)))
# per 8x8 tile maxima (phase-1 waves) 
t = img[..., 3].astype(np.int64).reshape(128, 8, 128, 8).max(axis=(1, 3))
print("  tile max: mean %.1f  p99 %d  max %d" % (t.mean(), np.percentile(t, 99), t.max()))
vr.close()
