#!/usr/bin/env python3
"""tools/march_stats.py [WORKLOAD] -- round statistics of the march kernel from a -DVR_MARCH_STATS
build (tools/mkvariant.sh mstats -DVR_MARCH_STATS; run on the GPU box): one frame, one renderer."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VRHIP_LIB_PATH"] = os.path.join(ROOT, "volumerenderercl_amd", "_variants", "libvrhip_mstats.so")
import torch  # noqa: F401,E402
import bench  # noqa: E402
from volumerenderercl_amd import VolumeRenderCL, frontend  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "shells2048"
kind, res, fmt_name, illum, tff_name, ess = bench.WORKLOADS[wl][:6]
vr = VolumeRenderCL()
vr.initialize()
vr.synthVolume(kind, (res, res, res), bench.FMT[fmt_name])
vr.setTransferFunction({"default": frontend.tff_from_stops, "haze": frontend.haze_tff}[tff_name]())
vr.setIllumination(illum)
vr.updateView(frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30.0)))
mt = frontend.Mt19937()
out = (C.c_ulonglong * 32)()
f = vr.lib.vrhip_debug_march_stats
f.argtypes = [C.c_void_p, C.c_int]
if os.environ.get("VRHIP_ROUND_BUDGET_STATS"):
    vr.setRoundBudget(int(os.environ["VRHIP_ROUND_BUDGET_STATS"]))
for k in range(3):
    vr.setSeed(mt())
    vr.setIteration(0)
    vr.runRaycast(1024, 1024)
    f(out, 1)
names = ["rounds", "live lanes x rounds", "A iterations", "  with DDA step", "  lanes in DDA", "  with sample step",
         "  lanes stepping", "samples queued", "B1 passes", "samples evaluated in vain", "opaque samples", "B2 passes"]
v = list(out)
for n, x in zip(names, v):
    print("%-28s %12d" % (n, x))
print("live lanes per round %.1f; DDA lanes per DDA exec %.1f; stepping lanes per exec %.1f; samples per B1 pass %.1f, per B2 pass %.1f" % (
    v[1] / max(v[0], 1), v[4] / max(v[3], 1), v[6] / max(v[5], 1), v[7] / max(v[8], 1), v[10] / max(v[11], 1)))

if any(v[16:]):
    names1 = ["rounds", "live lanes x rounds", "DDA step executions", "  lanes in them", "lookahead executions", "  lanes in them",
              "evaluation batches", "  lanes in them", "refills", "valid samples evaluated"]
    print("-- phase 1 on the ray list (vr_raycast_rays_kernel)")
    for n, x in zip(names1, v[16:]):
        print("%-28s %12d" % (n, x))

if any(v[28:]):
    print("-- pre-pass: patches that walk %d, DDA step executions %d (%.1f per patch), lanes in them %d (%.1f per execution), valid rays in those patches %d" % (
        v[28], v[29], v[29] / max(v[28], 1), v[30], v[30] / max(v[29], 1), v[31]))
