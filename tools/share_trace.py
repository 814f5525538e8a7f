#!/usr/bin/env python3
"""tools/share_trace.py [WORLD] [VIEWPORT] [RENDERERS] [FRAMES_PER_SET] [SETS] -- rank 0's tile share of a WORLD-rank
split rendered in launch sets (no gather), for a rocprofv3 --kernel-trace --stats run: which launches a
share's time goes to.  WORLD 1 = whole frames."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from volumerenderercl_amd import VolumeRenderCL, frontend, tiles

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
V = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
fif = int(sys.argv[3]) if len(sys.argv) > 3 else 2
fpl = int(sys.argv[4]) if len(sys.argv) > 4 else 32
nsets = int(sys.argv[5]) if len(sys.argv) > 5 else 16
dev = torch.device("cuda", 0)
vr = VolumeRenderCL(); vr.initialize()
vr.synthVolume("shells", (2048,) * 3, 0)
vr.setTransferFunction(frontend.tff_from_stops())
vr.updateView(frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30.0)))
vr.setRoundBudget(int(os.environ.get("BUDGET", "48")))
mt = frontend.Mt19937()
seeds = [mt() for _ in range(8192)]
twins = [vr] + [vr.shareVolumes() for _ in range(fif - 1)]
streams = [torch.cuda.Stream(dev) for _ in twins]
for r, s in zip(twins, streams):
    r.set_stream(s.cuda_stream)
split = tiles.TileSplit(V, V, 64, 64, world, 0)
ids = None if world == 1 else split.my_tiles
npix = V * V if world == 1 else len(ids) * 64 * 64
outs = [torch.empty((fpl, npix, 4), dtype=torch.float32, device=dev) for _ in range(fif)]
def run(n):
    k = 0
    for i in range(n):
        j = i % fif
        sd = seeds[k:k + fpl]; k += fpl
        if ids is None:
            twins[j].render_batch(V, V, sd, outs[j].data_ptr())
        else:
            twins[j].render_batch(V, V, sd, outs[j].data_ptr(), 64, 64, ids, frame_stride=npix)
run(fif); torch.cuda.synchronize()
t0 = time.perf_counter(); run(nsets); torch.cuda.synchronize()
print("world %d share (%d px) %d x %d: %.4f ms/frame, %.3f ms per set" % (
    world, npix, fif, fpl, (time.perf_counter() - t0) * 1e3 / (nsets * fpl), (time.perf_counter() - t0) * 1e3 / nsets))
