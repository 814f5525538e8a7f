#!/usr/bin/env python3
"""tools/split_frame.py -- ONE frame at a time, cut into K interleaved tile subsets rendered by K renderers in
flight on one GPU (shared volume), for phase-1 round budgets: does a frame's latency chain hide behind its
own other half?  One GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from volumerenderercl_amd import VolumeRenderCL, frontend, tiles

V = 1024
dev = torch.device("cuda", 0)
vr = VolumeRenderCL(); vr.initialize()
vr.synthVolume("shells", (2048,) * 3, 0)
vr.setTransferFunction(frontend.tff_from_stops())
vr.updateView(frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30.0)))
mt = frontend.Mt19937()
seeds = [mt() for _ in range(256)]
twins = [vr] + [vr.shareVolumes() for _ in range(3)]
streams = [torch.cuda.Stream(dev) for _ in twins]
for r, s in zip(twins, streams):
    r.set_stream(s.cuda_stream)
for K in (1, 2, 4):
    for budget in (10, 24, 48):
        for r in twins:
            r.setRoundBudget(budget)
        splits = [tiles.TileSplit(V, V, 64, 64, K, j) for j in range(K)]
        outs = [torch.empty((len(splits[j].my_tiles) * 64 * 64, 4), dtype=torch.float32, device=dev) for j in range(K)]
        def frame(seed):
            for j in range(K):
                twins[j].setSeed(seed); twins[j].setIteration(0)
                if K == 1:
                    twins[j].runRaycast(V, V, out_dev_ptr=outs[j].data_ptr())
                else:
                    twins[j].render_tiles(V, V, 64, 64, splits[j].my_tiles, outs[j].data_ptr())
            torch.cuda.synchronize()      # the caller waits for the frame
        for s in seeds[:4]:
            frame(s)
        t0 = time.perf_counter()
        for s in seeds[4:68]:
            frame(s)
        print("K=%d budget %2d: %.3f ms per frame (host clock, synchronised per frame)" % (
            K, budget, (time.perf_counter() - t0) * 1e3 / 64), flush=True)
