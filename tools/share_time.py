#!/usr/bin/env python3
"""tools/share_time.py [WORLD] [VIEWPORT] [RANKS] [OUT.json] -- GPU time per frame of a rank's tile share of a
WORLD-rank split (no gather; RANKS = comma-separated ranks, default 0), for several (renderers in flight,
frames per launch set): what image-tile strong scaling can reach before the collective.  One GPU.
TOTAL=K: the SHORT run instead of the steady state -- exactly K frames (the round driver's --steps 20), as
(2 renderers x K/2) and as (1 renderer x K) frames per launch set, under the round budgets BUDGETS=16,24,32,48:
the time from the first launch to the last frame, best of three, untimed warm-up sets of other seeds before."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from volumerenderercl_amd import VolumeRenderCL, frontend, tiles

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
V = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ranks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
out_json = sys.argv[4] if len(sys.argv) > 4 else None
results = []
T = int(os.environ.get("TILE", "64"))
only = os.environ.get("ONLY")   # e.g. "2x128": one schedule only
dev = torch.device("cuda", 0)
vr = VolumeRenderCL(); vr.initialize()
vr.synthVolume("shells", (2048,) * 3, 0)
vr.setTransferFunction(frontend.haze_tff() if os.environ.get("TFF") == "haze" else frontend.tff_from_stops())
vr.updateView(frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30.0)))
vr.setRoundBudget(int(os.environ.get("BUDGET", "48")))
mt = frontend.Mt19937()
seeds = [mt() for _ in range(16384)]
twins = [vr] + [vr.shareVolumes() for _ in range(3)]
streams = [torch.cuda.Stream(dev) for _ in twins]
for r, s in zip(twins, streams):
    r.set_stream(s.cuda_stream)
total = int(os.environ.get("TOTAL", "0"))
budgets = [int(x) for x in os.environ.get("BUDGETS", "16,24,32,48").split(",")]
for w, rk in [(1, 0)] + [(world, r_) for r_ in ranks]:
    split = tiles.TileSplit(V, V, T, T, w, rk)
    ids = None if w == 1 else split.my_tiles
    npix = V * V if w == 1 else len(ids) * T * T
    if total:      # the short run: K frames as 2 x K/2 and as 1 x K, per round budget
        for fif, fpl in ((2, (total + 1) // 2), (1, total)):
            outs = [torch.empty((fpl, npix, 4), dtype=torch.float32, device=dev) for _ in range(fif)]
            for budget in budgets:
                for r in twins:
                    r.setRoundBudget(budget)
                    r.setFrameTiming(False)

                def run(k0):
                    k = k0
                    for j in range(fif):
                        sd = seeds[k:k + fpl]; k += fpl
                        if ids is None:
                            twins[j].render_batch(V, V, sd, outs[j].data_ptr())
                        else:
                            twins[j].render_batch(V, V, sd, outs[j].data_ptr(), T, T, ids, frame_stride=npix)
                best = 1e9
                for rep in range(4):
                    run(4096 + rep * 64); torch.cuda.synchronize()          # warm-up sets of other seeds
                    t0 = time.perf_counter(); run(0); torch.cuda.synchronize()
                    if rep:
                        best = min(best, (time.perf_counter() - t0) * 1e3 / (fif * fpl))
                print("world %d rank %d share (%d px)  %d frames as %d renderer(s) x %2d frames/set, budget %2d: %.4f ms/frame"
                      % (w, rk, npix, fif * fpl, fif, fpl, budget, best), flush=True)
                results.append({"world": w, "rank": rk, "pixels": npix, "renderers": fif, "frames_per_set": fpl,
                                "total_frames": fif * fpl, "round_budget": budget, "ms_per_frame": best})
        continue
    cfgs = [(1, 1), (2, 8), (2, 16), (2, 32), (3, 16), (4, 8), (1, 32), (4, 16)]
    if w > 1:
        cfgs += [(2, 64), (2, 128), (1, 256), (2, 256)]
    if only:
        cfgs = [tuple(int(x) for x in only.split("x"))]
    for fif, fpl in cfgs:
        outs = [torch.empty((fpl, npix, 4), dtype=torch.float32, device=dev) for _ in range(fif)]
        def run(nsets):
            k = 0
            for i in range(nsets):
                j = i % fif
                sd = seeds[k:k + fpl]; k += fpl
                if ids is None:
                    twins[j].render_batch(V, V, sd, outs[j].data_ptr())
                else:
                    twins[j].render_batch(V, V, sd, outs[j].data_ptr(), T, T, ids, frame_stride=npix)
        run(fif); torch.cuda.synchronize()
        nsets = max(fif * 2, 512 // fpl)
        t0 = time.perf_counter(); run(nsets); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / (nsets * fpl)
        print("world %d rank %d share (%d px)  %d renderer(s) x %2d frames/set: %.4f ms/frame" % (w, rk, npix, fif, fpl, ms), flush=True)
        results.append({"world": w, "rank": rk, "pixels": npix, "renderers": fif, "frames_per_set": fpl, "ms_per_frame": ms})
if out_json:
    import json, subprocess
    head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    json.dump({"tool": "tools/share_time.py", "workload": "shells2048", "viewport": V, "world": world, "tile": T, "round_budget": budgets if total else 48, "short_run_frames": total or None,
               "head": head, "results": results}, open(out_json, "w"), indent=1)
