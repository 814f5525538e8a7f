#!/usr/bin/env python3
"""tools/cells_time.py -- time the one-time builders at 2048^3 (under rocprofv3 --kernel-trace --stats: ESS bricks,
footprint volume, cell grids): the build of both cell grids (cells of 4 and of 8 voxels, bounds, empty bits, per-brick
words) at 2048^3: separable streaming build vs the one-wave-per-cell kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from volumerenderercl_amd import VolumeRenderCL, frontend
for mode in ("stream", "wave"):
    if mode == "wave":
        os.environ["VRHIP_CELLS_PER_WAVE"] = "1"
    else:
        os.environ.pop("VRHIP_CELLS_PER_WAVE", None)
    vr = VolumeRenderCL(); vr.initialize()
    vr.synthVolume("shells", (2048, 2048, 2048), 0)
    vr.setTransferFunction(frontend.tff_from_stops())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = vr.lib.vrhip_download_cells(vr.handle, None, 0, None, None)
    assert rc == 0, vr.lib.vrhip_last_error(vr.handle)
    torch.cuda.synchronize()
    print(mode, "cell grid build incl. bounds + words: %.2f ms (host clock)" % ((time.perf_counter() - t0) * 1e3))
    if mode == "stream":   # one frame: the fine cell grid and the footprint volume are built on first use
        vr.updateView(frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30.0)))
        vr.setSeed(1)
        t0 = time.perf_counter()
        vr.runRaycast(256, 256)
        torch.cuda.synchronize()
        print("first frame incl. footprint volume + fine cell grid: %.2f ms (host clock)" % ((time.perf_counter() - t0) * 1e3))
    vr.close()
