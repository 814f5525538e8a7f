#!/usr/bin/env python3
"""tools/assemble_time.py [VIEWPORT] [FRAMES] [TILE] -- rank 0's side of an 8-rank sparse gather, rehearsed on one GPU:
the eight ranks' shares of FRAMES frames rendered one after the other, packed like TileDriver packs them, and the
batch assembled on "rank 0" by the fused kernel (vrhip_assemble_batch) and by the torch path (expand into the dense
staging + index_select + strided copy) -- bit-equal, each timed.  Prints the bytes a dense and the sparse gather
would move to rank 0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from volumerenderercl_amd import VolumeRenderCL, frontend, tiles

V = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
WORLD = 8
dev = torch.device("cuda", 0)
vr = VolumeRenderCL(); vr.initialize()
vr.synthVolume("shells", (2048,) * 3, 0)
vr.setTransferFunction(frontend.tff_from_stops())
vr.updateView(frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30.0)))
vr.setRoundBudget(48)
vr.set_stream(torch.cuda.current_stream().cuda_stream)
mt = frontend.Mt19937()
seeds = [mt() for _ in range(F)]


class Hub:            # one process plays all ranks: all_gather / gather over tensors the "ranks" deposit
    def __init__(self):
        self.counts, self.msgs = {}, {}

    class Done:
        def wait(self):
            pass

    def for_rank(self, rank):
        hub = self

        class D:
            def all_gather(self, out_list, t, async_op=False):
                hub.counts[rank] = t.clone()
                if len(hub.counts) == WORLD:
                    for r in range(WORLD):
                        out_list[r].copy_(hub.counts[r])
                self_out = out_list
                hub.last_out = getattr(hub, "last_out", {})
                hub.last_out[rank] = out_list
                return hub.Done()

            def gather(self, t, gather_list, dst=0, async_op=False):
                hub.msgs[rank] = t
                if gather_list is not None:
                    for r in range(WORLD):
                        gather_list[r].copy_(hub.msgs[r])
                return hub.Done()
        return D()


hub = Hub()
splits = [tiles.TileSplit(V, V, T, T, WORLD, k) for k in range(WORLD)]
drivers = [tiles.TileDriver(vr, splits[k], dev, dist=hub.for_rank(k), batch=F, sparse=True) for k in range(WORLD)]
order = list(range(1, WORLD)) + [0]
# phase A: every rank renders + packs (counts deposited); the last one sees all counts
for k in order:
    d = drivers[k]
    b = d.next_buf; d.next_buf ^= 1
    vr.render_batch(V, V, seeds, d.local[b].data_ptr(), T, T, splits[k].my_tiles, frame_stride=splits[k].cap * T * T)
    d._start_gather(b, F)
for k in order:      # every rank's counts list filled from the hub
    for r in range(WORLD):
        hub.last_out[k][r].copy_(hub.counts[r])
for k in order:      # payload gathers: peers deposit, rank 0 collects
    drivers[k]._issue_payloads()
e = drivers[0].pending[0]
torch.cuda.synchronize()
whole = sum(e["cs"])
S, P = F * splits[0].cap, T * T
print("viewport %d, %d frames, tile %d: whole tiles %d of %d (%.3f); per frame to rank 0: dense %.2f MB, sparse %.2f MB"
      % (V, F, T, whole, F * splits[0].n_tiles, whole / (F * splits[0].n_tiles), 16 * V * V * 7 / 8 / 1e6,
         4 * e["msg"].numel() * 7 / F / 1e6))
frames_a = torch.zeros((F, V, V, 4), dtype=torch.float32, device=dev)
frames_b = torch.zeros((F, V, V, 4), dtype=torch.float32, device=dev)
d0 = drivers[0]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    assert d0._assemble_fused(e, frames_a)
    torch.cuda.synchronize(); ta = time.perf_counter() - t0
    t0 = time.perf_counter()
    d0._unpack(e)
    rows = d0.staging[e["b"]].view(WORLD * F * splits[0].cap, T, T, 4)
    tv = rows.index_select(0, d0.perm[:F].reshape(-1)).view(F, splits[0].tiles_y, splits[0].tiles_x, T, T, 4).permute(0, 1, 3, 2, 4, 5)
    frames_b.view(F, splits[0].tiles_y, T, splits[0].tiles_x, T, 4).copy_(tv)
    torch.cuda.synchronize(); tb = time.perf_counter() - t0
assert torch.equal(frames_a, frames_b)
vr.setSeed(seeds[3]); vr.setIteration(0)
full = vr.runRaycastNoGL(V, V)
assert np.array_equal(frames_a[3].cpu().numpy(), full)
print("assembly per frame on rank 0: fused kernel %.4f ms, torch path %.4f ms (bit-equal, frame 3 equals the full-frame render)"
      % (ta * 1e3 / F, tb * 1e3 / F))
