"""Multi-GPU path on CPU: world_size-2 (and 3) `gloo` runs of the tile decomposition +
gather + reassembly (volumerenderercl_amd/tiles.py), with the oracle standing in for the
per-tile renderer.  The assembled frame must equal the single-rank frame exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vro
from volumerenderercl_amd import frontend, tiles


def _scene():
    vol = vro.synth_volume("sphere", [32, 32, 32], vro.UCHAR)
    cam = vro.CameraParams()
    cam.viewMat[:] = frontend.view_matrix(frontend.quat_from_axis_angle((1, 1, 0), 30))
    cam.bbox_bl[:] = [-1, -1, -1, 0]
    cam.bbox_tr[:] = [1, 1, 1, 0]
    rp = vro.RenderingParams()
    rp.backgroundColor[:] = [1, 1, 1, 1]
    rp.modelScale[:] = [1, 1, 1, 0]
    rp.illumType, rp.useLinear, rp.seed = 1, 1, 581869302
    rc = vro.RaycastParams()
    rc.samplingRate = 1.5
    _, brf, _ = vro.brick_layout([32, 32, 32])
    rc.brickRes[:] = brf + [0]
    return vol, frontend.tff_from_stops(), cam, rp, rc


BATCH_SEEDS = [11, 22222, 3333333, 44, 555]


def _worker(rank, world, port, W, H, T, q, sparse=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vol, tff, cam, rp, rc = _scene()
    split = tiles.TileSplit(W, H, T, T, world, rank)

    def render_tiles(ids, out, seed=None):
        if seed is not None:
            rp.seed = seed
        for k, t in enumerate(ids):
            x0, y0, w, h = split.tile_rect(t)
            img, _, _ = vro.render_tile(vol, vro.UCHAR, tff, cam, rp, rc, W=W, H=H,
                                        tile=(x0, y0, w, h), threads=1)
            out[k, :h, :w] = torch.from_numpy(img)

    drv = tiles.TileDriver(None, split, torch.device("cpu"), render_tiles_fn=render_tiles, dist=dist, sparse=sparse)
    frame = torch.zeros((H, W, 4)) if rank == 0 else None
    for _ in range(2):   # two frames: buffers are reusable
        out = drv.render_frame(frame)
    first = out.numpy().copy() if rank == 0 else None
    # pipelined: one frame in flight while the next is rendered
    outs = []
    drv.submit()
    for k in range(3):
        if k < 2:
            drv.submit()
        o = drv.collect(frame)
        if rank == 0:
            outs.append(o.numpy().copy())
    # batched: up to 3 independent frames (their own jitter seeds) per gather, one gather in flight
    drvb = tiles.TileDriver(None, split, torch.device("cpu"), render_tiles_fn=render_tiles, dist=dist,
                            batch=3, sparse=sparse)
    frames = torch.zeros((3, H, W, 4)) if rank == 0 else None
    batched = []

    def before(chunk):
        def set_seed(i):
            rp.seed = BATCH_SEEDS[chunk[i]]
        return set_seed

    drvb.submit_batch(3, before([0, 1, 2]))
    drvb.submit_batch(2, before([3, 4]))
    for n in (3, 2):
        o = drvb.collect_batch(frames)
        if rank == 0:
            batched += [o[i].numpy().copy() for i in range(n)]
    # the same frames through submit_frames (seeds handed to the renderer, one gather per call)
    drvf = tiles.TileDriver(None, split, torch.device("cpu"), render_tiles_fn=render_tiles, dist=dist,
                            batch=5, sparse=sparse)
    drvf.submit_frames(BATCH_SEEDS)
    o = drvf.collect_batch(torch.zeros((5, H, W, 4)) if rank == 0 else None)
    if rank == 0:
        for i in range(5):
            np.testing.assert_array_equal(o[i].numpy(), batched[i])
    if rank == 0:
        for o in outs:
            np.testing.assert_array_equal(o, first)
        q.put((first, batched, dict(drvf.gather_stats)))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,W,H,T,sparse", [(2, 96, 64, 32, False), (3, 80, 56, 16, False),
                                                 (2, 96, 64, 32, True), (3, 160, 112, 16, True)])
def test_gloo_tile_gather_matches_single_rank(world, W, H, T, sparse):
    """Tiles over `world` ranks, gathered densely or with uniform tiles travelling as one pixel (sparse):
    synchronous, pipelined and batched, every assembled frame equal to the single-rank render."""
    vol, tff, cam, rp, rc = _scene()
    ref, _, _ = vro.render_tile(vol, vro.UCHAR, tff, cam, rp, rc, W=W, H=H)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, T, q, sparse)) for r in range(world)]
    for p in procs:
        p.start()
    got, batched, stats = q.get(timeout=240)
    if sparse:   # the corners of the frame lie outside the volume's silhouette: whole tiles of one colour
        assert stats["batches"] == 1 and stats["sent_bytes"] > 0, stats
        if T == 16:   # (six 32 x 32 tiles all touch the silhouette; 16 x 16 tiles leave whole ones outside)
            assert stats["sent_bytes"] < 0.8 * stats["dense_bytes"], stats
    else:
        assert stats["batches"] == 0
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    np.testing.assert_array_equal(got, ref)
    assert len(batched) == len(BATCH_SEEDS)
    for seed, frame in zip(BATCH_SEEDS, batched):
        rp.seed = seed
        want, _, _ = vro.render_tile(vol, vro.UCHAR, tff, cam, rp, rc, W=W, H=H)
        np.testing.assert_array_equal(frame, want)


def _worker_img_ess(rank, world, port, W, H, T, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vol, tff, cam, rp, rc = _scene()
    vol = np.zeros_like(vol)
    vol[10:20, 12:22, 8:18] = 200          # small object: most groups end up skipped
    rp.imgEss = 1
    split = tiles.TileSplit(W, H, T, T, world, rank)
    state = {"hin": None, "hout": None}
    state["hin"], state["hout"] = vro.hit_image_init(W, H)

    def render_tiles(ids, out):
        for k, t in enumerate(ids):
            x0, y0, w, h = split.tile_rect(t)
            img, _, _ = vro.render_tile(vol, vro.UCHAR, tff, cam, rp, rc, W=W, H=H,
                                        tile=(x0, y0, w, h), threads=1, hit_in=state["hin"],
                                        hit_out=state["hout"])
            out[k, :h, :w] = torch.from_numpy(img)
        state["hin"], state["hout"] = state["hout"], state["hin"]   # the per-frame swap

    def put(h):
        state["hin"] = np.ascontiguousarray(h)

    drv = tiles.TileDriver(None, split, torch.device("cpu"), render_tiles_fn=render_tiles, dist=dist,
                           hit_io=(lambda: state["hin"], put))
    frame = torch.zeros((H, W, 4)) if rank == 0 else None
    frames, hits = [], []
    for _ in range(4):
        out = drv.render_frame(frame)
        if rank == 0:
            frames.append(out.numpy().copy())
            hits.append(state["hin"].copy())
    if rank == 0:
        q.put((frames, hits))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_image_order_ess_matches_single_rank():
    """imgEss across ranks: the hit image merged after every frame makes the 2-rank sequence
    equal to the single-rank one, frame by frame and texel by texel."""
    W, H, T = 112, 80, 16
    vol, tff, cam, rp, rc = _scene()
    vol = np.zeros_like(vol)
    vol[10:20, 12:22, 8:18] = 200
    rp.imgEss = 1
    hin, hout = vro.hit_image_init(W, H)
    want_frames, want_hits = [], []
    for _ in range(4):
        img, _, _ = vro.render_tile(vol, vro.UCHAR, tff, cam, rp, rc, W=W, H=H, hit_in=hin,
                                    hit_out=hout)
        hin, hout = hout, hin
        want_frames.append(img)
        want_hits.append(hin.copy())
    assert (want_hits[-1] == 0).sum() > want_hits[-1].size // 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_img_ess, args=(r, 2, port, W, H, T, q)) for r in range(2)]
    for p in procs:
        p.start()
    frames, hits = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k in range(4):
        assert np.array_equal(frames[k], want_frames[k]), "frame %d" % k
        assert np.array_equal(hits[k], want_hits[k]), "hit image %d" % k


def test_tile_split_is_a_partition():
    for world in (1, 2, 3, 4, 8):
        s0 = tiles.TileSplit(1024, 1000, 64, 48, world, 0)
        allt = np.concatenate(s0.tiles_of)
        assert sorted(allt.tolist()) == list(range(s0.n_tiles))
        counts = [len(t) for t in s0.tiles_of]
        assert max(counts) - min(counts) <= max(2, s0.tiles_y)
        assert s0.cap == max(counts)
    with pytest.raises(ValueError):
        tiles.TileSplit(64, 64, 24, 16, 2, 0)


def test_tiles_are_dealt_by_distance_and_the_cpp_host_deals_the_same():
    """deal_tiles: every tile has one owner, the counts differ by at most one pair of deals, every rank gets
    tiles near the centre and far from it, and the headless C++ host (vr_deal_tiles behind
    vrhost_deal_tiles) produces the same assignment."""
    import ctypes as C
    import os
    from volumerenderercl_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _lib.load()
    host = C.CDLL(os.path.join(root, "volumerenderercl_amd", "libvrhost.so"))
    host.vrhost_deal_tiles.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double,
                                       C.POINTER(C.c_uint32), C.c_uint32]
    for W, H, T, n in ((1024, 1024, 64, 8), (2048, 2048, 64, 8), (1024, 1024, 32, 4), (200, 136, 32, 3),
                       (200, 136, 16, 2), (640, 360, 48, 5), (64, 64, 64, 8), (1024, 1024, 64, 1)):
        for share in (1.0, 0.58, 0.9, 0.25, 0.0):
            owner = tiles.deal_tiles(W, H, T, T, n, share)
            tx, ty = (W + T - 1) // T, (H + T - 1) // T
            assert owner.shape == (tx * ty,) and owner.min() >= 0 and owner.max() <= n - 1
            counts = np.bincount(owner, minlength=n)
            if n > 1:
                assert counts[1:].max() - counts[1:].min() <= 2
                # rank 0 holds `share` of a peer's tiles, to within the two cards of a round
                assert abs(counts[0] - share * counts[1:].mean()) <= 2.0 + 1e-9, (W, H, T, n, share, counts)
            out = (C.c_uint32 * (tx * ty))()
            assert host.vrhost_deal_tiles(W, H, T, n, share, out, tx * ty) == 0
            np.testing.assert_array_equal(np.array(out[:], dtype=np.int64), owner)
    # the headline split: the four central tiles and the four corner tiles go to different ranks each
    owner = tiles.deal_tiles(1024, 1024, 64, 64, 8).reshape(16, 16)
    assert len({owner[7, 7], owner[7, 8], owner[8, 7], owner[8, 8]}) == 4
    ring = [(ty_, tx_) for ty_ in range(16) for tx_ in range(16) if 4 <= (2 * tx_ - 15) ** 2 + (2 * ty_ - 15) ** 2 <= 60]
    per_rank = np.bincount([owner[p] for p in ring], minlength=8)
    assert per_rank.min() >= 1 and per_rank.max() - per_rank.min() <= 2
