"""Image-tile decomposition of one frame over the GPUs of a node.

The reference is single-GPU; this layer is new.  Pixels are independent, so the frame is
cut into tile_w x tile_h tiles dealt to the ranks by their distance from the frame's centre
(deal_tiles: load balance under ESS/ERT -- every rank gets tiles of every distance), every rank holds the
whole volume, renders its tiles into a compact buffer and rank 0 gathers them over
RCCL/xGMI (torch.distributed `gather`, backend "nccl" == RCCL; "gloo" in the CPU tests).
One collective per frame, 16*W*H/N bytes per peer.

Image-order ESS (rendering_params.imgEss) is the one piece of inter-frame state that crosses
tile borders: a work-group looks at last frame's hit texels of its 3x3 neighbourhood.  Every
rank updates the texels of its own tiles; with `image_ess=True` the driver merges them after
each frame with one small all-reduce ((W/8+1)*(H/8+1) bytes) before the next frame reads them.
"""
import math
import os

import numpy as np


def deal_tiles(W, H, tw, th, world, root_share=1.0):
    """owner[t] for the tiles of a W x H frame, numbered row-major: the tiles are sorted by the distance
    of their centre from the frame's centre (integer arithmetic, ties by tile id) and dealt to the ranks
    like cards, back and forth (0 1 .. n-1 n-1 .. 1 0 0 1 ..): every rank gets tiles of every distance.
    What a tile costs follows the object in the middle of the view; a diagonal interleave ((tx + ty) mod
    n) hands whole anti-diagonals to one rank -- at 16 x 16 tiles for 8 ranks the one through the centre:
    measured shares of 0.012 .. 0.034 ms per frame on the headline, 0.72 of the possible speed-up; dealt
    by distance 0.96.

    root_share < 1: rank 0 also assembles every frame (time it does not render in), so it takes only that
    fraction of a peer's tiles: of its two cards per round of the deal it takes 2 * root_share on average
    (error diffusion over the rounds, first / last card alternating).  The same rule lives in the C++ host
    (vr_deal_tiles, csrc/host/vrhost_capi.cpp); tests/test_tiles_gloo.py compares the two."""
    tiles_x, tiles_y = (W + tw - 1) // tw, (H + th - 1) // th
    nt = tiles_x * tiles_y
    ids = np.arange(nt, dtype=np.int64)
    dx = (2 * (ids % tiles_x) + 1) * tw - W
    dy = (2 * (ids // tiles_x) + 1) * th - H
    order = np.lexsort((ids, dx * dx + dy * dy))          # by distance, then by id
    owner = np.zeros(nt, dtype=np.int64)
    if world == 1:
        return owner
    share = min(1.0, max(0.0, float(root_share)))
    seq, acc, cycle = [], 0.0, 0
    while len(seq) < nt:
        acc += 2.0 * share
        take = int(math.floor(acc + 1e-9))
        acc -= take
        first = take == 2 or (take == 1 and cycle % 2 == 0)
        last = take == 2 or (take == 1 and cycle % 2 == 1)
        seq += ([0] if first else []) + list(range(1, world)) + list(range(world - 1, 0, -1)) + ([0] if last else [])
        cycle += 1
    owner[order] = np.asarray(seq[:nt], dtype=np.int64)
    return owner


class TileSplit:
    def __init__(self, width, height, tile_w, tile_h, world, rank, root_share=1.0):
        if tile_w % 16 or tile_h % 16:
            raise ValueError("Tile size must be a positive multiple of 16.")
        self.W, self.H, self.tw, self.th = int(width), int(height), int(tile_w), int(tile_h)
        self.world, self.rank = int(world), int(rank)
        self.tiles_x = (self.W + self.tw - 1) // self.tw
        self.tiles_y = (self.H + self.th - 1) // self.th
        self.n_tiles = self.tiles_x * self.tiles_y
        ids = np.arange(self.n_tiles, dtype=np.uint32)
        self.root_share = float(root_share)
        self.owner = deal_tiles(self.W, self.H, self.tw, self.th, self.world, self.root_share)
        self.tiles_of = [ids[self.owner == r] for r in range(self.world)]
        self.cap = max(len(t) for t in self.tiles_of)     # slots per rank (equal-size gather)
        self.my_tiles = self.tiles_of[self.rank]

    def tile_rect(self, t):
        tx, ty = int(t) % self.tiles_x, int(t) // self.tiles_x
        x0, y0 = tx * self.tw, ty * self.th
        return x0, y0, min(self.tw, self.W - x0), min(self.th, self.H - y0)


class TileDriver:
    """Renders frames: full-frame launch at world == 1 (unless `force_gather`), tiles + gather otherwise.

    `render_tiles_fn(tile_ids, out)` fills out[k] (tile_h x tile_w x 4 floats) for tile k;
    the default drives vrhip_render_tiles on the renderer's GPU.

    Two ways to drive it:
      * render_frame(frame): one frame, synchronous (render, gather, assemble);
      * submit() / collect(frame): pipelined -- submit() renders this rank's tiles of the next
        frame into one of two buffers and starts the gather asynchronously (RCCL runs it on its
        own stream), collect() waits for the oldest gather in flight and assembles that frame
        on rank 0.  With one frame in flight the gather + assembly of frame k overlap the
        rendering of frame k + 1, so the frame rate is set by the slower of the two, not their sum;
      * submit_batch(n, before_frame) / collect_batch(frames): the same with `n` <= `batch`
        independent frames per collective (one gather and one assembly for all of them): at a few
        hundred microseconds of GPU work per rank and frame the host-side cost of a collective is
        as long as the rendering, so fewer, larger collectives keep the GPUs busy.

    sparse=True (the frame is mostly background): a tile whose pixels are all bit-identical -- the
    background outside the volume's silhouette, the cleared inside where nothing was sampled -- travels
    as ONE pixel; only the other tiles travel whole.  Per batch every rank packs its tile buffer ([slot
    numbers of its whole tiles | one pixel per slot | the whole tiles]), the ranks agree on the largest
    count with a tiny all-gather (messages of one size), and rank 0 expands what it receives into the
    same staging layout the dense gather fills -- so the assembly and the result are the dense path's,
    bit for bit.  Pure data compression: no assumption about what a background pixel is.  The count has
    to reach the host (one synchronisation per batch); the payload gather of batch k is therefore issued
    while batch k + 1 is already queued on the GPU (from the next submit, or from collect).  Headline
    frame, 64 x 64 tiles: 45 % of the tiles travel whole at 1024^2, 36 % at 2048^2.

    Stream ordering: a renderer enqueues on a stream of its own (vrhip_get_stream) and returns
    without synchronising, while torch.distributed orders a collective behind torch's CURRENT
    stream only.  The driver therefore brackets every render: the renderer's stream first waits
    for the current stream (the gather that last read the tile buffer was waited for there), and
    the current stream then waits for the renderer's stream before the gather is issued.  A
    renderer that already launches on the current stream (vr.set_stream) needs neither wait.
    """

    def __init__(self, vr, split, device, render_tiles_fn=None, dist=None, image_ess=False,
                 hit_io=None, batch=1, lanes=None, force_gather=False, sparse=False):
        import torch
        self.torch = torch
        self.vr, self.split, self.device = vr, split, device
        self.render_tiles_fn = render_tiles_fn
        s = split
        # force_gather: a world of ONE rank goes through everything a world of several does -- tile
        # buffers, the collective (a world-size-1 process group: RCCL on a one-GPU box), assembly --
        # instead of the full-frame launch
        self.gathering = s.world > 1 or bool(force_gather)
        # sparse: uniform tiles travel as one pixel (see _pack / _unpack below)
        self.sparse = bool(sparse) and self.gathering
        self.gather_stats = {"batches": 0, "dense_bytes": 0, "sent_bytes": 0}
        # frames in flight on this rank: [(renderer, torch stream), ...]; the frames of a batch are
        # dealt to them in turn (renderers sharing one volume, VolumeRenderCL.shareVolumes)
        self.lanes = list(lanes) if lanes else None
        if self.lanes and (image_ess or hit_io is not None):
            raise ValueError("image-order ESS chains its frames: one renderer per rank")
        # image-order ESS: (get, set) of the hit image the next frame reads
        self.hit_io = hit_io
        if image_ess and hit_io is None:
            self.hit_io = (lambda: vr.getImageEss(s.W, s.H)[0],
                           lambda h: vr.setImageEss(s.W, s.H, hit_in=h))
        if self.hit_io is not None:
            own = np.zeros((s.H // 8 + 1, s.W // 8 + 1), dtype=bool)
            owned_by_any = np.zeros_like(own)
            for t in range(s.n_tiles):
                x0, y0, w, h = s.tile_rect(t)
                rect = (slice(y0 // 8, (y0 + h + 7) // 8), slice(x0 // 8, (x0 + w + 7) // 8))
                owned_by_any[rect] = True
                if s.owner[t] == s.rank:
                    own[rect] = True
            self.hit_own, self.hit_owned_by_any = own, owned_by_any
        if dist is None and self.gathering:
            import torch.distributed as dist
        self.dist = dist
        self.pending = []          # (buffer index, frames, work handle) of the gathers in flight
        self.next_buf = 0
        self.batch = B = max(1, int(batch))
        if self.gathering:
            # [frame of the batch, slot, th, tw, 4]
            self.local = [torch.zeros((B, s.cap, s.th, s.tw, 4), dtype=torch.float32, device=device)
                          for _ in range(2)]
            if s.rank == 0:
                # one block per buffer: [rank, frame, slot, th, tw, 4]; the gather writes rank r's
                # tiles into staging[b][r]
                self.staging = [torch.zeros((s.world, B, s.cap, s.th, s.tw, 4), dtype=torch.float32,
                                            device=device) for _ in range(2)]
                # (frame f, tile id) -> row of staging.view(world * B * cap, ...)
                perm = np.zeros((B, s.n_tiles), dtype=np.int64)
                for f in range(B):
                    for r in range(s.world):
                        perm[f, s.tiles_of[r].astype(np.int64)] = (
                            (r * B + f) * s.cap + np.arange(len(s.tiles_of[r])))
                self.perm = torch.as_tensor(perm, device=device)
                self.exact = (s.W % s.tw == 0) and (s.H % s.th == 0)
                if not self.exact:
                    self.padded = torch.zeros((B, s.tiles_y * s.th, s.tiles_x * s.tw, 4),
                                              dtype=torch.float32, device=device)

    # ---- stream ordering between a renderer and the collective
    def _stream_of(self, r, given=None):
        """torch view of the stream renderer `r` launches on: asked from the renderer itself
        (vrhip_get_stream); `given` (a lane's stream as the caller named it) only serves stand-in
        renderers that cannot be asked."""
        if r is None or not hasattr(r, "get_stream") or getattr(self.device, "type", str(self.device)) != "cuda":
            return given
        return self.torch.cuda.ExternalStream(r.get_stream(), device=self.device)

    def _before_render(self, streams):
        cur = self.torch.cuda.current_stream(self.device) if any(s is not None for s in streams) else None
        for ls in streams:
            if ls is not None and ls != cur:
                ls.wait_stream(cur)
        return cur

    def _after_render(self, streams, cur):
        for ls in streams:
            if ls is not None and ls != cur:
                cur.wait_stream(ls)

    # ---- pipelined interface
    def submit(self):
        """Render this rank's tiles of the next frame and start its gather (world > 1)."""
        self.submit_batch(1)

    def submit_batch(self, n, before_frame=None):
        """Render this rank's tiles of the next `n` <= batch frames -- `before_frame(i)` (with
        `lanes`: `before_frame(i, renderer)`) is called ahead of frame i (jitter seed, iteration,
        ...) -- and start ONE gather for all of them."""
        s = self.split
        if not self.gathering:
            raise RuntimeError("submit/collect are for world > 1; use render_frame")
        if not 1 <= n <= self.batch:
            raise ValueError("1 <= n <= batch (%d) frames per gather" % self.batch)
        if len(self.pending) >= 2:
            raise RuntimeError("two gathers already in flight: collect first")
        b = self.next_buf
        self.next_buf ^= 1
        # the buffer may only be overwritten once the gather that last read it is done: that wait
        # sits on the current stream (collect_batch), so every renderer's stream joins it first
        if self.lanes:
            streams = [self._stream_of(r, ls) for r, ls in self.lanes]
        else:
            streams = [self._stream_of(self.vr)] if self.render_tiles_fn is None else []
        cur = self._before_render(streams)
        for i in range(n):
            if self.lanes:
                lane_vr = self.lanes[i % len(self.lanes)][0]
                if before_frame is not None:
                    before_frame(i, lane_vr)
                lane_vr.render_tiles(s.W, s.H, s.tw, s.th, s.my_tiles, self.local[b][i].data_ptr())
                continue
            if before_frame is not None:
                before_frame(i)
            if self.render_tiles_fn is None:
                self.vr.render_tiles(s.W, s.H, s.tw, s.th, s.my_tiles, self.local[b][i].data_ptr())
            else:
                self.render_tiles_fn(s.my_tiles, self.local[b][i])
            if self.hit_io is not None:
                self.merge_hit_image()      # the next frame reads the merged hit image
        self._after_render(streams, cur)    # the gather follows every renderer's frames
        self._start_gather(b, n)

    def submit_frames(self, seeds):
        """len(seeds) <= batch independent frames (frame i jittered by seeds[i]) rendered in as few
        launch sets as there are renderers on this rank (VolumeRenderCL.render_batch: the rank's
        tile share of several frames in one work queue), then ONE gather for all of them."""
        s = self.split
        n = len(seeds)
        if not self.gathering:
            raise RuntimeError("submit/collect are for world > 1; use render_frame")
        if not 1 <= n <= self.batch:
            raise ValueError("1 <= n <= batch (%d) frames per gather" % self.batch)
        if len(self.pending) >= 2:
            raise RuntimeError("two gathers already in flight: collect first")
        b = self.next_buf
        self.next_buf ^= 1
        lanes = self.lanes or [(self.vr, None)]
        per = -(-n // len(lanes))
        stride = s.cap * s.th * s.tw
        if self.render_tiles_fn is not None:      # stand-in renderer (CPU tests): frame by frame
            for i in range(n):
                self.render_tiles_fn(s.my_tiles, self.local[b][i], seed=seeds[i])
            lanes = []
        streams = [self._stream_of(r, ls) for r, ls in lanes]
        cur = self._before_render(streams)
        for j, (r, _) in enumerate(lanes):
            lo, hi = j * per, min(n, (j + 1) * per)
            if lo >= hi:
                break
            r.render_batch(s.W, s.H, seeds[lo:hi], self.local[b][lo].data_ptr(), s.tw, s.th,
                           s.my_tiles, frame_stride=stride)
        self._after_render(streams, cur)
        self._start_gather(b, n)

    # ---- the collective of a batch: dense, or uniform tiles as one pixel (sparse)
    def _gpu_pack(self):
        """True when the sparse message can be packed (and later read) by the library's kernels -- vrhip_pack_tiles /
        vrhip_message_positions / vrhip_assemble_batch, the C++ host's TileGather uses the same -- instead of a dozen
        torch ops per batch: a real renderer on a GPU."""
        lib = getattr(self.vr, "lib", None)
        return (lib is not None and self.render_tiles_fn is None and not os.environ.get("VRHIP_NO_FUSED_ASSEMBLY")
                and getattr(self.device, "type", str(self.device)) == "cuda" and self.split.world <= 64
                and self.split.cap <= 65536)

    def _start_gather(self, b, n):
        s = self.split
        if not self.sparse:
            glist = [self.staging[b][r] for r in range(s.world)] if s.rank == 0 else None
            work = self.dist.gather(self.local[b], glist, dst=0, async_op=True)
            self.pending.append({"b": b, "n": n, "work": work})
            return
        # payloads whose counts are known by now go first: the count of THIS batch reaches the host
        # only after its frames are rendered, and the next batch should be queued before anyone waits
        self._issue_payloads()
        torch = self.torch
        S, P = n * s.cap, s.th * s.tw
        counts = [torch.zeros(1, dtype=torch.int32, device=self.device) for _ in range(s.world)]
        if self._gpu_pack():
            # message = [spad slot numbers | S pixels | whole tiles], spad = S rounded up to 4: packed by three small
            # kernels on the current stream into a buffer sized for the worst case; only its used prefix travels
            import ctypes as C
            Smax = self.batch * s.cap
            if not hasattr(self, "_msg"):
                full = (Smax + 3) // 4 * 4 + 4 * Smax + 4 * Smax * P
                self._msg = [torch.empty(full, dtype=torch.float32, device=self.device) for _ in range(2)]
                self._scratch = [torch.empty(Smax, dtype=torch.int32, device=self.device) for _ in range(2)]
                self._count = [torch.zeros(1, dtype=torch.int32, device=self.device) for _ in range(2)]
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self.vr.lib.vrhip_pack_tiles(self.vr.handle, C.c_void_p(stream), C.c_void_p(self.local[b].data_ptr()), S, P,
                                              C.c_void_p(self._scratch[b].data_ptr()), C.c_void_p(self._msg[b].data_ptr()),
                                              C.c_void_p(self._count[b].data_ptr()))
            if rc != 0:
                raise RuntimeError("vrhip_pack_tiles failed (%d)" % rc)
            cwork = self.dist.all_gather(counts, self._count[b], async_op=True)
            self.pending.append({"b": b, "n": n, "work": None, "packed": True, "counts": counts, "cwork": cwork})
            return
        x = self.local[b].view(self.batch * s.cap, P, 4)[:S]
        xi = x.view(torch.int32)
        whole = (xi != xi[:, :1, :]).view(S, -1).any(dim=1)          # not all pixels bit-identical
        uni = x[:, 0, :].contiguous()                                 # one pixel per slot
        count = whole.sum().to(torch.int32).view(1)
        cwork = self.dist.all_gather(counts, count, async_op=True)
        self.pending.append({"b": b, "n": n, "work": None, "whole": whole, "uni": uni, "x": x,
                             "counts": counts, "cwork": cwork})

    def _issue_payloads(self):
        """Sparse: start the payload gather of every batch that has none yet (ONE host synchronisation per batch:
        the ranks' counts)."""
        s, torch = self.split, self.torch
        for e in self.pending:
            if e["work"] is not None or "cwork" not in e:
                continue
            e["cwork"].wait()
            cs = [int(v) for v in torch.cat(e["counts"]).cpu().tolist()]     # (the synchronisation)
            S, P = e["n"] * s.cap, s.th * s.tw
            if e.get("packed"):
                # the library's layout: the slot-number block is spad wide whatever the counts; all ranks send the
                # same length (a gather wants equal sizes): the prefix that holds the largest count's tiles
                maxc = (S + 3) // 4 * 4
                length = maxc + 4 * S + 4 * max(cs) * P
                msg = self._msg[e["b"]][:length]
                recv = None
                if s.rank == 0:
                    if not hasattr(self, "_recv") or self._recv[0].shape[1] < self._msg[0].numel():
                        self._recv = [torch.empty((s.world, self._msg[0].numel()), dtype=torch.float32, device=self.device)
                                      for _ in range(2)]
                    recv = [self._recv[e["b"]][r][:length] for r in range(s.world)]
            else:
                maxc = (max(1, max(cs)) + 3) // 4 * 4          # (a multiple of 4: the pixels behind it stay 16-byte aligned)
                slots = torch.nonzero(e["whole"]).view(-1).to(torch.int32)   # cs[rank] entries
                msg = torch.zeros(maxc + 4 * S + maxc * P * 4, dtype=torch.float32, device=self.device)
                k = cs[s.rank]
                if k:
                    msg[:k].view(torch.int32).copy_(slots)
                    msg[maxc + 4 * S: maxc + 4 * S + k * P * 4].view(k, P, 4).copy_(e["x"].index_select(0, slots.to(torch.int64)))
                msg[maxc: maxc + 4 * S].view(S, 4).copy_(e["uni"])
                recv = ([torch.empty_like(msg) for _ in range(s.world)] if s.rank == 0 else None)
            e["work"] = self.dist.gather(msg, recv, dst=0, async_op=True)
            e.update(cs=cs, maxc=maxc, recv=recv, msg=msg)
            for key in ("whole", "uni", "x", "counts", "cwork"):
                e.pop(key, None)
            self.gather_stats["batches"] += 1
            peers = max(1, s.world - 1)   # (a world of one, force_gather: its own message counts once)
            self.gather_stats["dense_bytes"] += 16 * S * P * peers
            self.gather_stats["sent_bytes"] += 4 * int(msg.numel()) * peers

    def _assemble_fused(self, e, frames):
        """Rank 0, sparse, on the GPU: the frames straight from the received messages with one kernel
        (vrhip_assemble_batch) instead of expanding them into the dense staging buffer and assembling that
        (index_select + strided copy: four passes over the frames' bytes, time rank 0 does not render in).
        False where it does not apply (CPU tensors, a stand-in renderer, odd layouts): the torch path."""
        s, torch = self.split, self.torch
        lib = getattr(self.vr, "lib", None)
        if os.environ.get("VRHIP_NO_FUSED_ASSEMBLY"):      # A/B: the torch path
            return False
        if (lib is None or frames is None or not frames.is_cuda or frames.dtype != torch.float32 or s.world > 64
                or s.cap > 65536 or not frames[:e["n"]].is_contiguous() or tuple(frames.shape[1:]) != (s.H, s.W, 4)):
            return False
        import ctypes as C
        n, S, maxc = e["n"], e["n"] * s.cap, e["maxc"]
        if not hasattr(self, "_pos") or self._pos.numel() < s.world * self.batch * s.cap:
            self._pos = torch.empty(s.world * self.batch * s.cap, dtype=torch.int32, device=self.device)
        pos = self._pos[: s.world * S]
        if not hasattr(self, "_rank_slot"):
            rs = np.zeros(s.n_tiles, dtype=np.uint32)
            for r in range(s.world):
                ids = s.tiles_of[r].astype(np.int64)
                rs[ids] = (np.uint32(r) << np.uint32(16)) | np.arange(len(ids), dtype=np.uint32)
            self._rank_slot = torch.as_tensor(rs.astype(np.int64), device=self.device).to(torch.int32)
        ptrs = (C.c_void_p * s.world)(*[m.data_ptr() for m in e["recv"]])
        counts = (C.c_uint32 * s.world)(*e["cs"])
        stream = torch.cuda.current_stream(self.device).cuda_stream
        # pos[rank][row] from the slot lists at the head of the messages (one kernel), then the frames (one kernel)
        rc = lib.vrhip_message_positions(self.vr.handle, C.c_void_p(stream), ptrs, counts, s.world, S,
                                         C.c_void_p(pos.data_ptr()))
        if rc == 0:
            rc = lib.vrhip_assemble_batch(self.vr.handle, C.c_void_p(stream), ptrs, s.world, n, s.cap, maxc,
                                          C.c_void_p(pos.data_ptr()), C.c_void_p(self._rank_slot.data_ptr()), s.W, s.H,
                                          s.tw, s.th, C.c_void_p(frames.data_ptr()))
        if rc != 0:
            raise RuntimeError("vrhip_message_positions / vrhip_assemble_batch failed (%d)" % rc)
        e["keep"] = (pos, ptrs)     # (alive until the kernel has run: the caller synchronises before reuse)
        self._last_assembled = e
        return True

    def _unpack(self, e):
        """Rank 0, sparse: the received messages expanded into the dense staging layout of buffer b."""
        s, torch = self.split, self.torch
        S, P, maxc = e["n"] * s.cap, s.th * s.tw, e["maxc"]
        for r in range(s.world):
            m = e["recv"][r]
            dense = self.staging[e["b"]][r].view(self.batch * s.cap, P, 4)[:S]
            dense.copy_(m[maxc: maxc + 4 * S].view(S, 1, 4).expand(S, P, 4))
            k = e["cs"][r]
            if k:
                slots = m[:k].view(torch.int32).to(torch.int64)
                dense.index_copy_(0, slots, m[maxc + 4 * S: maxc + 4 * S + k * P * 4].view(k, P, 4))

    def collect(self, frame):
        """Finish the oldest frame in flight; returns the assembled frame on rank 0."""
        out = self.collect_batch(None if frame is None else frame.unsqueeze(0))
        return None if out is None else frame

    def collect_batch(self, frames):
        """Finish the oldest gather in flight; on rank 0 `frames` ([>= n, H, W, 4]) receives its n
        assembled frames (one index_select + one strided copy for the whole batch)."""
        s = self.split
        if self.sparse:
            self._issue_payloads()
        e = self.pending.pop(0)
        b, n = e["b"], e["n"]
        e["work"].wait()
        if s.rank != 0:
            return None
        if self.sparse:
            if self._assemble_fused(e, frames):
                return frames
            self._unpack(e)
        rows = self.staging[b].view(s.world * self.batch * s.cap, s.th, s.tw, 4)
        tiles_sorted = rows.index_select(0, self.perm[:n].reshape(-1))
        tv = tiles_sorted.view(n, s.tiles_y, s.tiles_x, s.th, s.tw, 4).permute(0, 1, 3, 2, 4, 5)
        if self.exact:
            frames[:n].view(n, s.tiles_y, s.th, s.tiles_x, s.tw, 4).copy_(tv)
        else:
            self.padded[:n].view(n, s.tiles_y, s.th, s.tiles_x, s.tw, 4).copy_(tv)
            frames[:n].copy_(self.padded[:n, : s.H, : s.W])
        return frames

    def merge_hit_image(self):
        """Image-order ESS: give every rank the hit texels the tile owners produced this frame.
        Texels outside all tiles (the padded last row / column of groups) keep their value."""
        get, put = self.hit_io
        hit = np.ascontiguousarray(get(), dtype=np.uint8)
        mine = self.torch.from_numpy(np.where(self.hit_own, hit, 0).astype(np.int32)).to(self.device)
        self.dist.all_reduce(mine)      # sum: exactly one owner per texel
        merged = np.where(self.hit_owned_by_any, mine.cpu().numpy().astype(np.uint8), hit)
        put(merged)
        return merged

    # ---- one synchronous frame
    def render_frame(self, frame):
        """Returns the assembled H x W x 4 frame on rank 0 (None elsewhere)."""
        s = self.split
        if not self.gathering:
            if self.render_tiles_fn is None:
                streams = [self._stream_of(self.vr)]
                cur = self._before_render(streams)
                self.vr.runRaycast(s.W, s.H, out_dev_ptr=frame.data_ptr())
                self._after_render(streams, cur)     # `frame` is the caller's, on the current stream
                return frame
            out = self.torch.zeros((s.n_tiles, s.th, s.tw, 4), dtype=self.torch.float32,
                                   device=self.device)
            self.render_tiles_fn(s.my_tiles, out)
            for k, t in enumerate(s.my_tiles):
                x0, y0, w, h = s.tile_rect(t)
                frame[y0:y0 + h, x0:x0 + w] = out[k, :h, :w]
            return frame
        self.submit()
        return self.collect(frame)
