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
import numpy as np


def deal_tiles(W, H, tw, th, world):
    """owner[t] for the tiles of a W x H frame, numbered row-major: the tiles are sorted by the distance
    of their centre from the frame's centre (integer arithmetic, ties by tile id) and dealt to the ranks
    like cards, back and forth (0 1 .. n-1 n-1 .. 1 0 0 1 ..): every rank gets tiles of every distance.
    What a tile costs follows the object in the middle of the view; a diagonal interleave ((tx + ty) mod
    n) hands whole anti-diagonals to one rank -- at 16 x 16 tiles for 8 ranks the one through the centre:
    measured shares of 0.012 .. 0.034 ms per frame on the headline, 0.72 of the possible speed-up; dealt
    by distance 0.96.  The same rule lives in csrc/host/tilegather.cpp."""
    tiles_x, tiles_y = (W + tw - 1) // tw, (H + th - 1) // th
    ids = np.arange(tiles_x * tiles_y, dtype=np.int64)
    dx = (2 * (ids % tiles_x) + 1) * tw - W
    dy = (2 * (ids // tiles_x) + 1) * th - H
    order = np.lexsort((ids, dx * dx + dy * dy))          # by distance, then by id
    j = np.arange(ids.size) % (2 * world)
    owner = np.empty(ids.size, dtype=np.int64)
    owner[order] = np.where(j < world, j, 2 * world - 1 - j)
    return owner


class TileSplit:
    def __init__(self, width, height, tile_w, tile_h, world, rank):
        if tile_w % 16 or tile_h % 16:
            raise ValueError("Tile size must be a positive multiple of 16.")
        self.W, self.H, self.tw, self.th = int(width), int(height), int(tile_w), int(tile_h)
        self.world, self.rank = int(world), int(rank)
        self.tiles_x = (self.W + self.tw - 1) // self.tw
        self.tiles_y = (self.H + self.th - 1) // self.th
        self.n_tiles = self.tiles_x * self.tiles_y
        ids = np.arange(self.n_tiles, dtype=np.uint32)
        self.owner = deal_tiles(self.W, self.H, self.tw, self.th, self.world)
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

    Stream ordering: a renderer enqueues on a stream of its own (vrhip_get_stream) and returns
    without synchronising, while torch.distributed orders a collective behind torch's CURRENT
    stream only.  The driver therefore brackets every render: the renderer's stream first waits
    for the current stream (the gather that last read the tile buffer was waited for there), and
    the current stream then waits for the renderer's stream before the gather is issued.  A
    renderer that already launches on the current stream (vr.set_stream) needs neither wait.
    """

    def __init__(self, vr, split, device, render_tiles_fn=None, dist=None, image_ess=False,
                 hit_io=None, batch=1, lanes=None, force_gather=False):
        import torch
        self.torch = torch
        self.vr, self.split, self.device = vr, split, device
        self.render_tiles_fn = render_tiles_fn
        s = split
        # force_gather: a world of ONE rank goes through everything a world of several does -- tile
        # buffers, the collective (a world-size-1 process group: RCCL on a one-GPU box), assembly --
        # instead of the full-frame launch
        self.gathering = s.world > 1 or bool(force_gather)
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
        glist = [self.staging[b][r] for r in range(s.world)] if s.rank == 0 else None
        work = self.dist.gather(self.local[b], glist, dst=0, async_op=True)
        self.pending.append((b, n, work))

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
        glist = [self.staging[b][r] for r in range(s.world)] if s.rank == 0 else None
        work = self.dist.gather(self.local[b], glist, dst=0, async_op=True)
        self.pending.append((b, n, work))

    def collect(self, frame):
        """Finish the oldest frame in flight; returns the assembled frame on rank 0."""
        out = self.collect_batch(None if frame is None else frame.unsqueeze(0))
        return None if out is None else frame

    def collect_batch(self, frames):
        """Finish the oldest gather in flight; on rank 0 `frames` ([>= n, H, W, 4]) receives its n
        assembled frames (one index_select + one strided copy for the whole batch)."""
        s = self.split
        b, n, work = self.pending.pop(0)
        work.wait()
        if s.rank != 0:
            return None
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
