#!/usr/bin/env python3
"""bench.py -- frames of the ray-cast hot path on N MI355X GPUs of one node.

A "step" is one frame: the per-pixel front-to-back ray march over the whole viewport (one
pass of the hot path over one batch of synthetic input, volume resident in HBM).  At N>1
the frame is split into interleaved image tiles (volume replicated per GPU) and gathered
to rank 0 over RCCL -- strong scaling: the frame is fixed as N grows.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `value` = samples actually taken (inner-loop bodies,
after ESS/ERT) by all ranks in the K timed frames / wall time, in Msamples/s.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (kind, res, format, illum, tff, ess)
    # headline: BASELINE.json metric config -- 2048^3 UCHAR at a 1024^2 viewport, reference
    # defaults (default TF, central-difference Blinn-Phong, object-order ESS, ERT, rate 1.5)
    "shells2048": ("shells", 2048, "UCHAR", 1, "default", True),
    "sphere2048": ("sphere", 2048, "UCHAR", 1, "default", True),
    "haze2048": ("sphere", 2048, "UCHAR", 1, "haze", True),     # dense regime: no ERT
    "shells1024u16": ("shells", 1024, "USHORT", 1, "default", True),
    "shells1024": ("shells", 1024, "UCHAR", 1, "default", True),    # north_star's 1024^3 UCHAR grid
    "haze1024": ("sphere", 1024, "UCHAR", 1, "haze", True),     # mid-sized dense volumes
    "sphere512f": ("sphere", 512, "FLOAT", 1, "default", True),
    "sphere256": ("sphere", 256, "UCHAR", 1, "default", True),
    "sphere256_plain": ("sphere", 256, "UCHAR", 0, "default", False),
    "sphere64": ("sphere", 64, "UCHAR", 1, "default", True),    # CI-sized
    # BASELINE config 5: Woodcock-tracking path tracer (technique 1); a step = one sample per
    # pixel of the progressive render (iteration k of the running mean), 64 steps = 64 spp
    # -- on the field SURVEY 8(d) names for it (sphere), and on the shells field beside it
    "pt1024f_sphere": ("sphere", 1024, "FLOAT", 1, "default", True, 1),
    "pt1024f": ("shells", 1024, "FLOAT", 1, "default", True, 1),
    "pt256f_sphere": ("sphere", 256, "FLOAT", 1, "default", True, 1),
    "pt256f": ("shells", 256, "FLOAT", 1, "default", True, 1),
}
FMT = {"UCHAR": 0, "USHORT": 1, "FLOAT": 2}
FMT_BYTES = {"UCHAR": 1, "USHORT": 2, "FLOAT": 4}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="shells2048", choices=sorted(WORKLOADS))
    ap.add_argument("--viewport", type=int, default=1024)
    ap.add_argument("--tile", type=int, default=64, help="tile edge for the multi-GPU split")
    ap.add_argument("--frames-per-gather", type=int, default=0,
                    help="multi-GPU: independent frames per RCCL gather (each renderer of a rank renders "
                         "its half of them in one set of launches: <= 256 frames per set); 0 = 32 per rank, "
                         "at least 64: a rank's share of a set is then worth four whole frames or more")
    ap.add_argument("--frames-per-launch", type=int, default=32,
                    help="independent frames (own jitter seeds) rendered by one set of launches "
                         "(vrhip_render_batch, <= 256); 1 = one frame per launch set.  Larger sets amortise a set's "
                         "ramp and tail: 16 / 32 / 64 / 128 frames per set = 0.176 / 0.167 / 0.162 / 0.158 ms per frame "
                         "over 256 frames of the headline workload")
    ap.add_argument("--round-budget", type=int, default=0,
                    help="phase-1 sample rounds per ray when several frames are in flight (0 = 48 on one GPU; "
                         "on several, by the size of a rank's launch set: DESIGN.md 'Multi-GPU')")
    ap.add_argument("--frame-timing", type=int, default=0,
                    help="throughput mode on one GPU: 1 = keep the renderers' own events around every launch set "
                         "(vrhip_set_frame_timing; the region is timed as a whole either way)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="renderers (one stream each, sharing the volume) that take the launch sets in turn, so that "
                         "the tail of one set overlaps the head of the next; 0 = two, or one when all the timed frames "
                         "fit ONE launch set (--steps <= --frames-per-launch)")
    ap.add_argument("--root-share", default="auto",
                    help="multi-GPU: the fraction of a peer's tiles rank 0 renders (it also assembles every frame); "
                         "auto = 1 for a run of one batch, else (1 + a - N a) / (1 + a) with a = 0.03, the assembly's "
                         "share of a whole frame's time measured on one MI355X (tools/assemble_time.py: 0.006 of 0.19 ms "
                         "at 1024^2)")
    ap.add_argument("--dense-gather", action="store_true",
                    help="multi-GPU: gather every tile whole (default: tiles of one colour travel as one pixel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="target CPU work for the bounded cpu_baseline sample")
    ap.add_argument("--view", default="rot30", choices=["default", "rot30"])
    ap.add_argument("--profile-region", action="store_true",
                    help="warm-up + the timed region only (no one-frame-at-a-time pass, no instrumented "
                         "pass, no CPU baseline): a rocprofv3 --kernel-trace --stats of this command "
                         "holds the timed launches and nothing else after the warm-up")
    ap.add_argument("--out-json", default=None, help="also write the JSON line to this file")
    return ap.parse_args()


def oracle_renderer(vr, vol_host, bricks_host, tff, fmt, W, H, use_ess=True):
    """render(seed) -> (image, work counters): the oracle (CPU restatement: scalar fp32 C, OpenMP over 8-pixel row
    chunks) on the renderer's own kernel-argument structs, iteration 0.  The checker, and the cpu_baseline leg."""
    from oracle import vro
    cam, rp, rc, pt = vr.params()
    ocam = vro.CameraParams.from_buffer_copy(bytes(cam))
    orp = vro.RenderingParams.from_buffer_copy(bytes(rp))
    orc = vro.RaycastParams.from_buffer_copy(bytes(rc))
    opt = vro.PathtraceParams.from_buffer_copy(bytes(pt))
    prefix = vro.prefix_sum(tff)
    cores = host_cpu_share()

    def render(seed):
        orp.seed = seed
        orp.iteration = 0
        img, st, _ = vro.render_tile(vol_host, fmt, tff, ocam, orp, orc, opt, use_ess=use_ess,
                                     W=W, H=H, bricks=bricks_host, prefix=prefix, threads=cores)
        return img, st
    return render, cores


def cpu_baseline(render, cores, target_s, seeds, keep=()):
    """The oracle timed on whole frames of the same workload -- the timed frames' own jitter seeds, one after the
    other -- until ~target_s seconds of CPU work have been spent (bounded: at most len(seeds) frames).  Returns the
    frames of the seeds in `keep` (and of the first seed) for the parity check, and the cpu_baseline object."""
    frames, samples, secs = 0, 0, 0.0
    kept = {}
    for seed in seeds:
        t0 = time.perf_counter()
        img, st = render(seed)
        secs += time.perf_counter() - t0
        if frames == 0 or seed in keep:
            kept[seed] = (img, st)
        samples += st["samples_taken"]
        frames += 1
        if secs >= target_s:
            break
    return kept, {
        "value": samples / secs / 1e6 if secs > 0 else 0.0,
        "unit": "Msamples/s",
        "cores": int(cores),
        "kind": "port",
        "fps": frames / secs if secs > 0 else 0.0,
        "sample": "%d whole frames of the same workload (mt19937 jitter seeds continuing the timed frames' sequence), "
                  "%.1f s of CPU work, %d samples taken" % (frames, secs, samples),
    }


_BUILT_HASH = None


def source_hash():
    """sha256 (16 hex digits) over the kernel sources the measured code is built from
    (volumerenderercl_amd/csrc/*.hip, *.h, *.inc and the C ABI header), as the loaded library reports it -- an A/B
    build with extra compile flags carries "+<flags hash>" behind it (volumerenderercl_amd/_srchash.py).  A committed
    PMC profile is only used for a roofline figure when it was taken from the same build, and a library built from
    other sources than the tree holds is not measured at all (check_library_sources)."""
    from volumerenderercl_amd import _srchash
    return _BUILT_HASH or _srchash.source_hash()


def check_library_sources(lib):
    """The loaded libvrhip.so carries the hash of the sources it was built from: a stale build (sources edited or
    reverted without a rebuild) would be measured under the tree's name otherwise."""
    global _BUILT_HASH
    import ctypes
    from volumerenderercl_amd import _srchash
    lib.vrhip_build_source_hash.restype = ctypes.c_char_p
    built, tree = lib.vrhip_build_source_hash().decode(), _srchash.source_hash()
    if built.split("+")[0] != tree and os.environ.get("VRHIP_BENCH_ALLOW_STALE") != "1":   # (A/B against an older build)
        raise SystemExit("bench.py: libvrhip.so was built from kernel sources %s, the tree holds %s -- rebuild "
                         "(python -c 'import __graft_entry__ as g; g.build()') before measuring" % (built, tree))
    _BUILT_HASH = built


def schedule_key(workload, viewport, view, fif, fpl, round_budget):
    return {"workload": workload, "viewport": int(viewport), "view": view, "frames_in_flight": int(fif),
            "frames_per_launch": int(fpl), "round_budget": int(round_budget)}


def find_profile(kind, key, src_hash):
    """The committed rocprofv3 --pmc summary (profiles/r<N>/pmc_<kind>*.json, written by
    tools/pmc_issue.py / tools/pmc_traffic.py from `bench.py --profile-region` runs, each entry carrying
    the `meta` of the run it came from) whose workload, viewport, view, renderers in flight, frames per
    launch set and round budget equal this run's AND whose kernel sources hash like this run's.
    Returns (entry, None); (None, {...}) when only profiles of other sources / schedules exist."""
    import glob
    import re

    def round_no(path):
        m = re.search(r"profiles[/\\]r(\d+)[/\\]", path)
        return int(m.group(1)) if m else -1

    near = None
    paths = glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_%s*.json" % kind))
    for path in sorted(paths, key=lambda q: (-round_no(q), q)):
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:
            continue
        for entry in d.values():
            meta = entry.get("meta") if isinstance(entry, dict) else None
            if not meta or meta.get("workload") != key["workload"]:
                continue
            same_schedule = all(meta.get(k) == v for k, v in key.items())
            if same_schedule and meta.get("source_hash") == src_hash:
                return dict(entry, file=os.path.relpath(path, ROOT)), None
            if near is None or (same_schedule and not near["same_schedule"]):
                near = {"file": os.path.relpath(path, ROOT), "same_schedule": same_schedule,
                        "profile_meta": {k: meta.get(k) for k in list(key) + ["source_hash", "head"]},
                        "this_run": dict(key, source_hash=src_hash)}
    return None, near


def host_cpu_share():
    """Threads this process may actually use: min(affinity mask, cgroup cpu quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def main():
    args = parse()
    # stdout carries ONE line, the result: native libraries write banners to file descriptor 1 (RCCL prints its
    # version block there when the process group comes up), so everything but that line goes to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(text):
        os.write(result_fd, (text + "\n").encode())

    parity_failed = False
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
        args.gpus = world
    # rehearsal knob for a one-GPU box: every rank on device 0 over gloo (RCCL refuses two ranks
    # on one device); the driver's multi-GPU runs use one GPU per rank and RCCL
    rehearsal = os.environ.get("VRHIP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    # second rehearsal knob: VRHIP_BENCH_FORCE_GATHER=1 sends a world of ONE rank through everything a
    # world of several runs -- RCCL process group (world size 1), tile buffers, batched gathers with one
    # in flight, assembly, the collectives over the counters -- so that none of it executes for the
    # first time on the 8-GPU node.  `multi` below = "the distributed code path".
    force_gather = os.environ.get("VRHIP_BENCH_FORCE_GATHER") == "1" and world == 1
    multi = world > 1 or force_gather
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    _pool = torch.cuda.Stream(dev)   # torch creates its stream pool on first use (one 17 ms hipStreamCreateWithPriority
    del _pool                        # on this image): here, not between the warm-up and the first timed frame
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if force_gather:
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29541")):
                os.environ.setdefault(k, v)
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from volumerenderercl_amd import VolumeRenderCL, frontend
    from volumerenderercl_amd import tiles as vtiles

    kind, res, fmt_name, illum, tff_name, ess = WORKLOADS[args.workload][:6]
    technique = WORKLOADS[args.workload][6] if len(WORKLOADS[args.workload]) > 6 else 0
    fmt, b = FMT[fmt_name], FMT_BYTES[fmt_name]
    W = H = args.viewport
    tff = {"default": frontend.tff_from_stops, "haze": frontend.haze_tff,
           "opaque": frontend.opaque_ramp_tff}[tff_name]()

    vr = VolumeRenderCL()
    vr.initialize(device_id=local_rank)
    check_library_sources(vr.lib)
    stream = torch.cuda.current_stream(dev)
    vr.set_stream(stream.cuda_stream)            # kernels + HIP events on torch's stream
    vr.synthVolume(kind, (res, res, res), fmt)   # input generated in HBM
    vr.setTransferFunction(tff)                  # also builds the ESS bricks (reference order)
    bricks_s = vr.lastBricksSeconds()
    vr.setIllumination(illum)
    vr.setObjEss(ess)
    vr.setTechnique(technique)
    view = frontend.view_matrix() if args.view == "default" else frontend.view_matrix(
        frontend.quat_from_axis_angle((1, 1, 0), 30.0))
    vr.updateView(view)

    # frame k of the reference uses output k of a default-seeded std::mt19937
    mt = frontend.Mt19937()
    seeds = [mt() for _ in range(args.warmup + args.steps)]

    frame = torch.empty((H, W, 4), dtype=torch.float32, device=dev) if rank == 0 else None
    # world > 1: `--frames-per-gather` independent frames share one collective (fewer, larger
    # gathers: the host-side cost of a collective is comparable to a rank's rendering time)
    fpg_want = args.frames_per_gather if args.frames_per_gather > 0 else max(64, 32 * world)
    fif = (args.frames_in_flight if args.frames_in_flight > 0 else 2) if technique == 0 else 1
    fpl = max(1, min(args.frames_per_launch, 256)) if technique == 0 else 1
    throughput = fif > 1 or fpl > 1
    # A SHORT run -- all the timed frames fit one launch set (the round driver's --steps 20) -- goes to ONE renderer
    # as ONE set: two renderers with half the frames each pay a set's ramp and tail (pre-pass before anything marches,
    # the last rays of phase 1, phase 2's tail: ~0.5 ms) side by side without hiding each other's, which is what a
    # second renderer is for when sets follow one another.  Measured (tools/short_run_sweep.sh, tools/share_time.py
    # TOTAL=20): 20 frames as 1 x 20 against 2 x 10: 0.197 against 0.221 ms per frame on one box, 0.190 / 0.193 on
    # another; an 8-rank tile share 0.038 against 0.043.
    if throughput and args.frames_in_flight <= 0 and args.steps <= fpl:
        fif = 1
    fpg = max(1, min(fpg_want, args.steps, 256 * fif))
    # Rank 0 also assembles every batch (time it does not render in): when batches FOLLOW one another -- the assembly of
    # batch k shares rank 0's GPU with its rendering of batch k + 1 -- it takes a smaller share of the tiles, (1 + a -
    # N a) / (1 + a) of a peer's with a = the assembly's share of a whole frame's time (0.006 of 0.19 ms at 1024^2 since
    # round 4: tools/assemble_time.py).  A run of ONE batch (the round driver's --steps 20) has nothing for the
    # assembly to overlap: every rank renders, then the gather, then the assembly -- an equal share is the shortest.
    if args.root_share == "auto":
        a_asm = 0.03
        one_batch = args.steps <= fpg
        root_share = (max(0.25, (1.0 + a_asm - world * a_asm) / (1.0 + a_asm))
                      if (world > 1 and throughput and not one_batch) else 1.0)
    else:
        root_share = float(args.root_share)
    split = vtiles.TileSplit(W, H, args.tile, args.tile, world, rank, root_share)
    if args.round_budget <= 0:
        # One GPU: 48 rounds (rays stay in the leaner one-lane phase while other frames hide its latency).  A rank's
        # tile share of a short run cannot fill its GPU, and the set's time is the chain of its longest rays: fewer
        # rounds before the 4-lane phase, by the pixel-frames of a launch set (tools/share_time.py, BUDGET=...:
        # 8-rank shares of 2 x 10 frames 0.043 -> 0.036 ms per frame with 16-24 rounds, 4 ranks 0.061 -> 0.056 with
        # 24-32, 2 ranks 0.104 -> 0.095 with 32).  The dense "haze" regime keeps 48 at every size (measured).
        args.round_budget = 48
        if multi and throughput and tff_name != "haze":
            set_frames = -(-min(fpg, args.steps) // fif)
            p_set = set_frames * len(split.my_tiles) * args.tile * args.tile
            args.round_budget = 48 if p_set >= (8 << 20) else 32 if p_set >= (4 << 20) else 24 if p_set >= (2 << 20) else 16
    # (in throughput mode this driver only serves the warm-up and the untimed one-frame-at-a-time
    # passes; the timed loop has its own, below)
    sparse = not args.dense_gather
    driver = vtiles.TileDriver(vr, split, dev, batch=1 if throughput else fpg, force_gather=force_gather, sparse=sparse)
    frames = (torch.empty((fpg, H, W, 4), dtype=torch.float32, device=dev)
              if rank == 0 and multi else None)

    def render(seed, k=0):
        vr.setSeed(seed)
        vr.setIteration(k if technique == 1 else 0)   # path tracer: progressive running mean
        return driver.render_frame(frame)

    # Frames in flight (single GPU; the path tracer's running mean chains its frames, so it keeps
    # one): the extra renderers share the first one's voxels and bricks (vrhip_share_volumes) and
    # own a stream, a frame buffer and scratch each.
    if throughput:
        # throughput schedule: with other frames hiding the latency, rays stay longer in the
        # leaner one-lane phase (vrhip_set_round_budget; the serial pass below sets 10 again)
        vr.setRoundBudget(args.round_budget)
    lanes = [(vr, stream, frame)]
    for _ in range(fif - 1):
        s2 = torch.cuda.Stream(dev)
        twin = vr.shareVolumes()
        twin.set_stream(s2.cuda_stream)
        lanes.append((twin, s2, None))
    if not multi and throughput:   # one output block of fpl frames per renderer
        lanes = [(r, s_, torch.empty((fpl, H, W, 4), dtype=torch.float32, device=dev)) for r, s_, _ in lanes]
        # nobody reads a launch set's own kernel time here (the region is timed as a whole): without the two
        # events around every set the next set's pre-pass can overlap the tail of the one before
        # (vrhip_set_frame_timing; on again for the one-frame-at-a-time pass below)
        for r, _, _ in lanes:
            r.setFrameTiming(bool(args.frame_timing))
    if multi and throughput:
        driver_mt = vtiles.TileDriver(vr, split, dev, batch=fpg, lanes=[(r, s_) for r, s_, _ in lanes],
                                      force_gather=force_gather, sparse=sparse)

    # the throughput warm-up renders frames of its OWN seeds (a second generator), not the timed ones:
    # caches and the per-pixel cost map (phase 2's sort key) are primed by similar, not identical, frames
    warm_mt = frontend.Mt19937(20261004)
    warm_seeds = [warm_mt() for _ in range(max(fpl, fpg) * max(1, fif))]

    def render_block(j, frame_ids, warm=False):
        # renderer j % fif renders these frames (their own jitter seeds) with one set of launches
        r, _, out = lanes[j % fif]
        sd = ([warm_seeds[(j % fif) * fpl + i] for i in range(len(frame_ids))] if warm
              else [seeds[args.warmup + k] for k in frame_ids])
        r.render_batch(W, H, sd, out.data_ptr())

    # world > 1: one gather in flight -- the gather + assembly of a batch of frames overlap the
    # rendering of the next batch
    chunks = [list(range(c0, min(c0 + fpg, args.steps))) for c0 in range(0, args.steps, fpg)]
    # the timed frames in launch sets of <= fpl frames, as many sets as a multiple of the renderers in
    # flight and all of (nearly) the same size: no renderer is left with a short last set
    n_sets = -(-args.steps // fpl)
    n_sets = min(args.steps, -(-n_sets // fif) * fif)
    bounds = [round(i * args.steps / n_sets) for i in range(n_sets + 1)]
    blocks = [list(range(bounds[i], bounds[i + 1])) for i in range(n_sets) if bounds[i + 1] > bounds[i]]
    drv = driver_mt if (multi and throughput) else driver

    def submit_chunk(chunk, warm=False):
        if throughput:   # the rank's share of all the chunk's frames: one launch set per renderer
            drv.submit_frames(warm_seeds[:len(chunk)] if warm else [seeds[args.warmup + k] for k in chunk])
            return

        def before(i, r=vr):
            r.setSeed(seeds[args.warmup + chunk[i]])
            r.setIteration(chunk[i] if technique == 1 else 0)
        drv.submit_batch(len(chunk), before)

    for k in range(args.warmup):
        render(seeds[k])
    if throughput:   # every renderer once, untimed: buffers, work queue, skip bitmap, cell grid
        if not multi:
            for j in range(fif):
                render_block(j, blocks[0], warm=True)
        else:
            submit_chunk(chunks[0], warm=True)
            drv.collect_batch(frames)
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    gs0 = dict(drv.gather_stats)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _, s2, _ in lanes[1:]:
        s2.wait_event(ev0)
    if not multi and throughput:
        for j, blk in enumerate(blocks):
            render_block(j, blk)
    elif not multi:
        for k in range(args.steps):
            render(seeds[args.warmup + k], k)
    else:
        submit_chunk(chunks[0])
        for chunk in chunks[1:]:
            submit_chunk(chunk)
            drv.collect_batch(frames)
        drv.collect_batch(frames)
    for _, s2, _ in lanes[1:]:
        stream.wait_stream(s2)       # the region ends when every lane's last frame has
    ev1.record(stream)
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    wall = time.perf_counter() - t0
    gather_info = None
    if multi:
        gs1 = drv.gather_stats
        dense_b = 16 * W * H * (world - 1) / world if world > 1 else 16 * W * H
        gather_info = {"mode": "sparse (tiles of one colour travel as one pixel)" if drv.sparse else "dense",
                       "dense_bytes_per_frame_to_root": dense_b,
                       "sent_bytes_per_frame_to_root": ((gs1["sent_bytes"] - gs0["sent_bytes"]) / args.steps
                                                        if drv.sparse else dense_b),
                       "tile": args.tile, "root_share": root_share,
                       "tiles_per_rank": [int(len(t_)) for t_ in split.tiles_of]}
    gpu_region_s = ev0.elapsed_time(ev1) * 1e-3
    last_kernel_s = vr.getLastExecTime()
    last_phases = vr.getLastPhaseTimes()
    wall_t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if multi:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall = float(wall_t.item())

    # ---- the timed region's own output, copied to the host before anything else renders: the frames the
    # oracle is asked about below (parity.timed_frames) and what the last timed launch set launched
    timed_frames, timed_info = [], None      # [(label, jitter seed, host image)]
    set_frames_used = max(len(b_) for b_ in blocks) if throughput else 1
    if rank == 0 and technique == 0 and not args.profile_region:
        def pick(label, k, tensor):
            if all(k != k_ for _, k_, _ in timed_frames):
                timed_frames.append((label, k, tensor.detach().cpu().numpy().copy()))
        if not multi and throughput:
            # every renderer's output block still holds the last launch set it rendered
            last_of_lane = {j % fif: j for j in range(len(blocks))}
            j_first, j_last = min(last_of_lane.values()), len(blocks) - 1
            for j, i in ((j_first, 0), (j_last, len(blocks[j_last]) - 1)):
                pick("timed frame %d: frame %d of %d in launch set %d of %d (renderer %d)"
                     % (blocks[j][i], i, len(blocks[j]), j, len(blocks), j % fif), blocks[j][i], lanes[j % fif][2][i])
            timed_info = lanes[j_last % fif][0].lastLaunchInfo()
        elif not multi:
            pick("timed frame %d (the last of the region)" % (args.steps - 1), args.steps - 1, frame)
            timed_info = vr.lastLaunchInfo()
        else:
            n_last = len(chunks[-1])
            for i in (0, n_last - 1):
                pick("timed frame %d: frame %d of %d of the last gathered batch, assembled on rank 0"
                     % (chunks[-1][i], i, n_last), chunks[-1][i], frames[i])
            timed_info = vr.lastLaunchInfo()
        timed_frames = [(label, seeds[args.warmup + k], img) for label, k, img in timed_frames]

    if args.profile_region:
        # nothing but warm-up + timed launches has run: for rocprofv3 --kernel-trace --stats / --pmc
        if rank == 0:
            line = json.dumps({
                "profile_region": True, "workload": args.workload, "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
                "avg_launch_ms": gpu_region_s / args.steps * 1e3,
                "viewport": W, "view": args.view, "source_hash": source_hash(),
                "head": os.environ.get("VRHIP_HEAD"),
                "frames_in_flight": fif, "frames_per_launch": set_frames_used,
                "round_budget": args.round_budget if throughput else 10,
                "launch_sets_in_region": len(blocks) if (not multi and throughput) else args.steps,
                "warmup_launch_sets": (args.warmup + (fif if throughput else 0)) if not multi else None,
                "note": "warm-up + timed region and nothing else: the process's kernel dispatches are the warm-up's "
                        "(`warmup` single frames, then one launch set per renderer in throughput mode) followed by "
                        "the timed launch sets; work counters, roofline and cpu_baseline come from the full run"})
            emit(line)
            if args.out_json:
                open(args.out_json, "w").write(line + "\n")
        if multi:
            dist.barrier()
            dist.destroy_process_group()
        vr.close()
        return

    # ---- untimed: the same frames one at a time (what one launch takes when it has the GPU to itself)
    serial_s = None
    for r, _, _ in lanes:
        r.setFrameTiming(True)
    if throughput:
        vr.setRoundBudget(10)      # the single-frame schedule for everything that follows
    if not multi and throughput:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(args.steps):
            render(seeds[args.warmup + k], k)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        serial_s = e0.elapsed_time(e1) * 1e-3 / args.steps
        last_kernel_s = vr.getLastExecTime()
    if technique == 0:   # one more frame with the event between the phases (it costs GPU time: off otherwise)
        vr.setPhaseTiming(True)
        render(seeds[args.warmup + args.steps - 1], args.steps - 1)
        torch.cuda.synchronize(dev)
        last_phases = vr.getLastPhaseTimes()
        vr.setPhaseTiming(False)

    # ---- untimed: exact work counters of the K timed frames (instrumented kernel variant)
    vr.setStatsEnabled(True)
    tot = np.zeros(6, dtype=np.int64)
    names = ["samples_taken", "samples_nominal", "samples_shaded", "bricks_visited",
             "bricks_skipped", "rays_hit"]
    for k in range(args.steps):
        render(seeds[args.warmup + k], k)
        st = vr.getStats()
        tot += np.array([st[n] for n in names], dtype=np.int64)
    vr.setStatsEnabled(False)
    tot_t = torch.tensor(tot, dtype=torch.int64, device=dev)
    if multi:
        dist.all_reduce(tot_t, op=dist.ReduceOp.SUM)
    tot = tot_t.cpu().numpy()
    work = dict(zip(names, [int(x) for x in tot]))

    # ---- untimed: compulsory traffic of ONE launch on rank 0 (first timed seed)
    roofline = None
    roofline_valu = None
    cpu = None
    parity = None
    if rank == 0:
        vr.setSeed(seeds[args.warmup])
        vr.setIteration(0)
        if world == 1:
            mb, _ = vr.countTouched(W, H)
            n_pix = W * H
        else:
            mb = vr.countTouchedTiles(W, H, args.tile, args.tile, split.my_tiles)
            n_pix = len(split.my_tiles) * args.tile * args.tile
        st1 = vr.getStats()
        # B_frame = b*64*|micro-bricks touched| + 2b*|bricks visited| + 16 B per pixel written
        ref_set_bytes = b * 64 * mb + 2 * b * st1["bricks_visited"] * (0 if technique == 1 else 1) + 16 * n_pix
        alg_bytes, alg_note = ref_set_bytes, ("b*64*|micro-bricks touched by >= 1 fetch of the reference's sample set| + "
                                              "2b*|bricks visited| + 16 B per pixel (SURVEY 8d B_frame)")
        if technique == 1 and world == 1:
            # The path tracer does not move the reference's fetch set: a tracking step whose cell bound is
            # below the walk's threshold is a rejection whatever the voxels hold and is never fetched.  It is
            # priced against the micro-bricks its OWN fetches touch (vrhip_count_fetched: culling on, no
            # speculative steps) -- bytes the kernel can be shown to need.
            mbf = vr.countFetched(W, H)
            alg_bytes = b * 64 * mbf + 16 * n_pix
            alg_note = ("b*64*|micro-bricks touched by the fetches the opacity-bound culling lets through| + 16 B per "
                        "pixel; the reference's un-culled fetch set would be %d bytes" % ref_set_bytes)
        # The ray-cast pass is ONE logical kernel issued as two back-to-back launches (phase 1:
        # budgeted march of every ray; phase 2: the suspended long rays, 4 lanes per ray).  Its
        # duration = HIP events over the timed region on the launch stream / K (world == 1; with
        # the gather in the region at world > 1 the vrhip events of the last pass are used).
        # (multi: the renderer's own events around its last set of launches, which held its share of
        # ceil(len(last chunk) / renderers) frames in throughput mode)
        last_set_frames = (-(-len(chunks[-1]) // fif)) if (multi and throughput) else 1
        kernel_s = gpu_region_s / args.steps if not multi else last_kernel_s / last_set_frames
        achieved = alg_bytes / kernel_s / 1e9
        src = source_hash()
        key = schedule_key(args.workload, W, args.view, fif, set_frames_used, args.round_budget if throughput else 10)
        traffic, traffic_stale = find_profile("traffic", key, src) if not multi else (None, None)
        hbm_traffic = traffic["hbm_bytes_per_pass"] if traffic else None
        # a fraction of the HBM peak is only printed when the counters agree that the kernel moves at
        # least its algorithmic bytes (otherwise the accounting, not the kernel, is what was measured)
        frac_ok = hbm_traffic is None or hbm_traffic >= 0.95 * alg_bytes
        roofline = {
            "kernel": ("vr_pathtrace_kernel <%s>, one launch per sample-per-pixel pass" % fmt_name.lower())
                      if technique == 1 else
                      "ray-cast pass <%s, ESS=%s> = vr_dda_prepass_kernel + vr_raycast_rays_kernel + "
                      "(counting sort) + vr_raycast_split_kernel, back-to-back launches per set of frames"
                      % (fmt_name.lower(), ess),
            "bound": "hbm",
            "achieved": achieved if frac_ok else None,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS if frac_ok else None,
            "traffic": hbm_traffic,
            "traffic_detail": ({k: traffic.get(k) for k in ("file", "source", "fetch_bytes_raw_per_pass",
                                                             "write_bytes_per_pass", "fetch_correction",
                                                             "calibration", "note", "meta")}
                               if traffic else None),
            "stale_profile": traffic_stale,
            "algorithmic_bytes_per_launch": int(alg_bytes),
            "algorithmic_bytes_note": alg_note,
            "avg_launch_ms": kernel_s * 1e3,
            "frames_in_flight": fif,
            "frames_per_launch": set_frames_used,
            "round_budget": args.round_budget if throughput else 10,
            "viewport": W, "view": args.view, "source_hash": src,
            "serial_launch_ms": serial_s * 1e3 if serial_s else None,
            "launch_note": ("%d renderer(s) on as many streams over one shared volume, each rendering up to %d independent "
                            "frames (own jitter seeds) per set of launches (vrhip_render_batch; this run: sets of %d): avg_launch_ms = "
                            "HIP-event time of the timed region / frames (the renderers' own events around each "
                            "set off unless --frame-timing 1), phase-1 round budget %d (throughput "
                            "schedule); serial_launch_ms = the same frames one at a time with the single-frame "
                            "schedule (budget 10), which is what a rocprofv3 kernel trace of `--frames-in-flight 1 "
                            "--frames-per-launch 1` sums to" % (fif, fpl, set_frames_used, args.round_budget))
                           if throughput else "one frame at a time",
            "last_pass_ms_hip_events": {"phase1": last_phases[0] * 1e3, "phase2": last_phases[1] * 1e3,
                                        "total": last_kernel_s * 1e3},
            "request_bytes_per_launch": int(b * 8 * (st1["samples_taken"] +
                                                    6 * st1["samples_shaded"])),
            "note": ("not HBM-bound: bound by VALU issue (roofline_valu_issue; DESIGN.md 'Kernels'); the "
                     "streaming kernel of the path is vr_build_bricks (see bricks_build)") +
                    ("" if frac_ok else "; no fraction printed: the counters' traffic is BELOW the algorithmic bytes"),
        }
        # What actually bounds the march: VALU issue.  A gfx950 SIMD has 16 lanes: a wave64 VALU instruction occupies
        # it for FOUR cycles -- plain or packed (v_pk_fma_f32: two fp32 operations per lane in the same four cycles;
        # the guide's 157.3 TFLOP/s = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz x 2 (FMA) x 2 (packed) is the packed
        # figure, "2 cycles per instruction" is that figure read as plain fp32).  tools/micro_occ.hip measures it with
        # the shader clock read beside the wall clock (profiles/r4/micro_occ.txt: 4.1-4.2 cycles at 2.2-2.4 GHz under
        # an all-CU burn, v_fma_f32 and v_pk_fma_f32 alike: 5.8e11 wave-instructions/s, 148 TFLOP/s packed).
        issue, issue_stale = find_profile("issue", key, src) if not multi else (None, None)
        if issue:
            peak_wi = 256 * 4 * 2.4e9 / 4.0
            ach_wi = issue["valu_wave_insts_per_frame"] / kernel_s
            roofline_valu = {
                "bound": "valu_issue",
                "achieved": ach_wi / 1e9,
                "peak": peak_wi / 1e9,
                "unit": "G wave-instructions/s",
                "frac": ach_wi / peak_wi,
                "peak_note": "256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction on a 16-lane SIMD "
                             "(measured: tools/micro_occ.hip -> profiles/r4/micro_occ.txt, shader clock and wall clock "
                             "read together; packed fp32 costs the same slot); the guide's fp32 vector peak / 128 flops "
                             "would be 1228.8, which is the PACKED rate (two operations per lane and slot)",
                # what the chip sustains on independent v_fma_f32 chains with every CU busy
                # (tools/micro_occ.hip, MI355X: 4.4e11/s at two waves per SIMD, 5.8e11 at eight)
                "peak_measured": 582.0,
                "frac_of_measured": ach_wi / 5.82e11,
                "valu_wave_insts_per_frame": issue["valu_wave_insts_per_frame"],
                "valu_lane_utilisation": issue.get("valu_lane_utilisation"),
                "source": issue.get("file"),
                "profile_meta": issue.get("meta"),
                "note": "SQ_INSTS_VALU of the warm-up + timed launches of `bench.py --profile-region` (rocprofv3 --pmc, with "
                        "this run's workload, viewport, schedule and kernel sources, committed under profiles/; the warm-up's "
                        "launch sets render frames of the same schedule, its `warmup` single frames ride along) / all the "
                        "frames those launches rendered, over this run's avg_launch_ms",
            }
        elif issue_stale:
            roofline_valu = {"bound": "valu_issue", "achieved": None, "frac": None, "stale_profile": issue_stale,
                             "note": "no committed PMC profile matches this run's schedule and kernel sources "
                                     "(tools/profile_region.sh regenerates it)"}
        if not args.no_cpu_baseline:
            # ---- the oracle: the cpu_baseline leg (N = 1 only) and the checker
            vol_host = vr.downloadVolume()
            bricks_host = vr.downloadBricks()
            orender, cores = oracle_renderer(vr, vol_host, bricks_host, tff, fmt, W, H, use_ess=ess)
            seed0 = seeds[args.warmup]
            kept = {}
            if world == 1:
                cpu_seeds = seeds[args.warmup:] + [mt() for _ in range(2000)]   # bounded by cpu_seconds
                kept, cpu = cpu_baseline(orender, cores, args.cpu_seconds, cpu_seeds,
                                         keep=[sd for _, sd, _ in timed_frames])
            for sd in [seed0] + [sd for _, sd, _ in timed_frames]:
                if sd not in kept:
                    kept[sd] = orender(sd)     # (untimed: a frame the baseline leg did not reach)
            del vol_host
            # ---- (1) the frames the timed region itself produced -- the launch sets of the benchmark, with their
            # schedule (frames per set, round budget, waves per workgroup, lookahead: `timed_launch`) -- against the
            # oracle's frames of the same jitter seeds
            checked = []
            for label, sd, img in timed_frames:
                checked.append({"frame": label, "seed": int(sd),
                                "max_abs_diff": float(np.abs(img.astype(np.float64) - kept[sd][0]).max())})
            # ---- (2) one frame on its own for the first timed seed: the production (un-instrumented) kernels'
            # image and the instrumented kernels' image + six work counters (technique 1 chains its frames
            # through the running mean: iteration 0 = a frame on its own)
            ref_img, ref_st = kept[seed0]
            vr.setSeed(seed0)
            vr.setIteration(0)
            vr.setStatsEnabled(False)
            gpu_img = vr.runRaycastNoGL(W, H)
            single_info = vr.lastLaunchInfo()
            vr.setIteration(0)
            vr.setStatsEnabled(True)
            gpu_img_i = vr.runRaycastNoGL(W, H)
            gpu_st = vr.getStats()
            vr.setStatsEnabled(False)
            if technique == 1:   # the path tracer's brick counters count its culling, not bricks
                gpu_st = dict(gpu_st, bricks_visited=0, bricks_skipped=0, samples_nominal=0)   # (and its leaps)
            single = float(max(np.abs(gpu_img.astype(np.float64) - ref_img).max(),
                               np.abs(gpu_img_i.astype(np.float64) - ref_img).max()))
            parity = {
                "max_abs_diff": max([single] + [c["max_abs_diff"] for c in checked]),
                "tolerance": 1e-4,
                "counters_equal": gpu_st == ref_st,
                "timed_frames": checked,
                "timed_launch": timed_info,
                "single_frame": {"seed": int(seed0), "max_abs_diff": single, "launch": single_info},
                "frame": ("%d frame(s) of the timed region itself (copied from the renderers' output blocks when the "
                          "region ended) and, rendered on its own, the first timed seed %d through the production "
                          "(un-instrumented) and the instrumented kernels (+ six work counters), %dx%d, each against "
                          "the oracle's frame of the same seed" % (len(checked), seed0, W, H))
                         if checked else
                         ("first timed seed %d at iteration 0, %dx%d, production (un-instrumented) and instrumented "
                          "kernels vs the oracle frame (the timed launches are this kernel at iterations 0..%d of the "
                          "running mean)" % (seed0, W, H, args.steps - 1)),
            }

    if rank == 0:
        out = {
            "metric": ("Msamples/s (tracking steps taken, Woodcock path tracer, 1 spp per step) at %dx%d "
                       "viewport, %d^3 %s" if technique == 1 else
                       "Msamples/s (samples taken, rays x steps after ESS/ERT) at %dx%d viewport, "
                       "%d^3 %s") % (W, H, res, fmt_name),
            "value": work["samples_taken"] / wall / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "fps": args.steps / wall,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d^3 %s '%s' field generated in HBM, %dx%d viewport (launch grid "
                            "%dx%d), view %s, TF %s, illumType %d, object-order ESS %s, ERT 0.98, "
                            "samplingRate 1.5, per-frame mt19937 jitter seeds"
                            % (args.workload, res, fmt_name, kind, W, H, W + (8 - W % 8),
                               H + (8 - H % 8), args.view, tff_name, illum, "on" if ess else "off")
                            + (" -- technique 1 (path tracer, max_extinction 100): illumType/ESS/ERT/"
                               "samplingRate unused" if technique == 1 else ""),
                "parallelism": "tiles%dx%d/%d ranks, volume replicated, %d renderer(s) per rank, one RCCL "
                               "gather per %d frames (one gather in flight)" % (
                                   args.tile, args.tile, world, fif, fpg)
                               if multi else "single GPU, full frames, %d renderer(s) x %d frames per "
                                                 "launch set (%d launch sets in the timed region)" % (
                                                     fif, set_frames_used, len(blocks) if throughput else args.steps),
            },
            "value_note": "samples TAKEN = inner-loop bodies the reference executes after ESS/ERT; those that lie in "
                          "provably empty cells (opacity exactly 0) are stepped over without a voxel fetch and still "
                          "count, as the reference takes them",
            "msamples_nominal_per_s": work["samples_nominal"] / wall / 1e6,
            "work_per_frame": {k: v // args.steps for k, v in work.items()},
            "bricks_build": {
                "seconds": bricks_s,
                "GB/s": (b * res ** 3 + 2 * b * 64 ** 3) / bricks_s / 1e9 if bricks_s > 0 else None,
                "frac_hbm_peak": ((b * res ** 3) / bricks_s / 1e9 / HBM_PEAK_GBS
                                  if bricks_s > 0 else None),
            },
            "gather": gather_info,
            "roofline": roofline,
            "roofline_valu_issue": roofline_valu,
            "cpu_baseline": cpu,
            "parity": parity,
            "parity_max_abs_diff": parity["max_abs_diff"] if parity else None,
        }
        line = json.dumps(out)
        emit(line)
        if args.out_json:
            open(args.out_json, "w").write(line + "\n")
        if parity and not (parity["max_abs_diff"] <= parity["tolerance"] and parity["counters_equal"]):
            sys.stderr.write("bench.py: PARITY FAILURE against the oracle: %s\n" % json.dumps(parity))
            parity_failed = True
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    vr.close()
    if parity_failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
