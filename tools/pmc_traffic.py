#!/usr/bin/env python3
"""tools/pmc_traffic.py PROF_DIR WORKLOAD FRAMES OUT_JSON [VOLUME_BYTES] -- HBM traffic of the ray-cast
pass per FRAME from the rocprofv3 --pmc passes tools/profile_region.sh wrote for
`bench.py --profile-region` (FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes: they do not
fit one TCC pass together; every ray-cast dispatch of that command renders frames of the measured
schedule, FRAMES = all the frames it rendered).

Correction per MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
tallies 128-byte requests at 64 bytes, i.e. reads x2 for wide coalesced streams.  The same run
contains a kernel with a KNOWN byte count in that regime -- vr_build_bricks streams the whole volume
once with 16-byte loads -- which is used to check the factor; the ray-cast kernels gather single
bytes (an access width the guide calls uncalibrated), so their corrected number is an estimate and
the raw counter value is kept beside it."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

PASS = ("vr_dda_prepass_kernel", "vr_raycast_rays_kernel", "vr_raycast_kernel", "vr_raycast_split_kernel",
        "vr_cont_hist_kernel", "vr_cont_scatter_kernel", "vr_march_kernel", "vr_pathtrace_kernel",
        "vr_raycast_staged_kernel")


def short(name):
    m = re.search(r"(vr_\w+)(<.*?>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def main():
    prof, workload, frames, out_path = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
    volume_bytes = float(sys.argv[5]) if len(sys.argv) > 5 else None
    tot = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(prof, "p*", "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        meta = {}
        for row in csv.DictReader(open(f)):
            c = row["Counter_Name"]
            if c not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            key = (row["Dispatch_Id"], c)
            per_dispatch[key] += float(row["Counter_Value"])
            meta[key] = short(row["Kernel_Name"])
        for key, v in per_dispatch.items():
            tot[meta[key]][key[1]] += v
            cnt[meta[key]][key[1]] += 1
    res = {"source": os.path.basename(os.path.normpath(prof)), "frames": frames, "kernels": {}}
    for k, d in tot.items():
        res["kernels"][k] = {c: {"total_KiB": v, "dispatches": cnt[k][c]} for c, v in d.items()}
    factor = 2.0
    bb = [k for k in res["kernels"] if k.startswith("vr_build_bricks")]
    if bb and volume_bytes and "FETCH_SIZE" in res["kernels"][bb[0]]:
        e = res["kernels"][bb[0]]["FETCH_SIZE"]
        raw = e["total_KiB"] / e["dispatches"] * 1024.0
        res["calibration"] = {"kernel": bb[0], "known_read_bytes": volume_bytes,
                              "FETCH_SIZE_bytes_raw": raw, "measured_factor": volume_bytes / raw}
    ks = [k for k in res["kernels"] if k.startswith(PASS)]
    fetch = sum(res["kernels"][k].get("FETCH_SIZE", {}).get("total_KiB", 0.0) for k in ks) * 1024.0 / frames
    write = sum(res["kernels"][k].get("WRITE_SIZE", {}).get("total_KiB", 0.0) for k in ks) * 1024.0 / frames
    res["pass_kernels"] = ks
    res["fetch_bytes_raw_per_pass"] = fetch
    res["write_bytes_per_pass"] = write
    res["fetch_correction"] = factor
    res["hbm_bytes_per_pass"] = factor * fetch + write
    res["note"] = ("per frame: FETCH_SIZE x2 (gfx950 rule for wide streams; byte gathers uncalibrated, so an "
                   "estimate) + WRITE_SIZE, summed over every launch of the ray-cast pass in the region and "
                   "divided by the frames those launches rendered")
    # the run the counters belong to (bench.py --profile-region's own JSON line: workload, viewport, view,
    # renderers in flight, frames per launch set, round budget, hash of the kernel sources): bench.py
    # uses a profile only for a run with the same meta
    meta_path = os.path.join(prof, "bench.json")
    if os.path.exists(meta_path):
        res["meta"] = json.load(open(meta_path))
    allj = {}
    if os.path.exists(out_path):
        allj = json.load(open(out_path))
    allj[workload] = res
    json.dump(allj, open(out_path, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("frames", "fetch_bytes_raw_per_pass", "write_bytes_per_pass",
                                          "hbm_bytes_per_pass", "calibration") if k in res}, indent=1))


if __name__ == "__main__":
    main()
