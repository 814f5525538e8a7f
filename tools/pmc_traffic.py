#!/usr/bin/env python3
"""tools/pmc_traffic.py PROF_DIR WORKLOAD OUT_JSON -- HBM traffic of the ray-cast pass from the
rocprofv3 --pmc passes that tools/profile.sh wrote (FETCH_SIZE and WRITE_SIZE are collected in
SEPARATE passes: they do not fit one TCC pass together).

Correction per MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE tallies 128-byte requests at 64 bytes, i.e. reads x2 for wide coalesced streams.  The
same run contains a kernel with a KNOWN byte count in that regime -- vr_build_bricks streams the
whole volume once with 16-byte loads -- which is used to check the factor; the ray-cast kernels
gather single bytes (an access width the guide calls uncalibrated), so their corrected number is
an estimate and the raw counter value is kept beside it."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(vr_\w+)(<.*?>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def main():
    prof, workload, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
    volume_bytes = float(sys.argv[4]) if len(sys.argv) > 4 else None
    acc = defaultdict(lambda: defaultdict(list))
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
            acc[meta[key]][key[1]].append(v)
    res = {"source": os.path.basename(os.path.normpath(prof)), "kernels": {}}
    for k, d in acc.items():
        res["kernels"][k] = {c: {"avg_KiB_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
                             for c, v in d.items()}
    factor = 2.0
    bb = [k for k in res["kernels"] if k.startswith("vr_build_bricks")]
    if bb and volume_bytes:
        raw = res["kernels"][bb[0]]["FETCH_SIZE"]["avg_KiB_per_dispatch"] * 1024.0
        res["calibration"] = {"kernel": bb[0], "known_read_bytes": volume_bytes,
                              "FETCH_SIZE_bytes_raw": raw, "measured_factor": volume_bytes / raw}
    # every launch of the timed (un-instrumented) pass: pre-pass, phase 1, sort, phase 2
    p1 = [k for k in res["kernels"] if (k.startswith("vr_raycast_kernel") and ", 0, " in k)
          or k.startswith("vr_dda_prepass_kernel") or k.startswith("vr_pathtrace_kernel") and ", 0>" in k]
    p2 = [k for k in res["kernels"] if (k.startswith("vr_raycast_split_kernel") and ", 0, " in k)
          or k.startswith("vr_cont_")]
    fetch = sum(res["kernels"][k]["FETCH_SIZE"]["avg_KiB_per_dispatch"] for k in p1 + p2) * 1024.0
    write = sum(res["kernels"][k]["WRITE_SIZE"]["avg_KiB_per_dispatch"] for k in p1 + p2) * 1024.0
    res["fetch_bytes_raw_per_pass"] = fetch
    res["write_bytes_per_pass"] = write
    res["fetch_correction"] = factor
    res["hbm_bytes_per_pass"] = factor * fetch + write
    res["note"] = ("FETCH_SIZE x2 (gfx950 rule for wide streams; byte gathers uncalibrated, so an "
                   "estimate) + WRITE_SIZE, timed kernel variants only, both launches of the pass")
    allj = {}
    if os.path.exists(out_path):
        allj = json.load(open(out_path))
    allj[workload] = res
    json.dump(allj, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
