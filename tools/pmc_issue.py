#!/usr/bin/env python3
"""tools/pmc_issue.py PROF_DIR WORKLOAD FRAMES OUT_JSON -- where the cycles of the ray-cast pass
go, from the rocprofv3 --pmc passes tools/profile_region.sh wrote for `bench.py --profile-region`
(every ray-cast dispatch of that command renders frames of the same schedule; FRAMES = all the
frames it rendered: warm-up included).

Per kernel of the pass and summed: VALU wave-instructions (SQ_INSTS_VALU), cycles with a VALU
instruction issuing (SQ_ACTIVE_INST_VALU, quad-cycles), wave residency (SQ_WAVE_CYCLES,
quad-cycles), SQ busy cycles, waits, lane utilisation (SQ_THREAD_CYCLES_VALU / 64 /
SQ_ACTIVE_INST_VALU... see `derived`), LDS bank conflicts -- all divided by FRAMES."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

PASS = ("vr_dda_prepass_kernel", "vr_raycast_rays_kernel", "vr_raycast_kernel", "vr_raycast_split_kernel",
        "vr_cont_hist_kernel", "vr_cont_scatter_kernel", "vr_march_kernel", "vr_pathtrace_kernel")


def short(name):
    m = re.search(r"(vr_\w+)(<.*?>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def main():
    prof, workload, frames, out_path = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
    acc = defaultdict(lambda: defaultdict(float))
    ndisp = defaultdict(set)
    regs = {}
    for f in glob.glob(os.path.join(prof, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if not k.startswith(PASS):
                continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"] or 0)
            ndisp[k].add((f, row["Dispatch_Id"]))
            regs[k] = {"vgpr": row.get("VGPR_Count"), "accum_vgpr": row.get("Accum_VGPR_Count"),
                       "sgpr": row.get("SGPR_Count"), "lds_block_bytes": row.get("LDS_Block_Size"),
                       "workgroup": row.get("Workgroup_Size"), "grid": row.get("Grid_Size")}
    res = {"source": os.path.basename(os.path.normpath(prof)), "frames": frames, "kernels": {}}
    tot = defaultdict(float)
    for k, d in sorted(acc.items()):
        per = {c: v / frames for c, v in sorted(d.items())}
        der = {}
        if per.get("SQ_ACTIVE_INST_VALU") and per.get("SQ_THREAD_CYCLES_VALU"):
            # active lanes per VALU instruction: thread-cycles over the cycles a VALU instruction was
            # issuing (the ratio rocprof-compute reports as "VALU active threads"; both counters run in
            # the same unit, and SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU is 1.00-1.02 on these kernels)
            der["active_lanes_per_valu_inst"] = min(64.0, per["SQ_THREAD_CYCLES_VALU"] / per["SQ_ACTIVE_INST_VALU"])
        if per.get("SQ_ACTIVE_INST_VALU") and per.get("SQ_WAVE_CYCLES"):
            der["valu_issuing_share_of_wave_cycles"] = per["SQ_ACTIVE_INST_VALU"] / per["SQ_WAVE_CYCLES"]
        if per.get("SQ_WAIT_ANY") and per.get("SQ_WAVE_CYCLES"):
            der["waiting_share_of_wave_cycles"] = per["SQ_WAIT_ANY"] / per["SQ_WAVE_CYCLES"]
        if per.get("SQ_LDS_BANK_CONFLICT") is not None and per.get("SQ_LDS_IDX_ACTIVE"):
            der["lds_bank_conflict_share"] = per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"]
        res["kernels"][k] = {"per_frame": per, "derived": der, "resources": regs.get(k),
                             "dispatches_per_pmc_pass": len(ndisp[k]) / max(1, len({f for f, _ in ndisp[k]}))}
        for c, v in per.items():
            tot[c] += v
    res["per_frame_total"] = dict(sorted(tot.items()))
    res["valu_wave_insts_per_frame"] = tot.get("SQ_INSTS_VALU", 0.0)
    if tot.get("SQ_THREAD_CYCLES_VALU") and tot.get("SQ_INSTS_VALU"):
        res["valu_lane_utilisation"] = min(1.0, tot["SQ_THREAD_CYCLES_VALU"] / max(tot["SQ_ACTIVE_INST_VALU"], 1.0) / 64.0)
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
    print(json.dumps({k: res[k] for k in ("frames", "valu_wave_insts_per_frame", "valu_lane_utilisation",
                                          "per_frame_total") if k in res}, indent=1))
    for k, v in res["kernels"].items():
        print(k, json.dumps(v["derived"]), json.dumps(v["resources"]))


if __name__ == "__main__":
    main()
