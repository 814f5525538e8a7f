#!/usr/bin/env python3
"""tools/gaps.py KERNEL_TRACE.csv [SKIP_FRAMES] -- where one frame's time goes on the device when frames are
rendered one at a time: per kernel of the ray-cast pass its mean duration, and the mean idle gap between
the end of a launch and the start of the next one of the same frame, from a rocprofv3 --kernel-trace csv
of `bench.py --profile-region --frames-in-flight 1 --frames-per-launch 1`."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"vr_\w+", name) or re.search(r"\w+", name)
    return m.group(0) if m else name


rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]))
               for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
# a frame = everything from one vr_dda_prepass_kernel (or the launch before it that is not a vr_ kernel) to the next
starts = [i for i, r in enumerate(rows) if r[2] == "vr_dda_prepass_kernel"]
dur = defaultdict(list)
gap = defaultdict(list)
frame_t = []
for a, b in zip(starts[skip:-1], starts[skip + 1:]):
    fr = rows[a:b]
    for k, (s, e, n) in enumerate(fr):
        dur[n].append(e - s)
        if k + 1 < len(fr):
            gap["%s -> %s" % (n, fr[k + 1][2])].append(fr[k + 1][0] - e)
    gap["(last launch of the frame) -> next vr_dda_prepass_kernel"].append(rows[b][0] - fr[-1][1])
    frame_t.append(rows[b][0] - rows[a][0])
print("frames %d, mean frame period %.1f us" % (len(frame_t), sum(frame_t) / len(frame_t) / 1e3))
tk = tg = 0.0
for n, v in dur.items():
    print("  kernel %-40s %8.1f us" % (n, sum(v) / len(v) / 1e3)); tk += sum(v) / len(frame_t) / 1e3
for n, v in gap.items():
    print("  gap    %-70s %8.1f us" % (n, sum(v) / len(v) / 1e3)); tg += sum(v) / len(frame_t) / 1e3
print("kernels %.1f us + gaps %.1f us per frame" % (tk, tg))
