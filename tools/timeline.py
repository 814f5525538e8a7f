#!/usr/bin/env python3
"""tools/timeline.py KERNEL_TRACE.csv [SKIP_FRAMES] -- one frame at a time: when does every launch of a frame start
and end, relative to the start of the frame's pre-pass (means over the frames of a rocprofv3 --kernel-trace csv of
`bench.py --profile-region --frames-in-flight 1 --frames-per-launch 1`)?  Launches on a second stream (e.g. an experiment with
routing) overlap the others: durations alone (tools/gaps.py) do not show that."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"vr_\w+(<[^>]*>)?", name)
    return m.group(0) if m else name


rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]))
               for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
starts = [i for i, r in enumerate(rows) if r[2].startswith("vr_dda_prepass_kernel")]
acc = defaultdict(lambda: [0.0, 0.0, 0])
period = []
for a, b in zip(starts[skip:-1], starts[skip + 1:]):
    t0 = rows[a][0]
    seen = defaultdict(int)
    for s, e, n in rows[a:b]:
        seen[n] += 1
        k = "%s #%d" % (n, seen[n])
        acc[k][0] += (s - t0) / 1e3
        acc[k][1] += (e - t0) / 1e3
        acc[k][2] += 1
    period.append((rows[b][0] - t0) / 1e3)
print("frames %d, mean frame period %.1f us" % (len(period), sum(period) / max(1, len(period))))
for k, (s, e, n) in sorted(acc.items(), key=lambda kv: kv[1][0] / kv[1][2]):
    print("  %-110s start %7.1f  end %7.1f  (%6.1f us, %d launches)" % (k[:110], s / n, e / n, (e - s) / n, n))
