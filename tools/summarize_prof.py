#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: per-kernel stats + PMC sums per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    import re
    m = re.search(r"(vr_\w+)(<.*?>)?\(", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    return name.split("(")[0][-90:]


for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats (%s)" % os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-80s calls=%s total_ns=%s avg_ns=%s pct=%s" % (
            short(row.get("Name", "")), row.get("Calls"), row.get("TotalDurationNs"),
            row.get("AverageNs"), row.get("Percentage")))

for d in sorted(glob.glob(os.path.join(out, "p*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        for row in csv.DictReader(open(f)):
            k = short(row.get("Kernel_Name", ""))
            acc[k][row.get("Counter_Name")] += float(row.get("Counter_Value", 0) or 0)
            cnt[(k, row.get("Counter_Name"))] += 1
        print("== %s" % os.path.relpath(f, out))
        for k, d2 in acc.items():
            for c, v in sorted(d2.items()):
                n = cnt[(k, c)]
                print("  %-60s %-34s sum=%.6g per_dispatch=%.6g (n=%d)" % (k[-60:], c, v, v / n, n))
