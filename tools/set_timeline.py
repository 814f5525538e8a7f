#!/usr/bin/env python3
"""tools/set_timeline.py KERNEL_TRACE.csv [LAST_N] -- the last LAST_N (default 12) ray-cast launches of a kernel trace with
their start / end relative to the first of them, and the stream (queue) they ran on: the timed launch sets of a short
`bench.py --profile-region` run."""
import csv
import re
import sys

rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.search(r"vr_\w+(<[^>]*>)?", r["Kernel_Name"]),
                r.get("Queue_Id", "?")) for r in csv.DictReader(open(sys.argv[1]))), key=lambda x: x[0])
rows = [(s, e, m.group(0), q) for s, e, m, q in rows if m and re.match(r"vr_(dda|raycast|cont)", m.group(0))]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = rows[-n:]
t0 = rows[0][0]
for s, e, k, q in rows:
    print("  queue %-3s %-75s start %8.1f  end %8.1f  (%7.1f us)" % (q, k[:75], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
print("  span %.1f us" % ((max(e for _, e, _, _ in rows) - t0) / 1e3))
