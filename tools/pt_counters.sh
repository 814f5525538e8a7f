#!/bin/bash
# tools/pt_counters.sh TAG [bench args...] -- instruction counters of the path tracer's kernel (one PMC pass of
# `bench.py --profile-region ARGS`): VALU / SALU / VMEM wave-instructions and waves per launch
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ptc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc" -- python3 $ROOT/bench.py --profile-region "$@" > "$OUT/pmc.log" 2>&1
F=$(find "$OUT/pmc" -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "pathtrace" not in k and "raycast" not in k and "prepass" not in k:
        continue
    k = k.split("(")[0][-60:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen:
        seen.add(r["Dispatch_Id"]); n[k] += 1
for k in acc:
    print(k, "launches", n[k], {c: "%.4g" % (v / n[k]) for c, v in acc[k].items()})
PY
