#!/bin/bash
# tools/ab_march.sh TAG WORKLOAD... -- new march vs two-phase march (VRHIP_NO_MARCH=1), one frame at
# a time and in the default throughput mode, per workload.  Output: gpurun_out/ab_TAG/.
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_$TAG
mkdir -p "$OUT"
for WL in "$@"; do
  for MODE in march old; do
    if [ $MODE = march ]; then export VRHIP_MARCH=1; else unset VRHIP_MARCH; fi
    timeout -k 10 200 python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --steps 32 --warmup 2 --frames-in-flight 1 --frames-per-launch 1 --out-json $OUT/${WL}_${MODE}_single.json > $OUT/${WL}_${MODE}_single.log 2>&1 || exit 1
    timeout -k 10 200 python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --steps 64 --warmup 2 --out-json $OUT/${WL}_${MODE}_tput.json > $OUT/${WL}_${MODE}_tput.log 2>&1 || exit 1
    python3 - <<PY
import json
a=json.load(open("$OUT/${WL}_${MODE}_single.json")); b=json.load(open("$OUT/${WL}_${MODE}_tput.json"))
print("%-16s %-6s single %.3f ms   throughput %.3f ms  (serial %.3f)" % ("$WL", "$MODE", a["ms_per_step"], b["ms_per_step"], b["roofline"]["serial_launch_ms"] or 0))
PY
  done
done
