#!/bin/bash
# tools/profile_region.sh TAG WORKLOAD [bench args...] -- run on the GPU box (via gpurun).
# Profiles `python3 bench.py --profile-region --workload WORKLOAD ARGS`: that command runs the
# warm-up and the timed launches and nothing else, so the kernel stats and counters are those of
# the measured schedule.  One rocprofv3 --kernel-trace --stats run, then PMC passes each in a run
# of its own (never combined with trace domains other than --kernel-trace).
# Output: gpurun_out/region_TAG/ {trace/, pmc1..4/, bench.json, issue.json, stats.csv}
set -u
TAG=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/region_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --profile-region --workload $WL $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH --out-json "$OUT/bench.json" > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/stats.csv" \;
i=0
for PMC in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
  "FETCH_SIZE TCC_HIT_sum" \
  "WRITE_SIZE TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i rc=$? ($PMC)"
done
FRAMES=$(python3 -c "
import json,sys
d=json.load(open('$OUT/bench.json'))
thr = d['frames_in_flight']>1 or d['frames_per_launch']>1
print(d['warmup'] + (d['frames_in_flight']*min(d['frames_per_launch'], d['steps']) if thr else 0) + d['steps'])")
python3 "$ROOT/tools/pmc_issue.py" "$OUT" "$WL" "$FRAMES" "$OUT/issue.json"
VB=$(python3 -c "
import sys; sys.path.insert(0, '$ROOT'); import bench
w = bench.WORKLOADS['$WL']; print(w[1] ** 3 * bench.FMT_BYTES[w[2]])")
python3 "$ROOT/tools/pmc_traffic.py" "$OUT" "$WL" "$FRAMES" "$OUT/traffic.json" "$VB"
