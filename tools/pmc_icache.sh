#!/bin/bash
# tools/pmc_icache.sh [bench args] -- instruction-cache and issue-stall counters of the ray-cast kernels
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_icache
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 2 $*"
i=0
for PMC in \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
  "SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/p$i" -- $BENCH > "$OUT/p$i.log" 2>&1
  echo "p$i rc=$? ($PMC)"
done
python3 "$ROOT/tools/summarize_prof.py" "$OUT" 2>/dev/null | grep -E "vr_raycast(_split)?_kernel<[a-z ]+, (true|false), 0|vr_dda" 
