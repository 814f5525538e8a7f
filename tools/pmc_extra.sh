#!/bin/bash
# tools/pmc_extra.sh TAG [bench args] -- extra SQ counters: instruction fetch, branches, LDS/VMEM latency levels
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcx_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 1 $*"
i=0
for PMC in \
  "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_MISC" \
  "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VALU" \
  "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/p$i" -- $BENCH > "$OUT/p$i.log" 2>&1
  echo "p$i rc=$?"
done
python3 "$ROOT/tools/summarize_prof.py" "$OUT" | grep "true, 0, true" | awk '{print $(NF-3), $(NF-1)}'
