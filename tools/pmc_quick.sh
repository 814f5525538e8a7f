#!/bin/bash
# tools/pmc_quick.sh TAG [bench args...] -- two PMC passes (instruction mix, wait/active cycles)
# of `python3 bench.py ARGS` into gpurun_out/pmcq_TAG/, summarised for the vr_* kernels.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 $*"
i=0
for PMC in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i rc=$?"
done
python3 "$ROOT/tools/summarize_prof.py" "$OUT" 2>/dev/null | grep -E "vr_raycast(_split)?_kernel<[a-z ]+, (true|false), 0" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
