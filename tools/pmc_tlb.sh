#!/bin/bash
# tools/pmc_tlb.sh WORKLOAD -- address-translation counters of the ray-cast kernels with the
# plain layout and with the footprint volume forced on (VRHIP_FOOTPRINT_MAX_GB=100): evidence for
# the footprint-size cliff.  Run on the GPU box via gpurun.
WL=${1:-haze2048}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_tlb_$WL
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 --workload $WL"
PMC="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
export VRHIP_NO_FOOTPRINT=1
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/p_plain" -- $BENCH > "$OUT/plain.log" 2>&1
echo "plain rc=$?"
unset VRHIP_NO_FOOTPRINT
export VRHIP_FOOTPRINT_MAX_GB=100
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/p_fp" -- $BENCH > "$OUT/fp.log" 2>&1
echo "fp rc=$?"
python3 "$ROOT/tools/summarize_prof.py" "$OUT" 2>/dev/null | grep -E "^==|vr_raycast(_split)?_kernel<[a-z ]+, true, 0" 
