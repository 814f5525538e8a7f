#!/bin/bash
# tools/ktrace.sh TAG [bench args] -- rocprofv3 kernel trace + stats only
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kt_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 $* > "$OUT/trace.log" 2>&1
python3 "$ROOT/tools/summarize_prof.py" "$OUT" | grep calls= | grep vr_ | sed 's/ \+/ /g' | cut -c1-170
