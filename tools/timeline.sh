#!/bin/bash
# tools/timeline.sh TAG [bench args...] -- kernel trace of one frame at a time + tools/timeline.py
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/tl_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 $ROOT/bench.py --profile-region --frames-in-flight 1 --frames-per-launch 1 --steps 24 --warmup 4 "$@" > "$OUT/trace.log" 2>&1
F=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python3 "$ROOT/tools/timeline.py" "$F" 6
