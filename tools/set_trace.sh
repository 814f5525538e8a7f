#!/bin/bash
# tools/set_trace.sh TAG [bench args...] -- kernel trace of `bench.py --profile-region ARGS` + tools/set_timeline.py: start
# and end of the launches of the last launch set(s)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/st_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 $ROOT/bench.py --profile-region "$@" > "$OUT/trace.log" 2>&1
F=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python3 "$ROOT/tools/set_timeline.py" "$F" 6
