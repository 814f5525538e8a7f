#!/bin/bash
# tools/ab_libs.sh WORKLOAD REPS "bench args" VARIANT... -- bench.py with the default build and variant builds
# (_variants/libvrhip_<VARIANT>.so), alternating, REPS times
WL=$1; REPS=$2; ARGS=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq $REPS); do
  for v in default "$@"; do
    if [ "$v" = default ]; then unset VRHIP_LIB_PATH; else export VRHIP_LIB_PATH=$ROOT/volumerenderercl_amd/_variants/libvrhip_$v.so; fi
    python3 bench.py --workload $WL --no-cpu-baseline $ARGS --out-json /tmp/t.json > /dev/null 2>&1
    python3 -c "
import json; d=json.load(open('/tmp/t.json')); print('$WL [$ARGS] %-10s %.4f ms' % ('$v', d['ms_per_step']))"
  done
done
