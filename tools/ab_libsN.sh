#!/bin/bash
# tools/ab_libsN.sh REPS VARIANT... -- [bench args...]: same-box comparison of the product library ("default") and
# _variants/libvrhip_VARIANT.so, round robin: ms per step of `bench.py --profile-region ARGS`
REPS=$1; shift
VARS=()
while [ "$1" != "--" ]; do VARS+=("$1"); shift; done
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export VRHIP_BENCH_ALLOW_STALE=1
for i in $(seq $REPS); do
  for v in "${VARS[@]}"; do
    if [ $v = default ]; then unset VRHIP_LIB_PATH; else export VRHIP_LIB_PATH=$ROOT/volumerenderercl_amd/_variants/libvrhip_$v.so; fi
    python3 $ROOT/bench.py --profile-region "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %.4f ms/step' % ('$v', d['ms_per_step']))"
  done
done
