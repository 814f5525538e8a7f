#!/bin/bash
# tools/ab_libs2.sh VARIANT REPS [bench args...] -- same-box A/B of the product library against _variants/libvrhip_VARIANT.so:
# `bench.py --profile-region ARGS` alternating between the two, ms per step of every run
V=$1; REPS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq $REPS); do
  for v in default $V; do
    if [ $v = default ]; then unset VRHIP_LIB_PATH; else export VRHIP_LIB_PATH=$ROOT/volumerenderercl_amd/_variants/libvrhip_$v.so; fi
    python3 $ROOT/bench.py --profile-region "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s %.4f ms/step' % ('$v', d['ms_per_step']))"
  done
done
