#!/bin/bash
# tools/ab_variants.sh WORKLOAD VARIANT... -- like ab_env.sh for variant builds (_variants/libvrhip_<V>.so)
WL=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in default "$@"; do
  if [ "$v" = default ]; then unset VRHIP_LIB_PATH; else export VRHIP_LIB_PATH=$ROOT/volumerenderercl_amd/_variants/libvrhip_$v.so; fi
  bash tools/ab_env.sh $WL "VRHIP_VARIANT=$v"
done
