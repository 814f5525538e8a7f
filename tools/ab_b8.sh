#!/bin/bash
# tools/ab_b8.sh -- eight samples per lane and round (variant b8: -DVR_BATCH=8; skip bitmap from L2: the sample staging
# doubles) against the default four
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for wl in shells2048 haze2048 sphere256; do
  bash tools/ab_env.sh $wl "A=default" "VRHIP_SKIP_GLOBAL=1 VRHIP_LIB_PATH=$ROOT/volumerenderercl_amd/_variants/libvrhip_b8.so" "VRHIP_ROUND_BUDGET=5 VRHIP_SKIP_GLOBAL=1 VRHIP_LIB_PATH=$ROOT/volumerenderercl_amd/_variants/libvrhip_b8.so"
done
