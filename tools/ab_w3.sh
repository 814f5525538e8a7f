#!/bin/bash
# tools/ab_w3.sh -- three waves per SIMD (variant w3: -DVR_WAVES_PER_EU=3, skip bitmap from L2 so that three
# workgroups fit the LDS) against the default two
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for wl in shells2048 haze2048 sphere256 shells1024u16; do
  bash tools/ab_env.sh $wl "A=default" "VRHIP_SKIP_GLOBAL=1" "VRHIP_SKIP_GLOBAL=1 VRHIP_LIB_PATH=$ROOT/volumerenderercl_amd/_variants/libvrhip_w3.so"
done
