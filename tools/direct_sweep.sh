#!/bin/bash
# tools/direct_sweep.sh [WORKLOAD] -- one frame at a time under VRHIP_DIRECT_MIN = 0 (off) and a range of thresholds
WL=${1:-shells2048}
for T in 0 8 12 16 24 32 40; do
  echo "== VRHIP_DIRECT_MIN=$T"
  VRHIP_DIRECT_MIN=$T python3 tools/costmap.py $WL 2>&1 | head -1
done
