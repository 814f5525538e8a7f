#!/bin/bash
# tools/single_sweep.sh [WORKLOAD] -- one frame at a time under a set of schedule knobs (first line of tools/costmap.py)
WL=${1:-shells2048}
run() { echo "== $*"; env "$@" python3 tools/costmap.py $WL 2>&1 | head -1 | cut -c1-110; }
run A=0
run VRHIP_OCC_P2=3
run VRHIP_OCC_P1=3
run VRHIP_OCC=3
run VRHIP_REFILL_MIN=8
run VRHIP_REFILL_MIN=4
run VRHIP_REFILL_MIN=2
run VRHIP_REFILL_MIN=1
run VRHIP_ROUND_BUDGET=6
run VRHIP_ROUND_BUDGET=14
run VRHIP_ROUND_BUDGET=20
run VRHIP_NO_FOOTPRINT=1
run VRHIP_NO_SORT=1
run VRHIP_OCC_P2=3 VRHIP_REFILL_MIN=4
run VRHIP_OCC_P2=3 VRHIP_ROUND_BUDGET=16
