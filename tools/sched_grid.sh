#!/bin/bash
# tools/sched_grid.sh [WORKLOAD] -- throughput schedule grid: round budget x frames per launch set x renderers in flight
WL=${1:-shells2048}
for cfg in "32 16 2" "32 16 2" "36 16 2" "40 16 2" "28 16 2" "32 32 2" "32 16 3" "32 8 4" "40 32 2" "48 16 2"; do
  set -- $cfg
  python3 bench.py --workload $WL --no-cpu-baseline --steps 192 --warmup 2 --round-budget $1 --frames-per-launch $2 --frames-in-flight $3 --out-json /tmp/t.json > /dev/null 2>&1
  python3 -c "
import json; a=json.load(open('/tmp/t.json')); print('$WL budget $1, $2 frames per set, $3 in flight: %.4f ms' % a['ms_per_step'])"
done
