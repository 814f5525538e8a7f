#!/bin/bash
# tools/driver_args_sweep.sh -- the round driver's own command (--steps 20 --warmup 5) under schedule settings
WL=${1:-shells2048}
for cfg in "48 16 2" "32 16 2" "36 16 2" "32 16 3" "32 16 4" "32 32 1" "24 16 2" "32 16 2" "36 16 3"; do
  set -- $cfg
  for rep in 1 2; do
  python3 bench.py --workload $WL --no-cpu-baseline --steps 20 --warmup 5 --round-budget $1 --frames-per-launch $2 --frames-in-flight $3 --out-json /tmp/t.json > /dev/null 2>&1
  python3 -c "
import json; a=json.load(open('/tmp/t.json')); print('$WL steps 20: budget $1, <= $2 frames per set, $3 in flight: %.4f ms' % a['ms_per_step'])"
  done
done
