#!/bin/bash
# tools/budget_sweep_tput.sh [WORKLOAD] -- throughput mode (2 renderers x 16 frames per set) under phase-1 round budgets
WL=${1:-shells2048}
for b in 24 32 48 64 96; do
  python3 bench.py --workload $WL --no-cpu-baseline --steps 64 --warmup 2 --round-budget $b --out-json /tmp/t.json > /dev/null 2>&1
  python3 -c "
import json; a=json.load(open('/tmp/t.json')); print('$WL throughput, budget $b: %.4f ms' % a['ms_per_step'])"
done
