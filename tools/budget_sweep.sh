#!/bin/bash
# tools/budget_sweep.sh -- one frame at a time under phase-1 round budgets
for b in 4 6 8 10 14 20; do
  VRHIP_ROUND_BUDGET=$b python3 bench.py --workload shells2048 --no-cpu-baseline --steps 32 --warmup 2 --frames-in-flight 1 --frames-per-launch 1 --out-json /tmp/s.json > /dev/null 2>&1
  python3 -c "
import json; a=json.load(open('/tmp/s.json')); print('budget $b: single %.3f ms' % a['ms_per_step'])"
done
