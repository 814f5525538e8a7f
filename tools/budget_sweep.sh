#!/bin/bash
# tools/budget_sweep.sh [WORKLOAD] -- one frame at a time under phase-1 round budgets
WL=${1:-shells2048}
for b in 1 2 3 4 6 8 10 14; do
  VRHIP_ROUND_BUDGET=$b python3 bench.py --workload $WL --no-cpu-baseline --steps 32 --warmup 2 --frames-in-flight 1 --frames-per-launch 1 --out-json /tmp/s.json > /dev/null 2>&1
  python3 -c "
import json; a=json.load(open('/tmp/s.json')); p=a['roofline']['last_pass_ms_hip_events']; print('$WL budget $b: single %.3f ms (phase 1 %.3f, phase 2 %.3f)' % (a['ms_per_step'], p['phase1'], p['phase2']))"
done
