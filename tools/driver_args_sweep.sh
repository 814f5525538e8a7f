#!/bin/bash
# tools/driver_args_sweep.sh -- the round driver's command (--steps 20 --warmup 5) under schedule settings
for rb in 24 32 48 64 96; do
  for rep in 1 2; do
  python3 bench.py --workload shells2048 --no-cpu-baseline --steps 20 --warmup 5 --round-budget $rb --out-json /tmp/t.json > /dev/null 2>&1
  python3 -c "import json; a=json.load(open('/tmp/t.json')); print('round budget $rb: %.3f ms' % a['ms_per_step'])"
  done
done
