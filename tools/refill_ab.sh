#!/bin/bash
# tools/refill_ab.sh -- idle ray slots per wave before a refill (VRHIP_REFILL_MIN), both regimes
for RM in 16 8 4 2 1; do
  export VRHIP_REFILL_MIN=$RM
  python3 bench.py --workload shells2048 --no-cpu-baseline --steps 32 --warmup 2 --frames-in-flight 1 --frames-per-launch 1 --out-json /tmp/s.json > /dev/null 2>&1
  python3 bench.py --workload shells2048 --no-cpu-baseline --steps 64 --warmup 2 --out-json /tmp/t.json > /dev/null 2>&1
  python3 -c "
import json; a=json.load(open('/tmp/s.json')); b=json.load(open('/tmp/t.json')); print('refill_min $RM: single %.3f ms  throughput %.3f ms' % (a['ms_per_step'], b['ms_per_step']))"
done
