#!/bin/bash
# tools/ab_env.sh WORKLOAD "VAR=val ..." ... -- bench.py (single frame and throughput) under sets of env vars
WL=$1; shift
for cfg in "$@"; do
  ( export $cfg
    python3 bench.py --workload $WL --no-cpu-baseline --steps 32 --warmup 2 --frames-in-flight 1 --frames-per-launch 1 --out-json /tmp/s.json > /dev/null 2>&1
    python3 bench.py --workload $WL --no-cpu-baseline --steps 64 --warmup 2 --out-json /tmp/t.json > /dev/null 2>&1
    python3 -c "
import json; a=json.load(open('/tmp/s.json')); b=json.load(open('/tmp/t.json')); print('$WL [$cfg]: single %.3f ms  throughput %.3f ms' % (a['ms_per_step'], b['ms_per_step']))" )
done
