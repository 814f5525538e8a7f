#!/bin/bash
# tools/march_sweep.sh WORKLOAD "MICRO FILL REFILL" ... -- march-kernel knobs (env), single frame and throughput
WL=$1; shift
for cfg in "$@"; do
  set -- $cfg
  export VRHIP_MARCH_MICRO=$1 VRHIP_MARCH_FILL=$2 VRHIP_REFILL_MIN=$3
  python3 bench.py --workload $WL --no-cpu-baseline --steps 32 --warmup 2 --frames-in-flight 1 --frames-per-launch 1 --out-json /tmp/s.json > /dev/null 2>&1
  python3 bench.py --workload $WL --no-cpu-baseline --steps 64 --warmup 2 --out-json /tmp/t.json > /dev/null 2>&1
  python3 -c "
import json; a=json.load(open('/tmp/s.json')); b=json.load(open('/tmp/t.json')); print('$WL micro $1 fill $2 refill_min $3: single %.3f ms  throughput %.3f ms' % (a['ms_per_step'], b['ms_per_step']))"
done
