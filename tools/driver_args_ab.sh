#!/bin/bash
# tools/driver_args_ab.sh WORKLOAD "VAR=val ..." ... -- the round driver's own command (--steps 20 --warmup 5) under sets of env vars
WL=$1; shift
for cfg in "$@"; do
  ( export $cfg
    for rep in 1 2 3; do
    python3 bench.py --workload $WL --no-cpu-baseline --steps 20 --warmup 5 --out-json /tmp/t.json > /dev/null 2>&1
    python3 -c "
import json; a=json.load(open('/tmp/t.json')); print('$WL steps 20 [$cfg]: %.4f ms' % a['ms_per_step'])"
    done )
done
