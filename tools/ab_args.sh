#!/bin/bash
# tools/ab_args.sh WORKLOAD REPS "bench args A" "bench args B" ... -- bench.py under sets of arguments, alternating, REPS times
WL=$1; REPS=$2; shift 2
for rep in $(seq $REPS); do
  for a in "$@"; do
    python3 bench.py --workload $WL --no-cpu-baseline $a --out-json /tmp/t.json > /dev/null 2>&1
    python3 -c "
import json; d=json.load(open('/tmp/t.json')); print('$WL [$a]: %.4f ms' % d['ms_per_step'])"
  done
done
