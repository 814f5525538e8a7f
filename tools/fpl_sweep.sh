#!/bin/bash
# tools/fpl_sweep.sh -- the round driver's command (--steps 20 --warmup 5) under frames per launch set and renderers in flight
for fpl in 3 4 5 7 10 16; do
  python3 bench.py --workload shells2048 --no-cpu-baseline --steps 20 --warmup 5 --frames-per-launch $fpl --out-json /tmp/t.json > /dev/null 2>&1
  python3 -c "import json; a=json.load(open('/tmp/t.json')); print('fpl $fpl: %.3f ms' % a['ms_per_step'])"
done
for fif in 3 4; do for fpl in 3 5 7; do
  python3 bench.py --workload shells2048 --no-cpu-baseline --steps 20 --warmup 5 --frames-in-flight $fif --frames-per-launch $fpl --out-json /tmp/t.json > /dev/null 2>&1
  python3 -c "import json; a=json.load(open('/tmp/t.json')); print('fif $fif fpl $fpl: %.3f ms' % a['ms_per_step'])"
done; done
