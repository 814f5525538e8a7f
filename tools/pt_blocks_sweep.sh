#!/bin/bash
# tools/pt_blocks_sweep.sh -- path tracer (pt1024f) under persistent workgroups per CU
for b in 0 2 3 4 5 6 8; do
  if [ $b = 0 ]; then unset VRHIP_BLOCKS_PER_CU; else export VRHIP_BLOCKS_PER_CU=$b; fi
  python3 bench.py --workload pt1024f --no-cpu-baseline --steps 32 --warmup 2 --out-json /tmp/p.json > /dev/null 2>&1
  python3 -c "import json; a=json.load(open('/tmp/p.json')); print('blocks/CU $b: %.3f ms per spp' % a['ms_per_step'])"
done
