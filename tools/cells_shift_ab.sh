#!/bin/bash
# tools/cells_shift_ab.sh -- empty bits on cells of 8 voxels (VRHIP_CELL_SHIFT=3: one grid) against 4 (default):
# build time, path tracer, ray caster at 256^3 ... 2048^3
for s in 3 2; do
  export VRHIP_CELL_SHIFT=$s
  echo "== empty-bit cells of $((1 << s)) voxels"
  python3 tools/cells_time.py 2>&1 | grep stream
  python3 bench.py --workload pt1024f --no-cpu-baseline --steps 16 --warmup 2 --out-json /tmp/p.json > /dev/null 2>&1
  python3 -c "import json; a=json.load(open('/tmp/p.json')); print('pt1024f: %.3f ms per spp' % a['ms_per_step'])"
  for w in shells2048 haze2048 sphere2048 shells1024u16 sphere256; do bash tools/ab_env.sh $w "VRHIP_X=1"; done
done
