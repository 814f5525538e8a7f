#!/bin/bash
# tools/pt_batch_ab.sh -- path tracer under variant builds of VR_PT_BATCH (tools/mkvariant.sh ptbN -DVR_PT_BATCH=N): flat, 3: +4 %
for v in default ptb3 ptb6 ptb8; do
  if [ $v = default ]; then unset VRHIP_LIB_PATH; else export VRHIP_LIB_PATH=$PWD/volumerenderercl_amd/_variants/libvrhip_$v.so; fi
  python3 bench.py --workload pt1024f --no-cpu-baseline --steps 32 --warmup 2 --out-json /tmp/p.json > /dev/null 2>&1
  python3 -c "import json; a=json.load(open('/tmp/p.json')); print('$v: %.3f ms per spp' % a['ms_per_step'])"
done
