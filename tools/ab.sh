#!/bin/bash
# tools/ab.sh WORKLOAD VARIANT... -- bench the default build and variant builds of libvrhip
# (volumerenderercl_amd/_variants/libvrhip_<VARIANT>.so) on the GPU box; prints ms/step.
WL=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in default "$@"; do
  if [ "$v" = default ]; then unset VRHIP_LIB_PATH; else export VRHIP_LIB_PATH=$ROOT/volumerenderercl_amd/_variants/libvrhip_$v.so; fi
  python3 $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 3 --workload $WL 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$WL','$v','ms/step=%.3f kernel_ms=%.3f Msamples/s=%.0f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['value']))"
done
