#!/bin/bash
# tools/short_run_sweep.sh [STEPS] -- the driver's short run (--steps STEPS --warmup 5, default 20) under other
# schedules: renderers in flight x frames per launch set x phase-1 round budget; two runs each
K=${1:-20}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "2 32 48" "1 32 48" "1 20 32" "2 5 48" "4 5 48" "3 7 48" "2 10 32" "2 10 24" "2 10 64" "4 32 48" "1 10 48"; do
  set -- $cfg
  for rep in 1 2; do
    python3 $ROOT/bench.py --profile-region --steps $K --warmup 5 --frames-in-flight $1 --frames-per-launch $2 --round-budget $3 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fif %s fpl %-3s budget %-3s -> sets of %d: %.4f ms/step' % ('$1', '$2', '$3', d['frames_per_launch'], d['ms_per_step']))"
  done
done
