#!/bin/bash
# tools/ab_envN.sh REPS "ENV_A" "ENV_B" ... -- [bench args...]: same-box comparison of `bench.py --profile-region ARGS`
# under several environments ("-" = none), round robin; ms per step and the renderer's own event times of the last pass
REPS=$1; shift
ENVS=()
while [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq $REPS); do
  for e in "${ENVS[@]}"; do
    if [ "$e" = "-" ]; then ev=""; else ev="$e"; fi
    env $ev python3 $ROOT/bench.py --profile-region "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-44s %.4f ms/step   events of the last pass: %s' % ('$e', d['ms_per_step'], d.get('roofline',{}).get('last_pass_ms_hip_events')))"
  done
done
