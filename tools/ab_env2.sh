#!/bin/bash
# tools/ab_env2.sh "ENV_A" "ENV_B" REPS [bench args...] -- same-box A/B of `bench.py --profile-region ARGS` under two
# environments ("-" = none), alternating
A=$1; B=$2; REPS=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq $REPS); do
  for e in "$A" "$B"; do
    if [ "$e" = "-" ]; then ev=""; else ev="$e"; fi
    env $ev python3 $ROOT/bench.py --profile-region "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-22s %.4f ms/step' % ('$e', d['ms_per_step']))"
  done
done
