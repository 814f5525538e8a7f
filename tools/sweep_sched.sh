for cfg in "2 8 48" "2 16 48" "3 8 48" "4 8 48" "2 8 64" "2 12 48" "3 6 48" "4 4 32"; do
  set -- $cfg
  python3 bench.py --workload shells2048 --no-cpu-baseline --steps 96 --warmup 2 --frames-in-flight $1 --frames-per-launch $2 --round-budget $3 --out-json /tmp/t.json > /dev/null 2>&1
  python3 -c "
import json; b=json.load(open('/tmp/t.json')); print('fif $1 fpl $2 budget $3: %.4f ms' % b['ms_per_step'])"
done
