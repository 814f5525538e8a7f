R=r3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p "$OUT"
cd "$ROOT"
bash tools/profile_region.sh ${R}_shells2048_single shells2048 --warmup 3 --steps 32 --frames-in-flight 1 --frames-per-launch 1 > "$OUT/region_shells2048_single.log" 2>&1
D=$ROOT/gpurun_out/region_${R}_shells2048_single
cp "$D/stats.csv" "$OUT/shells2048_single_kernel_stats.csv"; cp "$D/bench.json" "$OUT/shells2048_single_region.json"; cp "$D/issue.json" "$OUT/pmc_issue_shells2048_single.json"; cp "$D/traffic.json" "$OUT/pmc_traffic_shells2048_single.json"
mkdir -p profiles/$R; cp "$OUT"/pmc_issue_shells2048_single.json "$OUT"/pmc_traffic_shells2048_single.json profiles/$R/
cp gpurun_out/profiles_$R/pmc_*.json profiles/$R/ 2>/dev/null
timeout -k 10 400 python3 bench.py --frames-in-flight 1 --frames-per-launch 1 --steps 32 --warmup 3 --out-json "$OUT/bench_shells2048_single.json" > /dev/null 2> "$OUT/bench_shells2048_single.err"; echo rc=$?
python3 -c "
import json; d=json.load(open('$OUT/bench_shells2048_single.json')); print(d['ms_per_step'], d['roofline']['frac'], (d.get('roofline_valu_issue') or {}).get('frac'))"
