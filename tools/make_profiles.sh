#!/bin/bash
# tools/make_profiles.sh ROUND -- run on the GPU box (via gpurun): everything profiles/<ROUND>/ keeps.
#  * rocprofv3 --kernel-trace --stats + PMC passes of `bench.py --profile-region` (timed launches only):
#    the default throughput schedule and one frame at a time, headline workload
#  * full bench.py JSON lines of the headline and of BASELINE configs 1, 2, 3, 5
R=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p "$OUT"
cd "$ROOT"
bash tools/profile_region.sh ${R}_tput shells2048 --warmup 0 --steps 64 > "$OUT/region_tput.log" 2>&1
bash tools/profile_region.sh ${R}_single shells2048 --warmup 1 --steps 32 --frames-in-flight 1 --frames-per-launch 1 > "$OUT/region_single.log" 2>&1
for t in tput single; do
  D=$ROOT/gpurun_out/region_${R}_$t
  cp "$D/stats.csv" "$OUT/shells2048_${t}_kernel_stats.csv"
  cp "$D/bench.json" "$OUT/shells2048_${t}_region.json"
  cp "$D/issue.json" "$OUT/pmc_issue_$t.json"
  cp "$D/traffic.json" "$OUT/pmc_traffic_$t.json"
done
echo "regions done"
# the committed PMC summaries feed bench.py's traffic / roofline_valu_issue fields
mkdir -p "$ROOT/profiles/$R"
cp "$OUT/pmc_issue_tput.json" "$ROOT/profiles/$R/pmc_issue.json"
cp "$OUT/pmc_traffic_tput.json" "$ROOT/profiles/$R/pmc_traffic.json"
timeout -k 10 400 python3 bench.py --out-json "$OUT/bench_shells2048.json" > /dev/null 2> "$OUT/bench_shells2048.err"; echo "shells2048 rc=$?"
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --out-json "$OUT/bench_shells2048_driver_args.json" > /dev/null 2> "$OUT/bench_shells2048_driver_args.err"; echo "shells2048 (driver arguments) rc=$?"
timeout -k 10 300 python3 bench.py --workload sphere256_plain --viewport 512 --out-json "$OUT/bench_sphere256_plain_512.json" > /dev/null 2> "$OUT/bench_sphere256_plain.err"; echo "config1 rc=$?"
timeout -k 10 300 python3 bench.py --workload sphere256 --out-json "$OUT/bench_sphere256.json" > /dev/null 2> "$OUT/bench_sphere256.err"; echo "config2 rc=$?"
timeout -k 10 400 python3 bench.py --workload shells1024u16 --out-json "$OUT/bench_shells1024u16.json" > /dev/null 2> "$OUT/bench_shells1024u16.err"; echo "config3 rc=$?"
timeout -k 10 400 python3 bench.py --workload pt1024f --steps 64 --out-json "$OUT/bench_pt1024f_64spp.json" > /dev/null 2> "$OUT/bench_pt1024f.err"; echo "config5 rc=$?"
timeout -k 10 400 python3 bench.py --workload haze2048 --out-json "$OUT/bench_haze2048.json" > /dev/null 2> "$OUT/bench_haze2048.err"; echo "haze2048 rc=$?"
timeout -k 10 400 python3 bench.py --workload shells2048 --viewport 2048 --steps 32 --out-json "$OUT/bench_shells2048_vp2048_1gpu.json" > /dev/null 2> "$OUT/bench_vp2048.err"; echo "config4@1gpu rc=$?"
# the build of the two cell grids at 2048^3 (cells of 4 and of 8 voxels)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cells_trace" -o c -- python3 "$ROOT/tools/cells_time.py" > "$OUT/cells_time.log" 2>&1 )
find "$OUT/cells_trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/cells_2048_kernel_stats.csv" \;
rm -rf "$OUT/cells_trace"
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    d = json.load(open(f))
    print("%-44s %8.4f ms/step %10.0f Msamples/s  serial %s  parity %s  cpu %s" % (
        os.path.basename(f), d["ms_per_step"], d["value"], d["roofline"].get("serial_launch_ms"),
        d.get("parity_max_abs_diff"), (d.get("cpu_baseline") or {}).get("value")))
PY
