#!/bin/bash
# tools/make_profiles.sh ROUND -- run on the GPU box (via gpurun): everything profiles/<ROUND>/ keeps.
#  * rocprofv3 --kernel-trace --stats + PMC passes of `bench.py --profile-region` (timed launches only) for the
#    schedules bench.py is run with: the default throughput schedule, the round driver's arguments
#    (--steps 20 --warmup 5), one frame at a time -- headline workload; the dense regime (haze2048); the path
#    tracer on both fields.  Every PMC summary carries the `meta` of its run (workload, viewport, schedule, hash
#    of the kernel sources): bench.py uses one only for a run with the same meta.
#  * full bench.py JSON lines of the headline and of BASELINE configs 1, 2, 3, 5 (sphere and shells) and 4 on one GPU
#  * kernel stats of the one-time builders (ESS bricks, footprint volume, cell grids)
R=${1:-r4}
PART=${2:-all}    # all | regions1 | regions2 | bench | extras  (a gpurun call is limited to 20 minutes: run the parts one by
                  # one, copying gpurun_out/profiles_<R>/pmc_*.json into profiles/<R>/ before `bench`)
want() { [ "$PART" = all ] || [ "$PART" = "$1" ]; }
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p "$OUT"
cd "$ROOT"
region() {   # TAG WORKLOAD ARGS...
  local tag=$1 wl=$2; shift 2
  bash tools/profile_region.sh ${R}_$tag $wl "$@" > "$OUT/region_$tag.log" 2>&1
  local D=$ROOT/gpurun_out/region_${R}_$tag
  cp "$D/stats.csv" "$OUT/${tag}_kernel_stats.csv"
  cp "$D/bench.json" "$OUT/${tag}_region.json"
  cp "$D/issue.json" "$OUT/pmc_issue_$tag.json"
  cp "$D/traffic.json" "$OUT/pmc_traffic_$tag.json"
  echo "region $tag done"
}
if want regions1; then
region shells2048_tput shells2048 --warmup 0 --steps 64
region shells2048_driver_args shells2048 --warmup 5 --steps 20
region shells2048_single shells2048 --warmup 3 --steps 32 --frames-in-flight 1 --frames-per-launch 1
region haze2048_tput haze2048 --warmup 0 --steps 64
region shells2048_vp2048 shells2048 --viewport 2048 --warmup 4 --steps 32
fi
if want regions2; then
region pt1024f_sphere pt1024f_sphere --warmup 1 --steps 16
region pt1024f pt1024f --warmup 1 --steps 16
# BASELINE configs 1, 2, 3 (config 4's frame on one GPU: above), with the schedule their bench lines below run
region sphere256_plain_512 sphere256_plain --viewport 512 --warmup 4 --steps 64
region sphere256 sphere256 --warmup 4 --steps 64
region shells1024u16 shells1024u16 --warmup 4 --steps 64
region shells1024 shells1024 --warmup 4 --steps 64
fi
b() {   # NAME ARGS...
  local name=$1; shift
  timeout -k 10 400 python3 bench.py "$@" --out-json "$OUT/bench_$name.json" > /dev/null 2> "$OUT/bench_$name.err"; echo "$name rc=$?"
}
mkdir -p "$ROOT/profiles/$R"
cp "$OUT"/pmc_issue_*.json "$OUT"/pmc_traffic_*.json "$ROOT/profiles/$R/" 2>/dev/null   # (the bench lines below look them up)
if want bench; then
b shells2048 --steps 64 --warmup 0
b shells2048_driver_args --steps 20 --warmup 5
b shells2048_single --frames-in-flight 1 --frames-per-launch 1 --steps 32 --warmup 3
b sphere256_plain_512 --workload sphere256_plain --viewport 512
b sphere256 --workload sphere256
b shells1024u16 --workload shells1024u16
b shells1024 --workload shells1024
b pt1024f_sphere_64spp --workload pt1024f_sphere --steps 64
b pt1024f_64spp --workload pt1024f --steps 64
b haze2048 --workload haze2048 --steps 64 --warmup 0
b shells2048_vp2048_1gpu --workload shells2048 --viewport 2048 --steps 32
fi
if want extras; then
# the VALU issue rate of the part (roofline_valu_issue.peak) and the short-run tile shares
hipcc --offload-arch=gfx950 -O3 tools/micro_occ.hip -o /tmp/micro_occ > /dev/null 2>&1 && /tmp/micro_occ > "$OUT/micro_occ.txt" 2>&1
TOTAL=20 python3 tools/share_time.py 8 1024 0,3,7 "$OUT/share_time_short_1024.json" > "$OUT/share_time_short_1024.txt" 2>&1
TOTAL=20 python3 tools/share_time.py 8 2048 0,7 "$OUT/share_time_short_2048.json" > "$OUT/share_time_short_2048.txt" 2>&1
./volumerenderercl_amd/vrhip_render --synth shells 2048 UCHAR --size 1024 1024 --rotate 1 1 0 30 --frames 64 --frames-per-launch 32 --bench --out /tmp/cppb > "$OUT/cpp_host_bench_64.json" 2> /dev/null
./volumerenderercl_amd/vrhip_render --synth shells 2048 UCHAR --size 1024 1024 --rotate 1 1 0 30 --frames 20 --frames-per-launch 32 --bench --out /tmp/cppb > "$OUT/cpp_host_bench_20.json" 2> /dev/null
# the one-time builders at 2048^3 (ESS bricks, footprint volume, both cell grids)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cells_trace" -o c -- python3 "$ROOT/tools/cells_time.py" > "$OUT/cells_time.log" 2>&1 )
find "$OUT/cells_trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/builders_2048_kernel_stats.csv" \;
rm -rf "$OUT/cells_trace"
fi
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    d = json.load(open(f))
    print("%-44s %8.4f ms/step %10.0f Msamples/s  serial %s  parity %s  cpu %s  hbm frac %s  issue %s" % (
        os.path.basename(f), d["ms_per_step"], d["value"], d["roofline"].get("serial_launch_ms"),
        d.get("parity_max_abs_diff"), (d.get("cpu_baseline") or {}).get("value"), d["roofline"].get("frac"),
        (d.get("roofline_valu_issue") or {}).get("frac")))
PY
