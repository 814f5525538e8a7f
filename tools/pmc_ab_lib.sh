#!/bin/bash
# tools/pmc_ab_lib.sh WORKLOAD [VARIANT...] -- SQ_INSTS_VALU / LDS / SALU per frame of the marching kernels in
# throughput mode, for the default build and variant builds (_variants/libvrhip_<VARIANT>.so).  One PMC pass each.
WL=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in default "$@"; do
  if [ "$v" = default ]; then unset VRHIP_LIB_PATH; else export VRHIP_LIB_PATH=$ROOT/volumerenderercl_amd/_variants/libvrhip_$v.so; fi
  OUT=$ROOT/gpurun_out/pmcab_$v; rm -rf "$OUT"; mkdir -p "$OUT"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --profile-region --workload $WL --steps 64 --warmup 3 > "$OUT/log.txt" 2>&1
  python3 - "$OUT" "$v" <<'PY'
import csv, glob, re, sys, collections
out, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"vr_\w+(<[^>]*>)?", r["Kernel_Name"])
        k = m.group(0)[:60] if m else r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(acc.items()):
    if "vr_" in k:
        print("%-10s %-62s VALU %8.2fM LDS %7.2fM SALU %7.2fM (sum over the run)" % (v, k, c["SQ_INSTS_VALU"] / 1e6, c["SQ_INSTS_LDS"] / 1e6, c["SQ_INSTS_SALU"] / 1e6))
PY
done
