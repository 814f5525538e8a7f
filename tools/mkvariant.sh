#!/bin/bash
# tools/mkvariant.sh NAME [extra hipcc flags...] -- build libvrhip with extra -D flags into
# volumerenderercl_amd/_variants/libvrhip_NAME.so (git-ignored; travels to the GPU box) for
# A/B runs with tools/ab.sh / VRHIP_LIB_PATH.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/volumerenderercl_amd/csrc
OBJ=$SRC/_obj/var_$NAME
mkdir -p "$OBJ" "$ROOT/volumerenderercl_amd/_variants"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function"
pids=()
for f in vr_raycast vr_pathtrace vr_cells vr_bricks vrhip_api; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c "$SRC/$f.hip" -o "$OBJ/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
printf 'extern "C" const char *vrhip_build_source_hash(void) { return "%s"; }\n' "$(python3 "$ROOT/volumerenderercl_amd/_srchash.py" "$@")" > "$OBJ/vr_srchash.cpp"
g++ -O2 -fPIC -c "$OBJ/vr_srchash.cpp" -o "$OBJ/vr_srchash.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/volumerenderercl_amd/_variants/libvrhip_$NAME.so" "$OBJ"/*.o
echo "built _variants/libvrhip_$NAME.so"
