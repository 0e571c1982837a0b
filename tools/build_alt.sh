#!/bin/bash
# A second build of libpbr_hip.so for in-process A/B runs (tools/tune.py --altlib):
#   tools/build_alt.sh NAME [extra hipcc flags, e.g. -DPBR_WAVES_PER_EU=4]   ->  tools/bin/NAME/libpbr_hip.so
set -eu
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/tools/bin/$NAME
mkdir -p "$D"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -fno-slp-vectorize -fno-gpu-rdc -Wno-unused-function $*"
pids=()
for f in cook_torrance ct_batch ct_tiled ct_backward ct_blend ct_loss map_ops resize blend; do
    /opt/rocm/bin/hipcc $FLAGS -c "$R/pypbr_amd/csrc/$f.hip" -o "$D/$f.o" &
    pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
g++ -O2 -fPIC -DPBR_SOURCE_HASH=\"alt:$NAME\" -c "$R/pypbr_amd/csrc/build_id.cpp" -o "$D/build_id.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$D/libpbr_hip.so" "$D"/*.o
echo "$D/libpbr_hip.so"
