#!/bin/bash
# Builds older revisions of the package at side paths for tools/ab_revisions.py (in this container: needs .git):
#   tools/ab_revisions.sh 8600504 1dbff36      ->  tools/bin/<rev>/pypbr_amd/libpbr_hip.so (git-ignored, travels with gpurun)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
for rev in "$@"; do
    D=$R/tools/bin/$rev
    rm -rf "$D"; mkdir -p "$D"
    git -C "$R" archive "$rev" pypbr_amd include | tar -x -C "$D"
    make -s -j4 -C "$D/pypbr_amd/csrc"
    echo "$rev -> $D/pypbr_amd/libpbr_hip.so ($(sha256sum "$D/pypbr_amd/libpbr_hip.so" | cut -c1-16))"
done
