#!/bin/bash
# workgroup-order sweep over shapes (fp32, one light): which `xcd` run length streams fastest where
cd ${GRAFT_REPO_ROOT:-.}
C="xcd=0;xcd=3;xcd=6;xcd=9;xcd=12"
for args in "--size 4096 --batch 1 --arena" "--size 2048 --batch 16" "--size 1024 --batch 64" "--size 4096 --height 1024 --batch 16" "--size 3072 --batch 4" "--size 2048 --batch 1 --arena" "--size 2048 --batch 16 --light directional"; do
    echo "== $args"
    python3 tools/tune.py $args --rounds 3 --iters 20 --configs "$C" 2>&1 | grep median | awk '{print $1, $2, $4, $5, $NF-1, $(NF-1)}'
done
