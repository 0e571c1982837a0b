#!/bin/bash
# Round-1 tree against HEAD with the DRIVER's literal command, alternating, on one box (VERDICT r4, next #1b).
#   build/ab/r01 = `git archive 5b1b55e` (the tree the round-1 driver line was measured on), built there by its own build().
# usage (on the GPU box): bash tools/ab_driver_line.sh [pairs=5] [outdir=gpurun_out/r5_ab]
set -o pipefail
PAIRS=${1:-5}
OUT=${2:-gpurun_out/r5_ab}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/$OUT"
: > "$ROOT/$OUT/r01.jsonl"; : > "$ROOT/$OUT/head.jsonl"
for i in $(seq 1 "$PAIRS"); do
    (cd "$ROOT/build/ab/r01" && python3 bench.py --gpus 1 --steps 20 --warmup 5 2>>"$ROOT/$OUT/r01.err") >> "$ROOT/$OUT/r01.jsonl" || exit 1
    echo "pair $i: r01 done"
    (cd "$ROOT" && python3 bench.py --gpus 1 --steps 20 --warmup 5 2>>"$ROOT/$OUT/head.err") >> "$ROOT/$OUT/head.jsonl" || exit 1
    echo "pair $i: head done"
done
python3 - "$ROOT/$OUT" <<'PY'
import json, sys
d = sys.argv[1]
for name in ("r01", "head"):
    rows = [json.loads(l) for l in open(f"{d}/{name}.jsonl") if l.strip().startswith("{")]
    print(name, "value", [r["value"] for r in rows], "frac", [r["roofline"]["frac"] for r in rows],
          "kernel_us", [r["roofline"].get("kernel_us") for r in rows])
PY
