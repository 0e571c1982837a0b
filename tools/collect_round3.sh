#!/bin/bash
# Round-3 evidence, part 1 (run ON the MI355X box):  gpurun --timeout 1150 -- 'bash tools/collect_round3.sh r3p'
# The headline set (tools/collect_evidence.sh: bench line, kernel-trace statistics of the same command, three PMC passes) and the
# named BASELINE configurations through bench.py --config, each with the rocprofv3 kernel-trace statistics of the same command.
set -u
TAG=${1:-r3p}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
bash "$R/tools/collect_evidence.sh" "$TAG" > "$OUT/collect_evidence.log" 2>&1
tail -3 "$OUT/collect_evidence.log"
export TMPDIR=/tmp
cd /tmp
for c in 3 4 5; do
    timeout 300 python3 "$R/bench.py" --config $c > "$OUT/bench_c$c.json" 2> "$OUT/bench_c$c.err"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c$c" -o run -- \
        python3 "$R/bench.py" --config $c --no-cpu-baseline > "$OUT/trace_c$c.log" 2>&1
    echo "config $c done: $(head -c 300 "$OUT/bench_c$c.json")"
done
find "$OUT" -name "*kernel_stats.csv" | head
