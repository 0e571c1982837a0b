#!/bin/bash
# Evidence set of one round, run ON the MI355X box:  gpurun --timeout 1500 -- 'bash tools/collect_evidence.sh r24'
# Writes gpurun_out/<tag>/: bench.json, kernel-trace stats, three PMC passes (never combined with other trace
# domains), the per-configuration table.  tools/pmc_to_json.py turns the PMC csv files into profiles/pmc_traffic.json.
set -u
TAG=${1:-evidence}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 400 python3 "$R/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
cat "$OUT/bench.json"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- \
    python3 "$R/bench.py" --no-cpu-baseline > "$OUT/trace.log" 2>&1
for pass in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$pass" -o run -- \
        python3 "$R/bench.py" --steps 20 --warmup 5 --settle 0 --no-cpu-baseline > "$OUT/pmc_$pass.log" 2>&1
done
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
    --output-format csv -d "$OUT/pmc_SQ" -o run -- python3 "$R/bench.py" --steps 50 --warmup 5 --settle 0 --no-cpu-baseline > "$OUT/pmc_SQ.log" 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
    --output-format csv -d "$OUT/pmc_SQ16" -o run -- python3 "$R/tools/run_multilight.py" 4 > "$OUT/pmc_SQ16.log" 2>&1
cd "$R"
timeout 900 python3 tools/bench_configs.py > "$OUT/configs.jsonl" 2> "$OUT/configs.err"
cat "$OUT/configs.jsonl"
timeout 300 python3 -m pytest tests/test_gpu_parity.py -q -s -k full_size 2>&1 | grep -E "4096x4096|passed|failed" > "$OUT/fullsize.log"
cat "$OUT/fullsize.log"
find "$OUT" -name "*.csv" | head -20
