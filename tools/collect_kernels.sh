#!/bin/bash
# Evidence for every kernel (run ON the MI355X box):  gpurun --timeout 1800 -- 'bash tools/collect_kernels.sh r2k'
# One kernel-trace pass and three PMC passes (never combined with other trace domains) over tools/run_kernels.py.
set -u
TAG=${1:-kernels}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 900 python3 "$R/tools/run_kernels.py" 100 "" 150 > "$OUT/cases.jsonl" 2> "$OUT/cases.err"
cat "$OUT/cases.jsonl"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python3 "$R/tools/run_kernels.py" 100 "" 150 > "$OUT/trace.log" 2>&1
for pass in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$pass" -o run -- python3 "$R/tools/run_kernels.py" 3 > "$OUT/pmc_$pass.log" 2>&1
done
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
    --output-format csv -d "$OUT/pmc_SQ" -o run -- python3 "$R/tools/run_kernels.py" 3 > "$OUT/pmc_SQ.log" 2>&1
find "$OUT" -name "*.csv" | head -20
