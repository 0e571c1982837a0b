#!/bin/bash
# The driver's literal GPU-suite command on a deliberately SLOW host (VERDICT r5 next #1d): pytest pinned to one core that three busy
# loops share with it (a quarter of a core for the suite), the burners bounded by `timeout` and ended by their own PIDs.
#     gpurun --timeout 1200 -- 'bash tools/throttled_suite.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R"
mkdir -p gpurun_out
pids=()
for i in 1 2 3; do
    timeout -k 5 1100 taskset -c 0 sh -c 'while :; do :; done' &
    pids+=($!)
done
start=$(date +%s)
taskset -c 0 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6_suite_quarter_core.log 2>&1
rc=$?
for p in "${pids[@]}"; do kill "$p" 2>/dev/null; done
wait 2>/dev/null
echo "exit code $rc after $(( $(date +%s) - start )) s on a quarter of one core" >> gpurun_out/r6_suite_quarter_core.log
tail -4 gpurun_out/r6_suite_quarter_core.log
exit $rc
