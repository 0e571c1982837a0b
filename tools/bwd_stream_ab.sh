#!/bin/bash
# streamed backward kernel (fp16 maps, one light): tiles per wave (PBR_TUNE_BWD_RUN; 0 = the one-tile kernel), steady state
cd ${GRAFT_REPO_ROOT:-.}
for run in ${RUNS:-0 2 3 4 6 8 0 4}; do
  echo "== PBR_TUNE_BWD_RUN=$run"
  env PBR_TUNE_BWD_RUN=$run python3 tools/run_kernels.py 50 bwd 100 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('   ', d['case'][:8], d['us_per_launch_hip_events'], d['frac_of_8TBps'])"
done
