#!/bin/bash
# A/B of one environment knob over tools/run_kernels.py cases, two alternating rounds:
#   tools/env_ab.sh PBR_TUNE_SCALAR_BASE "1 0" [case filter] [reps]
cd ${GRAFT_REPO_ROOT:-.}
VAR=$1; VALUES=$2; FILTER=${3:-}; REPS=${4:-100}
for round in 1 2; do
  for v in $VALUES; do
    env $VAR=$v python3 tools/run_kernels.py $REPS "$FILTER" 150 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$VAR=$v', d['case'][:22].ljust(22), d['us_per_launch_hip_events'], d['frac_of_8TBps'])"
  done
done | sort -k2,2 -k1,1 -s
