#!/bin/bash
# backward kernel, 4096^2 material, steady state: four-pixel vs two-pixel lanes (PBR_TUNE_BWD_VEC)
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "PBR_TUNE_BWD_VEC=4" "PBR_TUNE_BWD_VEC=2" "PBR_TUNE_BWD_VEC=0"; do
  echo "== $cfg"
  env $cfg python3 tools/run_kernels.py 50 bwd 100 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('   ', d['case'][:8], d['us_per_launch_hip_events'], d['frac_of_8TBps'])"
done
