#!/bin/bash
# Kernel trace of each user flow of tools/flow_trace.py on its own (run ON the GPU box): top kernels by total time.
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
for flow in 1 2 3 4; do
  rm -rf $R/gpurun_out/flow$flow
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/flow$flow -o run -- python3 $R/tools/flow_trace.py 20 $flow > $R/gpurun_out/flow$flow.log 2>&1
  echo "== flow $flow (20 iterations)"
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/flow$flow/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:9]: print("  ", r["Name"][:90].ljust(90), r["Calls"].rjust(5), str(round(float(r["AverageNs"])/1e3,1)).rjust(8), str(round(float(r["TotalDurationNs"])/1e3)).rjust(8))
PY
done
