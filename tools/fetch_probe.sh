#!/bin/bash
# HBM read bytes (FETCH_SIZE x 2, MI355X_MICROARCH.md) and write bytes of the kernels one run_kernels.py case launches:
#   bash tools/fetch_probe.sh resize tag      (knobs from the calling shell's environment)
set -u
ONLY=${1:-resize}; TAG=${2:-fetch}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for pass in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$pass" -o run -- python3 "$R/tools/run_kernels.py" 3 "$ONLY" > "$OUT/pmc_$pass.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, re, statistics, sys, collections
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob(sys.argv[1] + "/pmc_%s/**/*counter_collection.csv" % p, recursive=True):
        for row in csv.DictReader(open(path)):
            if "pbr::" not in row["Kernel_Name"]:
                continue
            k = (re.sub(r"\(.*$", "", row["Kernel_Name"].replace("pbr::", "").replace("void ", "")), int(row.get("Grid_Size_X") or row.get("Grid_Size") or 0))
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in sorted(acc.items()):
    rd = statistics.mean(c.get("FETCH_SIZE", [0])) * 1024 * 2
    wr = statistics.mean(c.get("WRITE_SIZE", [0])) * 1024
    print("%-60s grid %9d  read %8.1f MB  written %8.1f MB  total %8.1f MB" % (k[0][:60], k[1], rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
PY
