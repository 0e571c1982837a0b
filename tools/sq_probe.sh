#!/bin/bash
# SQ counters of the kernels one case of tools/run_kernels.py launches (run ON the MI355X box; knobs come from the
# environment of THIS shell, e.g. PBR_TUNE_BWD_RUN=8 bash tools/sq_probe.sh bwd tag):  per kernel -- shader clock, VALU
# wave-instructions, VALU busy fraction, resident waves per CU, wait fractions.
set -u
ONLY=${1:-bwd}; TAG=${2:-sq}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
    --output-format csv -d "$OUT/pmc_SQ" -o run -- python3 "$R/tools/run_kernels.py" 5 "$ONLY" > "$OUT/pmc_SQ.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, re, statistics, sys, collections
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/pmc_SQ/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "pbr::" not in row["Kernel_Name"]:
            continue
        k = re.sub(r"\(.*$", "", row["Kernel_Name"].replace("pbr::", "").replace("void ", ""))
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        acc[k]["_us"].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
for k, c in acc.items():
    m = {n: statistics.mean(v) for n, v in c.items()}
    cycles = m["GRBM_GUI_ACTIVE"] / 8
    us = statistics.median(c["_us"])
    print(k)
    print("   us %.1f  clock %.3f GHz  VALU insts %d  VALU active wave-cycles per SIMD cycle %.3f (ceilings: plain fp32 1.55, packed 0.91, trans 0.97)  waves/CU %.2f  wait_any %.3f  wait_inst %.3f  inst_any_active %.3f" % (
        us, cycles / (us * 1e-6) / 1e9, m["SQ_INSTS_VALU"], 4 * m["SQ_ACTIVE_INST_VALU"] / (1024 * cycles), 4 * m["SQ_WAVE_CYCLES"] / (256 * cycles),
        m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
PY
