#!/bin/bash
# Calibration of the SQ counters tools/kernels_summary.py derives "VALU busy" from: tools/bin/valu_occupancy (pure v_fma_f32 / v_pk_fma_f32 /
# v_exp_f32 streams at 1 ... 8 waves per SIMD, whose true issue rates its own timing prints) under rocprofv3 --pmc.  Run ON the GPU box.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${1:-valu_cal}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY \
    --output-format csv -d "$OUT/pmc" -o run -- "$R/tools/bin/valu_occupancy" > "$OUT/pmc.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
csv.field_size_limit(1 << 30)
rows = collections.OrderedDict()
for path in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"][:60], int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0)), r["Dispatch_Id"])
        d = rows.setdefault(k, collections.defaultdict(float))
        d[r["Counter_Name"]] += float(r["Counter_Value"])
        d["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
seen = set()
for (name, grid, _), d in rows.items():
    if (name, grid) in seen:
        continue
    seen.add((name, grid))
    cyc = d["GRBM_GUI_ACTIVE"] / 8           # summed over the 8 XCDs
    print("%-58s grid %6d  %8.1f us  clock %.2f GHz  INSTS_VALU %.3g  4*ACTIVE_INST_VALU/(1024*cycles) = %.3f  4*WAVE_CYCLES/(1024*cycles) = %.2f" % (
        name, grid, d["_us"], cyc / (d["_us"] * 1e-6) / 1e9, d["SQ_INSTS_VALU"], 4 * d["SQ_ACTIVE_INST_VALU"] / (1024 * cyc), 4 * d["SQ_WAVE_CYCLES"] / (1024 * cyc)))
PY
