#!/usr/bin/env python3
"""gpurun_out/<tag>/ of tools/collect_round3.sh -> profiles/r03_configs_kernel_trace.json: per BASELINE configuration the bench line's
value / step time / roofline object and the rocprofv3 kernel trace of the same command split by grid size (the full-batch launches of the
timed loop apart from the B/8 per-GPU-share launches).   python tools/configs_trace_summary.py gpurun_out/r3q > profiles/r03_configs_kernel_trace.json"""
import collections
import csv
import glob
import json
import re
import statistics
import sys

csv.field_size_limit(1 << 30)
root = sys.argv[1]
out = {}
for c in (3, 4, 5):
    line = json.loads(open(f"{root}/bench_c{c}.json").read().strip().splitlines()[-1])
    by = collections.defaultdict(list)
    for path in glob.glob(f"{root}/trace_c{c}/**/run_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "pbr::cook_torrance" not in r["Kernel_Name"]:
                continue
            name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void pbr::", ""))
            groups = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)) // max(1, int(r.get("Workgroup_Size", 1)))
            by[(name, groups)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out[f"config{c}"] = {
        "bench_line": {k: line[k] for k in ("value", "ms_per_step", "ms_per_step_cold") if k in line},
        "bench_roofline": line["roofline"],
        "kernel_trace": [{"kernel": k[0], "workgroups": k[1], "dispatches": len(v), "avg_us": round(statistics.mean(v), 2), "min_us": round(min(v), 2),
                          "median_us": round(statistics.median(v), 2)} for k, v in sorted(by.items(), key=lambda kv: -len(kv[1]) * statistics.mean(kv[1]))],
    }
    if "roofline_valu" in line:
        out[f"config{c}"]["bench_roofline_valu"] = line["roofline_valu"]
json.dump(out, sys.stdout, indent=1)
print()
