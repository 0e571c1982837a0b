#!/usr/bin/env python3
"""gpurun_out/<tag>/ of tools/collect_kernels.sh -> one JSON record per (kernel, grid): average / minimum duration from
the rocprofv3 kernel trace, HBM bytes per launch from the PMC passes (FETCH_SIZE doubled as MI355X_MICROARCH.md
prescribes for gfx950 16-byte streams, WRITE_SIZE exact), SQ counters, and -- where tools/run_kernels.py names the
kernel -- the algorithmic bytes and the roofline fraction.   python tools/kernels_summary.py gpurun_out/r2k > profiles/r02_kernels.json"""
import csv
import glob
import json
import re
import statistics
import sys

csv.field_size_limit(1 << 30)
root = sys.argv[1]


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("pbr::", "")
    return re.sub(r"\(.*$", "", name)


def is_ours(name):
    return "pbr::" in name


def grid_of(row):
    return int(row.get("Grid_Size_X") or row.get("Grid_Size") or 0)


rec = {}
for path in glob.glob(f"{root}/trace/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if not is_ours(row["Kernel_Name"]):
            continue
        key = (short(row["Kernel_Name"]), grid_of(row))
        r = rec.setdefault(key, {"us": [], "vgpr": int(row["VGPR_Count"]), "agpr": int(row["Accum_VGPR_Count"]), "lds": int(row["LDS_Block_Size"]),
                                 "scratch": int(row["Scratch_Size"]), "workgroup": int(row["Workgroup_Size_X"]),
                                 "first": int(row["Start_Timestamp"])})
        r["us"].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)

counters = {}
for pass_name in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
    for path in glob.glob(f"{root}/pmc_{pass_name}/**/*counter_collection.csv", recursive=True):
        acc, durs = {}, {}
        for row in csv.DictReader(open(path)):
            if not is_ours(row["Kernel_Name"]):
                continue
            key = (short(row["Kernel_Name"]), grid_of(row))
            k2 = (key, row["Dispatch_Id"], row["Counter_Name"])
            acc[k2] = acc.get(k2, 0.0) + float(row["Counter_Value"])       # one row per XCD / instance: summed
            durs[(key, row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
        for (key, _, cname), v in acc.items():
            counters.setdefault(key, {}).setdefault(cname, []).append(v)
        if pass_name == "SQ":
            for (key, _), us in durs.items():
                counters.setdefault(key, {}).setdefault("_sq_us", []).append(us)

cases = []
try:
    for line in open(f"{root}/cases.jsonl"):
        cases.append(json.loads(line))
except OSError:
    pass


def cases_of(name):
    """A case names its kernel by a prefix of the rocprof name, by a substring, or -- a template that has grown a parameter since the
    case was written -- by the name up to its closing bracket."""
    return ([cs for cs in cases if name.startswith(cs["kernel"])] or [cs for cs in cases if cs["kernel"] in name + "("]
            or [cs for cs in cases if cs["kernel"].endswith(">") and name.startswith(cs["kernel"][:-1] + ",")])

# one kernel, several cases, each with its own grid (the resize shapes): the n-th grid to appear belongs to the n-th case
by_order = {}
for name in {k[0] for k in rec}:
    keys = sorted((k for k in rec if k[0] == name), key=lambda k: rec[k]["first"])
    match = cases_of(name)
    if len(match) > 1 and len(match) == len(keys):
        for k, cs in zip(keys, match):
            by_order[k] = [cs]

out = []
for key in sorted(rec):
    name, grid = key
    r = rec[key]
    us = r["us"]
    reps = sum(cs.get("reps", 0) for cs in by_order.get(key, cases_of(name)))
    if reps and len(us) > reps:          # the timed launches come last; what precedes them is warm-up / clock settling
        us = us[-reps:]
    ent = {"kernel": name, "grid_threads": grid, "workgroup": r["workgroup"], "vgpr": r["vgpr"], "agpr": r["agpr"], "lds_bytes": r["lds"],
           "scratch": r["scratch"], "dispatches": len(r["us"]), "timed_dispatches": len(us), "avg_us": round(statistics.mean(us), 2), "min_us": round(min(us), 2),
           "median_us": round(statistics.median(us), 2)}
    c = counters.get(key, {})
    if "FETCH_SIZE" in c:
        ent["FETCH_SIZE_KiB_raw"] = round(statistics.mean(c["FETCH_SIZE"]), 1)
        ent["hbm_read_bytes"] = int(round(statistics.mean(c["FETCH_SIZE"]) * 1024 * 2))
    if "WRITE_SIZE" in c:
        ent["WRITE_SIZE_KiB_raw"] = round(statistics.mean(c["WRITE_SIZE"]), 1)
        ent["hbm_write_bytes"] = int(round(statistics.mean(c["WRITE_SIZE"]) * 1024))
    if "hbm_read_bytes" in ent and "hbm_write_bytes" in ent:
        ent["hbm_bytes"] = ent["hbm_read_bytes"] + ent["hbm_write_bytes"]
    if "SQ_INSTS_VALU" in c:
        m = {k: statistics.mean(v) for k, v in c.items()}
        sq = {"median_us_under_counters": round(statistics.median(c["_sq_us"]), 1)}
        cycles = m["GRBM_GUI_ACTIVE"] / 8
        # GRBM_GUI_ACTIVE spans the whole dispatch slot, idle head and tail included: under ~20 us the ratio is not a clock (round 5 printed
        # 15 GHz for a 2 us kernel) -- not derived there, and nothing that divides by `cycles` is either
        short_kernel = sq["median_us_under_counters"] < 20.0
        sq["shader_clock_GHz"] = None if short_kernel else round(cycles / (sq["median_us_under_counters"] * 1e-6) / 1e9, 3)
        sq["valu_wave_instructions"] = int(m["SQ_INSTS_VALU"])
        # SQ_ACTIVE_INST_VALU counts, per WAVE, the quad-cycles it spends in vector instructions; a SIMD overlaps the vector instructions
        # of two waves, so the sum per SIMD cycle is NOT a fraction of 1.  Calibrated with pure instruction streams (tools/valu_counters.sh,
        # profiles/r06_valu_counter_calibration.txt): it saturates at 1.5-1.6 for plain fp32 (v_fma / v_mul at 8 waves per SIMD), 0.91 for
        # packed fp32 (v_pk_*), 0.97 for transcendentals, 1.18 for compare + select pairs.  Round 5 called it `valu_busy_fraction` and printed
        # 1.054 for a plain-fp32 kernel: the number was right, the name was not.
        sq["valu_active_wave_cycles_per_simd_cycle"] = None if short_kernel else round(4 * m["SQ_ACTIVE_INST_VALU"] / (1024 * cycles), 3)
        sq["valu_active_ceiling_by_instruction_class"] = {"plain_fp32": 1.55, "packed_fp32": 0.91, "transcendental": 0.97, "compare_select": 1.18}
        sq["mean_resident_waves_per_cu"] = None if short_kernel else round(4 * m["SQ_WAVE_CYCLES"] / (256 * cycles), 2)
        sq["wait_any_fraction_of_wave_cycles"] = round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 3)
        sq["wait_inst_any_fraction_of_wave_cycles"] = round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 3)
        ent["sq"] = sq
    match = by_order.get(key, cases_of(name))
    if len(match) == 1 or (match and len({cs["algorithmic_bytes_per_launch"] for cs in match}) == 1):
        alg = match[0]["algorithmic_bytes_per_launch"]
        ent["case"] = match[0]["case"]
        ent["algorithmic_bytes"] = alg
        ent["algorithmic_GBps_at_avg"] = round(alg / ent["avg_us"] / 1e3, 1)
        ent["frac_of_8TBps_at_avg"] = round(alg / ent["avg_us"] / 1e3 / 8000.0, 4)
        if "hbm_bytes" in ent:
            ent["traffic_over_algorithmic"] = round(ent["hbm_bytes"] / alg, 4)
    elif match:
        ent["cases"] = [cs["case"] for cs in match]
    out.append(ent)
# a record whose stated bytes are not what moved is not evidence (round 2: a fraction of 1.24, a traffic ratio of 0.667): say so loudly
for ent in out:
    if ent.get("frac_of_8TBps_at_avg", 0) > 1.0 or not 0.9 <= ent.get("traffic_over_algorithmic", 1.0) <= 2.5:
        print("SUSPECT RECORD: %s frac %s traffic ratio %s" % (ent["kernel"], ent.get("frac_of_8TBps_at_avg"), ent.get("traffic_over_algorithmic")), file=sys.stderr)
print(json.dumps(out, indent=1))
