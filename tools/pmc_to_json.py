#!/usr/bin/env python3
"""gpurun_out/<tag>/pmc_*/…counter_collection.csv (tools/collect_evidence.sh) -> the JSON record bench.py reads
for roofline.traffic.   python tools/pmc_to_json.py gpurun_out/r24 > profiles/pmc_traffic.json"""
import csv
import glob
import json
import statistics
import sys

KERNEL = "cook_torrance_kernel<1, 0, float, float, 4, false, true, false>"
NAME = "ct_point_metallic_f32_f32_v4"
PIXELS, BPP = 4096 * 4096, 44


MULTI_KERNEL = "cook_torrance_batch_kernel<1, 0, __half, float, 2, 4, true>"
MULTI_NAME = "ctb_point_metallic_f16_f32_v2_b4"


def per_dispatch(root, pass_name, kernel=None):
    """counter -> list of per-dispatch values (rows of one dispatch are summed: one row per XCD/instance)."""
    kernel = kernel or KERNEL
    vals, dur = {}, []
    for path in glob.glob(f"{root}/pmc_{pass_name}/**/*counter_collection.csv", recursive=True):
        acc = {}
        for row in csv.DictReader(open(path)):
            if kernel not in row["Kernel_Name"]:
                continue
            key = (row["Dispatch_Id"], row["Counter_Name"])
            acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"])
            if "Start_Timestamp" in row:
                dur.append((row["Dispatch_Id"], (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3))
        for (_, counter), v in acc.items():
            vals.setdefault(counter, []).append(v)
    d = {}
    for disp, us in dur:
        d[disp] = us
    return vals, list(d.values())


def main():
    root = sys.argv[1]
    fetch, _ = per_dispatch(root, "FETCH_SIZE")
    write, _ = per_dispatch(root, "WRITE_SIZE")
    sq, sq_us = per_dispatch(root, "SQ")
    f_kib, w_kib = statistics.mean(fetch["FETCH_SIZE"]), statistics.mean(write["WRITE_SIZE"])
    rd, wr = int(round(f_kib * 1024 * 2)), int(round(w_kib * 1024))
    rec = {
        "run": f"{root} (tools/collect_evidence.sh on an MI355X via gpurun; committed as profiles/{sys.argv[2] if len(sys.argv) > 2 else 'pmc_traffic.json'})",
        "workload": "bench.py default: 1 x 4096x4096 BasecolorMetallicMaterial, point light, fp32, sRGB in/out",
        "collected": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes (gpurun, MI355X), per-dispatch mean over "
                     f"{len(fetch['FETCH_SIZE'])} / {len(write['WRITE_SIZE'])} launches; tools/collect_evidence.sh + tools/pmc_to_json.py",
        "FETCH_SIZE_KiB_raw": round(f_kib, 2), "WRITE_SIZE_KiB_raw": round(w_kib, 2),
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on 16-B/lane streams -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
        "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
        "algorithmic_bytes_per_launch": PIXELS * BPP, "ratio_traffic_to_algorithmic": round((rd + wr) / (PIXELS * BPP), 4),
    }
    out = {NAME: rec}
    if sq:
        rec["sq_counters_third_pass"] = sq_summary(sq, sq_us)
    sq16, sq16_us = per_dispatch(root, "SQ16", MULTI_KERNEL)
    if sq16:
        out[MULTI_NAME] = {"workload": "tools/run_multilight.py: 4 x 4096x4096 fp16 maps, 16 point lights, fp32 out (BASELINE.json configs[4] per-GPU share)",
                           "sq_counters": sq_summary(sq16, sq16_us)}
    print(json.dumps(out, indent=1))


def sq_summary(sq, sq_us):
    if True:
        m = {k: statistics.mean(v) for k, v in sq.items()}
        us = statistics.median(sq_us) if sq_us else None
        third = {"median_dispatch_us_under_counters": round(us, 1) if us else None}
        for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            if k in m:
                third[k] = m[k]
        if "SQ_INSTS_VALU" in m and "SQ_ACTIVE_INST_VALU" in m:
            third["cycles_per_valu_wave_instruction"] = round(4 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"], 2)   # counter ticks are quad-cycles
        if us and "GRBM_GUI_ACTIVE" in m:
            cycles = m["GRBM_GUI_ACTIVE"] / 8          # the per-dispatch value is the sum over the 8 XCDs
            third["shader_clock_GHz"] = round(cycles / (us * 1e-6) / 1e9, 3)
            if "SQ_ACTIVE_INST_VALU" in m:             # SQ_* tick in quad-cycles, summed over the 1024 SIMDs (256 CUs x 4)
                third["valu_active_wave_cycles_per_simd_cycle"] = round(4 * m["SQ_ACTIVE_INST_VALU"] / (1024 * cycles), 3)
            if "SQ_WAVE_CYCLES" in m:
                third["mean_resident_waves_per_cu"] = round(4 * m["SQ_WAVE_CYCLES"] / (256 * cycles), 2)
        return third


if __name__ == "__main__":
    main()
