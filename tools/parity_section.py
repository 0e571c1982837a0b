#!/usr/bin/env python3
"""INTEGRATION.md's section "Numerical differences from upstream" is GENERATED from the committed parity table of the GPU suite
(profiles/r06_parity_table.json = gpurun_out/parity_table.json of the round's last full `pytest -m gpu` run):
    python tools/parity_section.py            prints the section's table
    python tools/parity_section.py --write    rewrites the table between the markers in INTEGRATION.md
    python tools/parity_section.py --check    exit 1 if INTEGRATION.md does not hold exactly this table (tests/test_docs.py)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "profiles", "r06_parity_table.json")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- parity-table:begin (tools/parity_section.py --write) -->", "<!-- parity-table:end -->"

WHAT = {
    "4096x4096 vs the C oracle": "one 4096² material, point light (BASELINE configs[1]), every value",
    "8192^2 x 12": "12 × 8192² fp16 maps, roughness ≥ 0.2, bands",
    "baseline_cfg0_seed0": "configs[0]: 256² seed-0 material through `CookTorranceBRDF`",
    "baseline_cfg1_aten_bands": "configs[1]: bands of the 4096² image vs the ATen restatement",
    "cfg3": "configs[2]: 64 × 2048², converted workflow, directional, crops",
    "cfg4": "configs[3] share: 64 × 1024², point, bands",
    "cfg5": "configs[4] share: 4 × 4096², 16 lights, fp16 maps, bands",
    "edge_golden": "argument edges (`light_size` truthiness, negative sizes): golden vectors",
    "rand64": "golden: seed 1234, 64 × 64, every workflow / light / flag",
    "rand37x53": "golden: 37 × 53", "rand1x1": "golden: 1 × 1", "rand1x17": "golden: 1 × 17", "rand5x1": "golden: 5 × 1",
    "real48": "golden: low-roughness 48 × 48 crop",
}


def table():
    rows = json.load(open(TABLE))
    out = ["| input set | values N | over 1e-5 vs upstream fp32 (max) | upstream's own fp32 vs its float64: over 8e-6 (the suite's envelope count) | max vs float64 | every value within 1e-5 above roughness |",
           "|---|---|---|---|---|---|"]
    for r in rows:
        thr = "—" if r["threshold"] is None else ("every value" if r["threshold"] == 0 else "%.3f" % r["threshold"])
        out.append("| %s | %s | %d (%.1e) | %d | %.1e | %s |" % (
            WHAT.get(r["set"], r["set"]), format(r["N"], ","), r["count"], r["max_abs_vs_ref32"], r["reference_count"], r["max_abs_vs_ref64"], thr))
    return "\n".join(out)


def main():
    t = table()
    if len(sys.argv) < 2:
        print(t)
        return 0
    doc = open(DOC).read()
    if BEGIN not in doc or END not in doc:
        print("INTEGRATION.md has no parity-table markers", file=sys.stderr)
        return 1
    a, b = doc.index(BEGIN) + len(BEGIN), doc.index(END)
    if sys.argv[1] == "--write":
        open(DOC, "w").write(doc[:a] + "\n" + t + "\n" + doc[b:])
        return 0
    if sys.argv[1] == "--check":
        if doc[a:b].strip() != t.strip():
            print("INTEGRATION.md's parity table differs from profiles/r06_parity_table.json: run tools/parity_section.py --write", file=sys.stderr)
            return 1
        return 0
    return 2


if __name__ == "__main__":
    sys.exit(main())
