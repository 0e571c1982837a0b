"""Closes the suite's parity table (runs last: the file name sorts behind every other GPU test).

SURVEY.md 8c, "Parity criterion": (i) max-abs <= 1e-5 vs the reference's fp32 output [above a roughness threshold]; (ii) on
full-range random maps the count of values over 1e-5 must be <= 2e-5 * N and the build never further from the float64 evaluation than
the reference's own fp32 run.  parity_report asserts the per-value parts on every call; here, per input SET:
  * the survey's count bound is ASSERTED wherever the reference's own fp32-against-float64 count meets it (where the reference itself
    misses 2e-5 N -- low-roughness sets: its GGX denominator cancels -- the envelope criterion of parity_report stands alone);
  * the roughness threshold above which (i) holds on every value is a MEASURED output of the suite (printed, written); the GATE is the
    value recorded in tests/golden/parity_thresholds.json plus conftest.PARITY_GATE_MARGIN (0.01): a toolchain bump that moves one
    ill-conditioned value is not a regression, a threshold that rises by more is.  Re-recording: profiles/README.md.
The table goes to stdout and to gpurun_out/parity_table.json."""
import json
import os

import pytest

from conftest import PARITY_DEFAULT_THRESHOLD, PARITY_GATE_MARGIN, PARITY_RECORDED, PARITY_SETS, ROOT

pytestmark = pytest.mark.gpu


def test_parity_table_count_bound_and_thresholds():
    if not PARITY_SETS:
        pytest.skip("no parity set was evaluated in this session (run the whole -m gpu suite)")
    rows, failures = [], []
    for name in sorted(PARITY_SETS):
        r = PARITY_SETS[name]
        bound = 2e-5 * r["n"]
        reference_meets = r["n_ref"] <= bound
        recorded = PARITY_RECORDED.get(name)
        limit = PARITY_DEFAULT_THRESHOLD if recorded is None else float(recorded) + PARITY_GATE_MARGIN
        if reference_meets and r["n_hip"] > bound:
            failures.append(f"{name}: {r['n_hip']} values over 1e-5 vs the reference's fp32 output, bound 2e-5 N = {bound:.1f} (the reference meets it: {r['n_ref']})")
        if r["with_roughness"] and r["rough_needed"] > limit + 1e-9:
            failures.append(f"{name}: criterion (i) now needs roughness > {r['rough_needed']:.6f}; gate {limit:.6f} (recorded + {PARITY_GATE_MARGIN}): the threshold rose")
        rows.append(dict(set=name, N=r["n"], count=r["n_hip"], bound=round(bound, 1), reference_count=r["n_ref"], count_bound_asserted=bool(reference_meets),
                         threshold=round(r["rough_needed"], 6) if r["with_roughness"] else None, recorded_threshold=recorded,
                         max_abs_vs_ref32=r["max32"], max_abs_vs_ref64=r["max64"]))
    print("\n%-44s %12s %8s %10s %10s %9s %10s %10s" % ("set", "N", "count", "2e-5 N", "ref count", "asserted", "threshold", "recorded"))
    for x in rows:
        print("%-44s %12d %8d %10.1f %10d %9s %10s %10s" % (x["set"][:44], x["N"], x["count"], x["bound"], x["reference_count"], "yes" if x["count_bound_asserted"] else "envelope",
                                                           "-" if x["threshold"] is None else "%.4f" % x["threshold"], "-" if x["recorded_threshold"] is None else "%.4f" % x["recorded_threshold"]))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "parity_table.json"), "w") as f:
            json.dump(rows, f, indent=1)
    assert not failures, "\n".join(failures)
