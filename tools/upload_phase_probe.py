#!/usr/bin/env python3
"""Where `material.resize(...)` of a freshly loaded CPU material -- the statement of examples/example_brdf.py that uploads it -- spends its
time, phase by phase on the host clock: layout, page-locked staging, host copies into it, device allocation, H2D enqueue, unpack launches,
the rest of MaterialBase._resident (the samples are freed there), the resize launch.  `python tools/upload_phase_probe.py [torch threads]
[--aten-copies]`: --aten-copies stages with Tensor.copy_ as the library did before (functional.STAGE_MEMCPY_LIMIT = 0)."""
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pypbr_amd.materials as M  # noqa: E402
from pypbr_amd import functional as F  # noqa: E402
from pypbr_amd.io import load_material_from_folder  # noqa: E402
from pypbr_amd.models import CookTorranceBRDF  # noqa: E402

warnings.simplefilter("ignore")
V, L, I = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0])
marks = []


def wrapped(name, fn):
    def g(*a, **k):
        marks.append((name + ">", time.perf_counter()))
        r = fn(*a, **k)
        marks.append((name + "<", time.perf_counter()))
        return r
    return g


for name in ("upload_packed", "_upload_stage", "_aligned_arena", "unpack_image", "_resize_raw"):
    setattr(F, name, wrapped(name, getattr(F, name)))
M.MaterialBase._resident = wrapped("_resident", M.MaterialBase._resident)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if args:
    torch.set_num_threads(int(args[0]))
if "--aten-copies" in sys.argv:
    F.STAGE_MEMCPY_LIMIT = 0
print("torch threads %d, staging copies by %s" % (torch.get_num_threads(), "Tensor.copy_" if F.STAGE_MEMCPY_LIMIT == 0 else "memcpy"), flush=True)
for rep in range(12):
    t_load = time.perf_counter()
    m = load_material_from_folder(os.path.join(ROOT, "tests", "golden", "tiles"), preferred_workflow="metallic")
    t_load = time.perf_counter() - t_load
    torch.cuda.synchronize()
    marks.clear()
    t0 = time.perf_counter()
    m.resize((512, 512))
    t1 = time.perf_counter()
    d = {}
    for k, t in marks:
        d.setdefault(k, []).append(t)

    def ms(a, b, last=False):                  # from mark a to mark b; 0 where a phase did not happen (no staging on the direct path)
        if a not in d or b not in d:
            return 0.0
        return (d[b][-1 if last else 0] - d[a][0]) * 1e3
    staged = "_upload_stage>" in d
    print("rep %2d: load %6.2f | resize statement %6.2f ms = layout %.2f + staging %.2f + host copies %.2f + device allocation %.2f + H2D enqueue %.2f + "
          "unpack x%d %.2f + rest of _resident %.2f + resize %.2f%s" % (
              rep, t_load * 1e3, (t1 - t0) * 1e3, ms("upload_packed>", "_upload_stage>" if staged else "_aligned_arena>"), ms("_upload_stage>", "_upload_stage<"),
              ms("_upload_stage<", "_aligned_arena>"), ms("_aligned_arena>", "_aligned_arena<"), ms("_aligned_arena<", "unpack_image>"),
              len(d.get("unpack_image>", [])), ms("unpack_image>", "unpack_image<", last=True), ms("upload_packed<", "_resident<"),
              (t1 - d["_resident<"][0]) * 1e3, "" if staged else "   (samples sent from the loader's page-locked block)"), flush=True)
    m.tile(2)
    CookTorranceBRDF(light_type="point")(m, V, L, I, 1.0)
