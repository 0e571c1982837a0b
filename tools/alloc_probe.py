#!/usr/bin/env python3
"""Why do separately allocated maps stream slower than maps carved from one buffer?  python tools/alloc_probe.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import functional as F  # noqa: E402

S = 4096
dev = torch.device("cuda", 0)
P = S * S
kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
stream = torch.cuda.current_stream(dev).cuda_stream
variants = {}
# V4 first: fresh process, nothing cached yet -- four plain allocations back to back per set
fresh = [[torch.empty(c, S, S, device=dev) for c in (3, 3, 1, 1)] for _ in range(3)]
src = [synth_material(S, dev, 40 + i) for i in range(3)]
for f, m in zip(fresh, src):
    for d, t in zip(f, m):
        d.copy_(t)
variants["V4 fresh torch.empty per map"] = fresh
variants["V1 synth_material tensors"] = src
variants["V2 clones of V1"] = [[t.clone() for t in m] for m in src]
arena = []
for m in src:
    buf = torch.empty(8 * P, device=dev)
    views, k = [], 0
    for t in m:
        c = t.shape[0]
        v = buf[k * P:(k + c) * P].view(c, S, S)
        v.copy_(t)
        views.append(v)
        k += c
    arena.append(views)
variants["V3 one arena per material"] = arena
big = torch.empty(3 * 8 * P, device=dev)
shared = []
for i, m in enumerate(src):
    views, k = [], 0
    for t in m:
        c = t.shape[0]
        o = (i * 8 + k) * P
        v = big[o:o + c * P].view(c, S, S)
        v.copy_(t)
        views.append(v)
        k += c
    shared.append(views)
variants["V5 one arena for all three sets"] = shared
plans = {k: [F.plan_cook_torrance(*m, **kw) for m in v] for k, v in variants.items()}
for name, pl in plans.items():
    p_ = pl[0]
    ptrs = [t.data_ptr() for t in p_._keep if t is not None] + [p_.out.data_ptr()]
    base = min(ptrs)
    print(f"{name}: offsets MiB {[round((q - base) / 2**20, 1) for q in ptrs]}", flush=True)
times = {k: [] for k in plans}
order = list(plans.items())
if len(sys.argv) > 1 and sys.argv[1] == "reverse":
    order.reverse()
for rnd in range(7):
    for key, pl in order:
        for i in range(3):
            pl[i].launch(stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30):
            pl[i % 3].launch(stream)
        e1.record()
        torch.cuda.synchronize()
        times[key].append(e0.elapsed_time(e1) / 30 * 1e3)
for key, t in times.items():
    med = statistics.median(t)
    print(f"{key:36s} median {med:7.2f} us  min {min(t):7.2f} -> {44 * P / med / 1e3:7.1f} GB/s", flush=True)
