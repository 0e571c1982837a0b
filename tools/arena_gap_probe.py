#!/usr/bin/env python3
"""One allocation, maps at growing distances inside it: is it the single allocation or the proximity that makes
arena-carved maps stream faster than separately allocated ones?"""
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
src = [synth_material(S, dev, 40 + i) for i in range(3)]
plans = {"separate tensors": [F.plan_cook_torrance(*m, **kw) for m in src]}
for gap_mib in (0, 2, 64, 1024):
    gap = gap_mib * (1 << 20) // 4
    sets = []
    for m in src:
        buf = torch.empty(8 * P + 3 * gap, device=dev)
        views, o = [], 0
        for t in m:
            c = t.shape[0]
            v = buf[o:o + c * P].view(c, S, S)
            v.copy_(t)
            views.append(v)
            o += c * P + gap
        sets.append(views)
    plans[f"one buffer, {gap_mib} MiB between maps"] = [F.plan_cook_torrance(*v, **kw) for v in sets]
# separate hipMallocs that bypass nothing, but allocated as ONE torch.empty each of 4x the size (so each lives in a big block)
over = [[torch.empty(4 * t.numel(), device=dev)[:t.numel()].view_as(t).copy_(t) for t in m] for m in src]
plans["separate, each 4x over-allocated"] = [F.plan_cook_torrance(*m, **kw) for m in over]
times = {k: [] for k in plans}
for rnd in range(7):
    for key, pl in plans.items():
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
    print(f"{key:40s} median {med:7.2f} us  min {min(t):7.2f} -> {44 * P / med / 1e3:7.1f} GB/s", flush=True)
