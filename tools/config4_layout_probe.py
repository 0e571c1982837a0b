#!/usr/bin/env python3
"""BASELINE config 4 at N = 1 (512 x 1024^2 materials, point light) under the batch layouts pack_maps offers, alternating in one process:
map-major (five [B,C,H,W] tensors, what bench.py --config 4 builds) against material-major (each material's 8 planes and its result next to
each other), each under both workgroup orders.  python tools/config4_layout_probe.py [B] [S] [rounds]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import _native as N, functional as F  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
maps = [torch.empty((B, c, S, S), device=dev) for c in (3, 3, 1, 1)]
for b in range(B):
    for dst, src in zip(maps, synth_material(S, dev, 4000 + b)):
        dst[b].copy_(src)
kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
plans = {}
for sched_name, sched in (("auto", N.SCHEDULE_AUTO), ("linear", N.SCHEDULE_LINEAR), ("runs64", N.schedule_xcd(6))):
    plans[("map-major", sched_name)] = F.plan_cook_torrance(*maps, schedule=sched, **kw)
*mm, out = F.pack_maps(*maps, reserve_output=True, material_major=True)
for sched_name, sched in (("auto", N.SCHEDULE_AUTO), ("linear", N.SCHEDULE_LINEAR), ("runs64", N.schedule_xcd(6))):
    plans[("material-major", sched_name)] = F.plan_cook_torrance(*mm, out=out, schedule=sched, **kw)
assert torch.equal(plans[("map-major", "auto")].launch(), plans[("material-major", "auto")].launch())
stream = torch.cuda.current_stream(dev).cuda_stream
times = {k: [] for k in plans}
for r in range(rounds):
    for k, p in plans.items():
        for _ in range(3):
            p.launch(stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            p.launch(stream)
        e1.record()
        e1.synchronize()
        times[k].append(e0.elapsed_time(e1) * 100)
for k, t in times.items():
    med = statistics.median(t)
    print("%d x %d^2 %-15s %-7s median %8.1f us  min %8.1f  %6.1f Gpixel/s  %.3f of 8 TB/s" % (B, S, k[0], k[1], med, min(t), B * S * S / med / 1e3, 44 * B * S * S / med / 8e6), flush=True)
