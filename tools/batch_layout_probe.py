#!/usr/bin/env python3
"""Batched maps: map-major ([B,3,H,W] albedo tensor, [B,3,H,W] normal tensor, ...) against material-major (all planes of
material b next to each other; the ABI's per-map batch strides make both plain views).  python tools/batch_layout_probe.py B SIZE"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import functional as F  # noqa: E402

B, S = int(sys.argv[1]), int(sys.argv[2])
light = sys.argv[3] if len(sys.argv) > 3 else "point"
dev = torch.device("cuda", 0)
P = S * S
kw = dict(view_dir=[0, 0, 1], light_intensity=[1, 1, 1], light_type=light)
kw.update(dict(light=[0.1, 0.1, 1.0], light_size=1.0) if light == "point" else dict(light=[0.3, -0.2, 1.0]))
stream = torch.cuda.current_stream(dev).cuda_stream
one = synth_material(S, dev, 5)
major = [torch.stack([t] * B) for t in one]                      # map-major: four tensors
pitch = 8 * P
arena = torch.empty(B * pitch, device=dev)
views, k = [], 0
for t in one:
    c = t.shape[0]
    v = arena.as_strided((B, c, S, S), (pitch, P, S, 1), k * P)
    v.copy_(t.unsqueeze(0).expand(B, c, S, S))
    views.append(v)
    k += c
plans = {"map-major (4 tensors)": F.plan_cook_torrance(*major, **kw), "material-major inputs, contiguous result": F.plan_cook_torrance(*views, **kw)}
*packed, res = F.pack_maps(*major, reserve_output=True, material_major=True)
plans["material-major maps + result (F.pack_maps)"] = F.plan_cook_torrance(*packed, out=res, **kw)
assert torch.equal(plans["map-major (4 tensors)"].launch(), plans["material-major maps + result (F.pack_maps)"].launch())
assert torch.equal(plans["map-major (4 tensors)"].launch(), plans["material-major inputs, contiguous result"].launch())
times = {k: [] for k in plans}
for rnd in range(5):
    for key, p in plans.items():
        for _ in range(2):
            p.launch(stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            p.launch(stream)
        e1.record()
        torch.cuda.synchronize()
        times[key].append(e0.elapsed_time(e1) / 8 * 1e3)
for key, t in times.items():
    med = statistics.median(t)
    print(f"B={B} {S}^2 {light}: {key:44s} median {med:8.1f} us -> {44 * P * B / med / 1e3:7.1f} GB/s  ({plans[key].kernel_name})", flush=True)
