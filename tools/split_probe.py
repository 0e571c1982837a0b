#!/usr/bin/env python3
"""Does one very long launch stream slower than the same work cut into shorter launches?
python tools/split_probe.py BATCH SIZE [point|directional]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import maps  # noqa: E402
from pypbr_amd import functional as F  # noqa: E402

B, S = int(sys.argv[1]), int(sys.argv[2])
light = sys.argv[3] if len(sys.argv) > 3 else "point"
m = maps(B, S, S, seed=11)
kw = dict(view_dir=[0, 0, 1], light_intensity=[1, 1, 1], light_type=light)
kw.update(dict(light=[0.1, 0.1, 1.0], light_size=1.0) if light == "point" else dict(light=[0.3, -0.2, 1.0]))
out = torch.empty(B, 3, S, S, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
px = B * S * S
for parts in (1, 2, 4, 8, 16, 32, 64):
    if parts > B:
        break
    n = B // parts
    plans = [F.plan_cook_torrance(*[t[i * n:(i + 1) * n] for t in m], out=out[i * n:(i + 1) * n], **kw) for i in range(parts)]
    best = 1e9
    for r in range(5):
        for p in plans:
            p.launch(stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            for p in plans:
                p.launch(stream)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 3 * 1e-3)
    print(f"B={B} {S}^2 {light}: {parts:3d} launches of {n:3d} materials: {best * 1e6:8.1f} us  {plans[0].bytes_per_pixel * px / best / 1e9:7.1f} GB/s", flush=True)
