#!/usr/bin/env python3
"""Runs one batched configuration a few times: a target for rocprofv3 counter passes.
python tools/run_batch.py BATCH SIZE [point|directional] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import maps, timed  # noqa: E402
from pypbr_amd import functional as F  # noqa: E402

B, S = int(sys.argv[1]), int(sys.argv[2])
light = sys.argv[3] if len(sys.argv) > 3 else "point"
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
m = maps(B, S, S, seed=11)
kw = dict(view_dir=[0, 0, 1], light_intensity=[1, 1, 1], light_type=light)
kw.update(dict(light=[0.1, 0.1, 1.0], light_size=1.0) if light == "point" else dict(light=[0.3, -0.2, 1.0]))
p = F.plan_cook_torrance(*m, **kw)
dt = timed([p], iters, warm=2)
px = B * S * S
print(f"{p.kernel_name} B={B} {S}^2: {dt * 1e6:.1f} us, {p.bytes_per_pixel * px / dt / 1e9:.1f} GB/s")
