#!/usr/bin/env python3
"""Runs the 16-light fp16 configuration (BASELINE.json config 5, per-GPU share) a few times: a target
for rocprofv3 counter passes.  python tools/run_multilight.py [iters]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import maps, timed  # noqa: E402
from pypbr_amd import functional as F  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
s5 = maps(4, 4096, 4096, dtype=torch.float16, seed=5)
lights = [[math.cos(t), math.sin(t), 1.0] for t in [2 * math.pi * i / 16 for i in range(16)]]
p = F.plan_cook_torrance(*s5, view_dir=[0, 0, 1], light=lights, light_intensity=[[1.0 / 16] * 3] * 16,
                         light_type="point", light_size=1.0)
dt = timed([p], iters, warm=1)
px = 4 * 4096 * 4096
print(f"{p.kernel_name}: {dt * 1e6:.1f} us, {px * 16 / dt / 1e9:.1f} G light-evaluations/s")
