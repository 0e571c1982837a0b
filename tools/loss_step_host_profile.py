#!/usr/bin/env python3
"""Where the host time of one rendering-loss training step goes (small maps: the step is host-bound): cProfile over 300 steps of
RenderingLoss(material.tile(2), target) + backward on 256^2 maps.   python tools/loss_step_host_profile.py [size] [tile]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd.losses import RenderingLoss               # noqa: E402
from pypbr_amd.materials import BasecolorMetallicMaterial  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.rand(3, S, S, device="cuda", generator=g).requires_grad_(True)
n = torch.cat([torch.rand(2, S, S, device="cuda", generator=g) - 0.5, torch.ones(1, S, S, device="cuda")], 0).requires_grad_(True)
r = (torch.rand(1, S, S, device="cuda", generator=g) * 0.8 + 0.2).requires_grad_(True)
m = torch.rand(1, S, S, device="cuda", generator=g).requires_grad_(True)
target = torch.rand(3, T * S, T * S, device="cuda", generator=g)
crit = RenderingLoss(light_type="point", light_size=1.0)


def step():
    for t in (a, n, r, m):
        t.grad = None
    mat = BasecolorMetallicMaterial(albedo=a, roughness=r, metallic=m, device="cuda")
    mat._raw["normal"] = n
    if T > 1:
        mat.tile(T)
    crit(mat, target).backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    step()
torch.cuda.synchronize()
print("%.1f us per step" % ((time.perf_counter() - t0) / 300 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
