#!/usr/bin/env python3
"""The reference-shaped user flows at 2048^2, for a kernel trace that shows EVERYTHING a call enqueues (our kernels and
torch's glue):   rocprofv3 --kernel-trace --stats -- python3 tools/flow_trace.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd.materials import BasecolorMetallicMaterial  # noqa: E402
from pypbr_amd.models import CookTorranceBRDF  # noqa: E402
import pypbr_amd.blending as B  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # 1..4: that flow alone
dev = torch.device("cuda", 0)
S = 2048
view, light, inten = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0])
brdf = CookTorranceBRDF(light_type="point")


def material(seed):
    a, n, r, m = (t.cpu() for t in synth_material(S, dev, seed))
    return BasecolorMetallicMaterial(albedo=a, normal=n, roughness=r, metallic=m).to(dev)


m1, m2 = material(1), material(2)
torch.cuda.synchronize()
print("flow 1: forward x", reps)
for _ in range(reps if only in (0, 1) else 1):
    out = brdf(m1, view, light, inten, 1.0)
torch.cuda.synchronize()
print("flow 2: rendering loss step (albedo + light position learnable) x", reps)
target = out.detach()
albedo = m1.albedo.clone().requires_grad_()
lpos = light.clone().to(dev).requires_grad_()
for _ in range(reps if only in (0, 2) else 0):
    mat = BasecolorMetallicMaterial(albedo=albedo, normal=m1.normal, roughness=m1.roughness, metallic=m1.metallic).to(dev)
    loss = (brdf(mat, view, lpos, inten, 1.0) - target).square().mean()
    loss.backward()
torch.cuda.synchronize()
print("flow 3: height blend + render (lazy, fused) x", reps)
m1.height = torch.rand(1, S, S, device=dev)
m2.height = torch.rand(1, S, S, device=dev)
for _ in range(reps if only in (0, 3) else 0):
    with B.lazy_blending():
        blended, mask = B.HeightBlend(blend_width=0.1, shift=0.0)(m1, m2)
    out = brdf(blended, view, light, inten, 1.0)
torch.cuda.synchronize()
print("flow 4: resize 2048 -> 1024 + tile(2, lazy) + render x", reps)
for _ in range(reps if only in (0, 4) else 0):
    m = m1.clone().resize((1024, 1024))
    m.tile(2, lazy=True)
    out = brdf(m, view, light, inten, 1.0)
torch.cuda.synchronize()
print("done")
