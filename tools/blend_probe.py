#!/usr/bin/env python3
"""Throughput of the stand-alone blend kernels (N4) on 4096^2 maps."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import blending as B  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
for C, normal in ((3, False), (3, True), (1, False)):
    a = torch.rand(C, S, S, device=dev, generator=g) - (0.5 if normal else 0.0)
    b = torch.rand(C, S, S, device=dev, generator=g) - (0.5 if normal else 0.0)
    m = torch.rand(1, S, S, device=dev, generator=g)
    for _ in range(3):
        B.blend_maps(a, b, m, is_normal=normal)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        B.blend_maps(a, b, m, is_normal=normal)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    bytes_ = (3 * C + 1) * 4 * S * S
    print(f"blend_maps C={C} normal={normal}: {us:8.1f} us  {bytes_ / us / 1e3:7.1f} GB/s (incl. torch.empty per call)")

# ---- the example_blend.py pipeline on 4096^2 materials: blend -> (re-decode normal) -> render, unfused vs fused
from bench import synth_material  # noqa: E402
from pypbr_amd import functional as F  # noqa: E402
from pypbr_amd.materials import BasecolorMetallicMaterial  # noqa: E402
from pypbr_amd.models import CookTorranceBRDF  # noqa: E402


def material(seed):
    a, n, r, m = synth_material(S, dev, seed)
    mat = BasecolorMetallicMaterial(albedo=a, roughness=r, metallic=m, device=dev)
    mat._maps["normal"] = n
    return mat


m1, m2 = material(1), material(2)
mask = torch.rand(1, S, S, device=dev, generator=g)
brdf = CookTorranceBRDF("point")
args = (torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0]), 1.0)


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


t_unfused = timed(lambda: brdf(B.blend_with_mask(m1, m2, mask)[0], *args))
t_fused = timed(lambda: brdf(B.blend_with_mask(m1, m2, mask, lazy=True)[0], *args))
plan = F.plan_cook_torrance(m1.albedo, m1.normal, m1.roughness, m1.metallic, blend=(m2.albedo, m2.normal, m2.roughness, m2.metallic, None, mask),
                            view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
t_plan = timed(lambda: plan.launch(), 30)
px = S * S
print(f"blend+render {S}^2: unfused (4 blend kernels + normal re-decode + render, via the material API) {t_unfused:8.1f} us; "
      f"fused via the material API {t_fused:8.1f} us; fused launch alone {t_plan:8.1f} us = {px / t_plan / 1e3:6.1f} Gpix/s, "
      f"{80 * px / t_plan / 1e3:6.0f} GB/s of its 80 B/pixel")
