#!/usr/bin/env python3
"""Host-side cost of one evaluation of a small (256^2) material: CookTorranceBRDF call vs functional call vs
a prepared plan's launch vs a captured HIP graph of 16 launches."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import functional as F  # noqa: E402
from pypbr_amd.materials import BasecolorMetallicMaterial  # noqa: E402
from pypbr_amd.models import CookTorranceBRDF  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
a, n, r, m = synth_material(S, dev, 3)
mat = BasecolorMetallicMaterial(albedo=a, roughness=r, metallic=m, device=dev)
mat._maps["normal"] = n
brdf = CookTorranceBRDF("point")
view, light, inten = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0])
kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
plan = F.plan_cook_torrance(a, n, r, m, **kw)
stream = torch.cuda.current_stream(dev).cuda_stream


def wall(fn, iters=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


print(f"{S}^2 material, wall time per evaluation (2000 back-to-back calls, one sync at the end):")
print(f"  CookTorranceBRDF(material, tensors...)   {wall(lambda: brdf(mat, view, light, inten, 1.0)):8.1f} us")
print(f"  functional.cook_torrance(maps, lists)    {wall(lambda: F.cook_torrance(a, n, r, m, **kw)):8.1f} us")
print(f"  RenderPlan.launch()                      {wall(lambda: plan.launch(stream)):8.1f} us")
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    plan.launch(side.cuda_stream)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        for _ in range(16):
            plan.launch(side.cuda_stream)
print(f"  HIP graph of 16 launches, per launch     {wall(g.replay, 500) / 16:8.1f} us")
