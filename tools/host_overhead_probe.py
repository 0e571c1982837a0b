#!/usr/bin/env python3
"""Host time per call of the drop-in surface (CookTorranceBRDF.forward on a device-resident material) against the kernel time, by map size.
python tools/host_overhead_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import functional as F  # noqa: E402
from pypbr_amd.materials import BasecolorMetallicMaterial  # noqa: E402
from pypbr_amd.models import CookTorranceBRDF  # noqa: E402

dev = torch.device("cuda", 0)
view, light, inten = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0])
brdf = CookTorranceBRDF(light_type="point")
for S in (256, 512, 1024, 2048, 4096):
    a, n, r, m = synth_material(S, dev, 1)
    mat = BasecolorMetallicMaterial(albedo=a, normal=n, roughness=r, metallic=m).to(dev)
    for cached in (False, True):
        F.set_caching(cached)
        for _ in range(20):
            brdf(mat, view, light, inten, 1.0)
        torch.cuda.synchronize()
        iters = 200
        t0 = time.perf_counter()
        for _ in range(iters):
            brdf(mat, view, light, inten, 1.0)
        t_host = (time.perf_counter() - t0) / iters * 1e6
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / iters * 1e6
        print(f"{S}^2 caching={cached}: CookTorranceBRDF.forward host {t_host:7.1f} us per call, with the GPU drained {t_all:7.1f} us", flush=True)
    F.set_caching(False)
    plan = F.plan_cook_torrance(*[mat._maps[k] for k in ("albedo", "normal", "roughness", "metallic")], view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0],
                                light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
    for _ in range(20):
        plan.launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        plan.launch()
    t_host = (time.perf_counter() - t0) / 200 * 1e6
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 200 * 1e6
    print(f"{S}^2 plan.launch(): host {t_host:7.1f} us per call, with the GPU drained {t_all:7.1f} us", flush=True)
