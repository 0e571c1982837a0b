#!/usr/bin/env python3
"""Where does the time go when CookTorranceBRDF is called on a CPU-resident 4096^2 material?"""
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
a, n, r, m = [t.cpu() for t in synth_material(4096, dev, 6)]
kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)


def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


ms, packed = t(lambda: F.pack_maps(a, n, r, m, device=dev))
print(f"pack_maps (one device allocation + 4 H2D copies): {ms:6.1f} ms")
ms, _ = t(lambda: [x.to(dev) for x in (a, n, r, m)])
print(f"four separate .to(device):                        {ms:6.1f} ms")
ms, out = t(lambda: F.cook_torrance(*packed, **kw))
print(f"render on the device:                             {ms:6.2f} ms")
ms, _ = t(lambda: out.cpu())
print(f"result .cpu() (fresh pageable tensor each time):  {ms:6.1f} ms")
host = torch.empty(out.shape, pin_memory=True)
ms, _ = t(lambda: host.copy_(out))
print(f"result into a reused pinned host tensor:          {ms:6.1f} ms")
mat = BasecolorMetallicMaterial(albedo=a, roughness=r, metallic=m)
mat._maps["normal"] = n
brdf = CookTorranceBRDF("point")
args = (torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0]), 1.0)
ms, _ = t(lambda: brdf(mat, *args))
print(f"CookTorranceBRDF(material on the CPU):            {ms:6.1f} ms")
