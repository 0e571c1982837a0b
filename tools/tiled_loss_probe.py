#!/usr/bin/env python3
"""The tutorial's rendering loss (docs/source/tutorials/06_advanced.rst:73-107) over the example's material shape -- maps of SxS under tile(2)
(examples/example_brdf.py:11) -- per training step through autograd (loss + backward), eager:
  one pass      the recorded tile reaches pbr_cook_torrance_mse_step (repeat-inner kernel, map-sized gradients);
  three steps   PBR_TUNE_TILE_REPEAT = 0: render (wrap-around), torch's MSE, backward + fold;
  materialised  what upstream does: map.repeat(1, 2, 2) under autograd, then the untiled one-pass step on the repeated maps.
python tools/tiled_loss_probe.py [sizes ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import _native as N, functional as F      # noqa: E402
from pypbr_amd.losses import RenderingLoss               # noqa: E402
from pypbr_amd.materials import BasecolorMetallicMaterial  # noqa: E402


def step_time(fn, iters=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    sizes = [int(s) for s in sys.argv[1:]] or [256, 512, 1024, 2048]
    lib = N.lib()
    crit = RenderingLoss(light_type="point", light_size=1.0)
    for S in sizes:
        g = torch.Generator(device="cuda").manual_seed(S)
        a = torch.rand(3, S, S, device="cuda", generator=g)
        n = torch.cat([torch.rand(2, S, S, device="cuda", generator=g) - 0.5, torch.ones(1, S, S, device="cuda")], 0)
        r = torch.rand(1, S, S, device="cuda", generator=g) * 0.8 + 0.2
        m = torch.rand(1, S, S, device="cuda", generator=g)
        leaves = [t.requires_grad_(True) for t in (a, n, r, m)]
        target = torch.rand(3, 2 * S, 2 * S, device="cuda", generator=g)

        def material(lazy):
            mat = BasecolorMetallicMaterial(albedo=leaves[0], roughness=leaves[2], metallic=leaves[3], device="cuda")
            mat._raw["normal"] = leaves[1]
            mat.tile(2)
            if not lazy:
                mat.materialize_tile()
            return mat

        def run(lazy):
            for t in leaves:
                t.grad = None
            crit(material(lazy), target).backward()
        one = step_time(lambda: run(True))
        lib.pbr_set_tuning(N.TUNE_TILE_REPEAT, 0)
        three = step_time(lambda: run(True))
        lib.pbr_set_tuning(N.TUNE_TILE_REPEAT, -1)
        mat = step_time(lambda: run(False))
        print(f"{S}^2 maps, tile(2) -> {2 * S}^2: one pass {one:8.1f} us per step | three steps {three:8.1f} | materialised repeat {mat:8.1f}", flush=True)


if __name__ == "__main__":
    main()
