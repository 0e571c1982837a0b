#!/usr/bin/env python3
"""tile(n) fused -- the repeat-inner kernel (whole output, one light) and wrap-around addressing (PBR_TUNE_TILE_REPEAT = 0, row bands, several lights) -- on random shapes: the image equals the one evaluated on the materialised repeat bit for bit, for
every workgroup order, row bands included; gradients through the folded backward (one kernel, or backward + fold).  python tools/tile_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import _native as N, functional as F  # noqa: E402


def run(cases=60, seed=0, verbose=True):
    rng = random.Random(seed)
    lib = N.lib()
    try:
        for i in range(cases):
            half = rng.random() < 0.5
            dt = torch.float16 if half else torch.float32
            h = rng.choice([1, 3, 8, 16, 24, 32, 48, 64, 96, 128])
            w = rng.choice([4, 8, 36, 64, 128, 256, 512, 520, 1024, 1536]) if rng.random() < 0.7 else rng.choice([5, 7, 33, 130])
            ny, nx = rng.choice([1, 2, 3, 4]), rng.choice([1, 2, 3])
            B = rng.choice([1, 1, 1, 2])
            lights = rng.choice([1, 1, 2])
            g = torch.Generator().manual_seed(3000 + i)
            a = torch.rand(B, 3, h, w, generator=g).cuda().to(dt)
            n = torch.cat([torch.rand(B, 2, h, w, generator=g) - 0.5, torch.ones(B, 1, h, w)], 1).cuda().to(dt)
            r = (torch.rand(B, 1, h, w, generator=g) * 0.8 + 0.2).cuda().to(dt)
            m = torch.rand(B, 1, h, w, generator=g).cuda().to(dt)
            L = [[0.2, -0.1, 0.9], [-0.3, 0.3, 0.7]][:lights]
            kw = dict(view_dir=[0.1, 0, 1], light=L if lights > 1 else L[0], light_intensity=[[1, 0.9, 0.8]] * lights if lights > 1 else [1, 0.9, 0.8],
                      light_type=rng.choice(["point", "directional"]), light_size=2.0)
            rep = lambda t: t.repeat(1, 1, ny, nx)
            ref = F.cook_torrance(rep(a), rep(n), rep(r), rep(m), **kw)
            rng.choice([-1, 0, 1, 2, 3, 5, 8])                      # (the fold knob's draw of round 4: kept so that the cases stay the same)
            sched = rng.choice([N.SCHEDULE_AUTO, N.SCHEDULE_LINEAR, N.schedule_xcd(1), N.schedule_xcd(3), N.schedule_xcd(6)])
            repeat = rng.choice([-1, -1, 0])
            lib.pbr_set_tuning(N.TUNE_TILE_REPEAT, repeat)
            desc = f"case {i}: B={B} {h}x{w} tile=({ny},{nx}) {'f16' if half else 'f32'} lights={lights} schedule={sched} repeat={repeat}"
            out = F.cook_torrance(a, n, r, m, tile=(ny, nx), schedule=sched, **kw)
            if not torch.equal(out, ref):
                raise AssertionError(desc + f": differs from the materialised repeat by {float((out.float() - ref.float()).abs().max()):.2e}")
            if rng.random() < 0.35:
                # gradients: a map repeated by the fused tile owns the SUM of the per-output-pixel gradients (pbr_fold_gradient[_typed]) -- what
                # autograd gives through the materialised repeat
                wt = (torch.rand(ref.shape, generator=g) - 0.5).cuda()
                la = [t.clone().requires_grad_(True) for t in (a, n, r, m)]
                lb = [t.clone().requires_grad_(True) for t in (a, n, r, m)]
                (F.cook_torrance(*la, tile=(ny, nx), schedule=sched, **kw).float() * wt).sum().backward()
                (F.cook_torrance(*[rep(t) for t in lb], **kw).float() * wt).sum().backward()
                for name, x, y in zip(("albedo", "normal", "roughness", "metallic"), la, lb):
                    gx, gy = x.grad.float(), y.grad.float()
                    tol = (2e-3 if half else 2e-6) * float(gy.abs().max()) + (1e-3 if half else 0.0) * gy.abs()      # fp16: one rounding per repeat against one of the sum
                    if not bool(((gx - gy).abs() <= tol + 1e-12).all()):
                        raise AssertionError(desc + f": gradient of {name} differs from the materialised repeat's by {float((gx - gy).abs().max()):.2e} (max {float(gy.abs().max()):.2e})")
            if (ny, nx) == (1, 1):
                continue
            H = ny * h
            y0 = rng.randrange(0, H)
            rows = rng.randrange(1, H - y0 + 1)
            band = F.cook_torrance(a, n, r, m, tile=(ny, nx), y_offset=y0, rows=rows, schedule=sched, **kw)
            if not torch.equal(band, ref[:, :, y0:y0 + rows]):
                raise AssertionError(desc + f": band rows {y0}..{y0 + rows} differ")
            if verbose and i % 10 == 0:
                print(desc + ": ok", flush=True)
    finally:
        lib.pbr_set_tuning(N.TUNE_TILE_REPEAT, -1)
    if verbose:
        print(f"{cases} cases passed")


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
