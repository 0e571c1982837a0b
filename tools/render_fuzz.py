#!/usr/bin/env python3
"""Random shapes, workflows, flags, light counts and map dtypes through the evaluation and its gradients, against the ATen restatement of
the reference (oracle/torch_oracle.py; gradients: its float64 autograd).  Checker only.  python tools/render_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch_oracle as O  # noqa: E402
from pypbr_amd import functional as F  # noqa: E402

PICK = [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 66, 100, 127, 128, 129, 130, 255, 256, 257, 260]


def run(cases=150, seed=0, verbose=True):
    rng = random.Random(seed)
    worst = worst_g = 0.0
    for i in range(cases):
        B, H, W = rng.choice([1, 1, 2, 3]), rng.choice(PICK), rng.choice(PICK)
        half = rng.random() < 0.3
        wf = rng.choice(["metallic", "metallic", "specular", "converted"])
        lt = rng.choice(["point", "directional"])
        L = 1 if wf == "converted" else rng.choice([1, 1, 1, 2, 3])
        has_normal = rng.random() < 0.85
        flags = dict(albedo_is_srgb=rng.random() < 0.7, return_srgb=rng.random() < 0.7)
        if wf == "specular":
            flags["specular_is_srgb"] = rng.random() < 0.6
        size = rng.choice([None, 1.0, 2.5])
        want_grad = (not half) and rng.random() < 0.4
        g = torch.Generator().manual_seed(1000 + i)
        dt = torch.float16 if half else torch.float32
        q = lambda t: t.to(dt).float()                                       # the oracle gets the exact up-casts of fp16 maps
        a = q(torch.rand(B, 3, H, W, generator=g))
        n = q(torch.nn.functional.normalize(torch.cat([torch.rand(B, 2, H, W, generator=g) - 0.5, torch.rand(B, 1, H, W, generator=g) * 0.8 + 0.2], 1), dim=1)) \
            if has_normal else None
        r = q(torch.rand(B, 1, H, W, generator=g) * 0.8 + 0.2)
        m = q(torch.rand(B, 1, H, W, generator=g)) if wf != "specular" else None
        s = q(torch.rand(B, 3, H, W, generator=g)) if wf == "specular" else None
        view = [rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), 1.0]
        lights = [[rng.uniform(-0.6, 0.6), rng.uniform(-0.6, 0.6), rng.uniform(0.5, 1.5)] for _ in range(L)]
        inten = [[rng.uniform(0.3, 1.2) for _ in range(3)] for _ in range(L)]
        maps_cpu = [a, n, r, m, s]
        okw = dict(view=torch.tensor(view), light_type=lt, light_size=size, **{k: v for k, v in flags.items()})
        if wf == "converted":
            quirk = rng.random() < 0.5
            okw.pop("specular_is_srgb", None)

        def oracle(maps, dtype):
            c = [None if t is None else t.to(dtype) for t in maps]
            kw = dict(okw, view=okw["view"].to(dtype))
            if wf == "converted":
                return O.cook_torrance_batched(*c, converted=True, quirk_specular_srgb=quirk, light=torch.tensor(lights[0], dtype=dtype),
                                               intensity=torch.tensor(inten[0], dtype=dtype), **kw)
            if L > 1:
                return O.cook_torrance_batched(*c, lights=torch.tensor(lights, dtype=dtype), intensities=torch.tensor(inten, dtype=dtype), **kw)
            return O.cook_torrance_batched(*c, light=torch.tensor(lights[0], dtype=dtype), intensity=torch.tensor(inten[0], dtype=dtype), **kw)

        dev = [None if t is None else t.to(dt).cuda() for t in maps_cpu]
        if want_grad:
            dev = [None if t is None else t.requires_grad_(True) for t in dev]
        fkw = dict(view_dir=view, light=lights if L > 1 else lights[0], light_intensity=inten if L > 1 else inten[0], light_type=lt, light_size=size, **flags)
        want_params = want_grad and wf != "converted" and rng.random() < 0.5        # gradients of view / light / intensity too (the light-gradient kernels)
        if want_params:
            pv = torch.tensor(view, requires_grad=True)
            pl = torch.tensor(lights if L > 1 else lights[0], requires_grad=True)
            pi = torch.tensor(inten if L > 1 else inten[0], requires_grad=True)
            fkw.update(view_dir=pv, light=pl, light_intensity=pi)
        if wf == "converted":
            fkw.update(convert_to_diffuse_specular=True, specular_is_srgb=quirk)
        out = F.cook_torrance(*dev, **fkw)
        ref = oracle(maps_cpu, torch.float32)
        err = (out.detach().float().cpu() - ref).abs().max().item()
        worst = max(worst, err)
        desc = f"case {i}: B={B} {H}x{W} {wf} {lt} L={L} {'f16' if half else 'f32'} normal={has_normal} {flags} size={size}"
        if not err <= 1e-5:
            raise AssertionError(f"render fuzz {desc}: forward error {err:.2e}")
        eg = 0.0
        if want_grad:
            wt = torch.rand(out.shape, generator=g) - 0.5
            (out * wt.cuda()).sum().backward()
            leaves = [None if t is None else t.double().requires_grad_(True) for t in maps_cpu]
            if want_params:
                qv = torch.tensor(view, dtype=torch.float64, requires_grad=True)
                ql = torch.tensor(lights, dtype=torch.float64, requires_grad=True)
                qi = torch.tensor(inten, dtype=torch.float64, requires_grad=True)
                kw64 = {k: v for k, v in okw.items() if k != "view"}
                if L > 1:
                    ref64 = O.cook_torrance_batched(*leaves, lights=ql, intensities=qi, view=qv, **kw64)
                else:
                    ref64 = O.cook_torrance_batched(*leaves, light=ql[0], intensity=qi[0], view=qv, **kw64)
                (ref64 * wt.double()).sum().backward()
                for name, got, want in (("view", pv.grad, qv.grad), ("light", pl.grad, ql.grad if L > 1 else ql.grad[0]), ("intensity", pi.grad, qi.grad if L > 1 else qi.grad[0])):
                    e = float((got.double().cpu() - want).abs().max() / (1 + want.abs().max()))
                    eg = max(eg, e)
                    if not e <= 5e-5:
                        raise AssertionError(f"render fuzz {desc}: gradient of {name} off by {e:.2e}")
            else:
                (oracle(leaves, torch.float64) * wt.double()).sum().backward()
            for name, x, y in zip(("albedo", "normal", "roughness", "metallic", "specular"), dev, leaves):
                if x is None:
                    continue
                e = ((x.grad.cpu().double() - y.grad).abs() / (1 + y.grad.abs())).max().item()
                eg = max(eg, e)
                if not e <= 5e-5:
                    raise AssertionError(f"render fuzz {desc}: gradient of {name} off by {e:.2e}")
            worst_g = max(worst_g, eg)
        if verbose and i % 20 == 0:
            print(f"{desc}: forward {err:.2e}" + (f" gradients {eg:.2e}" if want_grad else ""), flush=True)
    if verbose:
        print(f"{cases} cases: worst forward error {worst:.2e}, worst relative gradient error {worst_g:.2e}")
    return worst, worst_g


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
