#!/usr/bin/env python3
"""Random shapes through the two fused training kernels of round 3 -- blend + render (forward and its one-pass backward) and the
rendering-loss step -- against the UNFUSED differentiable pieces of this library (each of which the tests hold to the oracles).
python tools/fused_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import functional as F  # noqa: E402

PICK = [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 46, 63, 64, 65, 100, 127, 128, 129, 130, 257]


def _close(a, b, what, rtol=2e-5):
    err = (a.double() - b.double()).abs()
    if not bool((err <= rtol * (1 + b.double().abs())).all()):
        raise AssertionError(f"{what}: off by {float(err.max()):.2e}")
    return float((err / (1 + b.double().abs())).max())


def run(cases=100, seed=0, verbose=True):
    rng = random.Random(seed)
    worst = 0.0
    for i in range(cases):
        B, H, W = rng.choice([1, 1, 2]), rng.choice(PICK), rng.choice(PICK)
        wf = rng.choice(["metallic", "specular", "converted"])
        lt = rng.choice(["point", "directional"])
        L = 1 if wf == "converted" else rng.choice([1, 1, 2])
        g = torch.Generator().manual_seed(5000 + i)
        rnd = lambda *s: torch.rand(*s, generator=g)

        def material(flat):
            n = torch.cat([(rnd(B, 2, H, W) * 0.3 + 0.1) if flat else (rnd(B, 2, H, W) - 0.5), torch.ones(B, 1, H, W)], 1) * (0.6 + rnd(B, 1, H, W))
            return [rnd(B, 3, H, W), n, rnd(B, 1, H, W) * 0.6 + 0.35, rnd(B, 1, H, W) if wf != "specular" else None,
                    rnd(B, 3, H, W) if wf == "specular" else None]
        flat = rng.random() < 0.3
        m1, m2 = material(flat), material(flat)
        mask = rnd(1, H, W) if rng.random() < 0.5 else rnd(B, 1, H, W)
        view = [rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), 1.0]
        lights = [[rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(0.6, 1.4)] for _ in range(L)]
        inten = [[rng.uniform(0.4, 1.1)] * 3 if rng.random() < 0.5 else [rng.uniform(0.4, 1.1) for _ in range(3)] for _ in range(L)]
        kw = dict(view_dir=view, light=lights if L > 1 else lights[0], light_intensity=inten if L > 1 else inten[0], light_type=lt,
                  light_size=rng.choice([None, 1.0, 2.0]), albedo_is_srgb=rng.random() < 0.7, return_srgb=rng.random() < 0.7)
        if wf == "converted":
            kw.update(convert_to_diffuse_specular=True, specular_is_srgb=rng.random() < 0.5)
        elif wf == "specular":
            kw.update(specular_is_srgb=rng.random() < 0.6)
        wt = (rnd(B, 3, H, W) - 0.5).cuda()
        leaf = lambda ts: [None if t is None else t.clone().cuda().requires_grad_(True) for t in ts]
        # ---- blend + render: fused forward / one-pass backward against blend_maps -> decode_normal -> cook_torrance with their backward kernels
        a1, a2, am = leaf(m1), leaf(m2), mask.clone().cuda().requires_grad_(True)
        b1, b2, bm = leaf(m1), leaf(m2), mask.clone().cuda().requires_grad_(True)
        fused = F.cook_torrance(*a1, blend=(*a2, am), **kw)
        if type(fused.grad_fn).__name__ != "_FusedBlendFnBackward":
            raise AssertionError(f"case {i}: the fused blend path was not taken")
        unfused = F._blend_then_render_with_grad(*b1, blend=(*b2, bm), **kw)
        desc = f"case {i}: B={B} {H}x{W} {wf} {lt} L={L} flat={flat} mask={tuple(mask.shape)}"
        worst = max(worst, _close(fused.detach(), unfused.detach(), desc + " blend forward", 5e-6))
        (fused * wt).sum().backward()
        (unfused * wt).sum().backward()
        for x, y, name in list(zip(a1 + a2, b1 + b2, ["albedo", "normal", "roughness", "metallic", "specular"] * 2)) + [(am, bm, "mask")]:
            if x is not None:
                worst = max(worst, _close(x.grad, y.grad, desc + " blend gradient of " + name, 4e-5))
        # ---- rendering-loss step: one kernel against evaluate + torch MSE + backward kernel
        half = rng.random() < 0.3
        dt = torch.float16 if half else torch.float32
        target = rnd(B, 3, H, W).cuda()
        p1 = [None if t is None else t.to(dt).cuda().requires_grad_(True) for t in m1]
        p2 = [None if t is None else t.to(dt).cuda().requires_grad_(True) for t in m1]
        l1 = F.rendering_loss_mse(*p1, target=target, **kw)
        if type(l1.grad_fn).__name__ != "_MseStepFnBackward":
            raise AssertionError(f"case {i}: the one-kernel loss step was not taken")
        l2 = torch.nn.functional.mse_loss(F.cook_torrance(*p2, **kw).float(), target)
        _close(l1.detach(), l2.detach(), desc + " loss", 2e-6)
        scale = float(rng.choice([1.0, 3.0]))
        (l1 * scale).backward()
        (l2 * scale).backward()
        for x, y, name in zip(p1, p2, ["albedo", "normal", "roughness", "metallic", "specular"]):
            if x is not None:
                gs = float(y.grad.float().abs().max()) + 1e-20             # gradients of a mean over B*3*H*W values are tiny: compare at their own scale
                # fp16 gradients: one fp16 rounding each way, and below 6.1e-5 fp16 is subnormal (steps of 6e-8, another rounding when the
                # upstream scale is applied to the stored values): gradients of a mean over many pixels live there
                tol = (3e-3 + 2.4e-7 / gs) if half else 4e-5
                worst = max(worst, _close(x.grad.float() / gs, y.grad.float() / gs, desc + f" loss gradient of {name} ({'f16' if half else 'f32'})", tol))
        if verbose and i % 10 == 0:
            print(desc + f": worst so far {worst:.2e}", flush=True)
    if verbose:
        print(f"{cases} cases: worst relative difference {worst:.2e}")
    return worst


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
