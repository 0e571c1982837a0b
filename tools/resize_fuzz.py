#!/usr/bin/env python3
"""Random shapes through the resize and its gradient (F.resize: strip kernel, two-tap kernel, two-pass fall-backs; one-pass and two-pass
gradient) against ATen (the closer of its float64 and its fp32 run: the tap positions are formed in fp32, as ATen does for float maps).  python tools/resize_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import functional as F  # noqa: E402

def run(cases=200, seed=0, verbose=True):
    rng = random.Random(seed)
    worst_f = worst_g = 0.0
    for i in range(cases):
        pick = lambda: rng.choice([1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 33, 63, 64, 65, 100, 127, 129, 200, 255, 257, 300, 511, 513])
        h, w, ho, wo = pick(), pick(), pick(), pick()
        planes = rng.choice([1, 3, 4])
        aa = rng.random() < 0.6
        if rng.random() < 0.25:                                          # a whole factor on both axes: the register-only down-scale (resize_down.hpp)
            S = rng.choice([2, 3, 4, 5, 6, 7, 8, 16])
            ho, wo = rng.choice([2, 3, 8, 13, 64, 100]), rng.choice([8, 12, 64, 260, 512])
            h, w, aa = S * ho, S * wo, True
        if aa and 1 in (ho, wo) and (h, w) != (ho, wo):
            # ATen's CPU antialias kernel returns its FIRST output row (column) in every row (column) when the other output extent is 1
            # (torch 2.10: interpolate(rand(1,1,200,1), (31,1), antialias=True) is constant) -- not a reference for these shapes
            aa = False
        g = torch.Generator().manual_seed(i)
        x = torch.rand(1, planes, h, w, generator=g)
        wt = torch.rand(1, planes, ho, wo, generator=g) - 0.5
        off = rng.choice([0, 0, 1, 3])                                   # views that start off a 16-byte boundary
        flat = torch.empty(x.numel() + 8, device="cuda")
        xd = flat[off:off + x.numel()].view_as(x).copy_(x).requires_grad_(True)
        out = F.resize(xd, (ho, wo), antialias=aa)
        (out * wt.cuda()).sum().backward()
        x64 = x.double().requires_grad_(True)
        ref = TF.interpolate(x64, size=(ho, wo), mode="bilinear", align_corners=False, antialias=aa)
        (ref * wt.double()).sum().backward()
        x32 = x.clone().requires_grad_(True)                              # ATen's own fp32 run: tap positions formed in fp32, as here
        r32 = TF.interpolate(x32, size=(ho, wo), mode="bilinear", align_corners=False, antialias=aa)
        (r32 * wt).sum().backward()
        ef = min((out.detach().cpu().double() - ref.detach()).abs().max().item(), (out.detach().cpu() - r32.detach()).abs().max().item())
        eg64 = ((xd.grad.cpu().double() - x64.grad).abs() / (1 + x64.grad.abs())).max().item()
        eg32 = ((xd.grad.cpu() - x32.grad).abs() / (1 + x32.grad.abs())).max().item()
        eg = min(eg64, eg32)
        worst_f, worst_g = max(worst_f, ef), max(worst_g, eg)
        big = max(h, w, ho, wo)                                          # a tap position near `big` carries big * 6e-8 of fp32 rounding into its weight
        bad = ef > 2e-6 + 2e-7 * big or eg > 1e-5 + 4e-7 * big or not bool(torch.isfinite(xd.grad).all())
        if verbose and (bad or i % 25 == 0):
            print(f"{'BAD ' if bad else ''}case {i}: {planes} x {h}x{w} -> {ho}x{wo} aa={aa} off={off}: forward {ef:.2e} gradient {eg:.2e}", flush=True)
        if bad:
            raise AssertionError(f"resize fuzz case {i}: {planes} x {h}x{w} -> {ho}x{wo} aa={aa} off={off}: forward {ef:.2e} gradient {eg:.2e}")
    if verbose:
        print(f"{cases} cases: worst forward error {worst_f:.2e}, worst relative gradient error {worst_g:.2e}")
    return worst_f, worst_g


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
