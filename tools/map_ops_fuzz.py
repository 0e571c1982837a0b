#!/usr/bin/env python3
"""Random shapes, dtypes and alignments through the stand-alone map operations -- colour transfers, both workflow conversions, normal
decode, blends and masks -- and their gradients, against the ATen restatements of the reference (oracle/torch_oracle.py,
oracle/blend_oracle.py; gradients: float64 autograd).  Checker only.  python tools/map_ops_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import blend_oracle as BO  # noqa: E402
import torch_oracle as O  # noqa: E402
from pypbr_amd import blending as B, functional as F  # noqa: E402

PICK = [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65, 100, 127, 128, 129, 255, 257, 300]


def _offset_copy(t, off):
    """The same values in a device tensor that starts `off` elements past an allocation's (256-byte aligned) start."""
    flat = torch.empty(t.numel() + 8, dtype=t.dtype, device="cuda")
    return flat[off:off + t.numel()].view(t.shape).copy_(t)


def _check(got, want, what, atol, rtol=0.0):
    err = (got.detach().double().cpu() - want.detach().double()).abs()
    lim = atol + rtol * want.detach().double().abs()
    if not bool((err <= lim).all()):
        raise AssertionError(f"{what}: off by {float((err - lim).max()) + atol:.2e}")


def run(cases=120, seed=0, verbose=True):
    rng = random.Random(seed)
    for i in range(cases):
        H, W = rng.choice(PICK), rng.choice(PICK)
        B_ = rng.choice([None, 1, 2])
        lead = () if B_ is None else (B_,)
        half = rng.random() < 0.25
        dt = torch.float16 if half else torch.float32
        off = rng.choice([0, 0, 1, 2, 3])
        g = torch.Generator().manual_seed(9000 + i)
        rnd = lambda *s: torch.rand(*s, generator=g)
        q = lambda t: t.to(dt).float()
        dev = lambda t, grad=False: _offset_copy(t.to(dt), off).requires_grad_(grad)
        tol = 2e-3 if half else 2e-6                                     # fp16: the result's own rounding
        desc = f"case {i}: {lead + (H, W)} {'f16' if half else 'f32'} off={off}"
        # ---- colour transfers (functions.py:31-66), values outside [0,1] included
        x = q(rnd(*lead, 3, H, W) * 1.4 - 0.2)
        for fn, ofn in ((F.srgb_to_linear, O.srgb_to_linear), (F.linear_to_srgb, O.linear_to_srgb)):
            _check(fn(dev(x)), ofn(x), desc + " " + fn.__name__, tol)
        if not half:
            xd, x64 = dev(x, True), x.double().requires_grad_(True)
            wt = rnd(*lead, 3, H, W) - 0.5
            (F.srgb_to_linear(xd) * wt.cuda()).sum().backward()
            (O.srgb_to_linear(x64) * wt.double()).sum().backward()
            safe = ((x - 0.04045).abs() > 1e-4) & (x.abs() > 1e-4) & ((x - 1).abs() > 1e-4)      # the knee and the clamp ends are kinks
            err = (xd.grad.cpu().double() - x64.grad).abs()
            if not bool((err <= 2e-5 * (1 + x64.grad.abs()))[safe].all()):
                raise AssertionError(desc + f" srgb_to_linear gradient off by {float(err[safe].max()):.2e}")
        # ---- workflow conversions (metallic.py:98-108, diffuse.py:128-147)
        a, m = q(rnd(*lead, 3, H, W)), q(rnd(*lead, 1, H, W))
        srgb = rng.random() < 0.5
        d_, s_ = F.metallic_to_diffuse_specular(dev(a), dev(m), albedo_is_srgb=srgb)
        od, os_ = O.metallic_to_diffuse_specular(O.srgb_to_linear(a) if srgb else a, m)
        _check(d_, od, desc + " metallic_to_diffuse_specular diffuse", tol)
        _check(s_, os_, desc + " metallic_to_diffuse_specular specular", tol)
        if not half:
            dd, ss = q(rnd(*lead, 3, H, W)) * 0.9 + 0.05, q(rnd(*lead, 3, H, W)) * 0.9 + 0.05
            bc, mm = F.diffuse_specular_to_basecolor_metallic(dev(dd), dev(ss), albedo_is_srgb=False)
            obc, omm = O.diffuse_specular_to_basecolor_metallic(dd, ss)
            # thresholded selects (den < eps, metallic >= 0.95): compare away from the thresholds
            den = dd - 0.04 + 1e-6
            ok = (den.abs() > 1e-3) & ((omm - 0.95).abs() > 1e-3)
            for got, want, name in ((bc, obc, "basecolor"), (mm, omm, "metallic")):
                err = (got.cpu() - want).abs()
                if not bool((err <= 1e-4)[ok if name == "metallic" else ok.expand_as(err)].all()):
                    raise AssertionError(desc + f" diffuse_specular_to_basecolor_metallic {name} off by {float(err[ok.expand_as(err)].max()):.2e}")
        # ---- normal decode (base.py:191-242): [0,1]-encoded 3 channels, signed 3 channels, 2 channels
        enc = q(rnd(*lead, 3, H, W))
        sgn = q(rnd(*lead, 3, H, W) - 0.5)
        two = q(rnd(*lead, 2, H, W) * 0.8 + 0.1)
        for nmap, name in ((enc, "encoded"), (sgn, "signed"), (two, "two channels")):
            if nmap.numel() == 0:
                continue
            one = nmap[0] if B_ is not None else nmap                     # (C,H,W) maps, as materials store them
            _check(F.decode_normal(dev(one)), O.decode_normal(one), desc + " decode_normal " + name,
                   3e-3 if half else (2e-5 if name == "two channels" else 3e-6))      # z = sqrt(1 - x^2 - y^2): cancellation near z = 0
        # ---- blends and masks (blending/functional.py:64-286), fp32 maps
        if not half:
            m1, m2 = rnd(3, H, W), rnd(3, H, W)
            k = rnd(1, H, W)
            _check(B.blend_maps(_offset_copy(m1, off), _offset_copy(m2, off), _offset_copy(k, off)), BO.blend_maps(m1, m2, k), desc + " blend_maps", 2e-6)
            n1 = torch.cat([rnd(2, H, W) - 0.5, torch.ones(1, H, W)], 0)
            n2 = torch.cat([rnd(2, H, W) - 0.5, torch.ones(1, H, W)], 0)
            _check(B.blend_maps(_offset_copy(n1, off), _offset_copy(n2, off), _offset_copy(k, off), is_normal=True), BO.blend_normals(n1, n2, k),
                   desc + " blend normals", 3e-6)
            h1, h2 = rnd(1, H, W), rnd(1, H, W)
            bw, sh = rng.choice([0.05, 0.2, 1.0]), rng.choice([0.0, -0.3, 0.4])
            _check(B.sigmoid_mask(_offset_copy(h1, off), _offset_copy(h2, off), bw, sh), BO.sigmoid_mask(h1, h2, bw, sh), desc + " sigmoid mask", 3e-6)
            for direction in ("horizontal", "vertical"):
                _check(B.gradient_mask(H, W, direction, "cuda"), BO.gradient_mask(H, W, direction), desc + " gradient mask " + direction, 1e-6)
            a1, a2, ak = (_offset_copy(t, off).requires_grad_(True) for t in (n1, n2, k))
            r1, r2, rk = (t.double().requires_grad_(True) for t in (n1, n2, k))
            wt = rnd(3, H, W) - 0.5
            (B.blend_maps(a1, a2, ak, is_normal=True) * wt.cuda()).sum().backward()
            (BO.blend_normals(r1, r2, rk) * wt.double()).sum().backward()
            for x_, y_, name in ((a1, r1, "normal 1"), (a2, r2, "normal 2"), (ak, rk, "mask")):
                _check(x_.grad, y_.grad, desc + " blend normals gradient of " + name, 2e-5, 2e-5)
        if verbose and i % 20 == 0:
            print(desc + ": ok", flush=True)
    if verbose:
        print(f"{cases} cases passed")


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
