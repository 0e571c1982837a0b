#!/usr/bin/env python3
"""Throughput of the stand-alone map kernels (N1/N2 rows) on 4096^2 maps: algorithmic bytes / time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import functional as F  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
a = torch.rand(3, S, S, device=dev, generator=g)
m = torch.rand(1, S, S, device=dev, generator=g)
n = torch.rand(3, S, S, device=dev, generator=g)
px = S * S


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


for name, fn, bpp in (("srgb_to_linear (3 ch)", lambda: F.srgb_to_linear(a), 24), ("linear_to_srgb (3 ch)", lambda: F.linear_to_srgb(a), 24),
                      ("srgb_to_linear fp16", lambda h=a.half(): F.srgb_to_linear(h), 12),
                      ("metallic -> diffuse/specular", lambda: F.metallic_to_diffuse_specular(a, m, True), 40),
                      ("diffuse/specular -> basecolor/metallic", lambda: F.diffuse_specular_to_basecolor_metallic(a, n, False), 48),
                      ("decode_normal 3 ch (flag pass + transform)", lambda: F.decode_normal(n), 36),
                      ("decode_normal 2 ch", lambda: F.decode_normal(n[:2]), 20),
                      ("resize -> half size, antialias", lambda: F.resize(a, (S // 2, S // 2)), 15),
                      ("resize -> same size x 1.5, no antialias", lambda: F.resize(a[:, :S // 2, :S // 2], (S * 3 // 4, S * 3 // 4), antialias=False), None)):
    us = timed(fn)
    rate = "" if bpp is None else f"{bpp * px / us / 1e3:7.0f} GB/s of {bpp} B/pixel"
    print(f"{name:46s} {us:8.1f} us  {rate}", flush=True)
