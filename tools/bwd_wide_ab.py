#!/usr/bin/env python3
"""A/B of the streamed backward kernel for fp16 maps (one 4096^2 material, point light, every gradient): 4-byte memory
instructions (PBR_TUNE_BWD_WIDE = 0) against 16-byte ones (1), alternating in one process after a clock-settle run; and the
one-tile kernels (PBR_TUNE_BWD_RUN = 0) for scale.   python tools/bwd_wide_ab.py [size] [workflow]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import _native as N, functional as F  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
spec = len(sys.argv) > 2 and sys.argv[2] == "specular"
dev = torch.device("cuda", 0)
a, n, r, m = synth_material(S, dev, 7, torch.float16)
s = torch.rand(3, S, S, device=dev).half() if spec else None
kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
plan = F.plan_cook_torrance(a, n, r, None if spec else m, s, **kw)
gout = torch.rand(1, 3, S, S, device=dev)
grads = [torch.empty_like(t) for t in (a, n, r)] + [torch.empty_like(s if spec else m)]
lib, stream = N.lib(), torch.cuda.current_stream(dev).cuda_stream


def bwd():
    N.check(lib.pbr_cook_torrance_backward(ctypes.byref(plan.desc), gout.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(), grads[2].data_ptr(),
                                           None if spec else grads[3].data_ptr(), grads[3].data_ptr() if spec else None, stream))


def timed(iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        bwd()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for _ in range(300):
    bwd()
px = S * S
bpp = (16 + 12 + 16) if not spec else (20 + 12 + 20)
res = {}
for rnd in range(4):
    for name, run, wide in (("one-tile", 0, 0), ("stream 4-byte", -1, 0), ("stream 16-byte", -1, 1)):
        lib.pbr_set_tuning(N.TUNE_BWD_RUN, run)
        lib.pbr_set_tuning(N.TUNE_BWD_WIDE, wide)
        for _ in range(20):
            bwd()
        res.setdefault(name, []).append(timed(100))
for name, ts in res.items():
    best = min(ts)
    print(f"{name:16s} {['%.1f' % t for t in ts]} us  -> best {best:.1f} us = {bpp * px / best / 1e3:.0f} GB/s ({bpp * px / best / 1e3 / 8000:.3f} of 8 TB/s), mean {sum(ts) / len(ts):.1f}")
