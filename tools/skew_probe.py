#!/usr/bin/env python3
"""Do batched 2048^2 launches stream faster when the planes' strides are not multiples of 8 MiB?  Same maps, (a) ordinary
contiguous [B,C,H,W] tensors, (b) views whose channel / batch strides carry `pad` extra elements per plane.
python tools/skew_probe.py [size] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import _native as N, functional as F  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
one = synth_material(S, dev, 3)
maps = [torch.stack([t] * B) for t in one]
kw = dict(view_dir=[0, 0, 1], light=[0.3, -0.2, 1.0], light_intensity=[1, 1, 1], light_type="directional")
stream = torch.cuda.current_stream(dev).cuda_stream


def skewed(t, pad):
    b, c, h, w = t.shape
    plane = h * w + pad
    buf = torch.empty(b * c * plane + 64, dtype=t.dtype, device=dev)
    v = buf.as_strided((b, c, h, w), (c * plane, plane, w, 1))
    v.copy_(t)
    return v


def timed(plan, iters=20):
    for _ in range(5):
        plan.launch(stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        plan.launch(stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


ref = None
for pad in (0, 1088, 4352, 16384 + 1088, 65536 + 1088):
    ms = [skewed(t, pad) if pad else t for t in maps]
    out = skewed(torch.empty(B, 3, S, S, device=dev), pad) if pad else None
    for sched, name in ((N.SCHEDULE_LINEAR, "linear"), (N.schedule_xcd(6), "runs64")):
        plan = F.plan_cook_torrance(*ms, out=out, schedule=sched, **kw)
        us = min(timed(plan) for _ in range(3))
        res = plan.launch().clone()
        ref = res if ref is None else ref
        assert torch.equal(res, ref)
        print(f"pad {pad:6d} elements  {name:7s} {us:8.1f} us  {44 * B * S * S / us / 1e3:7.1f} GB/s  ({plan.kernel_name})", flush=True)
