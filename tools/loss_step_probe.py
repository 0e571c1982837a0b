#!/usr/bin/env python3
"""The rendering-loss step on one 4096^2 material: ONE kernel (pbr_cook_torrance_mse_step: 76 B/pixel) against evaluate +
torch MSE + backward kernel (44 + 36 + 76 B/pixel).   python tools/loss_step_probe.py [size] [f16]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import _native as N, functional as F  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
half = len(sys.argv) > 2 and sys.argv[2] == "f16"
dev = torch.device("cuda", 0)
maps = synth_material(S, dev, 3, torch.float16 if half else torch.float32)
for light_type, light in (("point", [0.1, 0.1, 1.0]), ("directional", [0.3, -0.2, 1.0])):
    kw = dict(view_dir=[0, 0, 1], light=light, light_intensity=[1, 1, 1], light_type=light_type, light_size=1.0 if light_type == "point" else None)
    target = F.cook_torrance(*synth_material(S, dev, 4, maps[0].dtype), **kw).float()
    plan = F.plan_cook_torrance(*maps, **kw)
    grads = [torch.empty_like(t) for t in maps]
    loss = torch.empty((), device=dev)
    lib, stream = N.lib(), torch.cuda.current_stream(dev).cuda_stream
    ws = torch.empty(max(1, lib.pbr_mse_step_workspace_bytes(ctypes.byref(plan.desc)) // 4), device=dev)

    def fused():
        N.check(lib.pbr_cook_torrance_mse_step(ctypes.byref(plan.desc), target.data_ptr(), *[g.data_ptr() for g in grads], None, loss.data_ptr(),
                                               ws.data_ptr(), stream))

    def timed(fn, iters, warm):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    px = S * S
    bpp = (16 + 12 + 16) if half else 76
    ts = {}
    ts[0] = timed(fused, 100, 300)
    leaves = [t.clone().requires_grad_(True) for t in maps]

    def step(fused_path):
        for t in leaves:
            t.grad = None
        if fused_path:
            F.rendering_loss_mse(*leaves, target=target, **kw).backward()
        else:
            torch.nn.functional.mse_loss(F.cook_torrance(*leaves, **kw), target).backward()
    t_f, t_u = timed(lambda: step(True), 20, 10), timed(lambda: step(False), 20, 10)
    for vec, t in ts.items():
        print(f"{light_type:11s} {'fp16' if half else 'fp32'} maps, {vec or 'rule'} px/lane: fused step kernel {t:7.1f} us = {bpp * px / t / 1e3:5.0f} GB/s of its {bpp} B/pixel "
              f"({bpp * px / t / 1e3 / 8000:.3f} of 8 TB/s)")
    print(f"{light_type:11s} whole step through autograd: one kernel {t_f:7.1f} us; evaluate + torch MSE + backward kernel {t_u:7.1f} us ({t_u / t_f:.2f}x)")
