#!/usr/bin/env python3
"""Backward of the fused blend + render: the one-pass kernel (pbr_cook_torrance_blend_backward) against the unfused
differentiable pieces it replaces, 4096^2 materials.   python tools/blend_bwd_probe.py [size]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import _native as N, functional as F  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
m1, m2 = synth_material(S, dev, 1), synth_material(S, dev, 2)
mask = torch.rand(1, S, S, device=dev, generator=g)
kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
plan = F.plan_cook_torrance(*m1, blend=(*m2, None, mask), **kw)
plan.launch()
gout = torch.rand(1, 3, S, S, device=dev, generator=g)
g1 = [torch.empty_like(t) for t in m1]
g2 = [torch.empty_like(t) for t in m2]
gm = torch.empty_like(mask)
G1 = N.MapGrads(g1[0].data_ptr(), g1[1].data_ptr(), g1[2].data_ptr(), g1[3].data_ptr(), None)
G2 = N.MapGrads(g2[0].data_ptr(), g2[1].data_ptr(), g2[2].data_ptr(), g2[3].data_ptr(), None)
bd = N.BlendDesc.from_buffer_copy(plan._blend)
bd.sign_mode = N.BLEND_SIGN_GIVEN
lib, stream = N.lib(), torch.cuda.current_stream(dev).cuda_stream


def fused():
    N.check(lib.pbr_cook_torrance_blend_backward(ctypes.byref(plan.desc), ctypes.byref(bd), plan._workspace.data_ptr(), gout.data_ptr(),
                                                 ctypes.byref(G1), ctypes.byref(G2), gm.data_ptr(), stream))


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
for knob in (0,):
    t = timed(fused, 50, 100)
    print(f"fused blend backward {S}^2: {t:8.1f} us = {148 * px / t / 1e3:6.0f} GB/s of its 148 B/pixel (80 in, 68 out), {px / t / 1e3:6.1f} Gpix/s")

leaves1 = [t.clone().requires_grad_(True) for t in m1]
leaves2 = [t.clone().requires_grad_(True) for t in m2]
lm = mask.clone().requires_grad_(True)


def autograd_step(fused_path):
    for t in leaves1 + leaves2 + [lm]:
        t.grad = None
    second = (*leaves2, None, lm)
    out = F.cook_torrance(*leaves1, blend=second, **kw) if fused_path else F._blend_then_render_with_grad(*leaves1, None, blend=second, **kw)
    out.backward(gout[0])


print(f"forward + backward through autograd, fused kernels: {timed(lambda: autograd_step(True), 10, 5):8.1f} us; "
      f"unfused differentiable pieces: {timed(lambda: autograd_step(False), 10, 5):8.1f} us")
