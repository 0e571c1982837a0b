#!/usr/bin/env python3
"""Backward kernel, 4096^2 point/metallic: does it matter where the 19 plane streams (8 maps + 3 upstream-gradient planes
in, 8 gradient planes out) sit?  Separate tensors vs gradients in one allocation vs everything in one allocation."""
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import _native as N, functional as F  # noqa: E402

S = 4096
dev = torch.device("cuda", 0)
P = S * S
L = int(os.environ.get("PBR_PROBE_LIGHTS", "1"))          # several lights: the two-pass form of the backward kernel
kw = dict(view_dir=[0, 0, 1], light=[[0.1 + 0.05 * i, 0.1 - 0.03 * i, 1.0] for i in range(L)], light_intensity=[[1.0 / L] * 3] * L,
          light_type="point", light_size=1.0)
stream = torch.cuda.current_stream(dev).cuda_stream
lib = N.lib()


def case(maps, gout, grads):
    plan = F.plan_cook_torrance(*maps, **kw)
    return lambda: N.check(lib.pbr_cook_torrance_backward(ctypes.byref(plan.desc), gout.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(),
                                                          grads[2].data_ptr(), grads[3].data_ptr(), None, stream)), plan


src = synth_material(S, dev, 9)
g0 = torch.rand(1, 3, S, S, device=dev)
cases = {}
cases["separate tensors"] = case(src, g0, [torch.empty(1, c, S, S, device=dev) for c in (3, 3, 1, 1)])
one = torch.empty(8 * P, device=dev)
cases["gradients in one allocation"] = case(src, g0, [one[o * P:(o + c) * P].view(1, c, S, S) for o, c in ((0, 3), (3, 3), (6, 1), (7, 1))])
arena = torch.empty(19 * P, device=dev)
views, o = [], 0
for t in src:
    c = t.shape[0]
    v = arena[o * P:(o + c) * P].view(c, S, S)
    v.copy_(t)
    views.append(v)
    o += c
gv = arena[8 * P:11 * P].view(1, 3, S, S)
gv.copy_(g0)
cases["maps, upstream gradient and gradients in one allocation"] = case(views, gv, [arena[(11 + o) * P:(11 + o + c) * P].view(1, c, S, S)
                                                                                    for o, c in ((0, 3), (3, 3), (6, 1), (7, 1))])
times = {k: [] for k in cases}
for rnd in range(7):
    for key, (fn, _) in cases.items():
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[key].append(e0.elapsed_time(e1) / 20 * 1e3)
for key, t in times.items():
    med = statistics.median(t)
    print(f"{key:58s} median {med:7.2f} us -> {76 * P / med / 1e3:7.1f} GB/s", flush=True)
