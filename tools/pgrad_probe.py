#!/usr/bin/env python3
"""Backward with the light / view / intensity gradients (PGRAD kernels + the fp64 finish kernel) on one 4096^2 material."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import _native as N, functional as F  # noqa: E402

dev = torch.device("cuda", 0)
S = 4096
lib, stream = N.lib(), torch.cuda.current_stream(dev).cuda_stream
for dtype in (torch.float32, torch.float16):
    for lights in (1, 4):
        maps = [t.to(dtype) for t in synth_material(S, dev, 7)]
        lv = [[0.1 + 0.2 * i, 0.1, 1.0] for i in range(lights)]
        plan = F.plan_cook_torrance(*maps, view_dir=[0, 0, 1], light=lv, light_intensity=[[0.5, 0.5, 0.5]] * lights, light_type="point", light_size=1.0)
        gout = torch.rand(1, 3, S, S, device=dev)
        grads = [torch.empty_like(t) for t in maps]
        gp = torch.empty(3 + 6 * lights, device=dev)
        ws = torch.empty(lib.pbr_param_grad_workspace_bytes(ctypes.byref(plan.desc)) // 4 + 1, device=dev)

        def run(with_params):
            if with_params:
                N.check(lib.pbr_cook_torrance_backward_params(ctypes.byref(plan.desc), gout.data_ptr(), *[g.data_ptr() for g in grads], None,
                                                              gp.data_ptr(), ws.data_ptr(), stream))
            else:
                N.check(lib.pbr_cook_torrance_backward(ctypes.byref(plan.desc), gout.data_ptr(), *[g.data_ptr() for g in grads], None, stream))
        for with_params in (False, True):
            for _ in range(60):
                run(with_params)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                run(with_params)
            e1.record()
            torch.cuda.synchronize()
            print(f"{str(dtype):14s} {lights} light(s)  {'maps + view/light/intensity' if with_params else 'maps only':28s} {e0.elapsed_time(e1) / 30 * 1e3:8.1f} us", flush=True)
