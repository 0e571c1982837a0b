#!/usr/bin/env python3
"""Backward kernel by number of lights: time against the HBM time of its bytes, to see where it stops being bound by memory.
python tools/bwd_lights_probe.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import _native as N, functional as F  # noqa: E402

lib = N.lib()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
S = 4096
for dtype in (torch.float16, torch.float32):
    maps = [t.to(dtype) for t in synth_material(S, dev, 3)]
    gout = torch.rand(1, 3, S, S, device=dev)
    grads = [torch.empty_like(t) for t in maps]
    for L, inten in ((1, [1, 1, 1]), (1, [1, 0.9, 0.8]), (2, [1, 1, 1]), (2, [1, 0.9, 0.8]), (4, [1, 0.9, 0.8]), (8, [1, 0.9, 0.8]), (16, [1, 0.9, 0.8])):
        lights = [[0.1 + 0.05 * i, -0.2 + 0.03 * i, 1.0 + 0.1 * i] for i in range(L)]
        plan = F.plan_cook_torrance(*maps, view_dir=[0, 0, 1], light=lights if L > 1 else lights[0], light_intensity=[inten] * L if L > 1 else inten,
                                    light_type="point", light_size=1.0)
        call = lambda: N.check(lib.pbr_cook_torrance_backward(ctypes.byref(plan.desc), gout.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(),
                                                              grads[2].data_ptr(), grads[3].data_ptr(), None, stream))
        for _ in range(60 if L < 4 else 10):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100 if L < 4 else 20):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (100 if L < 4 else 20) * 1e3
        es = maps[0].element_size()
        mb = (16 * es + 12) * S * S / 1e6
        print(f"backward {str(dtype)[6:]} {L:2d} lights, intensity {inten}: {us:8.1f} us   {mb:.0f} MB -> {mb / us * 1e3 / 8000:.3f} of 8 TB/s   {us * 1e3 / (S * S / 1e6) / L / 1e3:.3f} ns per Mpixel-light", flush=True)
