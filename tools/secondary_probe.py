#!/usr/bin/env python3
"""Timing of the autograd-plumbing kernels beside the render backward: pbr_decode_normal_backward, pbr_blend_maps_backward.
python tools/secondary_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import _native as N  # noqa: E402

lib = N.lib()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
S = 4096
P = S * S


def timed(fn, reps=50):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


a, b, g = (torch.rand(3, S, S, device=dev) for _ in range(3))
mask = torch.rand(1, S, S, device=dev)
g1, g2, gm = torch.empty_like(a), torch.empty_like(a), torch.empty_like(mask)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
for normal in (0, 1):
    us = timed(lambda: N.check(lib.pbr_blend_maps_backward(a.data_ptr(), b.data_ptr(), mask.data_ptr(), g.data_ptr(), g1.data_ptr(), g2.data_ptr(),
                                                           gm.data_ptr(), 3, P, normal, 0, stream)))
    nbytes = 4 * P * (3 + 3 + 1 + 3 + 3 + 3 + 1)
    print(f"blend_maps_backward normal={normal} 3 ch 4096^2 (10 planes in, 7 out): {us:7.1f} us  {nbytes / us / 1e3:7.1f} GB/s")
us = timed(lambda: N.check(lib.pbr_decode_normal_backward(a.data_ptr(), g.data_ptr(), g1.data_ptr(), 3, P, flag.data_ptr(), stream)))
print(f"decode_normal_backward 3 ch 4096^2 (6 planes in, 3 out): {us:7.1f} us  {4 * P * 9 / us / 1e3:7.1f} GB/s")
