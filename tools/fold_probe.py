#!/usr/bin/env python3
"""pbr_fold_gradient (the sum over tile repeats / over a batch that shares a map) by launch shape.  python tools/fold_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import _native as N  # noqa: E402

lib = N.lib()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream


def timed(fn, iters=60):
    for _ in range(30):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, (B, C, h, w, ny, nx, fb) in (("tile(2) of 3 x 2048^2 (4 repeats)", (1, 3, 2048, 2048, 2, 2, 0)), ("a map shared by 8 materials, 3 x 2048^2", (8, 3, 2048, 2048, 1, 1, 1)),
                                        ("tile(3) of 3 x 1024^2 (9 repeats)", (1, 3, 1024, 1024, 3, 3, 0))):
    src = torch.rand(B, C, ny * h, nx * w, device=dev)
    dst = torch.empty(1 if fb else B, C, h, w, device=dev)
    nbytes = 4 * (src.numel() + dst.numel())
    call = lambda: N.check(lib.pbr_fold_gradient(src.data_ptr(), dst.data_ptr(), B, C, h, w, ny, nx, fb, stream))
    for shape in (0, 1, 2):
        for lds in (0, 20480):
            if shape != 2 and lds:
                continue
            lib.pbr_set_tuning(N.TUNE_STREAM_SHAPE, shape)
            lib.pbr_set_tuning(N.TUNE_STREAM_LDS, lds)
            us = timed(call)
            print(f"{name} shape={shape} lds={lds}: {us:7.1f} us  {nbytes / us / 1e3:6.0f} GB/s ({nbytes / us / 1e3 / 8000:.3f})", flush=True)
lib.pbr_set_tuning(N.TUNE_STREAM_SHAPE, -1)
lib.pbr_set_tuning(N.TUNE_STREAM_LDS, -1)
