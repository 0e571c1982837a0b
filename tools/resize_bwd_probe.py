#!/usr/bin/env python3
"""Gradient of the resize on 3 planes: whole-call time of pbr_resize_bilinear_backward, one pass (strip kernel with transposed tables)
against two passes through the workspace, and how far apart their results are.  python tools/resize_bwd_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import _native as N  # noqa: E402

lib = N.lib()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream


def timed(fn, iters=30):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for S, ho, aa in ((4096, 2048, 1), (4096, 1024, 1), (4096, 3000, 1), (4096, 6144, 1), (2048, 4096, 1), (1000, 2500, 0), (4090, 1365, 1)):
    gout = torch.rand(3, ho, ho, device=dev)
    gin = torch.empty(3, S, S, device=dev)
    ws = torch.empty(max(1, lib.pbr_resize_backward_workspace_bytes(3, S, S, ho, ho) // 4), device=dev)
    call = lambda: N.check(lib.pbr_resize_bilinear_backward(gout.data_ptr(), gin.data_ptr(), 3, S, S, ho, ho, aa, ws.data_ptr(), stream))
    res = {}
    for fused, quads in ((0, 1), (2, 0), (2, 1), (1, 1)):
        lib.pbr_set_tuning(N.TUNE_RESIZE_BWD_FUSED, fused)
        lib.pbr_set_tuning(N.TUNE_RESIZE_QUADS, quads)
        rows = (0, 4) if fused == 1 else ((0,) if fused != 2 else (0, 64, 128))
        for r in rows:
            lib.pbr_set_tuning(N.TUNE_RESIZE_ROWS, r)
            us = timed(call)
            res[(fused, r)] = gin.clone()
            mb = 12 * (ho * ho + S * S) / 1e6
            print(f"resize backward 3 x {ho}^2 gradient -> {S}^2 aa={aa} {('registers (gather / two-tap transpose)' if fused == 1 else ('strip, 16-byte stores' if quads else 'strip, 4-byte stores')) if fused else 'two passes'} rows={r:3d}: {us:8.1f} us, {mb:.0f} MB in + out = "
                  f"{mb / us * 1e3:5.0f} GB/s ({mb / us * 1e3 / 8000:.3f} of 8 TB/s)   max |one - two| {float((res[(fused, r)] - res[(0, 0)]).abs().max()):.2e}", flush=True)
    lib.pbr_set_tuning(N.TUNE_RESIZE_ROWS, 0)
    lib.pbr_set_tuning(N.TUNE_RESIZE_BWD_FUSED, 1)
