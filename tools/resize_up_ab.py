#!/usr/bin/env python3
"""Up-scales of 3 x 4096^2 (and 2048^2): the two-tap register kernel (PBR_TUNE_RESIZE_UP2 = 1) against the strip kernel (0),
alternating in one process.   python tools/resize_up_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import _native as N  # noqa: E402

lib = N.lib()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
for S, ho in ((4096, 6144), (4096, 8192), (4096, 4608), (2048, 4096), (4096, 4096)):
    a = torch.rand(3, S, S, device=dev)
    out = torch.empty(3, ho, ho, device=dev)
    ws = torch.empty(max(1, lib.pbr_resize_workspace_bytes(3, S, ho) // 4), device=dev)
    res = {0: [], 1: [], 2: [], 8: []}
    for rnd in range(3):
        for knob in (0, 1, 2, 8):
            lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 1 if knob else 0)
            lib.pbr_set_tuning(N.TUNE_RESIZE_ROWS, knob if knob > 1 else 0)
            for _ in range(30):
                lib.pbr_resize_bilinear(a.data_ptr(), out.data_ptr(), 3, S, S, ho, ho, 1, ws.data_ptr(), stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                lib.pbr_resize_bilinear(a.data_ptr(), out.data_ptr(), 3, S, S, ho, ho, 1, ws.data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            res[knob].append(e0.elapsed_time(e1) / 50 * 1e3)
    mb = 12 * (S * S + ho * ho) / 1e6
    print(f"3 x {S}^2 -> {ho}^2 ({mb:.0f} MB): strip {min(res[0]):7.1f} us = {mb / min(res[0]) * 1e3:5.0f} GB/s   two-tap {min(res[1]):7.1f} us = "
          f"{mb / min(res[1]) * 1e3:5.0f} GB/s ({mb / min(res[1]) * 1e3 / 8000:.3f} of 8 TB/s)   rows 2: {min(res[2]):.1f}  rows 8: {min(res[8]):.1f}")
lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 1)
lib.pbr_set_tuning(N.TUNE_RESIZE_ROWS, 0)
