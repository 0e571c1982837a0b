#!/usr/bin/env python3
"""Gradient of the resize on 3 planes: whole-call time of pbr_resize_bilinear_backward for a few shapes.  python tools/resize_bwd_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import _native as N  # noqa: E402

lib = N.lib()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
for S, ho in ((4096, 2048), (4096, 1024), (4096, 6144), (2048, 4096), (512, 4096)):
    gout = torch.rand(3, ho, ho, device=dev)
    gin = torch.empty(3, S, S, device=dev)
    ws = torch.empty(max(1, lib.pbr_resize_backward_workspace_bytes(3, S, S, ho, ho) // 4), device=dev)
    for _ in range(10):
        lib.pbr_resize_bilinear_backward(gout.data_ptr(), gin.data_ptr(), 3, S, S, ho, ho, 1, ws.data_ptr(), stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        lib.pbr_resize_bilinear_backward(gout.data_ptr(), gin.data_ptr(), 3, S, S, ho, ho, 1, ws.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    mb = 12 * (ho * ho + 2 * S * ho + S * S) / 1e6
    print(f"resize backward 3 x {ho}^2 gradient -> {S}^2: {us:8.1f} us for {mb:.0f} MB through both passes = {mb / us * 1e3:5.0f} GB/s ({mb / us * 1e3 / 8000:.3f} of 8 TB/s)")
