#!/usr/bin/env python3
"""Rows per workgroup of the resize kernel (knob PBR_TUNE_RESIZE_ROWS) over a set of scales:  python tools/resize_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import _native as N  # noqa: E402

lib = N.lib()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
S = 4096
a = torch.rand(3, S, S, device=dev)
cases = [(2048, True), (1024, True), (3000, True), (1365, True), (6144, False), (8192, False), (5000, True)]
rows = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,4,8,16,32,64").split(",")]
if len(sys.argv) > 2:                     # second argument: tile order, 1 = XCD-contiguous (default), 0 = identity
    lib.pbr_set_tuning(N.TUNE_RESIZE_XCD, int(sys.argv[2]))
if len(sys.argv) > 3:                     # third: width pass with 16-byte stores (1, default) or one column per lane (0)
    lib.pbr_set_tuning(N.TUNE_RESIZE_QUADS, int(sys.argv[3]))
for ho, aa in cases:
    out = torch.empty(3, ho, ho, device=dev)
    ws = torch.empty(max(1, lib.pbr_resize_workspace_bytes(3, S, ho) // 4), device=dev)
    line = []
    for r in rows:
        lib.pbr_set_tuning(N.TUNE_RESIZE_ROWS, r)
        for _ in range(30):
            lib.pbr_resize_bilinear(a.data_ptr(), out.data_ptr(), 3, S, S, ho, ho, int(aa), ws.data_ptr(), stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            lib.pbr_resize_bilinear(a.data_ptr(), out.data_ptr(), 3, S, S, ho, ho, int(aa), ws.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        line.append(f"{r}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f}")
    gb = 12 * (S * S + ho * ho) / 1e9
    print(f"3 x 4096^2 -> {ho}^2 aa={aa} ({gb * 1e3:.0f} MB)  us by rows  " + "   ".join(line))
lib.pbr_set_tuning(N.TUNE_RESIZE_ROWS, 0)
