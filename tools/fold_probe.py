#!/usr/bin/env python3
"""pbr_fold_gradient timing: the sums autograd would perform for a tiled (repeat) or batch-shared map.
python tools/fold_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import _native as N  # noqa: E402

lib = N.lib()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
for (B, C, h, w, ny, nx, fold_b) in [(1, 3, 2048, 2048, 2, 2, 0), (1, 1, 2048, 2048, 2, 2, 0), (8, 3, 2048, 2048, 1, 1, 1), (4, 3, 1024, 1024, 4, 4, 0), (2, 3, 1000, 1001, 2, 2, 0)]:
    src = torch.rand(B, C, ny * h, nx * w, device=dev)
    dst = torch.empty(1 if fold_b else B, C, h, w, device=dev)
    ref = src.view(B, C, ny, h, nx, w).sum(dim=(2, 4))
    if fold_b:
        ref = ref.sum(dim=0, keepdim=True)

    def run():
        N.check(lib.pbr_fold_gradient(src.data_ptr(), dst.data_ptr(), B, C, h, w, ny, nx, fold_b, stream))
    run()
    err = (dst - ref).abs().max().item()
    for _ in range(10):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    nbytes = 4 * (src.numel() + dst.numel())
    print(f"B={B} C={C} {h}x{w} tiles {ny}x{nx} fold_batch={fold_b}: {us:7.1f} us  {nbytes / us / 1e3:7.1f} GB/s  max err vs torch sum {err:.1e}")
