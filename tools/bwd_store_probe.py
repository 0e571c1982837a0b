#!/usr/bin/env python3
"""What bounds the fp16 backward kernels: timing with all / some / none of the gradient planes stored (the arithmetic and the
loads stay).   python tools/bwd_store_probe.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import _native as N, functional as F  # noqa: E402

DEV = torch.device("cuda", 0)
S = 4096
lib = N.lib()
stream = torch.cuda.current_stream(DEV).cuda_stream
maps = [t.half() for t in synth_material(S, DEV, 7)]
plan = F.plan_cook_torrance(*maps, view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
gout = torch.rand(1, 3, S, S, device=DEV)
grads = [torch.empty_like(t) for t in maps]


def timed(ptrs, reps=50, warm=150):
    def fn():
        N.check(lib.pbr_cook_torrance_backward(ctypes.byref(plan.desc), gout.data_ptr(), *ptrs, None, stream))
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


full = [g.data_ptr() for g in grads]
for rounds in (int(x) for x in (sys.argv[1:] or ["-1", "0"])):
    lib.pbr_set_tuning(N.TUNE_BWD_RUN, rounds)
    for name, ptrs in (("all 8 planes", full), ("albedo only (3)", [full[0], None, None, None]), ("roughness only (1)", [None, None, full[2], None]),
                       ("none", [None, None, None, None]), ("all 8 planes", full)):
        print(f"rounds {rounds:2d}  stores: {name:20s} {timed(ptrs):7.1f} us", flush=True)
