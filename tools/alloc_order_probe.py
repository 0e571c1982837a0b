#!/usr/bin/env python3
"""Does the streaming rate of a material depend on WHEN (where) in the process its buffers were allocated?
Allocates N materials one after the other (4 plain torch.empty each + output) and times each on its own."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import functional as F  # noqa: E402

S, N = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda", 0)
P = S * S
kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
stream = torch.cuda.current_stream(dev).cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
plans = []
for i in range(N):
    a = torch.empty(3, S, S, device=dev).uniform_(0, 1, generator=g)
    n = torch.empty(3, S, S, device=dev).uniform_(-0.5, 0.5, generator=g); n[2] = 1.0
    r = torch.empty(1, S, S, device=dev).uniform_(0.05, 1, generator=g)
    m = torch.empty(1, S, S, device=dev).uniform_(0, 1, generator=g)
    plans.append(F.plan_cook_torrance(a, n, r, m, **kw))
times = [[] for _ in plans]
for rnd in range(5):
    for i, p in enumerate(plans):
        for _ in range(3):
            p.launch(stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            p.launch(stream)
        e1.record()
        torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) / 20 * 1e3)
base = min(t.data_ptr() for p in plans for t in p._keep if t is not None)
for i, (p, t) in enumerate(zip(plans, times)):
    ptrs = [x.data_ptr() for x in p._keep if x is not None] + [p.out.data_ptr()]
    print(f"material {i:2d}: {statistics.median(t):7.2f} us  {44 * P / statistics.median(t) / 1e3:7.1f} GB/s  addresses (MiB from the lowest) "
          f"{[round((q - base) / 2**20) for q in ptrs]}", flush=True)
