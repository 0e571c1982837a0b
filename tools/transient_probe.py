#!/usr/bin/env python3
"""Launch time over the first ~800 launches after an idle GPU (chunks of 10): how long does the clock take to settle?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import functional as F  # noqa: E402

dev = torch.device("cuda", 0)
kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
plans = []
for i in range(3):
    *maps, out = F.pack_maps(*synth_material(4096, dev, i), reserve_output=True)
    plans.append(F.plan_cook_torrance(*maps, out=out.unsqueeze(0), **kw))
stream = torch.cuda.current_stream(dev).cuda_stream
torch.cuda.synchronize()
for idle in (0.0, 2.0):
    time.sleep(idle)
    n_chunks, per = 80, 10
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_chunks + 1)]
    evs[0].record()
    k = 0
    for c in range(n_chunks):
        for _ in range(per):
            plans[k % 3].launch(stream)
            k += 1
        evs[c + 1].record()
    torch.cuda.synchronize()
    t = [evs[c].elapsed_time(evs[c + 1]) / per * 1e3 for c in range(n_chunks)]
    print(f"after {idle:.0f} s idle, us per launch in chunks of {per}:")
    for r in range(0, n_chunks, 20):
        print("  launches %4d+: " % (r * per) + " ".join(f"{x:5.0f}" for x in t[r:r + 20]), flush=True)
