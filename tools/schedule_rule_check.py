#!/usr/bin/env python3
"""How good is PBR_SCHEDULE_AUTO (ct_launch.hpp: schedule_xcd_log2, a rule fitted to a dozen shapes)?  For a wider set of shapes: the launch time
under the linear workgroup order, under runs of 64 tiles per XCD, and under the rule; where the rule is more than 2 % off the better of the two.
python tools/schedule_rule_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import _native as N, functional as F  # noqa: E402

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
KW = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)


def timed(plan, iters):
    for _ in range(max(10, iters // 3)):
        plan.launch(stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        plan.launch(stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


shapes = [(1, s, s) for s in (512, 768, 1000, 1024, 1536, 2000, 2048, 2560, 3000, 3072, 4000, 4096, 5000, 6000, 6144, 8192)] + \
         [(1, 4096, 1024), (1, 1024, 4096), (1, 2048, 8192), (1, 8192, 2048), (1, 1080, 1920), (1, 2160, 3840), (4, 2048, 2048), (16, 1024, 1024), (8, 3000, 3000),
          (4, 4096, 4096), (64, 512, 512), (3, 1536, 2048)]
worst, off = 0.0, []
for dtype in (torch.float32, torch.float16):
    for B, H, W in shapes:
        g = torch.Generator(device=dev).manual_seed(1)
        a = torch.rand(B, 3, H, W, device=dev, generator=g).to(dtype)
        n = torch.rand(B, 3, H, W, device=dev, generator=g).to(dtype)
        r = (torch.rand(B, 1, H, W, device=dev, generator=g) * 0.8 + 0.2).to(dtype)
        m = torch.rand(B, 1, H, W, device=dev, generator=g).to(dtype)
        px = B * H * W
        iters = max(100, min(600, int(2e10 / (px * 44))))          # short runs produce outliers (one 37 % "miss" in a first sweep was a clock dip)
        t = {}
        for name, sched in (("linear", N.SCHEDULE_LINEAR), ("runs", N.schedule_xcd(6)), ("rule", N.SCHEDULE_AUTO)):
            t[name] = timed(F.plan_cook_torrance(a, n, r, m, schedule=sched, **KW), iters)
        best = min(t["linear"], t["runs"])
        loss = t["rule"] / best - 1.0
        worst = max(worst, loss)
        flag = "  <-- rule off by %.1f %%" % (100 * loss) if loss > 0.02 else ""
        if loss > 0.02:
            off.append((str(dtype)[6:], B, H, W, round(100 * loss, 1)))
        print(f"{str(dtype)[6:]:8s} {B:3d} x {H:5d} x {W:5d}: linear {t['linear']:8.1f}  runs {t['runs']:8.1f}  rule {t['rule']:8.1f} us{flag}", flush=True)
        del a, n, r, m
print(f"{2 * len(shapes)} shapes: the rule is at most {100 * worst:.1f} % off the better order; more than 2 % off on {len(off)}: {off}")
