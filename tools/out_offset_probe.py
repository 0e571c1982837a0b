#!/usr/bin/env python3
"""Where should the OUTPUT sit relative to the input planes?  All 8 input planes at multiples of the plane size in one
buffer; the contiguous 3-plane output at 8 planes + `off` bytes.  python tools/out_offset_probe.py [SIZE]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import functional as F  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
P = S * S
kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
stream = torch.cuda.current_stream(dev).cuda_stream
src = [synth_material(S, dev, 40 + i) for i in range(3)]
offs = [0, 256, 4096, 65536, 1 << 20, 2 << 20, 6 << 20, 16 << 20, 32 << 20]
in_shift = [0, 12 << 20]          # second variant: normal/rough/metal moved off the albedo's alignment by 12 MiB


def build(off, shift, maps):
    buf = torch.empty(12 * P + (64 << 20) // 4 + 64, device=dev)
    views, k = [], 0
    for j, t in enumerate(maps):
        c = t.shape[0]
        o = k * P + (shift // 4 if j > 0 else 0)
        v = buf[o:o + c * P].view(c, S, S)
        v.copy_(t)
        views.append(v)
        k += c
    o = 8 * P + (shift // 4) + off // 4
    out = buf[o:o + 3 * P].view(1, 3, S, S)
    return F.plan_cook_torrance(*views, out=out, **kw)


plans = {(off, sh): [build(off, sh, m) for m in src] for sh in in_shift for off in offs}
times = {k: [] for k in plans}
for rnd in range(5):
    for key, pl in plans.items():
        for i in range(3):
            pl[i].launch(stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30):
            pl[i % 3].launch(stream)
        e1.record()
        torch.cuda.synchronize()
        times[key].append(e0.elapsed_time(e1) / 30 * 1e3)
for (off, sh), t in times.items():
    med = statistics.median(t)
    print(f"inputs shift {sh >> 20:3d} MiB, out at planes + {off:9d} B: median {med:7.2f} us -> {44 * P / med / 1e3:7.1f} GB/s", flush=True)
