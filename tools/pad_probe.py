#!/usr/bin/env python3
"""Does a padded plane pitch (planes not a power-of-two apart) make the flagship launch stream faster?
All 8 input planes are carved out of ONE buffer with pitch H*W + pad elements; the output stays contiguous.
python tools/pad_probe.py [SIZE]"""
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


def carve(pad_bytes, maps):
    pad = pad_bytes // 4
    pitch = P + pad
    buf = torch.empty(8 * pitch + 64, device=dev)
    views, k = [], 0
    for t in maps:
        c = t.shape[0]
        v = buf[k * pitch:(k + c) * pitch].as_strided((c, S, S), (pitch, S, 1))
        v.copy_(t)
        views.append(v)
        k += c
    return views


pads = [0, 256, 1024, 4352, 8448, 16640, 65792, 1048832]
plans = {}
for pad in pads:
    plans[pad] = [F.plan_cook_torrance(*carve(pad, m), **kw) for m in src]
plans["separate tensors"] = [F.plan_cook_torrance(*m, **kw) for m in src]
ref = plans["separate tensors"][0].launch().clone()
for name, pl in (("separate tensors", plans["separate tensors"]), ("carved +0", plans[0])):
    for i, p_ in enumerate(pl):
        ptrs = [t.data_ptr() for t in p_._keep if t is not None] + [p_.out.data_ptr()]
        base = min(ptrs)
        print(f"{name} set {i}: offsets from the lowest map, MiB:", [round((q - base) / 2**20, 3) for q in ptrs], flush=True)
times = {k: [] for k in plans}
for rnd in range(7):
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
assert torch.equal(plans[4352][0].launch(), ref)
for key, t in times.items():
    med = statistics.median(t)
    print(f"plane pitch +{key!s:>18} B: median {med:7.2f} us  min {min(t):7.2f}  -> {44 * P / med / 1e3:7.1f} GB/s   ({plans[key][0].kernel_name}, schedule rule)")
