#!/usr/bin/env python3
"""A/B timing of kernel schedules inside ONE process, interleaved rounds (cdna_hip_programming.md
rule 24).  Each config is a dict of pbr_set_tuning knobs.
Usage: python tools/tune.py [--size 4096] [--rounds 7] [--iters 30]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import _native as N, functional as F  # noqa: E402

KNOBS = {"nt": 0, "blk": 1}

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--configs", type=str, default="blk=6;blk=7;blk=8;blk=6,nt=0")
args = ap.parse_args()
dev = torch.device("cuda", 0)
L = N.lib()
sets = [synth_material(args.size, dev, 1234 + i) for i in range(3)]
kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
plans = [F.plan_cook_torrance(*s, **kw) for s in sets]
stream = torch.cuda.current_stream(dev).cuda_stream
defaults = {"nt": 1, "blk": 6}
configs = [dict(kv.split("=") for kv in c.split(",")) for c in args.configs.split(";")]


def apply(cfg):
    for k, v in {**defaults, **{k: int(v) for k, v in cfg.items()}}.items():
        L.pbr_set_tuning(KNOBS[k], v)


times = [[] for _ in configs]
ref = None
for cfg in configs:                      # every schedule must produce identical bits
    apply(cfg)
    out = plans[0].launch(stream).clone()
    torch.cuda.synchronize()
    ref = out if ref is None else ref
    assert torch.equal(out, ref), cfg
for r in range(args.rounds):
    for ci, cfg in enumerate(configs):
        apply(cfg)
        for i in range(3):
            plans[i % 3].launch(stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.iters):
            plans[i % 3].launch(stream)
        e1.record()
        torch.cuda.synchronize()
        times[ci].append(e0.elapsed_time(e1) / args.iters * 1e3)
px = args.size * args.size
for cfg, t in zip(configs, times):
    med, mn = statistics.median(t), min(t)
    print(f"{str(cfg):60s} median {med:7.2f} us  min {mn:7.2f} us  -> {44 * px / med / 1e3:7.1f} GB/s")
