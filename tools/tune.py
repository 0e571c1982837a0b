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

KNOBS = {"nt": 0, "blk": 1, "f16vec": 2, "lds": 3}

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--configs", type=str, default="blk=6;blk=7;blk=8;blk=6,nt=0")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--light", type=str, default="point")
ap.add_argument("--dtype", type=str, default="float32")
args = ap.parse_args()
dev = torch.device("cuda", 0)
L = N.lib()
nsets = 3 if args.batch == 1 else 1
sets = [[torch.stack([t] * args.batch).to(getattr(torch, args.dtype)) for t in synth_material(args.size, dev, 1234 + i)]
        for i in range(nsets)]
kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0] if args.light == "point" else [0.3, -0.2, 1.0],
          light_intensity=[1, 1, 1], light_type=args.light, light_size=1.0)
plans = [F.plan_cook_torrance(*s, **kw) for s in sets]
stream = torch.cuda.current_stream(dev).cuda_stream
defaults = {"nt": 1, "blk": 6, "lds": -1}
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
            plans[i % nsets].launch(stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.iters):
            plans[i % nsets].launch(stream)
        e1.record()
        torch.cuda.synchronize()
        times[ci].append(e0.elapsed_time(e1) / args.iters * 1e3)
px = args.batch * args.size * args.size
bpp = plans[0].bytes_per_pixel
for cfg, t in zip(configs, times):
    med, mn = statistics.median(t), min(t)
    print(f"{str(cfg):60s} median {med:7.2f} us  min {mn:7.2f} us  -> {bpp * px / med / 1e3:7.1f} GB/s")
