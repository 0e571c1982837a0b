#!/usr/bin/env python3
"""A/B timing of kernel schedules inside ONE process, interleaved rounds (cdna_hip_programming.md
rule 24; box-to-box variation of the same binary is +-4 %, so nothing else is comparable).  Each
config is a set of pbr_set_tuning knobs; `alt=1` runs the config on a second build of the library
(--altlib), which makes two source revisions comparable in one process.
Usage: python tools/tune.py [--size 4096] [--rounds 7] [--iters 30] [--configs "blk=6;blk=8,nt=0"]"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import _native as N, functional as F  # noqa: E402

KNOBS = {"nt": 0, "blk": 1, "f16vec": 2, "lds": 3, "bwdvec": 4, "nb": 5, "sb": 6, "maxvec": 7}      # PBR_TUNE_* of ABI 7 (the workgroup order is the descriptor's `schedule`)
DEFAULTS = {"nt": 1, "blk": 6, "lds": -1, "nb": -1, "sb": 1, "maxvec": 8}

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--height", type=int, default=0, help="rows (default: --size); maps are the top rows of a size x size material")
ap.add_argument("--sets", type=int, default=0, help="rotating map sets (default 3 for --batch 1, else 1)")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--configs", type=str, default="blk=6;blk=7;blk=8;blk=6,nt=0")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--light", type=str, default="point")
ap.add_argument("--lights", type=int, default=1)
ap.add_argument("--dtype", type=str, default="float32")
ap.add_argument("--altlib", type=str, default="")
ap.add_argument("--arena", action="store_true", help="maps of a set + its result in one allocation (F.pack_maps)")
ap.add_argument("--nocheck", action="store_true", help="timing experiments whose alt build writes different values")
ap.add_argument("--linear", action="store_true", help="maps already linear, linear output: no sRGB transcendental work")
args = ap.parse_args()
dev = torch.device("cuda", 0)
libs = [N.lib()]
for path in filter(None, args.altlib.split(",")):       # alt=1 -> first path, alt=2 -> second, ...
    alt = ctypes.CDLL(os.path.abspath(path))
    alt.pbr_cook_torrance.argtypes = [ctypes.POINTER(N.RenderDesc), ctypes.c_void_p]
    alt.pbr_cook_torrance.restype = ctypes.c_int
    alt.pbr_set_tuning.argtypes = [ctypes.c_int, ctypes.c_int]
    alt.pbr_render_desc_size.restype = ctypes.c_size_t
    assert alt.pbr_render_desc_size() == ctypes.sizeof(N.RenderDesc), "descriptor layouts differ"
    libs.append(alt)
nsets = args.sets or (3 if args.batch == 1 else 1)
H = args.height or args.size
sets = [[torch.stack([t[:, :H].contiguous()] * args.batch).to(getattr(torch, args.dtype)) for t in synth_material(args.size, dev, 1234 + i)]
        for i in range(nsets)]
if args.lights > 1:
    import math
    light = [[math.cos(2 * math.pi * i / args.lights), math.sin(2 * math.pi * i / args.lights), 1.0] for i in range(args.lights)]
    inten = [[1.0 / args.lights] * 3] * args.lights
else:
    light, inten = ([0.1, 0.1, 1.0] if args.light == "point" else [0.3, -0.2, 1.0]), [1, 1, 1]
kw = dict(view_dir=[0, 0, 1], light=light, light_intensity=inten, light_type=args.light, light_size=1.0)
if args.linear:
    kw.update(albedo_is_srgb=False, return_srgb=False)
if args.arena:
    plans = []
    for s in sets:
        *packed, out = F.pack_maps(*s, reserve_output=True)
        plans.append(F.plan_cook_torrance(*packed, out=out, **kw))
else:
    plans = [F.plan_cook_torrance(*s, **kw) for s in sets]
stream = torch.cuda.current_stream(dev).cuda_stream
configs = [dict(kv.split("=") for kv in c.split(",")) for c in args.configs.split(";")]


def launcher(cfg):
    lib = libs[int(cfg.get("alt", 0))]
    for k, v in {**DEFAULTS, **{k: int(v) for k, v in cfg.items() if k != "alt"}}.items():
        lib.pbr_set_tuning(KNOBS[k], v)

    def launch(i):
        p = plans[i % nsets]
        rc = lib.pbr_cook_torrance(ctypes.byref(p.desc), stream)
        assert rc == 0, rc
        return p.result
    return launch


times = [[] for _ in configs]
ref = None
for cfg in configs:                      # every schedule / build must agree (bit-identical within a build)
    out = launcher(cfg)(0).clone()
    torch.cuda.synchronize()
    ref = out if ref is None else ref
    if args.nocheck:
        continue
    if int(cfg.get("alt", 0)):
        assert (out.float() - ref.float()).abs().max().item() <= 1e-5, cfg
    else:
        assert torch.equal(out, ref), cfg
for r in range(args.rounds):
    for ci, cfg in enumerate(configs):
        launch = launcher(cfg)
        for i in range(3):
            launch(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.iters):
            launch(i)
        e1.record()
        torch.cuda.synchronize()
        times[ci].append(e0.elapsed_time(e1) / args.iters * 1e3)
px = args.batch * H * args.size
bpp = plans[0].bytes_per_pixel
for cfg, t in zip(configs, times):
    med, mn = statistics.median(t), min(t)
    print(f"{str(cfg):60s} median {med:7.2f} us  min {mn:7.2f} us  -> {bpp * px / med / 1e3:7.1f} GB/s")
