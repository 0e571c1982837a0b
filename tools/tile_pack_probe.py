#!/usr/bin/env python3
"""Fused tile(n): the repeat-inner kernel (PBR_TUNE_TILE_REPEAT: texels loaded and decoded once, evaluated at every repeat) against
the wrap-around form with scalar / packed arithmetic (PBR_TUNE_PACK_SINGLE) in row / fold order (PBR_TUNE_TILE_FOLD); interleaved rounds
in one process; every variant checked bit-identical to the materialised repeat.
python tools/tile_pack_probe.py [SRC_SIZE] [N] [ROUNDS]     default 2048 2 5 -> 4096^2 image"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import _native as N, functional as F  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
lib = N.lib()
stream = torch.cuda.current_stream(dev).cuda_stream
KNOBS = (N.TUNE_TILE_REPEAT, N.TUNE_PACK_SINGLE, N.TUNE_TILE_FOLD, N.TUNE_NONTEMPORAL)


def setk(v):
    for k, x in zip(KNOBS, v):
        lib.pbr_set_tuning(k, x)


for dt in (torch.float32, torch.float16):
    maps = [t.to(dt) for t in synth_material(S, dev, 1)]
    for light_type, light in (("point", [0.1, 0.1, 1.0]), ("directional", [0.3, -0.2, 1.0])):
        kw = dict(view_dir=[0, 0, 1], light=light, light_intensity=[1, 1, 1], light_type=light_type, light_size=1.0)
        setk((-1, -1, -1, 1))
        eager = F.plan_cook_torrance(*[t.repeat(1, n, n) for t in maps], **kw)
        want = eager.launch().clone()
        #            repeat pack fold nt
        variants = [(1, 0, 0, 1), (1, 0, 0, 0), (1, 0, 0, 2), (1, 0, 0, 3), (1, 0, 0, 4), (0, 0, 0, 1), (0, 1, 0, 1), (0, 1, 5, 1), (0, -1, -1, 1)]
        plans, same, times = {}, {}, {v: [] for v in variants + ["materialised"]}
        for v in variants:
            setk(v)
            plans[v] = F.plan_cook_torrance(*maps, tile=n, **kw)
            same[v] = torch.equal(plans[v].launch(stream), want)
        plans["materialised"], same["materialised"] = eager, True
        for r in range(rounds):
            for v in variants + ["materialised"]:
                setk(v if v != "materialised" else (-1, -1, -1, 1))
                for _ in range(10):
                    plans[v].launch(stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100):
                    plans[v].launch(stream)
                e1.record()
                e1.synchronize()
                times[v].append(e0.elapsed_time(e1) * 10)
        esz = 4 if dt == torch.float32 else 2
        alg = 8 * S * S * esz + 12 * (S * n) ** 2
        for v in variants + ["materialised"]:
            setk(v if v != "materialised" else (-1, -1, -1, 1))
            med = statistics.median(times[v])
            print("%-11s %s %d^2 tile(%d) %-28s %-36s median %7.1f us  min %7.1f  %5.2f TB/s of the tiled launch's algorithmic bytes  bit-identical: %s"
                  % (light_type, str(dt)[6:], S, n, "repeat=%d pack=%d fold=%d nt=%d" % v if v != "materialised" else v, plans[v].kernel_name, med, min(times[v]),
                     alg / med / 1e6, same[v]), flush=True)
setk((-1, -1, -1, 1))
