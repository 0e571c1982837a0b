#!/usr/bin/env python3
"""tile(n) fused as wrap-around addressing vs evaluating the materialised repeat.
python tools/tile_probe.py [SRC_SIZE] [N]     default 2048 2 -> 4096^2 output"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import _native as N, functional as F  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
maps = synth_material(S, dev, 1)
kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
eager = F.plan_cook_torrance(*[t.repeat(1, n, n) for t in maps], **kw)
lazy = F.plan_cook_torrance(*maps, tile=n, **kw)
assert torch.equal(eager.launch(), lazy.launch())
stream = torch.cuda.current_stream(dev).cuda_stream
px = (S * n) ** 2


def timed(plan, iters=30):
    for _ in range(5):
        plan.launch(stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        plan.launch(stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for nt in (1, 0):          # the nt knob only reaches un-tiled launches: tiled ones never use the streaming hint
    N.lib().pbr_set_tuning(N.TUNE_NONTEMPORAL, nt)
    for lds in (-1, 0):
        N.lib().pbr_set_tuning(N.TUNE_LDS_BYTES, lds)
        te, tl = timed(eager), timed(lazy)
        print(f"{S}^2 tile({n}) nt={nt} lds={lds}: materialised {te:7.1f} us ({44 * px / te / 1e3:6.0f} GB/s alg.)   "
              f"fused {tl:7.1f} us ({px / tl / 1e3:6.1f} Gpix/s, {lazy.bytes_per_pixel} B/px -> {lazy.bytes_per_pixel * px / tl / 1e3:6.0f} GB/s)", flush=True)

# fold order (PBR_TUNE_TILE_FOLD): all vertical repeats of a band of source rows back to back; bit-identical results
N.lib().pbr_set_tuning(N.TUNE_NONTEMPORAL, 1)
N.lib().pbr_set_tuning(N.TUNE_LDS_BYTES, -1)
want = eager.launch().clone()
for dt in (torch.float32, torch.float16):
    mm = [t.to(dt) for t in maps]
    for xcd in (-1, 2):        # 2 here: the streaming hint on tiled launches too (TUNE_NONTEMPORAL = 2), built-in order
        N.lib().pbr_set_tuning(N.TUNE_NONTEMPORAL, 2 if xcd == 2 else 1)
        for fold in (0, -1, 5, 6, 7, 8):
            N.lib().pbr_set_tuning(N.TUNE_TILE_FOLD, fold)
            pl = F.plan_cook_torrance(*mm, tile=n, **kw)
            N.lib().pbr_set_tuning(N.TUNE_TILE_FOLD, 0)
            ref = F.plan_cook_torrance(*mm, tile=n, **kw).launch().clone()
            N.lib().pbr_set_tuning(N.TUNE_TILE_FOLD, fold)
            same = torch.equal(pl.launch(), ref)
            print(f"{S}^2 tile({n}) {str(dt)[6:]} nt={2 if xcd == 2 else 1} fold={fold:2d}: {timed(pl):7.1f} us  bit-identical to row order: {same}", flush=True)
N.lib().pbr_set_tuning(N.TUNE_NONTEMPORAL, 1)
N.lib().pbr_set_tuning(N.TUNE_TILE_FOLD, -1)
