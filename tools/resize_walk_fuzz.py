#!/usr/bin/env python3
"""Random shapes through the row-walking down-scale (csrc/resize_stream.hpp) against the strip kernel, bit for bit: factors 1.01 ... 17 that differ per axis,
1 ... 5 planes, widths that are whole 16-byte pieces (the walk's condition), bands and strips of every raggedness; knob PBR_TUNE_RESIZE_UP2 = 2 takes the walk
wherever the shape allows, 0 the strip kernel; pbr_resize_form says which family served (a case both knobs hand to the same family is counted, not compared).
    python tools/resize_walk_fuzz.py [cases] [seed]          # on an MI355X box; run it under `timeout`: a walk whose two waves disagreed on a barrier would hang"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import _native as N  # noqa: E402


def run(cases=300, seed=0, verbose=True):
    lib = N.lib()
    stream = torch.cuda.current_stream().cuda_stream
    rng = random.Random(seed)
    walked = same_family = 0

    def call(x, ho, wo, knob):
        planes, hi, wi = x.shape
        out = torch.full((planes, ho, wo), float("nan"), device="cuda")
        ws = torch.empty(max(1, lib.pbr_resize_workspace_bytes(planes, hi, wo) // 4), device="cuda")
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, knob)
        form = lib.pbr_resize_form(x.data_ptr(), out.data_ptr(), planes, hi, wi, ho, wo, 1, ws.data_ptr())
        N.check(lib.pbr_resize_bilinear(x.data_ptr(), out.data_ptr(), planes, hi, wi, ho, wo, 1, ws.data_ptr(), stream))
        torch.cuda.synchronize()
        return out, form

    try:
        for i in range(cases):
            ho, wo = rng.choice([4, 5, 7, 16, 17, 31, 33, 64, 65, 100, 127, 200, 257, 300, 400]), rng.choice([16, 17, 23, 31, 64, 65, 81, 100, 129, 200, 255, 341, 400, 512])
            sy, sx = rng.choice([1.02, 1.3, 1.5, 2.05, 3.0007, 4.1, 6.9, 7.1, 10.24, 13.3, 16.9]), rng.choice([1.02, 1.37, 2.0, 2.9, 3.3, 5.5, 7.01, 9.9, 12.5, 16.4])
            hi, wi = max(ho + 1, int(ho * sy)), max(wo + 4, int(wo * sx)) // 4 * 4
            planes = rng.choice([1, 2, 3, 5])
            if planes * hi * wi > 64 << 20:
                continue
            g = torch.Generator().manual_seed(seed * 100003 + i)
            x = (torch.rand(planes, hi, wi, generator=g) * 2 - 0.5).cuda()
            a, fa = call(x, ho, wo, 2)
            b, fb = call(x, ho, wo, 0)
            if fa == fb:
                same_family += 1
                continue
            walked += fa == N.RESIZE_ROW_WALK
            if not torch.equal(a, b):
                d = (a - b).abs()
                raise AssertionError("case %d: planes %d %dx%d -> %dx%d: families %d / %d differ, max %.3g, %d values" % (i, planes, hi, wi, ho, wo, fa, fb, float(d.max()), int((d > 0).sum())))
            if verbose and i % 50 == 0:
                print("case %d: planes %d %dx%d -> %dx%d family %d: equal" % (i, planes, hi, wi, ho, wo, fa), flush=True)
    finally:
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, -1)
    print("%d cases through the row walk bit-identical to the strip kernel; %d served by one family under both knobs" % (walked, same_family), flush=True)
    return walked


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
