"""mall_probe.py -- does a material that fits the 256 MB memory-side cache stream faster through cached (plain) loads / stores when the SAME
material is evaluated again and again (a light sweep, a training loop), and what do plain accesses cost one that does not fit?
Forward, backward and the loss step at several sizes with PBR_TUNE_NONTEMPORAL = 1 (the rule: streaming hints) against 0.

    python tools/mall_probe.py            # on an MI355X box (gpurun)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_material
from pypbr_amd import _native as N, functional as F

DEV = torch.device("cuda:0")
lib = N.lib()
PT = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)


def timed(fn, reps=200, warm=20):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for size in (1024, 1448, 2048, 2560, 2896, 3072, 4096):
    maps = F.pack_maps(*synth_material(size, DEV, 3))
    px = size * size
    line = [f"{size}^2 ({44 * px / 1e6:.0f} MB forward)"]
    for nt in (1, 0, 1, 0):
        lib.pbr_set_tuning(N.TUNE_NONTEMPORAL, nt)
        plan = F.plan_cook_torrance(*maps, **PT)
        us = timed(plan.launch)
        line.append(f"nt={nt}: {us:7.2f} us = {44 * px / us / 1e3:6.0f} GB/s")
        del plan
    lib.pbr_set_tuning(N.TUNE_NONTEMPORAL, -1)
    print(" | ".join(line), flush=True)
    # several different materials in turn (nothing of one survives until its next turn when their sum exceeds the cache)
    if size <= 2048:
        mats = [F.pack_maps(*synth_material(size, DEV, 10 + i)) for i in range(8)]
        line = [f"{size}^2, 8 materials in turn"]
        for nt in (1, 0):
            lib.pbr_set_tuning(N.TUNE_NONTEMPORAL, nt)
            plans = [F.plan_cook_torrance(*m, **PT) for m in mats]
            def all_of():
                for p in plans:
                    p.launch()
            us = timed(all_of, reps=30, warm=3) / 8
            line.append(f"nt={nt}: {us:7.2f} us = {44 * px / us / 1e3:6.0f} GB/s")
            del plans
        lib.pbr_set_tuning(N.TUNE_NONTEMPORAL, -1)
        print(" | ".join(line), flush=True)
        del mats
    del maps
