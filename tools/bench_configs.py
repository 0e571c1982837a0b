#!/usr/bin/env python3
"""Per-GPU shares of BASELINE.json configs 2-5 on ONE MI355X (the parity-test configurations that
are not bench.py's headline line), plus the PCIe-inclusive rate of the CPU-material path.
Prints one JSON object per line; `python tools/bench_configs.py > gpurun_out/configs.jsonl`."""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import functional as F  # noqa: E402

DEV = torch.device("cuda", 0)
PEAK = 8000.0


def maps(B, H, W, dtype=torch.float32, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    a = torch.rand(B, 3, H, W, device=DEV, generator=g)
    nxy = torch.rand(B, 2, H, W, device=DEV, generator=g) - 0.5
    n = torch.cat([nxy, torch.ones(B, 1, H, W, device=DEV)], 1)
    n = n / n.norm(dim=1, keepdim=True)
    r = torch.rand(B, 1, H, W, device=DEV, generator=g) * 0.95 + 0.05
    m = torch.rand(B, 1, H, W, device=DEV, generator=g)
    return [t.to(dtype) for t in (a, n, r, m)]


def timed(plans, iters, warm=None):
    stream = torch.cuda.current_stream(DEV).cuda_stream
    warm = max(3, iters // 2) if warm is None else warm          # the first launches after an idle period run slow
    for i in range(warm):
        plans[i % len(plans)].launch(stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        plans[i % len(plans)].launch(stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def report(name, plans, pixels, iters, extra=None):
    dt = timed(plans, iters)
    bpp = plans[0].bytes_per_pixel
    gbs = bpp * pixels / dt / 1e9
    line = {"config": name, "kernel": plans[0].kernel_name, "pixels_per_launch": pixels, "bytes_per_pixel": bpp,
            "us_per_launch": round(dt * 1e6, 1), "Mpixels_per_s": round(pixels / dt / 1e6, 1),
            "hbm_GBps_algorithmic": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 4)}
    if extra:
        line.update(extra)
    print(json.dumps(line), flush=True)


def main():
    pt = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
    # config 2: B=1 4096^2 point fp32 (bench.py's line), three rotating sets
    sets = [maps(1, 4096, 4096, seed=s) for s in range(3)]
    report("cfg2 B=1 4096^2 point fp32", [F.plan_cook_torrance(*s, **pt) for s in sets], 4096 * 4096, 200)
    del sets
    # config 3: B=64 2048^2 directional, sRGB->linear + metallic->diffuse/specular conversion fused (both F6 settings)
    s3 = maps(64, 2048, 2048, seed=3)
    for quirk in (True, False):
        p = F.plan_cook_torrance(*s3, view_dir=[0, 0, 1], light=[0.3, -0.2, 1.0], light_intensity=[1, 1, 1],
                                 light_type="directional", convert_to_diffuse_specular=True, specular_is_srgb=quirk)
        report(f"cfg3 B=64 2048^2 directional converted specular_is_srgb={quirk}", [p], 64 * 2048 * 2048, 5)
    p = F.plan_cook_torrance(*s3, view_dir=[0, 0, 1], light=[0.3, -0.2, 1.0], light_intensity=[1, 1, 1], light_type="directional")
    report("cfg3' B=64 2048^2 directional metallic (no conversion)", [p], 64 * 2048 * 2048, 5)
    del s3, p
    # config 4: B=512 1024^2 point over 8 GPUs -> 64 materials per GPU
    s4 = maps(64, 1024, 1024, seed=4)
    report("cfg4 per-GPU share B=64 1024^2 point fp32", [F.plan_cook_torrance(*s4, **pt)], 64 * 1024 * 1024, 20)
    del s4
    # config 5: B=32 4096^2, 16 point lights, fp16 maps, fp32 accumulate, 8 GPUs -> 4 materials per GPU
    s5 = maps(4, 4096, 4096, dtype=torch.float16, seed=5)
    lights = [[math.cos(t), math.sin(t), 1.0] for t in [2 * math.pi * i / 16 for i in range(16)]]
    inten = [[1.0 / 16] * 3] * 16
    kw5 = dict(view_dir=[0, 0, 1], light=lights, light_intensity=inten, light_type="point", light_size=1.0)
    for od in (torch.float32, torch.float16):
        p = F.plan_cook_torrance(*s5, out_dtype=od, **kw5)
        px = 4 * 4096 * 4096
        timed([p], 1, warm=60)          # the first ~50 launches of this 1.1 ms kernel run 7 % slower (clock transient, tools/sustain_multilight.py)
        dt = timed([p], 20, warm=0)
        report(f"cfg5 per-GPU share B=4 4096^2 16 point lights fp16 maps -> {str(od).split('.')[-1]}", [p], px, 20,
               {"light_evals_per_s_G": round(px * 16 / dt / 1e9, 1), "timing": "20 launches after 60 warm-up launches (steady state)"})
    p1 = F.plan_cook_torrance(*s5, view_dir=[0, 0, 1], light=lights[0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
    report("cfg5' same maps, ONE point light fp16 -> fp32", [p1], 4 * 4096 * 4096, 10)
    del s5, p, p1
    # backward (N3): gradients w.r.t. all four maps of one 4096^2 material, point light
    import ctypes
    from pypbr_amd import _native as N
    a, n, r, m = maps(1, 4096, 4096, seed=7)
    plan = F.plan_cook_torrance(a, n, r, m, **pt)
    gout = torch.rand(1, 3, 4096, 4096, device=DEV)
    grads = [torch.empty_like(t) for t in (a, n, r, m)]
    lib, stream = N.lib(), torch.cuda.current_stream(DEV).cuda_stream

    def bwd():
        N.check(lib.pbr_cook_torrance_backward(ctypes.byref(plan.desc), gout.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(),
                                               grads[2].data_ptr(), grads[3].data_ptr(), None, stream))
    for _ in range(3):
        bwd()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        bwd()
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / 20 * 1e-3
    px = 4096 * 4096
    print(json.dumps({"config": "backward B=1 4096^2 point fp32 (grads of albedo, normal, roughness, metallic)",
                      "bytes_per_pixel": 76, "us_per_launch": round(dt * 1e6, 1), "Mpixels_per_s": round(px / dt / 1e6, 1),
                      "hbm_GBps_algorithmic": round(76 * px / dt / 1e9, 1), "frac_of_8TBps": round(76 * px / dt / 1e9 / PEAK, 4)}), flush=True)
    # the same with fp16 maps and fp16 gradients (16 B maps + 12 B upstream gradient in, 16 B gradients out)
    h = [t.half() for t in (a, n, r, m)]
    plan16 = F.plan_cook_torrance(*h, **pt)
    grads16 = [torch.empty_like(t) for t in h]

    def bwd16():
        N.check(lib.pbr_cook_torrance_backward(ctypes.byref(plan16.desc), gout.data_ptr(), grads16[0].data_ptr(), grads16[1].data_ptr(),
                                               grads16[2].data_ptr(), grads16[3].data_ptr(), None, stream))
    for _ in range(3):
        bwd16()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        bwd16()
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / 20 * 1e-3
    print(json.dumps({"config": "backward B=1 4096^2 point, fp16 maps -> fp16 gradients", "bytes_per_pixel": 44,
                      "us_per_launch": round(dt * 1e6, 1), "Mpixels_per_s": round(px / dt / 1e6, 1),
                      "hbm_GBps_algorithmic": round(44 * px / dt / 1e9, 1), "frac_of_8TBps": round(44 * px / dt / 1e9 / PEAK, 4)}), flush=True)
    del a, n, r, m, plan, gout, grads, h, plan16, grads16
    # PCIe-inclusive: CPU-resident 4096^2 material through the reference-shaped callable
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    a, n, r, m = [t[0].cpu() for t in maps(1, 4096, 4096, seed=6)]
    mat = BasecolorMetallicMaterial(albedo=a, normal=None, roughness=r, metallic=m)
    mat._maps["normal"] = n
    brdf = CookTorranceBRDF("point")
    args = (torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0]), 1.0)
    brdf(mat, *args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = brdf(mat, *args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(json.dumps({"config": "PCIe-inclusive: CPU-resident 4096^2 material via CookTorranceBRDF (pageable host memory)",
                      "ms_per_call": round(dt * 1e3, 1), "Mpixels_per_s": round(4096 * 4096 / dt / 1e6, 1),
                      "out_device": str(out.device)}), flush=True)


if __name__ == "__main__":
    main()
