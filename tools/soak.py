#!/usr/bin/env python3
"""Soak run: random shapes / storage types / light counts / tiles / bands / workgroup orders for N seconds.  Every
configuration is evaluated twice with different workgroup orders and must give bit-identical, finite results; one in
three also on the one-pixel kernels and with the other plane-addressing mode (bit-identical); one in eight also runs
the backward kernel and the fused blend.  python tools/soak.py [seconds]"""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import _native as N, functional as F  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(12345)
g = torch.Generator(device="cuda").manual_seed(1)
dev = torch.device("cuda", 0)
t_end = time.time() + seconds
n_cfg = n_bwd = n_blend = n_stream = 0
t_note = time.time() + 60
while time.time() < t_end:
    if time.time() > t_note:                      # a line a minute: a silent GPU job is taken to be hung
        print(f"  ... {n_cfg} configurations so far", flush=True)
        t_note += 60
    B = rng.choice([1, 1, 2, 3, 4, 5, 8])
    h = rng.choice([1, 2, 7, 16, 33, 64, 100, 257, 512])
    w = rng.choice([1, 3, 4, 5, 8, 12, 13, 20, 37, 64, 100, 256, 1000, 1001, 1024])
    dtype = rng.choice([torch.float32, torch.float32, torch.float16])
    lights = rng.choice([1, 1, 1, 2, 5, 16])
    ny, nx = rng.choice([(1, 1), (1, 1), (2, 2), (1, 3), (3, 1)])
    wf = rng.choice(["metallic", "specular", "converted"])
    ltype = rng.choice(["point", "directional"])
    a = torch.rand(B, 3, h, w, device=dev, generator=g).to(dtype)
    n = torch.cat([torch.rand(B, 2, h, w, device=dev, generator=g) - 0.5, torch.ones(B, 1, h, w, device=dev)], 1).to(dtype)
    r = torch.rand(rng.choice([1, B]), 1, h, w, device=dev, generator=g).to(dtype)
    m = torch.rand(B, 1, h, w, device=dev, generator=g).to(dtype) if wf != "specular" else None
    s = torch.rand(B, 3, h, w, device=dev, generator=g).to(dtype) if wf == "specular" else None
    lv = [[rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.05, 1.5)] for _ in range(lights)]
    kw = dict(view_dir=[rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), 1.0], light=lv, light_intensity=[[rng.uniform(0, 1)] * 3] * lights,
              light_type=ltype, light_size=rng.choice([None, 1.0, 3.0]), convert_to_diffuse_specular=(wf == "converted"),
              return_srgb=rng.random() < 0.7, albedo_is_srgb=rng.random() < 0.7)
    if (ny, nx) != (1, 1):
        H = ny * h
        y0 = rng.randrange(0, H)
        kw.update(tile=(ny, nx), y_offset=y0, rows=rng.randrange(1, H - y0 + 1))
    out1 = F.cook_torrance(a, n, r, m, s, schedule=N.SCHEDULE_LINEAR, **kw)
    out2 = F.cook_torrance(a, n, r, m, s, schedule=N.schedule_xcd(rng.randrange(1, 9)), **kw)
    assert bool(torch.isfinite(out1).all()) and torch.equal(out1, out2), (B, h, w, dtype, lights, ny, nx, wf, ltype)
    # a material of the batch evaluated on its own (one-material kernels) == its slice of the batched launch (several lights:
    # the batch-inner kernel, 2 or 4 materials per lane)
    b = rng.randrange(B)
    one = F.cook_torrance(a[b], n[b], r[b if r.shape[0] == B else 0], None if m is None else m[b], None if s is None else s[b], **kw)
    assert torch.equal(one, out1[b]), ("batch slice", B, b, h, w, dtype, lights, ny, nx, wf, ltype)
    if dtype == torch.float16 and lights == 1:       # 8-pixel lanes (LDS piece exchange) == 4-pixel lanes
        lib = N.lib()
        lib.pbr_set_tuning(N.TUNE_F16_VEC, 4)
        out4 = F.cook_torrance(a, n, r, m, s, schedule=N.SCHEDULE_LINEAR, **kw)
        lib.pbr_set_tuning(N.TUNE_F16_VEC, 8)
        assert torch.equal(out4, out1), ("f16 vec", B, h, w, lights, ny, nx, wf, ltype)
    if n_cfg % 3 == 0:       # the one-pixel kernels == the vector kernels (ragged rows overlap their last two lanes); scalar plane bases
        lib = N.lib()
        lib.pbr_set_tuning(N.TUNE_MAX_VEC, 1)
        out_1px = F.cook_torrance(a, n, r, m, s, schedule=N.SCHEDULE_LINEAR, **kw)
        lib.pbr_set_tuning(N.TUNE_MAX_VEC, 8)
        lib.pbr_set_tuning(N.TUNE_SCALAR_BASE, rng.choice([0, 2]))
        out_sb = F.cook_torrance(a, n, r, m, s, schedule=N.SCHEDULE_LINEAR, **kw)
        lib.pbr_set_tuning(N.TUNE_SCALAR_BASE, 1)
        assert torch.equal(out_1px, out1) and torch.equal(out_sb, out1), ("max_vec / scalar base", B, h, w, dtype, lights, ny, nx, wf, ltype)
    n_cfg += 1
    if n_cfg % 8 == 0 and (ny, nx) == (1, 1) and r.shape[0] == B:
        leaves = [None if t is None else t.clone().requires_grad_(True) for t in (a, n, r, m, s)]
        lt = torch.tensor(lv, device=dev, requires_grad=True)            # light / view gradients too (PGRAD kernels)
        F.cook_torrance(*leaves, **{**kw, "light": lt}).sum().backward()
        assert all(t is None or bool(torch.isfinite(t.grad).all()) for t in leaves + [lt]), ("backward", B, h, w, dtype, lights, wf, ltype)
        n_bwd += 1
        if dtype == torch.float32 and wf != "converted":
            mask = torch.rand(1, 1, h, w, device=dev, generator=g)
            second = (a.flip(0), n, r, None if m is None else m.flip(0), None if s is None else s.flip(0), mask)
            ob = F.cook_torrance(a, n, r, m, s, blend=second, **kw)
            assert bool(torch.isfinite(ob).all()), ("blend", B, h, w, lights, wf, ltype)
            n_blend += 1
    if dtype == torch.float16 and lights == 1 and w % 128 == 0 and (ny, nx) == (1, 1) and r.shape[0] == B:
        # the streamed backward kernel (rounds 1 ... 4) == the one-tile kernels, bit for bit
        lib = N.lib()
        grads = []
        for knob in (0, rng.choice([-1, 1, 2, 3])):
            lib.pbr_set_tuning(N.TUNE_BWD_RUN, knob)
            leaves = [None if t is None else t.clone().requires_grad_(True) for t in (a, n, r, m, s)]
            F.cook_torrance(*leaves, **kw).square().sum().backward()
            grads.append([None if t is None else t.grad for t in leaves])
        lib.pbr_set_tuning(N.TUNE_BWD_RUN, -1)
        assert all(x is None or torch.equal(x, y) for x, y in zip(*grads)), ("streamed backward", B, h, w, wf, ltype)
        n_stream += 1
torch.cuda.synchronize()
print(f"soak ok: {n_cfg} configurations x 2 orders, {n_bwd} backward passes, {n_blend} fused blends, {n_stream} streamed backward passes in {seconds:.0f} s")
