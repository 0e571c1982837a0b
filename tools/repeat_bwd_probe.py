#!/usr/bin/env python3
"""Gradients of tiled maps: the repeat-inner backward (pbr_cook_torrance_backward_folded, one kernel) against backward + fold
(PBR_TUNE_TILE_REPEAT = 0), and the rendering-loss step over tiled maps against its three-step form.  2048^2 maps, tile(2) -> 4096^2.
  python tools/repeat_bwd_probe.py [--size 2048] [--tile 2] [--iters 50]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import _native as N, functional as F      # noqa: E402


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--tile", type=int, default=2)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--schedule", type=int, default=0, help="pbr_render_desc.schedule: 0 = the rule, 1 = linear, 1 + c = runs of 1 << c tiles per XCD")
    args = ap.parse_args()
    S, T = args.size, args.tile
    lib = N.lib()
    st = torch.cuda.current_stream().cuda_stream
    for dtype in (torch.float32, torch.float16):
        for light_type, light in (("point", [0.1, 0.1, 1.0]), ("directional", [0.3, -0.2, 1.0])):
            g = torch.Generator(device="cuda").manual_seed(1)
            a = torch.rand(3, S, S, device="cuda", generator=g)
            n = torch.cat([torch.rand(2, S, S, device="cuda", generator=g) - 0.5, torch.ones(1, S, S, device="cuda")], 0)
            r = torch.rand(1, S, S, device="cuda", generator=g) * 0.8 + 0.2
            m = torch.rand(1, S, S, device="cuda", generator=g)
            a, n, r, m = F.pack_maps(*[t.to(dtype) for t in (a, n, r, m)])
            kw = dict(view_dir=[0.0, 0.0, 1.0], light=light, light_intensity=[1.0, 1.0, 1.0], light_type=light_type, light_size=1.0)
            plan = F.plan_cook_torrance(a, n, r, m, tile=T, schedule=args.schedule, **kw)
            d = plan.desc
            gout = torch.rand(1, 3, S * T, S * T, device="cuda", generator=g)
            grads = [torch.empty_like(t) for t in (a, n, r, m)]
            esz = 4 if dtype == torch.float32 else 2
            bytes_one = 12 * (S * T) ** 2 + 2 * 8 * esz * S * S
            rows = {}
            for knob, name in ((-1, "one kernel"), (0, "backward + fold")):
                lib.pbr_set_tuning(N.TUNE_TILE_REPEAT, knob)
                ws_bytes = lib.pbr_backward_folded_workspace_bytes(ctypes.byref(d))
                ws = torch.empty(max(1, ws_bytes), dtype=torch.uint8, device="cuda")

                def run():
                    N.check(lib.pbr_cook_torrance_backward_folded(ctypes.byref(d), gout.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(),
                                                                  grads[2].data_ptr(), grads[3].data_ptr(), None, ws.data_ptr(), st))
                rows[name] = timed(run, args.iters)
                del ws
            lib.pbr_set_tuning(N.TUNE_TILE_REPEAT, -1)
            target = torch.rand(1, 3, S * T, S * T, device="cuda", generator=g)
            loss = torch.empty((), device="cuda")
            wsl = torch.empty(max(1, lib.pbr_mse_step_workspace_bytes(ctypes.byref(d))), dtype=torch.uint8, device="cuda")

            def step():
                N.check(lib.pbr_cook_torrance_mse_step(ctypes.byref(d), target.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(), grads[2].data_ptr(),
                                                       grads[3].data_ptr(), None, loss.data_ptr(), wsl.data_ptr(), st))
            rows["loss step"] = timed(step, args.iters)
            fwd = timed(lambda: plan.launch(st), args.iters)
            print("%s %-11s  forward %.1f us | folded backward: one kernel %.1f us (%.3f of HBM at %d MB), backward + fold %.1f us | loss step %.1f us"
                  % ("fp32" if dtype == torch.float32 else "fp16", light_type, fwd, rows["one kernel"], bytes_one / rows["one kernel"] / 8e6, bytes_one >> 20,
                     rows["backward + fold"], rows["loss step"]), flush=True)


if __name__ == "__main__":
    main()
