"""resize_down_probe.py -- the register-only whole-factor down-scale (csrc/resize_down.hpp) against the strip kernel it replaces:
bit-identity (knob RESIZE_UP2 = 0 selects the strip form), distance from ATen's own antialiased interpolate on the CPU, and time.

    python tools/resize_down_probe.py            # on an MI355X box (gpurun)
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pypbr_amd import _native as N

DEV = torch.device("cuda:0")
lib = N.lib()
stream = torch.cuda.current_stream(DEV).cuda_stream


def resize(a, ho, wo, knob):
    planes, hi, wi = a.shape
    out = torch.full((planes, ho, wo), float("nan"), device=DEV)
    ws = torch.empty(max(1, lib.pbr_resize_workspace_bytes(planes, hi, wo) // 4), device=DEV)
    lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, knob)
    try:
        N.check(lib.pbr_resize_bilinear(a.data_ptr(), out.data_ptr(), planes, hi, wi, ho, wo, 1, ws.data_ptr(), stream))
        torch.cuda.synchronize()
    finally:
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, -1)
    return out


def timed(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator(device=DEV).manual_seed(5)
bad = 0
for S in (2, 3, 4, 5, 6, 7, 8, 16):
    for planes, ho, wo in ((1, 2, 8), (3, 8, 256), (2, 13, 260), (1, 64, 1028), (3, 37, 12), (1, 512, 512), (2, 100, 2048 // S // 4 * 4)):
        a = torch.rand(planes, S * ho, S * wo, device=DEV, generator=g) * 2 - 0.5
        new, strip = resize(a, ho, wo, 1), resize(a, ho, wo, 0)
        ref = torch.nn.functional.interpolate(a.cpu()[None], size=(ho, wo), mode="bilinear", antialias=True, align_corners=False)[0]
        same = torch.equal(new, strip)
        err = float((new.cpu() - ref).abs().max())
        ok = same and err <= 2e-6 and not bool(torch.isnan(new).any())
        bad += not ok
        print(f"S={S} planes={planes} {S * ho}x{S * wo} -> {ho}x{wo}: bit-identical to the strip kernel: {same}; max |new - ATen| = {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
        if not same:
            d = (new - strip).abs()
            idx = torch.nonzero(d > 0)[:6].tolist()
            print("   first differences at", idx, [float(d[tuple(i)]) for i in idx])
print("failures:", bad, flush=True)

Sz = 4096
P = int(os.environ.get("PLANES", "3"))
a = torch.rand(P, Sz, Sz, device=DEV, generator=g)
variants = [int(v) for v in os.environ.get("VARIANTS", "0").split(",")]
for rnd in range(2):
    for S in (2, 4, 8, 16):
        ho = Sz // S
        out = torch.empty(P, ho, ho, device=DEV)
        ws = torch.empty(max(1, lib.pbr_resize_workspace_bytes(P, Sz, ho) // 4), device=DEV)
        nbytes = 4 * P * (Sz * Sz + ho * ho)
        line = []
        for v in variants:
            os.environ["PBR_DOWN_VARIANT"] = str(v)
            us = timed(lambda: lib.pbr_resize_bilinear(a.data_ptr(), out.data_ptr(), P, Sz, Sz, ho, ho, 1, ws.data_ptr(), stream))
            line.append(f"v{v} {us:.1f} us ({nbytes / us / 8e6:.3f})")
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 0)
        us = timed(lambda: lib.pbr_resize_bilinear(a.data_ptr(), out.data_ptr(), P, Sz, Sz, ho, ho, 1, ws.data_ptr(), stream), reps=10)
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, -1)
        print(f"{P} x 4096^2 -> {ho}^2: " + " | ".join(line) + f" | strip / two-pass {us:.1f} us ({nbytes / us / 8e6:.3f})", flush=True)
sys.exit(1 if bad else 0)
