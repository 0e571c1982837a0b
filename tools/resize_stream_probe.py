"""resize_stream_probe.py -- the row-streaming down-scale (csrc/resize_stream.hpp) against the strip kernel it replaces on antialiased down-scales
that are not a whole factor: bit-identity (knob RESIZE_UP2 = 2 takes the walk at every factor, 0 selects the strip form), distance from ATen's own antialiased interpolate on the
CPU, non-finite inputs, and time (the whole call: tables kernel + walk).

    python tools/resize_stream_probe.py            # on an MI355X box (gpurun);  CHECK=0 skips the parity part, PLANES=8 the 537 MB input
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


WARM_MS = float(os.environ.get("WARM_MS", "150"))


def timed(fn, reps=50, warm=5):
    """After an idle moment the GPU runs ~20 launches at boost clocks and the next ones up to 25 % slower (tools/transient_probe.py): launch for WARM_MS first, as
    tools/run_kernels.py does -- the walk loses more to the settled clocks than the strip kernel does (8 x 4096^2 -> 1365^2: 107 -> 132 us against 119 -> 122)."""
    import time
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + WARM_MS * 1e-3
    while time.perf_counter() < t_end:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator(device=DEV).manual_seed(7)
bad = 0
if os.environ.get("CHECK", "1") != "0":
    shapes = [(1, 64, 64, 40, 40), (3, 512, 512, 341, 341), (2, 1024, 1024, 100, 100), (1, 1000, 1024, 333, 700), (3, 256, 2048, 77, 1365),
              (1, 2048, 2048, 1365, 1365), (2, 2048, 2048, 200, 200), (1, 4096, 4096, 400, 400), (1, 4096, 4096, 3000, 3000), (1, 4096, 4096, 4000, 4000),
              (2, 777, 1024, 123, 321), (1, 96, 128, 17, 16), (3, 600, 800, 37, 49), (1, 4096, 4096, 249, 249), (1, 1024, 1024, 1000, 1000),
              (4, 300, 512, 7, 500), (1, 2048, 4096, 2047, 1366), (2, 128, 5000, 50, 1234)]
    for planes, hi, wi, ho, wo in shapes:
        a = torch.rand(planes, hi, wi, device=DEV, generator=g) * 2 - 0.5
        new, strip = resize(a, ho, wo, 2), resize(a, ho, wo, 0)
        ref = torch.nn.functional.interpolate(a.cpu()[None], size=(ho, wo), mode="bilinear", antialias=True, align_corners=False)[0]
        same = torch.equal(new, strip)
        err = float((new.cpu() - ref).abs().max())
        ok = same and err <= 2e-6 and not bool(torch.isnan(new).any())
        bad += not ok
        print(f"planes={planes} {hi}x{wi} -> {ho}x{wo}: bit-identical to the strip kernel: {same}; max |new - ATen| = {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
        if not same:
            d = (new - strip).abs()
            d[torch.isnan(d)] = 1e9
            idx = torch.nonzero(d > 0)[:6].tolist()
            print("   first differences at", idx, [float(d[tuple(i)]) for i in idx], "count", int((d > 0).sum()))
    # non-finite inputs poison exactly the outputs whose windows hold them: the same set as the strip kernel's (and ATen's)
    a = torch.rand(1, 512, 512, device=DEV, generator=g)
    a[0, 100, 200] = float("inf"); a[0, 300, 17] = float("nan"); a[0, 511, 511] = float("-inf")
    new, strip = resize(a, 150, 150, 2), resize(a, 150, 150, 0)
    ref = torch.nn.functional.interpolate(a.cpu()[None], size=(150, 150), mode="bilinear", antialias=True, align_corners=False)[0]
    same = torch.equal(torch.isfinite(new), torch.isfinite(strip)) and torch.equal(torch.isfinite(new).cpu(), torch.isfinite(ref))
    same = same and torch.equal(new[torch.isfinite(new)], strip[torch.isfinite(strip)])
    print(f"non-finite inputs: the same outputs poisoned as the strip kernel and ATen: {same} ({int((~torch.isfinite(new)).sum())} outputs)", flush=True)
    bad += not same
    print("failures:", bad, flush=True)

Sz = 4096
P = int(os.environ.get("PLANES", "3"))
REPS = int(os.environ.get("REPS", "4"))
keep, res = [], {}
# the time of a launch moves with WHERE its buffers lie (by +-15 % between processes): every repetition allocates its input, result and workspace anew
# (the old ones stay alive, so the addresses differ); both forms run on the same buffers; min and median over the repetitions are printed.
for rep in range(REPS):
    a = torch.rand(P, Sz, Sz, device=DEV, generator=g)
    keep.append(a)
    for ho in (3000, 2731, 2000, 1365, 1000, 700, 400, 300):
        out = torch.empty(P, ho, ho, device=DEV)
        ws = torch.empty(max(1, lib.pbr_resize_workspace_bytes(P, Sz, ho) // 4), device=DEV)
        keep += [out, ws]
        call = lambda: lib.pbr_resize_bilinear(a.data_ptr(), out.data_ptr(), P, Sz, Sz, ho, ho, 1, ws.data_ptr(), stream)
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 2)
        us = timed(call, reps=20)
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 0)
        us0 = timed(call, reps=20)
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, -1)
        res.setdefault(ho, []).append((us, us0))
for ho, v in res.items():
    nbytes = 4 * P * (Sz * Sz + ho * ho)
    st, sp = sorted(x[0] for x in v), sorted(x[1] for x in v)
    med = lambda z: z[len(z) // 2]
    print(f"{P} x 4096^2 -> {ho}^2: stream min {st[0]:.1f} med {med(st):.1f} us ({nbytes / med(st) / 8e6:.3f}) | strip min {sp[0]:.1f} med {med(sp):.1f} us ({nbytes / med(sp) / 8e6:.3f})", flush=True)
sys.exit(1 if bad else 0)
