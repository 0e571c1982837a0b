#!/usr/bin/env python3
"""Launch shape of the streaming map kernels (PBR_TUNE_STREAM_SHAPE / _LDS): 2048 x 256 workgroups walking the data against one item
per lane (256-lane or one-wave workgroups), with and without an occupancy cap.  python tools/stream_shape_probe.py [op-substring]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypbr_amd import _native as N  # noqa: E402

lib = N.lib()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
S = 4096
PX = S * S
ONLY = sys.argv[1] if len(sys.argv) > 1 else ""
g = torch.Generator(device=dev).manual_seed(0)
a, b3, c3 = (torch.rand(3, S, S, device=dev, generator=g) for _ in range(3))
m, m2 = torch.rand(1, S, S, device=dev, generator=g), torch.rand(1, S, S, device=dev, generator=g)
o3, o3b, o3c, o1, o1b = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a), torch.empty_like(m), torch.empty_like(m)


def timed(fn, iters=60):
    for _ in range(30):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


P = lambda t: t.data_ptr()
ops = (
    ("metallic_to_specular (4 in, 6 out)", 40 * PX, lambda: lib.pbr_metallic_to_specular(P(a), P(m), P(o3), P(o3b), 1, PX, 1, N.F32, stream)),
    ("specular_to_metallic (6 in, 6 out)", 48 * PX, lambda: lib.pbr_specular_to_metallic(P(a), P(b3), P(o3), P(o3b), 3 * PX, 1, N.F32, stream)),
    ("srgb_to_linear (3 in, 3 out)", 24 * PX, lambda: lib.pbr_srgb_to_linear(P(a), P(o3), 3 * PX, N.F32, stream)),
    ("srgb_to_linear backward (6 in, 3 out)", 36 * PX, lambda: lib.pbr_srgb_to_linear_backward(P(a), P(b3), P(o3), 3 * PX, N.F32, stream)),
    ("metallic_to_specular backward (10 in, 4 out)", 56 * PX,
     lambda: lib.pbr_metallic_to_specular_backward(P(a), P(m), P(b3), P(c3), P(o3), P(o1), 1, PX, 1, N.F32, stream)),
    ("specular_to_metallic backward (12 in, 6 out)", 72 * PX,
     lambda: lib.pbr_specular_to_metallic_backward(P(a), P(b3), P(c3), P(o3c), P(o3), P(o3b), 3 * PX, 1, N.F32, stream)),
    ("blend_maps 3 ch (7 in, 3 out)", 40 * PX, lambda: lib.pbr_blend_maps(P(a), P(b3), P(m), P(o3), 3, PX, 0, stream)),
    ("blend_maps normals (7 in, 3 out)", 40 * PX, lambda: lib.pbr_blend_maps(P(a), P(b3), P(m), P(o3), 3, PX, 1, stream)),
    ("blend_maps backward 3 ch (10 in, 7 out)", 68 * PX,
     lambda: lib.pbr_blend_maps_backward(P(a), P(b3), P(m), P(c3), P(o3), P(o3b), P(o1), 3, PX, 0, 0, stream)),
    ("blend_maps backward normals (10 in, 7 out)", 68 * PX,
     lambda: lib.pbr_blend_maps_backward(P(a), P(b3), P(m), P(c3), P(o3), P(o3b), P(o1), 3, PX, 1, 0, stream)),
    ("sigmoid mask (2 in, 1 out)", 12 * PX, lambda: lib.pbr_blend_sigmoid_mask(P(m), P(m2), P(o1), PX, -0.5, 0.1, stream)),
)
for name, nbytes, fn in ops:
    if ONLY not in name:
        continue
    N.check(fn())
    best = None
    for shape in (-1, 0, 1, 2):                     # -1: the launcher's rule
        for lds in (0, 10240, 13312, 16384, 20480, 26624):
            if shape <= 0 and lds:
                continue
            lib.pbr_set_tuning(N.TUNE_STREAM_SHAPE, shape)
            lib.pbr_set_tuning(N.TUNE_STREAM_LDS, lds if shape >= 0 else -1)
            us = timed(fn)
            if best is None or us < best[0]:
                best = (us, shape, lds)
            print(f"{name} shape={shape:2d} lds={lds:6d}: {us:7.1f} us  {nbytes / us / 1e3:6.0f} GB/s  {nbytes / us / 1e3 / 8000:.3f}", flush=True)
    print(f"  -> best for {name}: shape {best[1]} lds {best[2]} {best[0]:.1f} us", flush=True)
lib.pbr_set_tuning(N.TUNE_STREAM_SHAPE, -1)
lib.pbr_set_tuning(N.TUNE_STREAM_LDS, -1)
