#!/usr/bin/env python3
"""Launches every kernel of libpbr_hip.so a few times at a probe shape large enough to stream from HBM: the target of
the rocprofv3 passes of tools/collect_kernels.sh (kernel trace + FETCH_SIZE / WRITE_SIZE / SQ counters in separate
passes).  Prints one JSON line per case: which kernel (substring of the rocprof name), the algorithmic bytes of one
launch (SURVEY.md 8d accounting: every map plane read once, every result plane written once) and its own HIP-event
timing.   python tools/run_kernels.py [reps] [only-substring]"""
import ctypes
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_material  # noqa: E402
from pypbr_amd import _native as N, blending as B, functional as F  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""
WARM_MS = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0      # settle the clocks before timing: launch for this long first
DEV = torch.device("cuda", 0)
S = 4096
PX = S * S


def timed(fn, reps=REPS, warm=3):
    """`warm` launches, then -- if WARM_MS -- launches until that much time has passed (after an idle moment the GPU runs
    ~20 launches at boost clocks and the next ~100 up to 25 % slower: tools/transient_probe.py), then `reps` timed ones.
    kernels_summary.py looks at the LAST `reps` dispatches of each kernel only."""
    import time
    for _ in range(warm):
        fn()
    if WARM_MS:
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


def report(case, kernel, alg_bytes, us, **extra):
    line = {"case": case, "kernel": kernel, "reps": extra.pop("reps", REPS), "algorithmic_bytes_per_launch": int(alg_bytes), "us_per_launch_hip_events": round(us, 1),
            "GBps_algorithmic": round(alg_bytes / us / 1e3, 1), "frac_of_8TBps": round(alg_bytes / us / 1e3 / 8000.0, 4)}
    line.update(extra)
    print(json.dumps(line), flush=True)


def batch(B_, size, dtype=torch.float32, seed=0):
    ms = [synth_material(size, DEV, seed + i) for i in range(B_)]
    return [torch.stack([m[k] for m in ms]).to(dtype) for k in range(4)]


def want(name):
    return ONLY in name


PT = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
stream = torch.cuda.current_stream(DEV).cuda_stream
lib = N.lib()

if want("fwd_f32"):
    a, n, r, m, out = F.pack_maps(*synth_material(S, DEV, 1), reserve_output=True)
    p = F.plan_cook_torrance(a, n, r, m, out=out.unsqueeze(0), **PT)
    report("fwd_f32: 1 x 4096^2 point metallic fp32 (bench.py workload)", "cook_torrance_kernel<1, 0, float, float, 4, false, true, false>",
           p.bytes_per_pixel * PX, timed(lambda: p.launch(stream), reps=max(REPS, 50), warm=20), reps=max(REPS, 50))
    del a, n, r, m, out, p
if want("fwd_f16"):
    h = batch(4, S, torch.float16, 10)
    p = F.plan_cook_torrance(*h, **PT)
    report("fwd_f16: 4 x 4096^2 point metallic, fp16 maps -> fp32", "cook_torrance_kernel<1, 0, __half, float, 8, false, true, false>",
           p.bytes_per_pixel * 4 * PX, timed(lambda: p.launch(stream)))
    p2 = F.plan_cook_torrance(*h, out_dtype=torch.float16, **PT)
    report("fwd_f16_f16: same, fp16 result", "cook_torrance_kernel<1, 0, __half, __half, 8, false, true, false>",
           p2.bytes_per_pixel * 4 * PX, timed(lambda: p2.launch(stream)))
    lights = [[math.cos(t), math.sin(t), 1.0] for t in [2 * math.pi * i / 16 for i in range(16)]]
    p16 = F.plan_cook_torrance(*h, view_dir=[0, 0, 1], light=lights, light_intensity=[[1.0 / 16] * 3] * 16, light_type="point", light_size=1.0)
    us = timed(lambda: p16.launch(stream), reps=min(REPS, 20), warm=60)     # steady state: the first ~50 launches run 7 % slower
    report("fwd_16_lights: 4 x 4096^2, 16 point lights, fp16 maps -> fp32 (config 5 share)", "cook_torrance_batch_kernel<1, 0, __half, float, 2, 4, true>",
           p16.bytes_per_pixel * 4 * PX, us, Gpixels_per_s=round(4 * PX / us / 1e3, 1), reps=min(REPS, 20))
    del h, p, p2, p16
if want("bwd"):
    for dtype, tag, kern in ((torch.float32, "bwd_f32", "cook_torrance_backward_kernel<1, 0, 4, false, float, false>"),
                             (torch.float16, "bwd_f16", "cook_torrance_backward_stream_kernel<1, 0, true>")):      # PBR_TUNE_BWD_RUN=0: the one-tile kernel <1, 0, 2, false, __half, false>
        maps = [t.to(dtype) for t in synth_material(S, DEV, 7)]
        plan = F.plan_cook_torrance(*maps, **PT)
        gout = torch.rand(1, 3, S, S, device=DEV)
        grads = [torch.empty_like(t) for t in maps]

        def bwd():
            N.check(lib.pbr_cook_torrance_backward(ctypes.byref(plan.desc), gout.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(),
                                                   grads[2].data_ptr(), grads[3].data_ptr(), None, stream))
        es = maps[0].element_size()
        report(f"{tag}: backward 1 x 4096^2 point metallic ({dtype}) maps 8 planes + grad_out 3 fp32 in, 8 gradient planes out", kern,
               (8 * es + 12 + 8 * es) * PX, timed(bwd))
        del maps, plan, gout, grads
if want("bwd_dir"):
    maps = [t.half() for t in synth_material(S, DEV, 7)]
    plan = F.plan_cook_torrance(*maps, view_dir=[0, 0, 1], light=[0.3, -0.2, 1.0], light_intensity=[1, 1, 1], light_type="directional")
    gout = torch.rand(1, 3, S, S, device=DEV)
    grads = [torch.empty_like(t) for t in maps]

    def bwd_dir():
        N.check(lib.pbr_cook_torrance_backward(ctypes.byref(plan.desc), gout.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(),
                                               grads[2].data_ptr(), grads[3].data_ptr(), None, stream))
    report("bwd_dir_f16: backward 1 x 4096^2 directional metallic fp16 maps", "cook_torrance_backward_stream_kernel<0, 0, true>", 44 * PX, timed(bwd_dir))
    del maps, plan, gout, grads
if want("blend_fused"):
    m1, m2 = synth_material(S, DEV, 21), synth_material(S, DEV, 22)
    mask = torch.rand(1, S, S, device=DEV)
    p = F.plan_cook_torrance(*m1, blend=(m2[0], m2[1], m2[2], m2[3], None, mask), **PT)
    report("blend_fused: blend_with_mask + re-decode + render, 4096^2 (17 planes in, 3 out)", "cook_torrance_blend_kernel<1, 0, 4, false>",
           80 * PX, timed(lambda: p.launch(stream)))
    del m1, m2, mask, p
if want("blend_tiled_fwd"):
    # round 6: the fused blend over TILED maps walks the source -- both materials' texels and the mask loaded once, blended once per texel, evaluated at
    # every repeat (until then: wrap-around addressing, the blend at every output pixel: 154 us)
    m1, m2 = synth_material(2048, DEV, 21), synth_material(2048, DEV, 22)
    mask = torch.rand(1, 2048, 2048, device=DEV)
    p = F.plan_cook_torrance(*m1, blend=(m2[0], m2[1], m2[2], m2[3], None, mask), tile=2, **PT)
    report("blend_tiled_fwd: blend_with_mask + re-decode + render over 2048^2 maps under tile(2) -> 4096^2 (17 planes of 2048^2 in once, 3 planes of 4096^2 out)",
           "cook_torrance_repeat_blend_kernel<1, 0, false>", 68 * 2048 * 2048 + 12 * PX, timed(lambda: p.launch(stream)))
    del m1, m2, mask, p
if want("tiled"):
    maps = synth_material(2048, DEV, 31)
    p = F.plan_cook_torrance(*maps, tile=2, **PT)
    report("tiled: 2048^2 maps, fused tile(2) -> 4096^2 image (8 planes of 2048^2 in once, 3 planes of 4096^2 out): repeat-inner kernel",
           "cook_torrance_repeat_kernel<1, 0, float, float, false, true>", 32 * 2048 * 2048 + 12 * PX, timed(lambda: p.launch(stream)))
    # a THIN row band of the same tiled image -- rows [1536, 2560) of 4096: a quarter of the image across the period boundary, what one of four
    # ranks holds of ONE tiled material -- on the same kernel through its window of source rows (round 6; until then the wrap-around form)
    p = F.plan_cook_torrance(*maps, tile=2, y_offset=1536, rows=1024, **PT)
    report("tiled_band: rows [1536, 2560) of the same tiled image (a band thinner than a period: 1024 source rows x 2048 in once, 3 planes of 1024 x 4096 out)",
           "cook_torrance_repeat_kernel<1, 0, float, float, false, true>", 32 * 1024 * 2048 + 12 * 1024 * S, timed(lambda: p.launch(stream)))
    del maps, p
    maps = [t.half() for t in synth_material(2048, DEV, 32)]
    p = F.plan_cook_torrance(*maps, tile=2, **PT)
    report("tiled_f16: 2048^2 fp16 maps, fused tile(2) -> 4096^2 fp32 image (8 planes of 2048^2 in once, 3 planes of 4096^2 out): repeat-inner kernel",
           "cook_torrance_repeat_kernel<1, 0, __half, float, false, true>", 16 * 2048 * 2048 + 12 * PX, timed(lambda: p.launch(stream)))
    del maps, p
if want("tiled_multi"):
    # round 5: several lights over tiled maps take the repeat-inner walk too (texels read and decoded once); beside it the wrap-around form
    lights4 = [[math.cos(t), math.sin(t), 1.0] for t in [2 * math.pi * i / 4 for i in range(4)]]
    kw4 = dict(view_dir=[0, 0, 1], light=lights4, light_intensity=[[0.25] * 3] * 4, light_type="point", light_size=1.0)
    maps = F.pack_maps(*synth_material(2048, DEV, 31))
    p = F.plan_cook_torrance(*maps, tile=2, **kw4)
    report("tiled_multi: 2048^2 maps, tile(2) -> 4096^2, 4 point lights: repeat-inner kernel (8 planes of 2048^2 in once, 3 planes of 4096^2 out)",
           "cook_torrance_repeat_kernel<1, 0, float, float, false, true, true>", 32 * 2048 * 2048 + 12 * PX, timed(lambda: p.launch(stream)))
    leaves = maps
    plan4 = F.plan_cook_torrance(*leaves, tile=2, **kw4)
    gout = torch.rand(1, 3, S, S, device=DEV)
    grads = [torch.empty_like(t) for t in leaves]
    report("tiled_multi_bwd: folded gradient of the same tiled maps under 4 point lights: 8 map planes + 3 upstream planes of 4096^2 in, 8 gradient planes of 2048^2 out",
           "cook_torrance_repeat_backward_kernel<1, 0, float, false, true>", 12 * PX + 64 * 2048 * 2048,
           timed(lambda: N.check(lib.pbr_cook_torrance_backward_folded(ctypes.byref(plan4.desc), gout.data_ptr(), *[t.data_ptr() for t in grads], None, None, stream))))
    del maps, p, plan4, gout, grads, leaves
if want("tiled_bwd"):
    # round 5: gradients of tiled maps folded in registers (pbr_cook_torrance_backward_folded) and the loss step over tiled maps
    for dtype, tag in ((torch.float32, "f32"), (torch.float16, "f16")):
        maps = F.pack_maps(*[t.to(dtype) for t in synth_material(2048, DEV, 41)])
        plan = F.plan_cook_torrance(*maps, tile=2, **PT)
        gout = torch.rand(1, 3, S, S, device=DEV)
        grads = [torch.empty_like(t) for t in maps]
        es = maps[0].element_size()
        tm = "float" if dtype == torch.float32 else "__half"
        report(f"tiled_bwd_{tag}: folded gradient of 2048^2 maps under tile(2) -> 4096^2 ({dtype}): 8 map planes + 3 upstream planes of 4096^2 in, 8 gradient planes of 2048^2 out",
               f"cook_torrance_repeat_backward_kernel<1, 0, {tm}, false>", 12 * PX + 16 * es * 2048 * 2048,
               timed(lambda: N.check(lib.pbr_cook_torrance_backward_folded(ctypes.byref(plan.desc), gout.data_ptr(), *[t.data_ptr() for t in grads], None, None, stream))))
        loss = torch.empty((), device=DEV)
        ws = torch.empty(max(1, lib.pbr_mse_step_workspace_bytes(ctypes.byref(plan.desc)) // 4), device=DEV)
        report(f"tiled_bwd_loss_{tag}: rendering-loss step over the same tiled maps: 8 map planes + target image of 4096^2 in, 8 gradient planes of 2048^2 out",
               f"cook_torrance_repeat_backward_kernel<1, 0, {tm}, true>", 12 * PX + 16 * es * 2048 * 2048,
               timed(lambda: N.check(lib.pbr_cook_torrance_mse_step(ctypes.byref(plan.desc), gout.data_ptr(), *[t.data_ptr() for t in grads], None,
                                                                    loss.data_ptr(), ws.data_ptr(), stream))))
        del maps, plan, gout, grads, ws
    # one directional light: the repeats' upstream values are summed before the chain rule (one evaluation per texel): memory-bound
    maps = F.pack_maps(*synth_material(2048, DEV, 41))
    sun = dict(view_dir=[0, 0, 1], light=[0.3, -0.2, 1.0], light_intensity=[1, 1, 1], light_type="directional")
    plan = F.plan_cook_torrance(*maps, tile=2, **sun)
    gout = torch.rand(1, 3, S, S, device=DEV)
    grads = [torch.empty_like(t) for t in maps]
    report("tiled_bwd_dir_f32: folded gradient of 2048^2 maps under tile(2) -> 4096^2, ONE directional light (fp32): 8 map planes + 3 upstream planes of 4096^2 in, 8 gradient planes of 2048^2 out",
           "cook_torrance_repeat_backward_kernel<0, 0, float, false, false>", 12 * PX + 64 * 2048 * 2048,
           timed(lambda: N.check(lib.pbr_cook_torrance_backward_folded(ctypes.byref(plan.desc), gout.data_ptr(), *[t.data_ptr() for t in grads], None, None, stream))))
    del maps, plan, gout, grads
if want("map_ops"):
    g = torch.Generator(device=DEV).manual_seed(0)
    a = torch.rand(3, S, S, device=DEV, generator=g)
    n = torch.rand(3, S, S, device=DEV, generator=g)
    m = torch.rand(1, S, S, device=DEV, generator=g)
    o3, o3b, o1 = torch.empty_like(a), torch.empty_like(a), torch.empty_like(m)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    report("map_ops srgb_to_linear 3 x 4096^2 fp32", "colour_kernel<float, true>", 24 * PX,
           timed(lambda: lib.pbr_srgb_to_linear(a.data_ptr(), o3.data_ptr(), a.numel(), N.F32, stream)))
    report("map_ops linear_to_srgb 3 x 4096^2 fp32", "colour_kernel<float, false>", 24 * PX,
           timed(lambda: lib.pbr_linear_to_srgb(a.data_ptr(), o3.data_ptr(), a.numel(), N.F32, stream)))
    report("map_ops metallic -> diffuse/specular 4096^2 (4 planes in, 6 out)", "metallic_to_specular_kernel", 40 * PX,
           timed(lambda: lib.pbr_metallic_to_specular(a.data_ptr(), m.data_ptr(), o3.data_ptr(), o3b.data_ptr(), 1, PX, 1, N.F32, stream)))
    report("map_ops diffuse/specular -> basecolor/metallic 4096^2 (6 planes in, 6 out)", "specular_to_metallic_kernel", 48 * PX,
           timed(lambda: lib.pbr_specular_to_metallic(a.data_ptr(), n.data_ptr(), o3.data_ptr(), o3b.data_ptr(), a.numel(), 0, N.F32, stream)))
    report("map_ops decode_normal 3 ch [0,1]-encoded 4096^2, one pass: probe, decode + record any negative (3 planes in, 3 out), fix-up kernel that returns at once", "decode_normal_speculative_kernel", 24 * PX,
           timed(lambda: lib.pbr_decode_normal(n.data_ptr(), o3.data_ptr(), 3, PX, N.F32, flag.data_ptr(), stream)))
    # fp16 here, so that the copy kernel of THIS case has a name of its own: keep_normal_kernel<float> also runs -- and returns at
    # once -- in the encoded-map case above, and the counter averages are per kernel name (round 2's record mixed the two: 0.667)
    n_signed = (n * 2 - 1).half()
    o3h = torch.empty_like(n_signed)
    report("map_ops decode_normal 3 ch already signed 4096^2 fp16, one pass: the probe sees a negative value, the decode returns at once, the map is copied as it is (3 planes in, 3 out)", "keep_normal_kernel<__half>", 12 * PX,
           timed(lambda: lib.pbr_decode_normal(n_signed.data_ptr(), o3h.data_ptr(), 3, PX, N.F16, flag.data_ptr(), stream)))
    n_inplace = n.clone()
    n_late = n.clone()
    n_late.view(-1)[-1] = -1.0
    report("map_ops decode_normal 3 ch, the only negative value is one the probe does not see: full decode, then the copy (6 planes in, 6 out)", "decode_normal_speculative_kernel", 48 * PX,
           timed(lambda: lib.pbr_decode_normal(n_late.data_ptr(), o3.data_ptr(), 3, PX, N.F32, flag.data_ptr(), stream)))
    report("map_ops decode_normal in place (the map is signed after the first call: the probe settles the flag, the flag pass leaves at once, the transform copies 3 planes in, 3 out)", "decode_normal_kernel", 24 * PX,
           timed(lambda: lib.pbr_decode_normal(n_inplace.data_ptr(), n_inplace.data_ptr(), 3, PX, N.F32, flag.data_ptr(), stream)))
    # round 4: maps that come out of image files arrive as samples (3 bytes per texel in, 12 out)
    rgb = torch.randint(0, 256, (S, S, 3), dtype=torch.uint8, device=DEV, generator=g)
    report("unpack_image 4096^2 RGB uint8 samples -> 3 float32 planes (3 B in, 12 B out per texel)", "unpack_dense_kernel<unsigned char, 3, false>", 15 * PX,
           timed(lambda: lib.pbr_unpack_image(rgb.data_ptr(), 8, 3, S, S, 1, 3 * S, 3, o3.data_ptr(), 0, stream)))
    report("unpack_image 4096^2 RGB uint8 normal map -> decoded unit normals (base.py:191-242 in the same pass)", "unpack_dense_kernel<unsigned char, 3, true>", 15 * PX,
           timed(lambda: lib.pbr_unpack_image(rgb.data_ptr(), 8, 3, S, S, 1, 3 * S, 3, o3.data_ptr(), 1, stream)))
    report("blend_maps 3 ch 4096^2 (7 planes in, 3 out)", "blend_kernel<false>", 40 * PX,
           timed(lambda: lib.pbr_blend_maps(a.data_ptr(), n.data_ptr(), m.data_ptr(), o3.data_ptr(), 3, PX, 0, stream)))
    report("blend_maps normals 4096^2 (7 planes in, 3 out)", "blend_kernel<true>", 40 * PX,
           timed(lambda: lib.pbr_blend_maps(a.data_ptr(), n.data_ptr(), m.data_ptr(), o3.data_ptr(), 3, PX, 1, stream)))
    m_other = torch.rand(1, S, S, device=DEV, generator=g)            # two DISTINCT property maps (round 2 passed one buffer twice: 8 B/pixel moved, 12 counted)
    report("sigmoid mask 4096^2 (2 in, 1 out)", "sigmoid_mask_kernel", 12 * PX,
           timed(lambda: lib.pbr_blend_sigmoid_mask(m.data_ptr(), m_other.data_ptr(), o1.data_ptr(), PX, 0.0, 0.1, stream)))
    # round 3: gradients of the map ops
    g3 = torch.rand(3, S, S, device=DEV, generator=g)
    report("map_ops srgb_to_linear backward 3 x 4096^2 fp32 (map + upstream gradient in, gradient out)", "colour_backward_kernel<float, true>", 36 * PX,
           timed(lambda: lib.pbr_srgb_to_linear_backward(a.data_ptr(), g3.data_ptr(), o3.data_ptr(), a.numel(), N.F32, stream)))
    report("map_ops metallic -> diffuse/specular backward 4096^2 (4 map planes + 6 gradient planes in, 4 out)", "metallic_to_specular_backward_kernel<float, true>", 56 * PX,
           timed(lambda: lib.pbr_metallic_to_specular_backward(a.data_ptr(), m.data_ptr(), g3.data_ptr(), n.data_ptr(), o3.data_ptr(), o1.data_ptr(), 1, PX, 1, N.F32, stream)))
    g3b = torch.rand(3, S, S, device=DEV, generator=g)                # two DISTINCT upstream gradients
    report("map_ops diffuse/specular -> basecolor/metallic backward 4096^2 (6 + 6 planes in, 6 out)", "specular_to_metallic_backward_kernel<float>", 72 * PX,
           timed(lambda: lib.pbr_specular_to_metallic_backward(a.data_ptr(), n.data_ptr(), g3.data_ptr(), g3b.data_ptr(), o3.data_ptr(), o3b.data_ptr(), a.numel(), 0, N.F32, stream)))
    del a, n, m, o3, o3b, o1, g3, g3b, m_other
if want("resize"):
    g = torch.Generator(device=DEV).manual_seed(0)
    a = torch.rand(3, S, S, device=DEV, generator=g)
    # whole factors 2 ... 8 | 16 down (what resize(512) of a 1024^2 ... 4096^2 texture is): the register-only band walk (round 5); any other down-scale: the
    # strip kernel (4096 -> 1365: 3.0007x) below 7 x, the walk down the input rows (round 6: resize_stream.hpp; its time here is the call's: tables kernel + walk) from there up
    # (4096 -> 400: 10.24x); up-scales: the two-tap register kernel.  THREE planes = one map, 201 MB: between launches it stays in the 256 MB
    # memory-side cache; EIGHT planes (537 MB) is the same kernel with nothing left from the launch before -- the HBM figure.
    for planes in (3, 8):
        if planes == 8:
            del a
            a = torch.rand(8, S, S, device=DEV, generator=g)
        # (8 planes: lanes of 16 bytes per row and non-temporal loads for 2x and 4x -- resize.hip: launch_down)
        k2, k4 = ("resize_down_kernel<2, 4, 4, 1, false>", "resize_down_kernel<4, 2, 2, 1, false>") if planes == 3 else \
                 ("resize_down_kernel<2, 4, 2, 3, true>", "resize_down_kernel<4, 2, 1, 3, true>")
        for (ho, wo), aa, kern in (((S // 2, S // 2), True, k2), ((S // 4, S // 4), True, k4),
                                   ((S // 8, S // 8), True, "resize_down_kernel<8, 2, 1, 1, false>"), ((1365, 1365), True, "resize_strip_kernel<false, false, false>"),
                                   ((400, 400), True, "resize_stream_kernel<1, 4, %s>" % ("false" if planes == 3 else "true")),
                                   ((S * 3 // 2, S * 3 // 2), False, "resize_up2_kernel<8>")):
            if planes == 8 and ho > S:
                continue
            out = torch.empty(planes, ho, wo, device=DEV)
            ws = torch.empty(max(1, lib.pbr_resize_workspace_bytes(planes, S, wo) // 4), device=DEV)
            report(f"resize {planes} x 4096^2 -> {ho}x{wo} antialias={aa}", kern, 4 * planes * (PX + ho * wo),
                   timed(lambda: lib.pbr_resize_bilinear(a.data_ptr(), out.data_ptr(), planes, S, S, ho, wo, int(aa), ws.data_ptr(), stream)))
            del out, ws
        # 4096^2 -> 1365^2 by the row walk, which is NOT the rule below 7 x (knob value 2): the figure behind that rule (resize.hip)
        out = torch.empty(planes, 1365, 1365, device=DEV)
        ws = torch.empty(max(1, lib.pbr_resize_workspace_bytes(planes, S, 1365) // 4), device=DEV)
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 2)
        report(f"resize_walk{planes}: {planes} x 4096^2 -> 1365x1365 antialias=True by the row walk (knob value 2; the rule keeps the strip kernel below 7 x)",
               "resize_stream_kernel<1, 4, %s>" % ("false" if planes == 3 else "true"), 4 * planes * (PX + 1365 * 1365),
               timed(lambda: lib.pbr_resize_bilinear(a.data_ptr(), out.data_ptr(), planes, S, S, 1365, 1365, 1, ws.data_ptr(), stream)))
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, -1)
        del out, ws
    del a
if want("blend_bwd"):
    m1, m2 = synth_material(S, DEV, 21), synth_material(S, DEV, 22)
    mask = torch.rand(1, S, S, device=DEV)
    p = F.plan_cook_torrance(*m1, blend=(m2[0], m2[1], m2[2], m2[3], None, mask), **PT)
    p.launch(stream)
    gout = torch.rand(1, 3, S, S, device=DEV)
    g1, g2, gm = [torch.empty_like(t) for t in m1], [torch.empty_like(t) for t in m2], torch.empty_like(mask)
    G1 = N.MapGrads(g1[0].data_ptr(), g1[1].data_ptr(), g1[2].data_ptr(), g1[3].data_ptr(), None)
    G2 = N.MapGrads(g2[0].data_ptr(), g2[1].data_ptr(), g2[2].data_ptr(), g2[3].data_ptr(), None)
    bd = N.BlendDesc.from_buffer_copy(p._blend)
    bd.sign_mode = N.BLEND_SIGN_GIVEN
    report("blend_bwd: backward of the fused blend + render, 4096^2 (17 map planes + 3 gradient planes in, 17 gradient planes out)",
           "cook_torrance_blend_backward_kernel<1, 0, 2, false>", 148 * PX,
           timed(lambda: N.check(lib.pbr_cook_torrance_blend_backward(ctypes.byref(p.desc), ctypes.byref(bd), p._workspace.data_ptr(), gout.data_ptr(),
                                                                      ctypes.byref(G1), ctypes.byref(G2), gm.data_ptr(), stream))))
    del m1, m2, mask, p, gout, g1, g2, gm
if want("blend_bwd_tiled"):
    # round 6: the fused blend's backward over TILED maps -- the blend example's material under tile(2) inside a rendering loss: one kernel blends
    # once per texel, walks the repeats and runs the folded gradients through the blend's chain rule (MAP-sized gradients of both materials + the mask)
    m1, m2 = synth_material(2048, DEV, 21), synth_material(2048, DEV, 22)
    mask = torch.rand(1, 2048, 2048, device=DEV)
    p = F.plan_cook_torrance(*m1, blend=(m2[0], m2[1], m2[2], m2[3], None, mask), tile=2, **PT)
    p.launch(stream)
    gout = torch.rand(1, 3, S, S, device=DEV)
    g1, g2, gm = [torch.empty_like(t) for t in m1], [torch.empty_like(t) for t in m2], torch.empty_like(mask)
    G1 = N.MapGrads(g1[0].data_ptr(), g1[1].data_ptr(), g1[2].data_ptr(), g1[3].data_ptr(), None)
    G2 = N.MapGrads(g2[0].data_ptr(), g2[1].data_ptr(), g2[2].data_ptr(), g2[3].data_ptr(), None)
    bd = N.BlendDesc.from_buffer_copy(p._blend)
    bd.sign_mode = N.BLEND_SIGN_GIVEN
    assert lib.pbr_blend_backward_serves(ctypes.byref(p.desc)) == 1
    report("blend_bwd_tiled: backward of the fused blend + render over 2048^2 maps under tile(2) -> 4096^2 (17 map planes of 2048^2 + 3 upstream planes of 4096^2 in, 17 gradient planes of 2048^2 out)",
           "cook_torrance_repeat_blend_backward_kernel<1, 0>", 12 * PX + 136 * 2048 * 2048,
           timed(lambda: N.check(lib.pbr_cook_torrance_blend_backward(ctypes.byref(p.desc), ctypes.byref(bd), p._workspace.data_ptr(), gout.data_ptr(),
                                                                      ctypes.byref(G1), ctypes.byref(G2), gm.data_ptr(), stream))))
    del m1, m2, mask, p, gout, g1, g2, gm
if want("loss_step"):
    for dtype, tag, kern, bpp in ((torch.float32, "loss_step_f32", "cook_torrance_mse_step_kernel<1, 0, 2, false, float>", 76),
                                  (torch.float16, "loss_step_f16", "cook_torrance_mse_stream_kernel<1, 0, true>", 44)):
        maps = synth_material(S, DEV, 3, dtype)
        target = torch.rand(1, 3, S, S, device=DEV)                   # any image serves the traffic measurement (no extra kernel under the counters)
        plan = F.plan_cook_torrance(*maps, **PT)
        grads = [torch.empty_like(t) for t in maps]
        loss = torch.empty((), device=DEV)
        ws = torch.empty(max(1, lib.pbr_mse_step_workspace_bytes(ctypes.byref(plan.desc)) // 4), device=DEV)
        report(f"{tag}: rendering-loss step 1 x 4096^2 point metallic ({dtype}): 8 map planes + target image in, 8 gradient planes out (+ two small reduction kernels)",
               kern, bpp * PX,
               timed(lambda: N.check(lib.pbr_cook_torrance_mse_step(ctypes.byref(plan.desc), target.data_ptr(), *[t.data_ptr() for t in grads], None,
                                                                    loss.data_ptr(), ws.data_ptr(), stream))))
        del maps, target, plan, grads
if want("resize_bwd"):
    g = torch.Generator(device=DEV).manual_seed(0)
    ho = S // 2
    gout = torch.rand(3, ho, ho, device=DEV, generator=g)
    gin = torch.empty(3, S, S, device=DEV)
    ws = torch.empty(max(1, lib.pbr_resize_backward_workspace_bytes(3, S, S, ho, ho) // 4), device=DEV)
    us = timed(lambda: lib.pbr_resize_bilinear_backward(gout.data_ptr(), gin.data_ptr(), 3, S, S, ho, ho, 1, ws.data_ptr(), stream))
    # one pass: the register gather over the transposed tap tables (plus the small kernel that writes the tables); `us` is the whole call
    report("resize backward 3 x 2048^2 gradient -> 4096^2 (gradient of a 2x down-scale: 3 planes of 2048^2 in, 3 of 4096^2 out)", "resize_backward_gather_kernel<8, 8, true>", 12 * (ho * ho + PX), us,
           whole_call_us=round(us, 1))
    del gout, gin, ws
    # gradient of a 2x up-scale: a whole factor -> the band walk of resize_down.hpp with the transposed two-tap weights (round 5).  SIX planes (403 MB of upstream
    # gradient: the streaming form), so that the launch is told from the forward down-scales of the `resize` case (same kernel) by its grid
    gout = torch.rand(6, S, S, device=DEV, generator=g)
    gin = torch.empty(6, ho, ho, device=DEV)
    ws = torch.empty(max(1, lib.pbr_resize_backward_workspace_bytes(6, ho, ho, S, S) // 4), device=DEV)
    us = timed(lambda: lib.pbr_resize_bilinear_backward(gout.data_ptr(), gin.data_ptr(), 6, ho, ho, S, S, 1, ws.data_ptr(), stream))
    report("resize backward 6 x 4096^2 gradient -> 2048^2 (gradient of a 2x up-scale: 6 planes of 4096^2 in, 6 of 2048^2 out): the band walk with the transposed two-tap weights",
           "resize_down_kernel<2, 4, 2, 3, true>", 24 * (ho * ho + PX), us)
    del gout, gin, ws
    # gradient of a 1.5x up-scale (no whole factor): the two-tap transpose
    hu = S * 3 // 2
    gout = torch.rand(3, hu, hu, device=DEV, generator=g)
    gin = torch.empty(3, S, S, device=DEV)
    ws = torch.empty(max(1, lib.pbr_resize_backward_workspace_bytes(3, S, S, hu, hu) // 4), device=DEV)
    us = timed(lambda: lib.pbr_resize_bilinear_backward(gout.data_ptr(), gin.data_ptr(), 3, S, S, hu, hu, 1, ws.data_ptr(), stream))
    report("resize backward 3 x 6144^2 gradient -> 4096^2 (gradient of a 1.5x up-scale: 3 planes of 6144^2 in, 3 of 4096^2 out): the two-tap transpose", "resize_up2_backward_kernel<8, 4>",
           12 * (hu * hu + PX), us)
