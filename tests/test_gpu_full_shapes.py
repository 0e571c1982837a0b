"""BASELINE.json configs 3, 4 (per-GPU share) and 5 (per-GPU share) at their FULL shapes: one launch over the whole
batch -- the XCD-run workgroup order, 8-pixel lanes and rows = B * H arithmetic at sizes the small tests never reach --
then sampled parity: crops / row bands (first and LAST material, first and LAST rows included) against the ATen
restatement of the reference (fp32, pinned bit-equal to it) and against float64 (the plain-C oracle), under the
criterion of tests/test_gpu_parity.py.  Plus the size-independent properties: finite, in [0,1], a second launch is
bit-identical, a band evaluated on its own equals the rows of the full launch."""
import math

import numpy as np
import pytest
import torch

import c_oracle as C
import torch_oracle as O
from test_gpu_parity import parity_report

pytestmark = pytest.mark.gpu


def _maps(B, H, W, seed, dtype=torch.float32):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.rand(B, 3, H, W, device="cuda", generator=g)
    nxy = torch.rand(B, 2, H, W, device="cuda", generator=g) - 0.5
    n = torch.cat([nxy, torch.ones(B, 1, H, W, device="cuda")], 1)
    n = n / n.norm(dim=1, keepdim=True)
    r = torch.rand(B, 1, H, W, device="cuda", generator=g) * 0.95 + 0.05
    m = torch.rand(B, 1, H, W, device="cuda", generator=g)
    return [t.to(dtype) for t in (a, n, r, m)]


def _windows(B, H, W, h, w, count, seed):
    """(b, y0, x0) of `count` windows: the four corners of the batch (first/last material, first/last rows and columns)
    plus seeded random ones."""
    rng = np.random.default_rng(seed)
    fixed = [(0, 0, 0), (B - 1, H - h, W - w), (B - 1, 0, W - w), (0, H - h, 0)]
    rand = [(int(rng.integers(B)), int(rng.integers(0, H - h + 1)), int(rng.integers(0, (W - w) // 8 + 1)) * 8) for _ in range(count - len(fixed))]
    return fixed + rand


def _check_properties(F, out, maps, kw):
    assert bool(torch.isfinite(out).all()) and float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    again = F.cook_torrance(*maps, **kw)
    assert torch.equal(out, again)
    del again


@pytest.mark.parametrize("quirk", [True, False])
def test_config3_full_shape_64x2048_converted_directional(quirk):
    """B=64 2048^2, directional light, sRGB decode + metallic -> diffuse/specular conversion fused, both settings of
    the upstream specular_is_srgb quirk."""
    from pypbr_amd import functional as F
    B, H, W = 64, 2048, 2048
    maps = _maps(B, H, W, seed=3)
    view, light, inten = [0.0, 0.0, 1.0], [0.3, -0.2, 1.0], [1.0, 1.0, 1.0]
    kw = dict(view_dir=view, light=light, light_intensity=inten, light_type="directional",
              convert_to_diffuse_specular=True, specular_is_srgb=quirk)
    plan = F.plan_cook_torrance(*maps, **kw)
    assert plan.kernel_name == "ct_directional_converted_f32_f32_v4"
    out = plan.launch()
    _check_properties(F, out, maps, kw)
    worst32 = worst64 = 0.0
    n_hip = n_ref = 0
    for b, y0, x0 in _windows(B, H, W, 128, 128, 10, seed=33):
        crop = [t[b, :, y0:y0 + 128, x0:x0 + 128].cpu() for t in maps]
        got = out[b, :, y0:y0 + 128, x0:x0 + 128].cpu().numpy()
        okw = dict(view=torch.tensor(view), light=torch.tensor(light), intensity=torch.tensor(inten), light_type="directional")
        ref32 = O.cook_torrance_converted(*crop, quirk_specular_srgb=quirk, **okw).numpy()
        ref64 = O.cook_torrance_converted(*[t.double() for t in crop], quirk_specular_srgb=quirk,
                                          **{k: (v.double() if isinstance(v, torch.Tensor) else v) for k, v in okw.items()}).numpy()
        rep = parity_report(got, ref32, ref64, crop[2].numpy(), what=("cfg3", quirk, b, y0, x0))
        worst32, worst64 = max(worst32, rep["max32"]), max(worst64, rep["max64"])
        n_hip, n_ref = n_hip + rep["n_hip"], n_ref + rep["n_ref"]
    print(f"\n[cfg3 64x2048^2 converted directional quirk={quirk}] 10 crops of 128^2: max|hip-ref32| {worst32:.2e}, "
          f"max|hip-ref64| {worst64:.2e}, values > 1e-5 vs ref32: {n_hip} (reference vs its own float64: {n_ref})")


def test_config4_share_full_shape_64x1024_point():
    """Per-GPU share of config 4: B=64 1024^2, point light.  Row bands span the full width (the point-light grid)."""
    from pypbr_amd import functional as F
    B, H, W = 64, 1024, 1024
    maps = _maps(B, H, W, seed=4)
    view, light, inten = [0.0, 0.0, 1.0], [0.1, 0.1, 1.0], [1.0, 1.0, 1.0]
    kw = dict(view_dir=view, light=light, light_intensity=inten, light_type="point", light_size=1.0)
    out = F.cook_torrance(*maps, **kw)
    _check_properties(F, out, maps, kw)
    band = F.cook_torrance(*[t[B - 1:, :, H - 24:] for t in maps], y_offset=H - 24, height_total=H, **kw)
    assert torch.equal(band, out[B - 1:, :, H - 24:])
    worst32 = worst64 = 0.0
    n_hip = n_ref = 0
    for b, y0, _ in _windows(B, H, W, 16, W, 10, seed=44):
        crop = [t[b, :, y0:y0 + 16].cpu() for t in maps]
        got = out[b, :, y0:y0 + 16].cpu().numpy()
        ref32 = O.cook_torrance(*crop, None, view=torch.tensor(view), light=torch.tensor(light), intensity=torch.tensor(inten),
                                light_type="point", light_size=1.0, y_offset=y0, H_total=H).numpy()
        ref64 = C.render(*[t.numpy() for t in crop], None, view=view, lights=light, intensities=inten, light_type="point",
                         light_size=1.0, y_offset=y0, H_total=H, dtype=np.float64)
        rep = parity_report(got, ref32, ref64, crop[2].numpy(), what=("cfg4", b, y0))
        worst32, worst64 = max(worst32, rep["max32"]), max(worst64, rep["max64"])
        n_hip, n_ref = n_hip + rep["n_hip"], n_ref + rep["n_ref"]
    print(f"\n[cfg4 share 64x1024^2 point] 10 bands of 16 rows: max|hip-ref32| {worst32:.2e}, max|hip-ref64| {worst64:.2e}, "
          f"values > 1e-5 vs ref32: {n_hip} (reference vs its own float64: {n_ref})")


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.float16])
def test_config5_share_full_shape_4x4096_16_lights_fp16(out_dtype):
    """Per-GPU share of config 5: B=4 4096^2, 16 point lights on a ring, fp16 maps, fp32 accumulate.  The oracle is fed
    the exact fp32 up-casts of the fp16 maps (SURVEY.md 8c iii); an fp16 result adds its own rounding (<= 4.9e-4)."""
    from pypbr_amd import functional as F
    B, H, W = 4, 4096, 4096
    maps = _maps(B, H, W, seed=5, dtype=torch.float16)
    lights = [[math.cos(2 * math.pi * i / 16), math.sin(2 * math.pi * i / 16), 1.0] for i in range(16)]
    inten = [[1.0 / 16] * 3] * 16
    view = [0.0, 0.0, 1.0]
    kw = dict(view_dir=view, light=lights, light_intensity=inten, light_type="point", light_size=1.0, out_dtype=out_dtype)
    out = F.cook_torrance(*maps, **kw)
    assert out.dtype == out_dtype
    _check_properties(F, out, maps, kw)
    worst32 = worst64 = 0.0
    for b, y0, _ in _windows(B, H, W, 4, W, 8, seed=55):
        crop = [t[b, :, y0:y0 + 4].float().cpu() for t in maps]
        got = out[b, :, y0:y0 + 4].float().cpu().numpy()
        ref32 = O.cook_torrance_multi(*crop, None, lights=torch.tensor(lights), intensities=torch.tensor(inten), view=torch.tensor(view),
                                      light_type="point", light_size=1.0, y_offset=y0, H_total=H).numpy()
        ref64 = C.render(*[t.numpy() for t in crop], None, view=view, lights=lights, intensities=inten, light_type="point",
                         light_size=1.0, y_offset=y0, H_total=H, dtype=np.float64)
        if out_dtype == torch.float32:
            rep = parity_report(got, ref32, ref64, crop[2].numpy(), what=("cfg5", b, y0))
            worst32, worst64 = max(worst32, rep["max32"]), max(worst64, rep["max64"])
        else:       # fp16 storage of the result: half an fp16 ulp below 1 on top of the fp32 criterion
            e64 = np.abs(got.astype(np.float64) - ref64)
            worst64 = max(worst64, float(e64.max()))
            assert e64.max() <= 4.9e-4 + 2e-6, (b, y0, float(e64.max()))
    print(f"\n[cfg5 share 4x4096^2 16 lights fp16 -> {out_dtype}] 8 bands of 4 rows: max|hip-ref32| {worst32:.2e}, max|hip-ref64| {worst64:.2e}")


def test_offsets_beyond_2_to_31_elements():
    """Maximum sizes: a batch whose planes lie more than 2^31 ELEMENTS into their tensors (12 x 8192^2: the last
    material's albedo starts 2.2e9 elements in, its result 2.2e9 floats in) -- every index in the kernels must be 64-bit.
    fp16 maps to keep it at 12 GiB in, 9 GiB out; bands of the first and the LAST material against the oracles."""
    from pypbr_amd import functional as F
    B, H, W = 12, 8192, 8192
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * 2 ** 30:
        pytest.skip("needs 40 GiB of free device memory")
    g = torch.Generator(device="cuda").manual_seed(8)
    a = torch.rand(B, 3, H, W, device="cuda", generator=g, dtype=torch.float16)
    n = torch.rand(B, 3, H, W, device="cuda", generator=g, dtype=torch.float16)
    n[:, 2] += 1.0                                                    # z-dominant, un-normalised (the kernel normalises)
    n[:, :2] -= 0.5
    r = torch.rand(B, 1, H, W, device="cuda", generator=g, dtype=torch.float16) * 0.8 + 0.2
    m = torch.rand(B, 1, H, W, device="cuda", generator=g, dtype=torch.float16)
    assert a[B - 1].storage_offset() > 2 ** 31
    view, light, inten = [0.0, 0.0, 1.0], [0.1, 0.1, 1.0], [1.0, 1.0, 1.0]
    out = F.cook_torrance(a, n, r, m, view_dir=view, light=light, light_intensity=inten, light_type="point", light_size=1.0)
    assert out.shape == (B, 3, H, W) and out[B - 1].storage_offset() > 2 ** 31
    worst = 0.0
    for b, y0 in ((0, 0), (B - 1, 0), (B - 1, H - 4), (B // 2, 4097)):
        crop = [t[b, :, y0:y0 + 4].float().cpu() for t in (a, n, r, m)]
        got = out[b, :, y0:y0 + 4].cpu().numpy()
        ref64 = C.render(*[t.numpy() for t in crop], None, view=view, lights=light, intensities=inten, light_type="point",
                         light_size=1.0, y_offset=y0, H_total=H, dtype=np.float64)
        ref32 = O.cook_torrance(*crop, None, view=torch.tensor(view), light=torch.tensor(light), intensity=torch.tensor(inten),
                                light_type="point", light_size=1.0, y_offset=y0, H_total=H).numpy()
        rep = parity_report(got, ref32, ref64, crop[2].numpy(), what=("8192^2 x 12", b, y0))
        worst = max(worst, rep["max64"])
    # the tail of the last plane was written, and nothing is left unwritten in between (an empty() buffer would show garbage / NaN)
    assert bool(torch.isfinite(out[B - 1, 2, H - 1]).all()) and float(out[B - 1].min()) >= 0.0 and float(out[B - 1].max()) <= 1.0
    print(f"\n[12 x 8192^2 fp16 maps, offsets > 2^31 elements] max|hip-ref64| {worst:.2e}")


def test_streamed_backward_full_shape_2x4096_fp16():
    """The streamed backward kernel (fp16 maps, one light) at a full shape: 2 x 4096^2, i.e. 262 144 tiles over a grid of
    a few thousand waves, both materials -- every gradient plane bit-equal to the one-tile kernels', whatever the number of
    rounds; and a crop against the float64 autograd of the ATen restatement (the criterion of tests/test_gpu_backward.py)."""
    from pypbr_amd import _native as N, functional as F
    lib = N.lib()
    B, H, W = 2, 4096, 4096
    maps = _maps(B, H, W, 31, torch.float16)
    maps[2] = maps[2].clamp(min=0.3)                   # gradients come back in fp16: keep the specular peak inside its range
    kw = dict(view_dir=[0.0, 0.1, 1.0], light=[0.1, 0.1, 1.0], light_intensity=[1.0, 0.9, 0.8], light_type="point", light_size=1.0)
    wt = torch.rand(B, 3, H, W, device="cuda", generator=torch.Generator(device="cuda").manual_seed(32)) - 0.3

    def grads(rounds):
        lib.pbr_set_tuning(N.TUNE_BWD_RUN, rounds)
        leaves = [t.clone().requires_grad_() for t in maps]
        (F.cook_torrance(*leaves, **kw) * wt).sum().backward()
        return [t.grad for t in leaves]
    try:
        want = grads(0)
        for rounds in (-1, 1, 3):
            got = grads(rounds)
            for name, x, y in zip(("albedo", "normal", "roughness", "metallic"), want, got):
                assert torch.equal(x, y), (rounds, name)
            del got
    finally:
        lib.pbr_set_tuning(N.TUNE_BWD_RUN, -1)
    assert all(bool(torch.isfinite(x).all()) for x in want)
    # last rows of the last material against float64 autograd of the oracle
    b, y0, h = B - 1, H - 8, 8
    crop = [t[b, :, y0:].float().cpu().double().requires_grad_() for t in maps]
    ref = O.cook_torrance(*crop, None, view=torch.tensor(kw["view_dir"], dtype=torch.float64), light=torch.tensor(kw["light"], dtype=torch.float64),
                          intensity=torch.tensor(kw["light_intensity"], dtype=torch.float64), light_type="point", light_size=1.0,
                          y_offset=y0, H_total=H)
    (ref * wt[b, :, y0:].cpu().double()).sum().backward()
    for name, x, leaf in zip(("albedo", "normal", "roughness", "metallic"), want, crop):
        g64 = leaf.grad
        err = (x[b, :, y0:].float().cpu().double() - g64).abs()
        assert bool((err <= 1e-3 * (1e-3 + g64.abs()) + 2e-5 * (1 + g64.abs())).all()), (name, float(err.max()))      # fp16 gradient storage


@pytest.mark.parametrize("B,dtype", [(1, torch.float32), (2, torch.float16)])
def test_loss_step_full_shape_4096(B, dtype):
    """The one-kernel rendering-loss step at full size: fp32 maps (131 072 one-wave workgroups, the two-stage reduction of their partial
    sums) and 2 x 4096^2 fp16 maps (the streamed form: a few thousand persistent waves, one partial each).  The loss against
    evaluate + torch's MSE; the gradients against the backward kernel fed with that loss's upstream gradient -- the same chain rule on the
    same pixels -- over whole planes; and the streamed form against the one-tile kernels bit for bit."""
    from pypbr_amd import _native as N, functional as F
    lib = N.lib()
    H = W = 4096
    maps = _maps(B, H, W, 41, dtype)
    kw = dict(view_dir=[0.0, 0.1, 1.0], light=[0.1, 0.1, 1.0], light_intensity=[1.0, 0.9, 0.8], light_type="point", light_size=1.0)
    target = torch.rand(B, 3, H, W, device="cuda", generator=torch.Generator(device="cuda").manual_seed(42))

    def fused(stream_knob):
        lib.pbr_set_tuning(N.TUNE_MSE_STREAM, stream_knob)
        leaves = [t.clone().requires_grad_() for t in maps]
        loss = F.rendering_loss_mse(*leaves, target=target, **kw)
        assert type(loss.grad_fn).__name__ == "_MseStepFnBackward"
        loss.backward()
        return loss.detach(), [t.grad for t in leaves]
    try:
        loss1, g1 = fused(1)
        loss0, g0 = fused(0)
    finally:
        lib.pbr_set_tuning(N.TUNE_MSE_STREAM, 1)
    assert abs(float(loss1) - float(loss0)) <= 2e-6 * float(loss0)                  # another summation order of the partial sums
    for x, y in zip(g1, g0):
        assert torch.equal(x, y)                                                     # (fp32 maps: the same kernel twice -- deterministic)
    leaves = [t.clone().requires_grad_() for t in maps]
    out = F.cook_torrance(*leaves, **kw)
    ref_loss = torch.nn.functional.mse_loss(out.float(), target)
    ref_loss.backward()
    assert abs(float(loss1) - float(ref_loss)) <= 2e-6 * float(ref_loss)
    for name, x, leaf in zip(("albedo", "normal", "roughness", "metallic"), g1, leaves):
        scale = float(leaf.grad.float().abs().max())
        err = float((x.float() - leaf.grad.float()).abs().max())
        # fp32: the two paths form 2 (out - target) / N with different roundings (a few ulp); fp16: gradients of a mean over 10^8 values sit in
        # fp16's subnormal range, where one rounding is 6e-8 absolute
        assert err <= (2e-5 * scale if dtype == torch.float32 else 2e-3 * scale + 1.3e-7), (name, err, scale)
        assert bool(torch.isfinite(x).all())


@pytest.mark.parametrize("light_type", ["point", "directional"])
def test_tiled_gradients_full_shape_2048_tile2(light_type):
    """The example's material shape at full size -- 2048^2 maps under tile(2) -> a 4096^2 image (examples/example_brdf.py:11 scaled to
    BASELINE's 4K): the folded gradients from ONE kernel that walks the maps (32 768 one-wave workgroups, 4 positions per texel pair)
    against the two-kernel form (wrap-around backward over 16.8 M output pixels + pbr_fold_gradient): bit-identical under the point
    light, to fp32 rounding under the directional one (its repeats are summed before the chain rule); finite; and a window of texels
    against float64 autograd of the reference's ops through the materialised repeat of that window's rows (y_offset / H_total keep
    the point-light grid the full image's)."""
    from pypbr_amd import functional as F, _native as N
    H = W = 2048
    a, n, r, m = [t[0] for t in _maps(1, H, W, seed=77)]
    r = r.clamp(min=0.2)
    light = [0.1, 0.1, 1.0] if light_type == "point" else [0.3, -0.2, 1.0]
    kw = dict(view_dir=[0.0, 0.1, 1.0], light=light, light_intensity=[1.0, 0.9, 0.8], light_type=light_type, light_size=1.0 if light_type == "point" else None)
    gout = torch.rand(3, 2 * H, 2 * W, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)) - 0.3

    def grads(knob):
        leaves = [t.clone().requires_grad_(True) for t in (a, n, r, m)]
        try:
            N.lib().pbr_set_tuning(N.TUNE_TILE_REPEAT, knob)
            (F.cook_torrance(*leaves, tile=2, **kw) * gout).sum().backward()
        finally:
            N.lib().pbr_set_tuning(N.TUNE_TILE_REPEAT, -1)
        return [t.grad for t in leaves]
    one, two = grads(-1), grads(0)
    for name, x, y in zip(("albedo", "normal", "roughness", "metallic"), one, two):
        assert x.shape == y.shape == (x.shape[0], H, W) and bool(torch.isfinite(x).all()), name
        if light_type == "point":
            assert torch.equal(x, y), (name, float((x - y).abs().max()))
        else:
            assert (x - y).abs().max().item() <= 2e-6 * float(y.abs().max()) + 1e-9, name
    del two
    # the LAST texel rows, all columns of a 64-wide window at the right edge: float64 autograd through repeat() of those rows
    y0, x0, h, w = H - 4, W - 64, 4, 64
    crop = [t[:, y0:, x0:x0 + w].cpu().double().requires_grad_(True) for t in (a, n, r, m)]
    total = None
    for ry in range(2):
        for rx in range(2):
            # the window's repeat (ry, rx) sits at rows ry*H + y0 .., columns rx*W + x0 .. of the 4096^2 image: the oracle evaluates that row
            # band at full width (its point-light grid needs the real columns) and only the window's columns meet the upstream gradient
            cols = slice(rx * W + x0, rx * W + x0 + w)
            rows = slice(ry * H + y0, ry * H + y0 + h)

            def across(c, fill):                    # the window's texels at their columns of a full-width row band; neutral texels elsewhere
                left = torch.zeros(c.shape[0], h, cols.start, dtype=torch.float64) + fill.reshape(-1, 1, 1)
                right = torch.zeros(c.shape[0], h, 2 * W - cols.stop, dtype=torch.float64) + fill.reshape(-1, 1, 1)
                return torch.cat([left, c, right], dim=2)
            args = [across(crop[0], torch.zeros(3, dtype=torch.float64)), across(crop[1], torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)),
                    across(crop[2], torch.tensor([0.5], dtype=torch.float64)), across(crop[3], torch.zeros(1, dtype=torch.float64))]
            out = O.cook_torrance(args[0], args[1], args[2], args[3], None, view=torch.tensor(kw["view_dir"], dtype=torch.float64),
                                  light=torch.tensor(light, dtype=torch.float64), intensity=torch.tensor(kw["light_intensity"], dtype=torch.float64),
                                  light_type=light_type, light_size=kw["light_size"], y_offset=rows.start, H_total=2 * H)
            term = (out[:, :, cols] * gout[:, rows, cols].cpu().double()).sum()
            total = term if total is None else total + term
    total.backward()
    for name, x, c in zip(("albedo", "normal", "roughness", "metallic"), one, crop):
        got = x[:, y0:, x0:x0 + w].cpu().double()
        assert bool(((got - c.grad).abs() <= 2e-5 * (1 + c.grad.abs())).all()), (name, float((got - c.grad).abs().max()))
