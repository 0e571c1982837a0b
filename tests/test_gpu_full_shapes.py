"""Full shapes beyond the five BASELINE.json configurations (those: tests/test_gpu_00_baseline_configs.py): offsets past 2^31
elements, the streamed backward, the loss step and tiled gradients at sizes the small tests never reach -- with
sampled parity: crops / row bands (first and LAST material, first and LAST rows included) against the ATen
restatement of the reference (fp32, pinned bit-equal to it) and against float64 (the plain-C oracle), under the
criterion of tests/test_gpu_parity.py.  Plus the size-independent properties: finite, in [0,1], a second launch is
bit-identical, a band evaluated on its own equals the rows of the full launch."""

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
        # equal to rounding: the one-kernel form sums adjoints (point) / upstream values (directional) before the light-independent tail
        assert (x - y).abs().max().item() <= 4e-6 * float(y.abs().max()) + 1e-9, (name, float((x - y).abs().max()), float(y.abs().max()))
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
