"""Gradients of the stand-alone map ops (round 3): the reference's colour transfers (utils/functions.py:31-66), workflow
conversions (materials/metallic.py:98-108, materials/diffuse.py:128-147), resize (materials/base.py:490-504 -> F.interpolate) and
blend masks (blending/functional.py:184-193) are plain torch ops, so its autograd differentiates through them; here each has its
own backward kernel.  Checked against float64 autograd of the ATen restatement (oracle/torch_oracle.py, pinned bit-equal to the
reference) under the gradient criterion of tests/test_gpu_backward.py: |g - g64| <= 2e-5 (1 + |g64|) -- and through the material
API inside a rendering loss, as docs/source/tutorials/06_advanced.rst:73-107 uses it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

import torch_oracle as O

pytestmark = pytest.mark.gpu


def _close(g, g64, what, rtol=2e-5):
    err = (g.detach().cpu().double() - g64).abs()
    ok = err <= rtol * (1 + g64.abs())
    assert bool(ok.all()), (what, float(err.max()), float((err / (1 + g64.abs())).max()))


@pytest.mark.parametrize("binding", ["functional", "torch_op"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_colour_transfer_gradients(binding, dtype):
    from pypbr_amd import functional as F, torch_ops
    if binding == "torch_op":
        assert torch_ops.available()
    g = torch.Generator().manual_seed(1)
    # inside, at and beyond the clamp; both sides of both knees
    x = torch.cat([torch.linspace(-0.2, 1.2, 2801), torch.rand(3 * 37 * 53 - 2801, generator=g)]).reshape(3, 37, 53)
    x = x.to(dtype).float()                                                       # the values the device sees
    wt = torch.rand(3, 37, 53, generator=g) - 0.3
    for name, ref in (("srgb_to_linear", O.srgb_to_linear), ("linear_to_srgb", O.linear_to_srgb)):
        x64 = x.double().requires_grad_(True)
        (ref(x64) * wt.double()).sum().backward()
        xd = x.to(dtype).cuda().requires_grad_(True)
        fn = getattr(F, name) if binding == "functional" else getattr(torch.ops.pbr_hip, name)
        out = fn(xd)
        assert out.requires_grad and out.dtype == dtype
        (out.float() * wt.cuda()).sum().backward()
        assert xd.grad.dtype == dtype
        # away from the knees (a value within fp32 rounding of a knee may sit on the other side in float64) and, for fp16
        # gradient storage, to fp16 precision
        knee = 0.04045 if name == "srgb_to_linear" else 0.0031308
        safe = ((x - knee).abs() > 1e-6) & ((x - 1.0).abs() > 1e-6) & (x.abs() > 1e-6)
        err = (xd.grad.float().cpu().double() - x64.grad).abs()
        tol = (2e-5 if dtype == torch.float32 else 1e-3) * (1 + x64.grad.abs())
        assert bool((err <= tol)[safe].all()), (name, dtype, float((err - tol)[safe].max()))
        # sub-gradient conventions at the clamp: zero outside [0, 1]
        assert float(xd.grad.float().cpu()[(x < 0) | (x > 1)].abs().max()) == 0.0


@pytest.mark.parametrize("srgb", [True, False])
@pytest.mark.parametrize("binding", ["functional", "torch_op"])
def test_metallic_to_diffuse_specular_gradients(srgb, binding):
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(2)
    B, H, W = 2, 21, 36
    a, m = torch.rand(B, 3, H, W, generator=g), torch.rand(B, 1, H, W, generator=g)
    wd, ws = torch.rand(B, 3, H, W, generator=g) - 0.4, torch.rand(B, 3, H, W, generator=g) - 0.6
    a64, m64 = a.double().requires_grad_(True), m.double().requires_grad_(True)
    d64, s64 = O.metallic_to_diffuse_specular(O.srgb_to_linear(a64) if srgb else a64, m64)
    ((d64 * wd.double()).sum() + (s64 * ws.double()).sum()).backward()
    ad, md = a.cuda().requires_grad_(True), m.cuda().requires_grad_(True)
    fn = F.metallic_to_diffuse_specular if binding == "functional" else torch.ops.pbr_hip.metallic_to_diffuse_specular
    d, s = fn(ad, md, srgb)
    assert (d.detach().cpu().double() - d64.detach()).abs().max().item() <= 2e-6
    ((d * wd.cuda()).sum() + (s * ws.cuda()).sum()).backward()
    _close(ad.grad, a64.grad, "albedo")
    _close(md.grad, m64.grad, "metallic")
    # only ONE of the two outputs used (the other's upstream gradient is absent), only one input wanting a gradient
    a2 = a.cuda().requires_grad_(True)
    d2, _ = fn(a2, m.cuda(), srgb)
    (d2 * wd.cuda()).sum().backward()
    a64b = a.double().requires_grad_(True)
    (O.metallic_to_diffuse_specular(O.srgb_to_linear(a64b) if srgb else a64b, m.double())[0] * wd.double()).sum().backward()
    _close(a2.grad, a64b.grad, "albedo, diffuse only")


@pytest.mark.parametrize("srgb", [True, False])
@pytest.mark.parametrize("binding", ["functional", "torch_op"])
def test_diffuse_specular_to_basecolor_metallic_gradients(srgb, binding):
    """Thresholded selects (den < 1e-6, metallic >= 0.95, two clamps): compared away from the ties, where fp32 and float64 may
    stand on different sides; every branch is exercised (dead denominators, saturated metallic, clamped basecolor)."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(3)
    H, W = 40, 56
    d, s = torch.rand(3, H, W, generator=g), torch.rand(3, H, W, generator=g)
    d[:, :4] = 0.02                                   # den < eps: metallic forced to 0
    s[:, 4:8] = d[:, 4:8] * 0.9 + 0.1                 # q near / above 1
    s[:, 8:12] = 0.01                                 # num < 0: q clamps at 0
    wb, wm = torch.rand(3, H, W, generator=g) - 0.4, torch.rand(3, H, W, generator=g) - 0.5
    d64, s64 = d.double().requires_grad_(True), s.double().requires_grad_(True)
    b64, m64 = O.diffuse_specular_to_basecolor_metallic(O.srgb_to_linear(d64) if srgb else d64, s64)
    ((b64 * wb.double()).sum() + (m64 * wm.double()).sum()).backward()
    dd, sd = d.cuda().requires_grad_(True), s.cuda().requires_grad_(True)
    fn = F.diffuse_specular_to_basecolor_metallic if binding == "functional" else torch.ops.pbr_hip.diffuse_specular_to_basecolor_metallic
    b, m = fn(dd, sd, srgb)
    ((b * wb.cuda()).sum() + (m * wm.cuda()).sum()).backward()
    with torch.no_grad():
        lin = O.srgb_to_linear(d.double()) if srgb else d.double()
        den = lin - 0.04 + 1e-6
        q = (s.double() - 0.04) / (den + 1e-6)
        mm = torch.where(den < 1e-6, torch.zeros_like(q), q.clamp(0, 1))
        bc1 = torch.where(mm >= 0.95, s.double(), lin / (1 - mm + 1e-6))
        safe = ((den - 1e-6).abs() > 1e-4) & (q.abs() > 1e-4) & ((q - 1).abs() > 1e-4) & ((mm - 0.95).abs() > 1e-4) & \
               (bc1.abs() > 1e-4) & ((bc1 - 1).abs() > 1e-4) & (den.abs() > 0.02)
        assert 0.5 < float(safe.float().mean()) < 1.0
        for branch in (den < 1e-6, mm >= 0.95, q > 1, q < 0, bc1 > 1):
            assert bool((branch & safe).any())
    for name, got, want in (("diffuse", dd.grad, d64.grad), ("specular", sd.grad, s64.grad)):
        err = (got.cpu().double() - want).abs()
        tol = 5e-5 * (1 + want.abs())              # two chained divisions: the quotient's rounding enters squared terms
        assert bool((err <= tol)[safe].all()), (name, float((err - tol)[safe].max()))


@pytest.mark.parametrize("size,antialias", [((20, 31), True), ((80, 97), True), ((20, 31), False), ((37, 53), True), ((9, 120), True),
                                            ((100, 13), False), (24, True)])
def test_resize_gradient_is_the_transposed_tap_matrix(size, antialias):
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 3, 37, 53, generator=g)
    xd = x.cuda().requires_grad_(True)
    out = F.resize(xd, size, antialias=antialias)
    ho, wo = out.shape[-2:]
    wt = torch.rand(2, 3, ho, wo, generator=g) - 0.5
    (out * wt.cuda()).sum().backward()
    x64 = x.double().requires_grad_(True)
    ref = TF.interpolate(x64, size=(ho, wo), mode="bilinear", align_corners=False, antialias=antialias)
    # tap positions are formed in fp32 (scale * (i + 0.5), as ATen does for float maps): a few 1e-6 from the float64 taps
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() <= 1e-5
    (ref * wt.double()).sum().backward()
    _close(xd.grad, x64.grad, ("resize vs float64 autograd", size, antialias), rtol=2e-5)
    x32 = x.clone().requires_grad_(True)                                 # ATen's own fp32 run: the same fp32 taps
    (TF.interpolate(x32, size=(ho, wo), mode="bilinear", align_corners=False, antialias=antialias) * wt).sum().backward()
    _close(xd.grad, x32.grad.double(), ("resize vs ATen fp32 autograd", size, antialias), rtol=5e-6)
    # linear map: <resize(x), w> == <x, resize^T(w)> with the kernel's own forward, to fp32 accuracy (adjoint test)
    lhs = float((out.detach().double() * wt.cuda().double()).sum())
    rhs = float((x.cuda().double() * xd.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


def test_resize_gradient_full_size_adjoint():
    """4096^2 -> 1365^2 (non-integer scale, 9 taps) and 1024^2 -> 4096^2 (up-scale): the adjoint identity at full size."""
    from pypbr_amd import functional as F
    g = torch.Generator(device="cuda").manual_seed(5)
    for (h, w), size in (((4096, 4096), (1365, 1365)), ((1024, 1024), (4096, 4096))):
        x = torch.rand(1, h, w, device="cuda", generator=g, requires_grad=True)
        out = F.resize(x, size)
        wt = torch.rand(out.shape, device="cuda", generator=g) - 0.5
        (out * wt).sum().backward()
        lhs, rhs = float((out.detach().double() * wt.double()).sum()), float((x.detach().double() * x.grad.double()).sum())
        assert abs(lhs - rhs) <= 2e-5 * max(1.0, abs(lhs)), (size, lhs, rhs)
        assert bool(torch.isfinite(x.grad).all())


@pytest.mark.parametrize("shape,size,antialias", [((37, 53), (20, 31), True), ((64, 256), (32, 128), True), ((200, 260), (67, 90), True),
                                                  ((40, 64), (100, 160), False), ((130, 131), (129, 64), True), ((256, 512), (128, 256), True),
                                                  ((40, 64), (9, 12), True), ((200, 260), (20, 26), True), ((24, 40), (60, 13), True)])
def test_resize_gradient_forms_agree_with_float64_autograd(shape, size, antialias):
    """pbr_resize_bilinear_backward picks its form by shape (ABI 7: the A/B knobs of round 4 are gone, profiles/EXPERIMENTS.md): the
    register gather over the transposed tap tables (banded and looked-up rows), the two-tap transpose for up-scales, the LDS strip
    kernel (outputs narrower than 16 columns), the generic two passes (more than 16 taps).  Every form against float64 autograd of
    F.interpolate, and -- the up-scale forms -- the register kernel against the strip kernel (PBR_TUNE_RESIZE_UP2 = 0)."""
    from pypbr_amd import _native as N
    lib = N.lib()
    g = torch.Generator().manual_seed(14)
    (h, w), (ho, wo) = shape, size
    gout = (torch.rand(3, ho, wo, generator=g) - 0.5).cuda()
    ws = torch.empty(max(1, lib.pbr_resize_backward_workspace_bytes(3, h, w, ho, wo) // 4), device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    got = {}
    try:
        for up2 in (1, 0):
            lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, up2)
            gin = torch.full((3, h, w), float("nan"), device="cuda")
            N.check(lib.pbr_resize_bilinear_backward(gout.data_ptr(), gin.data_ptr(), 3, h, w, ho, wo, int(antialias), ws.data_ptr(), stream))
            got[up2] = gin
    finally:
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 1)
    x = torch.zeros(1, 3, h, w, dtype=torch.float64, requires_grad=True)
    (TF.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=False, antialias=antialias)[0] * gout.cpu().double()).sum().backward()
    for up2, gin in got.items():
        assert bool(torch.isfinite(gin).all()), up2
        assert (gin.cpu().double() - x.grad[0]).abs().max().item() <= 2e-5, up2
    assert (got[1] - got[0]).abs().max().item() <= 2e-6 * max(1.0, float(got[0].abs().max()))


@pytest.mark.parametrize("shape,size", [((64, 96), (128, 192)), ((37, 53), (80, 97)), ((50, 70), (50, 70)), ((33, 130), (97, 131)), ((40, 44), (57, 128)),
                                        ((9, 16), (10, 16)), ((128, 256), (300, 700)), ((5, 4), (11, 16)), ((24, 250), (31, 251)),
                                        ((2, 8), (4, 16)), ((13, 260), (52, 1040)), ((16, 8), (128, 64)), ((301, 512), (602, 1024)), ((64, 96), (256, 384)),
                                        ((37, 12), (111, 36)), ((20, 64), (100, 320)), ((9, 40), (54, 240)), ((6, 8), (42, 56)), ((5, 16), (80, 256))])
def test_gradient_of_an_upscale_in_registers(shape, size):
    """pbr_resize_bilinear_backward for up-scales (round 4: resize_up2_backward_kernel, the register-only transpose of the two-tap forward):
    against float64 autograd of F.interpolate, and within rounding of the table-driven strip kernel it replaces on these shapes; exact 2x,
    ragged and unaligned widths, 1:1, widths whose last lane is partial, a 3x up-scale across (16 upstream columns per lane).  Whole factors 2 | 4 | 8
    on both axes (round 5): the band walk of csrc/resize_down.hpp with the transposed two-tap weights -- the smallest shape, ragged bands, idle lanes
    (3 x, 5 x, 6 x, 7 x stay with the two-tap transpose: their fp32 tap positions are not periodic)."""
    from pypbr_amd import _native as N
    lib = N.lib()
    g = torch.Generator().manual_seed(15)
    (h, w), (ho, wo) = shape, size
    gout = (torch.rand(3, ho, wo, generator=g) - 0.5).cuda()
    ws = torch.empty(max(1, lib.pbr_resize_backward_workspace_bytes(3, h, w, ho, wo) // 4), device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    got = {}
    try:
        for up2 in (1, 0):
            lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, up2)
            gin = torch.full((3, h, w), float("nan"), device="cuda")
            N.check(lib.pbr_resize_bilinear_backward(gout.data_ptr(), gin.data_ptr(), 3, h, w, ho, wo, 1, ws.data_ptr(), stream))
            got[up2] = gin
    finally:
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 1)
    x = torch.zeros(1, 3, h, w, dtype=torch.float64, requires_grad=True)
    (TF.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=False, antialias=True)[0] * gout.cpu().double()).sum().backward()
    assert bool(torch.isfinite(got[1]).all())
    assert (got[1].cpu().double() - x.grad[0]).abs().max().item() <= 3e-5      # float64 tap positions against float32 ones: the bound of the strip kernel's test
    assert (got[1] - got[0]).abs().max().item() <= 2e-6 * max(1.0, float(got[0].abs().max()))      # (a gradient element of an S x up-scale sums S^2 upstream values)
    # the whole of an upstream gradient of ones comes back: every output's weights sum to one
    ones = torch.ones(3, ho, wo, device="cuda")
    gin = torch.empty(3, h, w, device="cuda")
    N.check(lib.pbr_resize_bilinear_backward(ones.data_ptr(), gin.data_ptr(), 3, h, w, ho, wo, 1, ws.data_ptr(), stream))
    assert abs(float(gin.double().sum()) - 3.0 * ho * wo) <= 1e-3 * ho * wo / 1000 + 1e-2


def test_sigmoid_mask_gradient_and_height_blend_through_the_mask():
    from pypbr_amd import blending as B
    g = torch.Generator().manual_seed(6)
    h1, h2 = torch.rand(1, 30, 44, generator=g), torch.rand(1, 30, 44, generator=g)
    wt = torch.rand(1, 30, 44, generator=g) - 0.5
    a, b = h1.cuda().requires_grad_(True), h2.cuda().requires_grad_(True)
    mask = B.sigmoid_mask(a, b, 0.1, -0.2)
    (mask * wt.cuda()).sum().backward()
    a64, b64 = h1.double().requires_grad_(True), h2.double().requires_grad_(True)
    (torch.sigmoid((a64 + (-0.2) - b64) / (0.1 + 1e-6)) * wt.double()).sum().backward()
    _close(a.grad, a64.grad, "height 1")
    _close(b.grad, b64.grad, "height 2")


def test_rendering_loss_through_the_material_conversions():
    """material.to_linear() / to_diffuse_specular_material() / resize() inside a rendering loss (verdict r2, item 3): gradients
    of the predicted maps against float64 autograd of the same chain of oracle ops."""
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    g = torch.Generator().manual_seed(7)
    H, W = 48, 64
    a0, m0 = torch.rand(3, H, W, generator=g), torch.rand(1, H, W, generator=g)
    r0 = torch.rand(1, H, W, generator=g) * 0.6 + 0.3
    n0 = TF.normalize(torch.cat([torch.rand(2, H, W, generator=g) - 0.5, torch.ones(1, H, W)]), dim=0)
    target = torch.rand(3, 24, 32, generator=g)
    view, light, inten = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.3, -0.2, 1.0]), torch.tensor([1.0, 0.9, 0.8])

    pred = {k: t.clone().cuda().requires_grad_(True) for k, t in (("albedo", a0), ("roughness", r0), ("metallic", m0))}
    mat = BasecolorMetallicMaterial(albedo=pred["albedo"], normal=None, roughness=pred["roughness"], metallic=pred["metallic"],
                                    device=torch.device("cuda"))
    mat._maps["normal"] = n0.cuda()
    mat.to_linear()                                                  # srgb_to_linear, in place on the material
    assert mat.albedo_is_srgb is False and mat._maps["albedo"].requires_grad
    conv = mat.to_diffuse_specular_material()                        # metallic -> diffuse / specular
    conv.specular_is_srgb = False
    conv.resize((24, 32))                                            # antialiased down-scale of every map
    assert conv._maps["albedo"].requires_grad and conv._maps["specular"].requires_grad and conv._maps["roughness"].requires_grad
    out = CookTorranceBRDF("directional")(conv, view, light, inten)
    loss = TF.mse_loss(out, target.cuda())
    loss.backward()

    a64, m64, r64 = a0.double().requires_grad_(True), m0.double().requires_grad_(True), r0.double().requires_grad_(True)
    d64, s64 = O.metallic_to_diffuse_specular(O.srgb_to_linear(a64), m64)
    rs = lambda t: TF.interpolate(t[None], size=(24, 32), mode="bilinear", align_corners=False, antialias=True)[0]
    # the material stores the resized normal as it comes out of the resize (no re-decode: resize acts on _maps directly)
    ref = O.cook_torrance(rs(d64), rs(n0.double()), rs(r64), None, rs(s64), view=view.double(), light=light.double(), intensity=inten.double(),
                          light_type="directional", albedo_is_srgb=False, specular_is_srgb=False)
    loss64 = TF.mse_loss(ref, target.double())
    loss64.backward()
    assert abs(loss.item() - loss64.item()) <= 1e-6
    for name, got, want in (("albedo", pred["albedo"].grad, a64.grad), ("metallic", pred["metallic"].grad, m64.grad),
                            ("roughness", pred["roughness"].grad, r64.grad)):
        err = (got.cpu().double() - want).abs()
        scale = float(want.abs().max())
        assert float(err.max()) <= 2e-5 * (scale + 1e-6) + 2e-9, (name, float(err.max()), scale)


def test_to_basecolor_metallic_and_to_srgb_keep_the_graph():
    from pypbr_amd.materials import DiffuseSpecularMaterial
    g = torch.Generator().manual_seed(8)
    d = (torch.rand(3, 16, 24, generator=g) * 0.8 + 0.1).cuda().requires_grad_(True)
    s = (torch.rand(3, 16, 24, generator=g) * 0.5).cuda().requires_grad_(True)
    mat = DiffuseSpecularMaterial(albedo=d, roughness=torch.rand(1, 16, 24, generator=g).cuda(), specular=s, albedo_is_srgb=False,
                                  specular_is_srgb=False, device=torch.device("cuda"))
    back = mat.to_basecolor_metallic_material()
    back.to_srgb()
    (back._maps["albedo"].sum() + back._maps["metallic"].sum()).backward()
    assert d.grad is not None and s.grad is not None and bool(torch.isfinite(d.grad).all()) and float(s.grad.abs().sum()) > 0


def test_resize_and_its_gradient_on_random_shapes():
    """tools/resize_fuzz.py: 150 random shapes (extents 1 ... 513, 1-4 planes, views off a 16-byte boundary, with and without
    antialiasing) through every resize path and its gradient, against ATen."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("resize_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "resize_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.run(150, 11, verbose=False)


def test_map_operations_on_random_shapes_dtypes_and_alignments():
    """tools/map_ops_fuzz.py: 60 random cases (extents 1 ... 300, unbatched and batched, fp32 / fp16, views that start 1-3 elements off an
    allocation's start) through the colour transfers, both workflow conversions, the normal decode, blends and masks, with gradients,
    against the ATen restatements of the reference."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("map_ops_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "map_ops_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.run(60, 13, verbose=False)
