"""GPU parity of the backward kernel: gradients of sum(out * W) w.r.t. the maps
against the REAL reference's autograd gradients (tests/golden/grad.npz, fp32 and float64 runs) and
against autograd through the ATen oracle for the build extensions (batch, several lights).

Tolerance: gradients are not bounded by 1 like the colours, so the bound is relative to the
gradient scale: |g - g64| <= 2e-5 * (1 + |g64|) against the float64 reference, and the fp32
reference must lie in the same band (it does, by construction of the fixture: roughness >= 0.2)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LIGHTS = {"pt1": ("point", [0.1, 0.1, 1.0], 1.0), "dir": ("directional", [0.3, -0.2, 1.0], None)}


def _leaf(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda().requires_grad_(True)


@pytest.mark.parametrize("kind", ["metallic", "specular"])
@pytest.mark.parametrize("lk", ["pt1", "dir"])
@pytest.mark.parametrize("srgb", [True, False])
def test_gradients_match_reference_autograd(kind, lk, srgb, golden):
    from pypbr_amd import functional as F
    z = golden("grad")
    a, n, r = _leaf(z["in_albedo"]), _leaf(z["in_normal"]), _leaf(z["in_roughness"])
    m = _leaf(z["in_metallic"]) if kind == "metallic" else None
    s = _leaf(z["in_specular"]) if kind == "specular" else None
    w = torch.from_numpy(z["in_weight"]).cuda()
    ltype, lvec, lsize = LIGHTS[lk]
    out = F.cook_torrance(a, n, r, m, s, view_dir=[0, 0, 1], light=lvec, light_intensity=[1, 1, 1], light_type=ltype,
                          light_size=lsize, return_srgb=srgb)
    assert out.requires_grad
    (out * w).sum().backward()
    tag = f"{kind}_{lk}_{'srgb' if srgb else 'lin'}"
    got = {"albedo": a.grad, "normal": n.grad, "roughness": r.grad, "metallic" if m is not None else "specular": (m if m is not None else s).grad}
    for name, g in got.items():
        g = g.cpu().numpy()
        ref32, ref64 = z[f"grad_{tag}_{name}"], z[f"g64_{tag}_{name}"]
        assert g.shape == ref32.shape and np.isfinite(g).all()
        band = 2e-5 * (1.0 + np.abs(ref64))
        assert (np.abs(g.astype(np.float64) - ref64) <= band).all(), (tag, name, float(np.abs(g - ref64).max()))
        # and against the reference's own fp32 gradient, allowing its own rounding envelope
        env = np.abs(ref32.astype(np.float64) - ref64)
        assert (np.abs(g - ref32) <= env + band).all(), (tag, name, float((np.abs(g - ref32) - env).max()))


def test_backward_of_batched_multilight_against_oracle_autograd():
    import torch_oracle as O
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(99)
    B, H, W = 2, 20, 36
    a = torch.rand(B, 3, H, W, generator=g)
    n = torch.cat([(torch.rand(B, 2, H, W, generator=g) - 0.5), torch.ones(B, 1, H, W)], 1)
    r = torch.rand(1, 1, H, W, generator=g) * 0.7 + 0.3                 # ONE roughness map shared by the batch
    m = torch.rand(B, 1, H, W, generator=g)
    wt = torch.rand(B, 3, H, W, generator=g) - 0.4
    lights = torch.tensor([[0.3, 0.2, 0.8], [-0.4, 0.1, 0.6], [0.0, -0.5, 1.0]])
    inten = torch.tensor([[0.6, 0.5, 0.4], [0.3, 0.3, 0.5], [0.4, 0.4, 0.4]])
    # float64 oracle autograd = ground truth for the build extension
    leaves = [t.double().requires_grad_(True) for t in (a, n, r, m)]
    ref = O.cook_torrance_batched(leaves[0], leaves[1], leaves[2].expand(B, 1, H, W), leaves[3], None, lights=lights.double(),
                                  intensities=inten.double(), view=torch.tensor([0.0, 0.1, 1.0], dtype=torch.float64),
                                  light_type="point", light_size=1.5)
    (ref * wt.double()).sum().backward()
    dev = [t.clone().cuda().requires_grad_(True) for t in (a, n, r, m)]
    out = F.cook_torrance(dev[0], dev[1], dev[2], dev[3], view_dir=[0.0, 0.1, 1.0], light=lights, light_intensity=inten,
                          light_type="point", light_size=1.5)
    assert (out.detach().cpu() - ref.detach().float()).abs().max().item() <= 1e-5
    (out * wt.cuda()).sum().backward()
    for name, d, l in zip(("albedo", "normal", "roughness", "metallic"), dev, leaves):
        assert d.grad.shape == d.shape
        err = (d.grad.cpu().double() - l.grad).abs()
        assert (err <= 2e-5 * (1 + l.grad.abs())).all(), (name, float(err.max()))


def test_rendering_loss_step_through_the_module():
    """docs/source/tutorials/06_advanced.rst:73-107: MSE between two renders, gradient reaches the maps."""
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    g = torch.Generator().manual_seed(3)
    H = W = 32
    pred = {k: torch.rand(c, H, W, generator=g).cuda().requires_grad_(True) for k, c in (("albedo", 3), ("roughness", 1), ("metallic", 1))}
    normal = torch.cat([torch.zeros(2, H, W), torch.ones(1, H, W)], 0).cuda()
    mat = BasecolorMetallicMaterial(albedo=pred["albedo"], normal=None, roughness=pred["roughness"], metallic=pred["metallic"],
                                    device=torch.device("cuda"))
    mat._maps["normal"] = normal
    target = torch.rand(3, H, W, generator=g).cuda()
    brdf = CookTorranceBRDF("point")
    out = brdf(mat, torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0]), 1.0)
    loss = torch.nn.functional.mse_loss(out, target)
    loss.backward()
    for k, t in pred.items():
        assert t.grad is not None and bool(torch.isfinite(t.grad).all()) and float(t.grad.abs().sum()) > 0, k
    with pytest.raises(NotImplementedError):
        from pypbr_amd import functional as F
        F.cook_torrance(pred["albedo"].half(), normal.half(), pred["roughness"].half(), pred["metallic"].half(),
                        view_dir=[0, 0, 1], light=[0, 0, 1], light_intensity=[1, 1, 1], out_dtype=torch.float16)   # fp16 RESULT: forward only


@pytest.mark.parametrize("light_type,light,size,lights", [("point", [0.1, 0.1, 1.0], 1.0, 1), ("directional", [0.3, -0.2, 1.0], None, 1),
                                                          ("point", [[0.1, 0.1, 1.0], [-0.3, 0.2, 0.8]], 1.5, 2)])
def test_gradients_of_fp16_maps(light_type, light, size, lights):
    """fp16 map storage (BASELINE config 5): the backward kernel reads fp16 maps and returns fp16 gradients; same arithmetic
    as for the fp32 up-casts of those maps, so the two agree to fp16 rounding of the gradient."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(61)
    H, W = 20, 40
    a = torch.rand(3, H, W, generator=g).half()
    n = torch.cat([(torch.rand(2, H, W, generator=g) - 0.5), torch.ones(1, H, W)], 0).half()
    r = (torch.rand(1, H, W, generator=g) * 0.7 + 0.3).half()
    m = torch.rand(1, H, W, generator=g).half()
    wt = (torch.rand(3, H, W, generator=g) - 0.3).cuda()
    kw = dict(view_dir=[0, 0.1, 1], light=light, light_intensity=[[0.8, 0.7, 0.6]] * lights, light_type=light_type, light_size=size)
    h16 = [t.clone().cuda().requires_grad_(True) for t in (a, n, r, m)]
    out16 = F.cook_torrance(*h16, **kw)
    assert out16.dtype == torch.float32
    (out16 * wt).sum().backward()
    f32 = [t.float().cuda().requires_grad_(True) for t in (a, n, r, m)]
    out32 = F.cook_torrance(*f32, **kw)
    assert (out16 - out32).abs().max().item() <= 2e-6          # packed two-pixel body (fp16 maps) vs the scalar one
    (out32 * wt).sum().backward()
    for name, x, y in zip(("albedo", "normal", "roughness", "metallic"), h16, f32):
        assert x.grad.dtype == torch.float16 and x.grad.shape == x.shape
        assert bool(torch.isfinite(x.grad).all()), name
        err = (x.grad.float() - y.grad).abs()
        assert (err <= 1e-3 * (1e-3 + y.grad.abs())).all(), (name, float(err.max()))      # fp16: 2^-11 relative, denormal floor


@pytest.mark.parametrize("quirk", [True, False])
@pytest.mark.parametrize("light_type,light,size", [("point", [0.1, 0.1, 1.0], 1.0), ("directional", [0.3, -0.2, 1.0], None)])
def test_backward_through_the_fused_conversion(quirk, light_type, light, size):
    """convert_to_diffuse_specular=True: gradients w.r.t. the METALLIC-workflow maps through
    to_diffuse_specular_material (metallic.py:98-108) and the render, both settings of the F6 flag,
    against float64 autograd through the ATen oracle."""
    import torch_oracle as O
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(17)
    H, W = 20, 28
    a = torch.rand(3, H, W, generator=g)
    n = torch.cat([(torch.rand(2, H, W, generator=g) - 0.5) * 1.2, torch.ones(1, H, W)], 0)
    r = torch.rand(1, H, W, generator=g) * 0.7 + 0.3
    m = torch.rand(1, H, W, generator=g)
    wt = torch.rand(3, H, W, generator=g) - 0.3
    leaves = [t.double().requires_grad_(True) for t in (a, n, r, m)]
    ref = O.cook_torrance_converted(*leaves, quirk_specular_srgb=quirk, view=torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64),
                                    light=torch.tensor(light, dtype=torch.float64), intensity=torch.tensor([1.0, 0.9, 0.8], dtype=torch.float64),
                                    light_type=light_type, light_size=size)
    (ref * wt.double()).sum().backward()
    dev = [t.clone().cuda().requires_grad_(True) for t in (a, n, r, m)]
    out = F.cook_torrance(*dev, view_dir=[0, 0, 1], light=light, light_intensity=[1.0, 0.9, 0.8], light_type=light_type,
                          light_size=size, convert_to_diffuse_specular=True, specular_is_srgb=quirk)
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() <= 1e-5
    (out * wt.cuda()).sum().backward()
    for name, d, l in zip(("albedo", "normal", "roughness", "metallic"), dev, leaves):
        err = (d.grad.cpu().double() - l.grad).abs()
        assert (err <= 2e-5 * (1 + l.grad.abs())).all(), (name, float(err.max()))


def test_in_place_edit_between_forward_and_backward_is_detected():
    """The backward kernel re-reads the maps, so autograd's version check must guard them."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(5)
    a = torch.rand(3, 16, 16, generator=g).cuda().requires_grad_(True)
    n = torch.cat([torch.zeros(2, 16, 16), torch.ones(1, 16, 16)], 0).cuda()
    r = (torch.rand(1, 16, 16, generator=g) * 0.5 + 0.4).cuda()
    m = torch.rand(1, 16, 16, generator=g).cuda()
    out = F.cook_torrance(a, n, r, m, view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_size=1.0)
    r.mul_(0.5)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out.sum().backward()


@pytest.mark.parametrize("kind", ["signed3", "unit3", "xy2"])
def test_normal_decode_is_differentiable(kind):
    """MaterialBase._process_normal_map (base.py:191-242) keeps the gradient of a predicted normal map: pbr_decode_normal_backward
    against torch autograd through the reference's formulas in float64."""
    import torch.nn.functional as TF
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed({"signed3": 1, "unit3": 2, "xy2": 3}[kind])
    H, W = 12, 20
    if kind == "signed3":
        x = torch.rand(3, H, W, generator=g) * 2 - 1
    elif kind == "unit3":
        x = torch.rand(3, H, W, generator=g)
    else:
        x = torch.rand(2, H, W, generator=g)
        x[:, 0, :4] = 0.98                                           # 1 - x^2 - y^2 below the 1e-6 clamp: zero z-gradient there
    wt = torch.rand(3, H, W, generator=g) - 0.5

    def reference(t):                                                # the reference's ops, verbatim semantics
        if t.shape[0] == 2:
            t = t * 2 - 1
            z = torch.sqrt(torch.clamp(1.0 - (t[0:1] ** 2 + t[1:2] ** 2), min=1e-6))
            return TF.normalize(torch.cat([t[0:1], t[1:2], z], 0), dim=0)
        if t.min() < 0:
            return t
        return TF.normalize(t * 2.0 - 1.0, dim=0)
    x64 = x.double().requires_grad_(True)
    (reference(x64) * wt.double()).sum().backward()
    xd = x.clone().cuda().requires_grad_(True)
    out = F.decode_normal(xd)
    assert out.requires_grad and (out.detach().cpu().double() - reference(x.double())).abs().max().item() <= 2e-6
    (out * wt.cuda()).sum().backward()
    err = (xd.grad.cpu().double() - x64.grad).abs()
    assert (err <= 2e-5 * (1 + x64.grad.abs())).all(), float(err.max())
    # through the material: a predicted normal assigned in the constructor reaches the rendering loss
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    pred = x.clone().cuda().requires_grad_(True)
    mat = BasecolorMetallicMaterial(albedo=torch.rand(3, H, W, generator=g).cuda(), normal=pred, roughness=(torch.rand(1, H, W, generator=g) * 0.6 + 0.3).cuda(),
                                    metallic=torch.rand(1, H, W, generator=g).cuda(), device=torch.device("cuda"))
    loss = CookTorranceBRDF("point")(mat, torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0]), 1.0).mean()
    loss.backward()
    assert pred.grad is not None and bool(torch.isfinite(pred.grad).all()) and float(pred.grad.abs().sum()) > 0
    # (round 3) the stand-alone map ops are differentiable too: tests/test_gpu_map_op_gradients.py
    assert F.srgb_to_linear(torch.rand(3, 4, 4, device="cuda", requires_grad=True)).requires_grad


def _param_case(light_type, n_lights, workflow, seed):
    g = torch.Generator().manual_seed(seed)
    B, H, W = 2, 24, 40
    a = torch.rand(B, 3, H, W, generator=g)
    n = torch.cat([(torch.rand(B, 2, H, W, generator=g) - 0.5), torch.ones(B, 1, H, W)], 1)
    r = torch.rand(B, 1, H, W, generator=g) * 0.7 + 0.3
    m = torch.rand(B, 1, H, W, generator=g) if workflow != "specular" else None
    s = torch.rand(B, 3, H, W, generator=g) if workflow == "specular" else None
    wt = torch.rand(B, 3, H, W, generator=g) - 0.4
    base = torch.tensor([[0.3, 0.2, 0.8], [-0.4, 0.1, 0.6], [0.0, -0.5, 1.0]])[:n_lights]
    lights = base if light_type == "point" else base * 1.7            # directional: not unit length on purpose
    inten = torch.tensor([[0.6, 0.5, 0.4], [0.3, 0.3, 0.5], [0.4, 0.4, 0.4]])[:n_lights]
    view = torch.tensor([0.05, 0.1, 0.9])                             # not unit length either (F.normalize, :95)
    return a, n, r, m, s, wt, lights, inten, view


@pytest.mark.parametrize("light_type", ["point", "directional"])
@pytest.mark.parametrize("n_lights", [1, 3])
@pytest.mark.parametrize("workflow", ["metallic", "specular"])
def test_gradients_of_view_light_and_intensity(light_type, n_lights, workflow):
    """The reference's forward is plain torch ops on view_dir / light_dir_or_position / light_intensity
    (cooktorrance.py:95-96, :126-140), so its autograd reaches them.  Ground truth: float64 autograd through the pinned
    ATen oracle.  The kernel sums per-pixel fp32 adjoints (fp64 across workgroups), so the bound is relative to the
    gradient's own scale."""
    import torch_oracle as O
    from pypbr_amd import functional as F
    a, n, r, m, s, wt, lights, inten, view = _param_case(light_type, n_lights, workflow, seed=7 + n_lights)
    size = 1.5 if light_type == "point" else None
    # float64 oracle autograd
    P = [t.double().requires_grad_(True) for t in (view, lights, inten)]
    M = [None if t is None else t.double().requires_grad_(True) for t in (a, n, r, m, s)]
    if n_lights == 1:
        ref = O.cook_torrance_batched(*M, view=P[0], light=P[1][0], intensity=P[2][0], light_type=light_type, light_size=size)
    else:
        ref = O.cook_torrance_batched(*M, lights=P[1], intensities=P[2], view=P[0], light_type=light_type, light_size=size)
    (ref * wt.double()).sum().backward()
    # HIP: parameters as device tensors that require grad, maps too (both kinds of gradient from one backward launch)
    p = [t.clone().cuda().requires_grad_(True) for t in (view, lights, inten)]
    d = [None if t is None else t.clone().cuda().requires_grad_(True) for t in (a, n, r, m, s)]
    out = F.cook_torrance(*d, view_dir=p[0], light=p[1] if n_lights > 1 else p[1][0], light_intensity=p[2] if n_lights > 1 else p[2][0],
                          light_type=light_type, light_size=size)
    assert (out.detach().cpu() - ref.detach().float()).abs().max().item() <= 1e-5
    (out * wt.cuda()).sum().backward()
    for name, got, want in zip(("view_dir", "light", "light_intensity"), p, P):
        assert got.grad is not None and got.grad.shape == got.shape and got.grad.device == got.device, name
        err = (got.grad.cpu().double() - want.grad).abs()
        scale = want.grad.abs().max().item()
        assert (err <= 2e-5 * (1.0 + scale)).all(), (name, err.max().item(), scale, got.grad.cpu(), want.grad)
    for name, got, want in zip(("albedo", "normal", "roughness", "metallic", "specular"), d, M):
        if got is not None:
            err = (got.grad.cpu().double() - want.grad).abs()
            assert (err <= 2e-5 * (1 + want.grad.abs())).all(), (name, float(err.max()))


def test_gradient_of_a_light_position_only_and_host_tensors():
    """Optimising a light against a photograph: only the light requires grad (no map gradient buffers at all); CPU
    parameter tensors get CPU gradients; an intensity given once for several lights owns the sum; ragged width ->
    the 1-pixel-per-lane PGRAD kernel with partially filled workgroups."""
    import torch_oracle as O
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(5)
    H, W = 19, 37
    a, n = torch.rand(3, H, W, generator=g), torch.cat([(torch.rand(2, H, W, generator=g) - 0.5), torch.ones(1, H, W)], 0)
    r, m = torch.rand(1, H, W, generator=g) * 0.7 + 0.3, torch.rand(1, H, W, generator=g)
    wt = torch.rand(3, H, W, generator=g) - 0.4
    lights = torch.tensor([[0.3, 0.2, 0.8], [-0.4, 0.1, 0.6]])
    inten = torch.tensor([0.5, 0.4, 0.3])
    view = torch.tensor([0.0, 0.0, 1.0])
    L64, I64 = lights.double().requires_grad_(True), inten.double().requires_grad_(True)
    ref = O.cook_torrance_multi(a.double(), n.double(), r.double(), m.double(), None, lights=L64, intensities=I64.expand(2, 3),
                                view=view.double(), light_type="point", light_size=1.0)
    (ref * wt.double()).sum().backward()
    L, I = lights.clone().requires_grad_(True), inten.clone().requires_grad_(True)              # CPU leaves
    out = F.cook_torrance(a.cuda(), n.cuda(), r.cuda(), m.cuda(), view_dir=view, light=L, light_intensity=I, light_type="point", light_size=1.0)
    (out * wt.cuda()).sum().backward()
    assert L.grad.device.type == "cpu" and L.grad.shape == (2, 3) and I.grad.shape == (3,)
    for got, want in ((L.grad, L64.grad), (I.grad, I64.grad)):
        err = (got.double() - want).abs()
        assert (err <= 2e-5 * (1.0 + want.abs().max())).all(), (got, want)
    # determinism: the per-workgroup sums are added in a fixed order
    L2 = lights.clone().requires_grad_(True)
    out2 = F.cook_torrance(a.cuda(), n.cuda(), r.cuda(), m.cuda(), view_dir=view, light=L2, light_intensity=inten, light_type="point", light_size=1.0)
    (out2 * wt.cuda()).sum().backward()
    assert torch.equal(L2.grad, L.grad)


def test_light_gradients_with_fused_tile_and_fp16_maps():
    """The light / view adjoints compose with the fused tile() (wrap-around addressing: every output pixel contributes,
    whatever texel it reads) and with fp16 map storage.  Ground truth: float64 oracle autograd on the materialised repeat
    / on the exact fp32 up-casts of the fp16 maps."""
    import torch_oracle as O
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(17)
    h, w = 12, 20
    a, n = torch.rand(3, h, w, generator=g), torch.cat([(torch.rand(2, h, w, generator=g) - 0.5), torch.ones(1, h, w)], 0)
    r, m = torch.rand(1, h, w, generator=g) * 0.7 + 0.3, torch.rand(1, h, w, generator=g)
    wt = torch.rand(3, 2 * h, 2 * w, generator=g) - 0.4
    light, view = torch.tensor([0.3, 0.2, 0.8]), torch.tensor([0.0, 0.1, 1.0])
    for half in (False, True):
        src = [t.half() for t in (a, n, r, m)] if half else [a, n, r, m]
        L64, V64 = light.double().requires_grad_(True), view.double().requires_grad_(True)
        ref = O.cook_torrance(*[t.float().double().repeat(1, 2, 2) for t in src], None, view=V64, light=L64,
                              intensity=torch.ones(3, dtype=torch.float64), light_type="point", light_size=2.0)
        (ref * wt.double()).sum().backward()
        L, V = light.clone().cuda().requires_grad_(True), view.clone().requires_grad_(True)
        out = F.cook_torrance(*[t.cuda() for t in src], tile=2, view_dir=V, light=L, light_intensity=[1, 1, 1], light_type="point", light_size=2.0)
        assert out.shape == (3, 2 * h, 2 * w) and (out.detach().cpu() - ref.detach().float()).abs().max().item() <= 1e-5
        (out * wt.cuda()).sum().backward()
        for name, got, want in (("light", L.grad.cpu(), L64.grad), ("view", V.grad, V64.grad)):
            err = (got.double() - want).abs()
            assert (err <= 2e-5 * (1.0 + want.abs().max())).all(), (half, name, got, want)


@pytest.mark.parametrize("quirk", [True, False])
@pytest.mark.parametrize("light_type,light,size", [("point", [0.1, 0.1, 1.0], 1.0), ("directional", [0.3, -0.2, 1.0], None)])
def test_gradients_through_the_fused_conversion_and_without_a_normal_map(quirk, light_type, light, size):
    """CONVERTED workflow (to_diffuse_specular_material fused in front, metallic.py:98-108, both settings of the upstream
    specular_is_srgb quirk) and the +Z default normal (cooktorrance.py:147-152): gradients w.r.t. the metallic-workflow maps
    against float64 autograd through the oracle's composition of the same reference steps."""
    import torch_oracle as O
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(23)
    H, W = 20, 36
    a, r, m = torch.rand(3, H, W, generator=g), torch.rand(1, H, W, generator=g) * 0.7 + 0.3, torch.rand(1, H, W, generator=g)
    wt = torch.rand(3, H, W, generator=g) - 0.4
    okw = dict(view=torch.tensor([0.0, 0.1, 1.0], dtype=torch.float64), light=torch.tensor(light, dtype=torch.float64),
               intensity=torch.tensor([0.9, 0.8, 0.7], dtype=torch.float64), light_type=light_type, light_size=size)
    leaves64 = [t.double().requires_grad_(True) for t in (a, r, m)]
    ref = O.cook_torrance_converted(leaves64[0], None, leaves64[1], leaves64[2], quirk_specular_srgb=quirk, **okw)
    (ref * wt.double()).sum().backward()
    leaves = [t.clone().cuda().requires_grad_(True) for t in (a, r, m)]
    out = F.cook_torrance(leaves[0], None, leaves[1], leaves[2], view_dir=[0.0, 0.1, 1.0], light=light, light_intensity=[0.9, 0.8, 0.7],
                          light_type=light_type, light_size=size, convert_to_diffuse_specular=True, specular_is_srgb=quirk)
    assert (out.detach().cpu() - ref.detach().float()).abs().max().item() <= 1e-5
    (out * wt.cuda()).sum().backward()
    for name, x, y in zip(("albedo", "roughness", "metallic"), leaves, leaves64):
        err = (x.grad.cpu().double() - y.grad).abs()
        assert (err <= 2e-5 * (1 + y.grad.abs())).all(), (name, float(err.max()))


@pytest.mark.parametrize("kind", ["metallic", "specular"])
@pytest.mark.parametrize("lk", ["pt1", "dir"])
def test_light_view_intensity_gradients_match_the_reference_autograd(kind, lk, golden):
    """tests/golden/grad_params.npz: the REAL reference's autograd gradients w.r.t. view_dir / light_dir_or_position /
    light_intensity on the maps of grad.npz (oracle/gen_golden.py --only grad_params)."""
    from pypbr_amd import functional as F
    z, zg = golden("grad_params"), golden("grad")
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    a, n, r = dev(zg["in_albedo"]), dev(zg["in_normal"]), dev(zg["in_roughness"])
    m = dev(zg["in_metallic"]) if kind == "metallic" else None
    s = dev(zg["in_specular"]) if kind == "specular" else None
    ltype, lvec, lsize = LIGHTS[lk]
    V = torch.tensor([0.05, 0.1, 0.9], requires_grad=True)
    L = torch.tensor(lvec, device="cuda", requires_grad=True)
    I = torch.tensor([0.9, 0.8, 0.7], requires_grad=True)
    out = F.cook_torrance(a, n, r, m, s, view_dir=V, light=L, light_intensity=I, light_type=ltype, light_size=lsize)
    assert (out.detach().cpu().numpy() - z[f"out_{kind}_{lk}"]).__abs__().max() <= 1e-5
    (out * dev(zg["in_weight"])).sum().backward()
    for name, leaf in (("view", V), ("light", L), ("intensity", I)):
        g = leaf.grad.cpu().double().numpy()
        ref32, ref64 = z[f"grad_{kind}_{lk}_{name}"].astype(np.float64), z[f"g64_{kind}_{lk}_{name}"]
        band = 2e-5 * (1.0 + np.abs(ref64).max())
        assert (np.abs(g - ref64) <= band).all(), (name, g, ref64)
        # never further from the float64 gradient than the reference's own fp32 gradient is, plus the band
        assert (np.abs(g - ref32) <= np.abs(ref32 - ref64) + band).all(), (name, g, ref32)


@pytest.mark.gpu
def test_streamed_backward_is_bit_identical_to_the_one_tile_kernels():
    """fp16 maps with one light and rows that are a whole number of 128-pixel tiles take the streamed backward kernel
    (ct_backward.hpp: a wave walks a run of tiles, the next tile's texels travel global -> LDS while the current one is
    differentiated).  Same backward_body as the one-tile kernel, so every gradient must agree bit for bit -- for runs
    that do not divide the tile count, batches, every workflow, both light types, a missing normal map, sRGB off, and
    only some of the gradients wanted."""
    from pypbr_amd import _native as N, functional as F
    lib = N.lib()
    g = torch.Generator(device="cuda").manual_seed(77)

    def maps(b, h, w, spec=False):
        a = torch.rand(b, 3, h, w, device="cuda", generator=g)
        n = torch.nn.functional.normalize(torch.rand(b, 3, h, w, device="cuda", generator=g) * 2 - 1, dim=1)
        r = torch.rand(b, 1, h, w, device="cuda", generator=g) * 0.9 + 0.1
        m = torch.rand(b, 3 if spec else 1, h, w, device="cuda", generator=g)
        return [t.half() for t in (a, n, r, m)]
    point = dict(view_dir=[0.1, -0.2, 1.0], light=[0.1, 0.1, 1.0], light_intensity=[1.0, 0.9, 0.8], light_type="point", light_size=1.0)
    sun = dict(view_dir=[0, 0.1, 1.0], light=[0.3, -0.2, 1.0], light_intensity=[0.8, 0.7, 0.6], light_type="directional")

    def grads(mp, kw, run, wanted=(True, True, True, True), spec=False, no_normal=False):
        lib.pbr_set_tuning(N.TUNE_BWD_RUN, run)
        leaves = [t.clone().requires_grad_(w) for t, w in zip(mp, wanted)]
        a, n, r, m = leaves
        args = dict(kw)
        if spec:
            out = F.cook_torrance(a, None if no_normal else n, r, None, specular=m, **args)
        else:
            out = F.cook_torrance(a, None if no_normal else n, r, m, **args)
        wt = torch.rand(out.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)) - 0.3
        (out * wt).sum().backward()
        return [t.grad.clone() if t.grad is not None else None for t in leaves]
    cases = [("single 5 x 384", maps(1, 5, 384), point, {}),
             ("batch 3 x 7 x 128", maps(3, 7, 128), point, {}),
             ("directional", maps(2, 9, 256), sun, {}),
             ("linear in and out", maps(1, 6, 256), dict(point, albedo_is_srgb=False, return_srgb=False), {}),
             ("specular", maps(2, 5, 256, spec=True), point, dict(spec=True)),
             ("converted", maps(2, 5, 256), dict(sun, convert_to_diffuse_specular=True), {}),
             ("no normal map", maps(1, 11, 128), point, dict(no_normal=True)),
             ("albedo and roughness only", maps(1, 8, 256), sun, dict(wanted=(True, False, True, False))),
             ("row band of a taller map", maps(2, 6, 256), dict(point, y_offset=5, height_total=32), {}),
             ("full rows 64 x 1024", maps(1, 64, 1024), point, {})]
    try:
        for name, mp, kw, extra in cases:
            want = grads(mp, kw, 0, **extra)                  # the one-tile kernels
            for run in (1, 2, 7, 1000):          # rounds: grid = run x the waves the chip holds; 1000 = one tile per wave
                got = grads(mp, kw, run, **extra)
                for x, y in zip(want, got):
                    assert (x is None) == (y is None), (name, run)
                    if x is not None:
                        assert torch.equal(x, y), (name, run, float((x.float() - y.float()).abs().max()))
    finally:
        lib.pbr_set_tuning(N.TUNE_BWD_RUN, -1)


@pytest.mark.parametrize("binding", ["torch_op", "ctypes"])
def test_gradients_of_fp16_maps_that_are_tiled_or_shared_by_the_batch(binding):
    """Round 3: a fused tile() repeat, or one map shared by the whole batch, owns the SUM of the per-pixel gradients also when
    the maps (hence their gradients) are fp16: pbr_fold_gradient_typed adds in fp32 and rounds once.  Ground truth: float64
    oracle autograd on the exact fp32 up-casts of the fp16 maps (materialised repeat / broadcast)."""
    import torch_oracle as O
    from pypbr_amd import functional as F
    F.USE_TORCH_OPS = binding == "torch_op"
    try:
        g = torch.Generator().manual_seed(23)
        h, w, B = 12, 24, 3
        mk = lambda c, lo=0.0, sc=1.0: (torch.rand(B, c, h, w, generator=g) * sc + lo).half()
        a, r, m = mk(3), mk(1, 0.35, 0.6), mk(1)
        n = torch.cat([torch.rand(B, 2, h, w, generator=g) - 0.5, torch.ones(B, 1, h, w)], 1).half()
        rshared = r[:1].clone()                                          # ONE roughness map for the whole batch
        view, light, inten = torch.tensor([0.0, 0.1, 1.0]), torch.tensor([0.2, 0.1, 0.9]), torch.tensor([1.0, 0.9, 0.8])
        kw = dict(view_dir=view, light=light, light_intensity=inten, light_type="point", light_size=2.0)
        okw = dict(view=view.double(), light=light.double(), intensity=inten.double(), light_type="point", light_size=2.0)
        # (1) tiled 2 x 2, batch of 3, roughness shared: both folds at once
        wt = torch.rand(B, 3, 2 * h, 2 * w, generator=g) - 0.4
        leaves64 = [t.float().double().requires_grad_(True) for t in (a, n, rshared, m)]
        ref = torch.stack([O.cook_torrance(leaves64[0][b].repeat(1, 2, 2), leaves64[1][b].repeat(1, 2, 2), leaves64[2][0].repeat(1, 2, 2),
                                           leaves64[3][b].repeat(1, 2, 2), None, **okw) for b in range(B)])
        (ref * wt.double()).sum().backward()
        leaves = [t.clone().cuda().requires_grad_(True) for t in (a, n, rshared, m)]
        out = F.cook_torrance(*leaves, tile=2, **kw)
        assert out.shape == (B, 3, 2 * h, 2 * w) and (out.detach().cpu().double() - ref.detach()).abs().max().item() <= 1e-5
        (out * wt.cuda()).sum().backward()
        for name, x, y in zip(("albedo", "normal", "roughness (shared, tiled)", "metallic"), leaves, leaves64):
            assert x.grad.dtype == torch.float16 and x.grad.shape == x.shape
            err = (x.grad.float().cpu().double() - y.grad).abs()
            # fp16 storage of each of the up to 12 per-pixel gradients of a sum (2^-11 relative each) and of the sum itself
            assert bool((err <= 1.5e-3 * y.grad.abs() + 5e-3).all()), (name, float(err.max()), float(y.grad.abs().max()))
        # (2) the fold itself, exactly: the same evaluation on materialised repeats and an expanded roughness map gives the
        # per-pixel fp16 gradients; their float64 sum, rounded once to fp16, is what the typed fold kernel must return
        rep = [t.clone().cuda() for t in (a, n, rshared.expand(B, -1, -1, -1).contiguous(), m)]
        rep = [t.repeat(1, 1, 2, 2).requires_grad_(True) for t in rep]
        (F.cook_torrance(*rep, **kw) * wt.cuda()).sum().backward()
        for i, (name, x) in enumerate(zip(("albedo", "normal", "roughness", "metallic"), leaves)):
            per_pixel = rep[i].grad.float().cpu().double()                                      # [B, C, 2h, 2w]
            want = per_pixel.reshape(B, -1, 2, h, 2, w).sum(dim=(2, 4))
            if i == 2:
                want = want.sum(dim=0, keepdim=True)
            diff = (x.grad.float().cpu().double() - want).abs()                  # fp32 accumulation, then ONE rounding to fp16
            assert bool((diff <= 2.0 ** -11 * want.abs() + 2.0 ** -24).all()), (name, float(diff.max()))
    finally:
        F.USE_TORCH_OPS = True
