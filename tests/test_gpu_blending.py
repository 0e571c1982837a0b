"""GPU parity of the blending pre-stage against pypbr.blending run by the real
reference on 96x96 crops of its two PNG materials (tests/golden/blend.npz), through the class API
the reference's examples/example_blend.py uses."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _materials(z, device):
    from pypbr_amd.materials import BasecolorMetallicMaterial
    mats = []
    for i in (1, 2):
        m = BasecolorMetallicMaterial(device=torch.device(device))
        for key in z:
            if key.startswith(f"in_m{i}_"):
                m._maps[key[len(f"in_m{i}_"):]] = torch.from_numpy(z[key]).to(device)    # maps exactly as the reference loaded them
        mats.append(m)
    return mats


@pytest.mark.parametrize("device", ["cuda", "cpu"])
def test_blends_match_reference(device, golden):
    import pypbr_amd.blending as B
    z = golden("blend")
    m1, m2 = _materials(z, device)
    blends = {"height": B.HeightBlend(blend_width=0.1, shift=-0.5), "mask": B.MaskBlend(torch.from_numpy(z["in_mask"]).to(device)),
              "prop": B.PropertyBlend(property_name="roughness", blend_width=0.1),
              "gradh": B.GradientBlend("horizontal"), "gradv": B.BlendFactory.get_blend_method("gradient", direction="vertical")}
    for name, blender in blends.items():
        out, mask = blender(m1, m2)
        assert type(out) is type(m1) and mask.shape == (1, 96, 96)
        assert np.abs(mask.cpu().numpy() - z[f"out_{name}_mask"]).max() <= 2e-6, name
        expected = sorted(k[len(f"out_{name}_"):] for k in z if k.startswith(f"out_{name}_") and not k.endswith("_mask"))
        assert sorted(out._maps) == expected
        for k, v in out._maps.items():
            assert v.device.type == device
            assert np.abs(v.cpu().numpy() - z[f"out_{name}_{k}"]).max() <= 3e-6, (name, k)
    assert out.albedo_is_srgb == m1.albedo_is_srgb


def test_blend_then_render_pipeline(golden):
    """example_blend.py's sequence on the crops: HeightBlend -> resize -> tile -> point-light render, against
    the same sequence evaluated by the oracle on the reference's blended maps."""
    import torch_oracle as O
    import pypbr_amd.blending as B
    from pypbr_amd.models import CookTorranceBRDF
    z = golden("blend")
    m1, m2 = _materials(z, "cuda")
    material, mask = B.HeightBlend(blend_width=0.1, shift=-0.5)(m1, m2)
    material.tile(2)
    out = CookTorranceBRDF("point")(material, torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]),
                                    torch.tensor([1.0, 1.0, 1.0]), 1.0)
    ref_maps = {k: torch.from_numpy(z[f"out_height_{k}"]).repeat(1, 2, 2) for k in ("albedo", "normal", "roughness", "metallic")}
    ref = O.cook_torrance(ref_maps["albedo"], ref_maps["normal"], ref_maps["roughness"], ref_maps["metallic"], None,
                          view=torch.tensor([0.0, 0.0, 1.0]), light=torch.tensor([0.1, 0.1, 1.0]),
                          intensity=torch.tensor([1.0, 1.0, 1.0]), light_type="point", light_size=1.0)
    assert (out.cpu() - ref).abs().max().item() <= 1e-5


def test_blend_errors():
    import pypbr_amd.blending as B
    from pypbr_amd.materials import BasecolorMetallicMaterial
    a = BasecolorMetallicMaterial(albedo=torch.rand(3, 8, 8), roughness=torch.rand(1, 8, 8))
    with pytest.raises(ValueError, match="height maps"):
        B.blend_on_height(a, a)
    with pytest.raises(ValueError, match="'metallic' maps"):
        B.blend_on_properties(a, a)
    with pytest.raises(ValueError, match="Mask must have shape"):
        B.blend_with_mask(a, a, torch.rand(2, 8, 8))
    with pytest.raises(ValueError, match="Direction must be"):
        B.blend_with_gradient(a, a, "diagonal")
    with pytest.raises(ValueError, match="Unknown blending method"):
        B.blend_materials(a, a, method="magic")
    with pytest.raises(ValueError, match="Mask must be provided"):
        B.blend_materials(a, a, method="mask")


BLEND_LIGHTS = {"pt1": ("point", [0.1, 0.1, 1.0], 1.0), "dir": ("directional", [0.3, -0.2, 1.0], None)}


@pytest.mark.parametrize("device", ["cuda", "cpu"])
def test_fused_blend_and_render_matches_the_reference_pipeline(device, golden):
    """N4 as the survey words it: the blend fused in front of the BRDF.  Lazy blends rendered by CookTorranceBRDF
    (pbr_cook_torrance_blend: both materials read once, no blended copy) against the REAL reference's
    blend -> render (tests/golden/blend.npz render_*), and bit-for-bit against this package's own unfused path."""
    import pypbr_amd.blending as B
    from pypbr_amd.models import CookTorranceBRDF
    z = golden("blend")
    m1, m2 = _materials(z, device)
    blends = {"height": lambda: B.HeightBlend(blend_width=0.1, shift=-0.5), "mask": lambda: B.MaskBlend(torch.from_numpy(z["in_mask"]).to(device)),
              "prop": lambda: B.PropertyBlend(property_name="roughness", blend_width=0.1),
              "gradh": lambda: B.GradientBlend("horizontal"), "gradv": lambda: B.GradientBlend("vertical")}
    view, inten = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([1.0, 1.0, 1.0])
    for name, make in blends.items():
        with B.lazy_blending():
            lazy, mask = make()(m1, m2)
        assert lazy.__dict__.get("_lazy_blend") is not None and type(lazy) is type(m1)
        eager, _ = make()(m1, m2)
        for lk, (ltype, lvec, lsize) in BLEND_LIGHTS.items():
            brdf = CookTorranceBRDF(ltype)
            got = brdf(lazy, view, torch.tensor(lvec), inten, lsize)
            assert lazy.__dict__.get("_lazy_blend") is not None            # rendering does not materialise the blend
            assert got.device.type == device
            assert np.abs(got.cpu().numpy() - z[f"render_{name}_{lk}"]).max() <= 1e-5, (name, lk)
            unfused = brdf(eager, view, torch.tensor(lvec), inten, lsize)
            assert (got - unfused).abs().max().item() <= 2e-7, (name, lk)
        # the first look at the maps blends them for real; they equal the eager blend
        assert lazy.albedo.shape == (3, 96, 96) and lazy.__dict__.get("_lazy_blend") is None
        for k, v in eager._maps.items():
            assert torch.equal(lazy._maps[k], v), (name, k)


def test_fused_blend_flat_normals_follow_the_redecode_quirk():
    """Two flat +Z normal maps blend to (0,0,1) everywhere: no negative component, so on assignment the reference
    re-reads the blended normal as [0,1]-encoded (base.py:212-216).  The fused path carries the same per-map flag."""
    import pypbr_amd.blending as B
    from pypbr_amd import functional as F
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    g = torch.Generator().manual_seed(12)
    H, W = 24, 40
    dev = torch.device("cuda")

    def mat(seed_shift, flat):
        m = BasecolorMetallicMaterial(albedo=torch.rand(3, H, W, generator=g), roughness=torch.rand(1, H, W, generator=g) * 0.7 + 0.3,
                                      metallic=torch.rand(1, H, W, generator=g), device=dev)
        n = torch.zeros(3, H, W); n[2] = 1.0
        if not flat:
            n[:2] = torch.rand(2, H, W, generator=g) - 0.5
        m._maps["normal"] = n.to(dev)
        return m
    mask = torch.rand(1, H, W, generator=g).to(dev)
    args = (torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0]), 1.0)
    brdf = CookTorranceBRDF("point")
    for flat in (True, False):
        m1, m2 = mat(0, flat), mat(1, flat)
        lazy, _ = B.blend_with_mask(m1, m2, mask, lazy=True)
        eager, _ = B.blend_with_mask(m1, m2, mask)
        assert bool((eager.normal[:2] < 0).all()) == flat                   # flat: re-decoded to (-1,-1,1)/sqrt(3)
        assert (brdf(lazy, *args) - brdf(eager, *args)).abs().max().item() <= 2e-7
    # batched call through the functional API: one flag per material
    a = torch.rand(2, 3, H, W, generator=g).to(dev); r = (torch.rand(2, 1, H, W, generator=g) * 0.7 + 0.3).to(dev)
    m = torch.rand(2, 1, H, W, generator=g).to(dev)
    n = torch.zeros(2, 3, H, W); n[:, 2] = 1.0; n[1, :2] = torch.rand(2, H, W, generator=g) - 0.5
    n = n.to(dev)
    kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
    fused = F.cook_torrance(a, n, r, m, blend=(a.flip(0), n, r.flip(0), m.flip(0), None, mask), **kw)
    for b in range(2):
        one = F.cook_torrance(a[b], n[b], r[b], m[b], blend=(a.flip(0)[b], n[b], r.flip(0)[b], m.flip(0)[b], None, mask), **kw)
        assert torch.equal(fused[b], one)
    # with a gradient attached the same call runs the differentiable (unfused) pieces and gives the same image
    leaf = a.clone().requires_grad_(True)
    with_grad = F.cook_torrance(leaf, n, r, m, blend=(a.flip(0), n, r.flip(0), m.flip(0), None, mask), **kw)
    assert with_grad.requires_grad and (with_grad.detach() - fused).abs().max().item() <= 2e-6
    with_grad.sum().backward()
    assert leaf.grad is not None and bool(torch.isfinite(leaf.grad).all()) and float(leaf.grad.abs().sum()) > 0


def test_fused_blend_composes_with_fused_tile_and_several_lights():
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(21)
    h, w, n_t = 24, 32, 2
    dev = torch.device("cuda")
    mk = lambda c: torch.rand(c, h, w, generator=g).to(dev)
    nrm = lambda: torch.cat([torch.rand(2, h, w, generator=g) - 0.5, torch.ones(1, h, w)], 0).to(dev)
    a1, a2, r1, r2, m1, m2, mask = mk(3), mk(3), mk(1) * 0.7 + 0.3, mk(1) * 0.7 + 0.3, mk(1), mk(1), mk(1)
    n1, n2 = nrm(), nrm()
    kw = dict(view_dir=[0, 0, 1], light=[[0.1, 0.1, 1.0], [-0.3, 0.2, 0.8]], light_intensity=[[0.6, 0.6, 0.6], [0.4, 0.5, 0.6]],
              light_type="point", light_size=1.5)
    rep = lambda t: t.repeat(1, n_t, n_t)
    tiled = F.cook_torrance(a1, n1, r1, m1, blend=(a2, n2, r2, m2, None, mask), tile=n_t, **kw)
    full = F.cook_torrance(rep(a1), rep(n1), rep(r1), rep(m1), blend=(rep(a2), rep(n2), rep(r2), rep(m2), None, rep(mask)), **kw)
    assert tiled.shape == (3, n_t * h, n_t * w) and torch.equal(tiled, full)
    # ragged width -> the one-pixel-per-lane instantiation of the blend kernel
    cut = lambda t: t[..., :w - 3].contiguous()
    ragged = F.cook_torrance(cut(a1), cut(n1), cut(r1), cut(m1), blend=(cut(a2), cut(n2), cut(r2), cut(m2), None, cut(mask)), **kw)
    assert (ragged - F.cook_torrance(a1, n1, r1, m1, blend=(a2, n2, r2, m2, None, mask), **kw)[..., :w - 3]).abs().max().item() > 0  # light grid differs
    import pypbr_amd.blending as B
    blended = {k: B.blend_maps(cut(x), cut(y), cut(mask), is_normal=(k == "n")) for k, x, y in (("a", a1, a2), ("n", n1, n2), ("r", r1, r2), ("m", m1, m2))}
    ref = F.cook_torrance(blended["a"], blended["n"], blended["r"], blended["m"], **kw)
    assert (ragged - ref).abs().max().item() <= 2e-7


def test_fused_blend_guard_bands():
    """Both materials, the mask and the result inside NaN / sentinel margins: no read or write outside them."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(77)

    def guarded(t, fill, G):
        flat = torch.full((t.numel() + 2 * G,), fill, dtype=t.dtype, device="cuda")
        flat[G:G + t.numel()] = t.reshape(-1).cuda()
        return flat, flat[G:G + t.numel()].view(t.shape)

    for trial in range(12):
        G = 256 if trial % 2 else 255
        B = 1 + trial % 2
        h, w = int(torch.randint(1, 30, (1,), generator=g)), [8, 20, 64, 5, 33, 12][trial % 6]
        mk = lambda c: torch.rand(B, c, h, w, generator=g)
        nrm = lambda: torch.cat([torch.rand(B, 2, h, w, generator=g) - 0.5, torch.ones(B, 1, h, w)], 1)
        mats = [mk(3), nrm(), mk(1) * 0.7 + 0.3, mk(1), mk(3), nrm(), mk(1) * 0.7 + 0.3, mk(1), torch.rand(1, 1, h, w, generator=g)]
        kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point" if trial % 3 else "directional",
                  light_size=1.0)
        dev = [t.cuda() for t in mats]
        ref = F.cook_torrance(*dev[:4], blend=(dev[4], dev[5], dev[6], dev[7], None, dev[8]), **kw)
        bufs, v = zip(*[guarded(t, float("nan"), G) for t in mats])
        obuf = torch.full((B * 3 * h * w + 2 * G,), -7.0, device="cuda")
        out = obuf[G:G + B * 3 * h * w].view(B, 3, h, w)
        got = F.cook_torrance(*v[:4], blend=(v[4], v[5], v[6], v[7], None, v[8]), out=out, **kw)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(got).all()) and (got - ref).abs().max().item() <= 2e-6, (trial, B, h, w)
        assert bool((obuf[:G] == -7.0).all()) and bool((obuf[-G:] == -7.0).all()), (trial, B, h, w)


def test_blend_maps_gradients_against_oracle_autograd():
    """pbr_blend_maps_backward: the reference's blend is plain torch arithmetic (functional.py:103-110, :119-145), so its
    autograd reaches both maps and the mask.  Ground truth: float64 autograd through oracle/blend_oracle.py."""
    import blend_oracle as BO
    from pypbr_amd.blending import blend_maps
    g = torch.Generator().manual_seed(31)
    H, W = 24, 40
    for C, normal in ((3, False), (1, False), (3, True)):
        a, b = torch.rand(C, H, W, generator=g) - (0.5 if normal else 0), torch.rand(C, H, W, generator=g) - (0.5 if normal else 0)
        m, wt = torch.rand(1, H, W, generator=g), torch.rand(C, H, W, generator=g) - 0.4
        ref_leaves = [t.double().requires_grad_(True) for t in (a, b, m)]
        ref = (BO.blend_normals if normal else BO.blend_maps)(*ref_leaves)
        (ref * wt.double()).sum().backward()
        leaves = [t.clone().cuda().requires_grad_(True) for t in (a, b, m)]
        out = blend_maps(*leaves, is_normal=normal)
        assert out.requires_grad and (out.detach().cpu().double() - ref.detach()).abs().max().item() <= 2e-6
        (out * wt.cuda()).sum().backward()
        for name, x, y in zip(("map1", "map2", "mask"), leaves, ref_leaves):
            err = (x.grad.cpu().double() - y.grad).abs()
            assert x.grad.shape == x.shape and (err <= 2e-5 * (1 + y.grad.abs())).all(), (C, normal, name, float(err.max()))
    # only the mask wants a gradient (a learned blend mask): no map gradient buffers
    a, b = torch.rand(3, H, W, generator=g).cuda(), torch.rand(3, H, W, generator=g).cuda()
    m = torch.rand(1, H, W, generator=g).cuda().requires_grad_(True)
    blend_maps(a, b, m).sum().backward()
    assert (m.grad - (a - b).sum(dim=0, keepdim=True)).abs().max().item() <= 1e-5


def test_rendering_loss_through_a_blend_reaches_both_materials_and_the_mask():
    """examples/example_blend.py:14-32 inside a training loop: F.cook_torrance(blend=...) with gradients attached runs the
    differentiable pieces (blend, re-decode of the blended normal, evaluation) and matches float64 autograd through the
    oracles; without gradients the same call is the fused kernel, and both give the same image."""
    import blend_oracle as BO
    import torch_oracle as O
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(41)
    H, W = 24, 48

    def material():
        n = torch.cat([(torch.rand(2, H, W, generator=g) - 0.5), torch.ones(1, H, W)], 0)
        return {"albedo": torch.rand(3, H, W, generator=g), "normal": n / n.norm(dim=0, keepdim=True),
                "roughness": torch.rand(1, H, W, generator=g) * 0.7 + 0.3, "metallic": torch.rand(1, H, W, generator=g)}
    m1, m2 = material(), material()
    mask, wt = torch.rand(1, H, W, generator=g), torch.rand(3, H, W, generator=g) - 0.4
    view, light, inten = torch.tensor([0.0, 0.1, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 0.9, 0.8])
    # float64 ground truth
    r1 = {k: v.double().requires_grad_(True) for k, v in m1.items()}
    r2 = {k: v.double().requires_grad_(True) for k, v in m2.items()}
    rm = mask.double().requires_grad_(True)
    bl = BO.blend_materials(r1, r2, rm)
    ref = O.cook_torrance(bl["albedo"], bl["normal"], bl["roughness"], bl["metallic"], None, view=view.double(), light=light.double(),
                          intensity=inten.double(), light_type="point", light_size=1.0)
    (ref * wt.double()).sum().backward()
    d1 = {k: v.clone().cuda().requires_grad_(True) for k, v in m1.items()}
    d2 = {k: v.clone().cuda().requires_grad_(True) for k, v in m2.items()}
    dm = mask.clone().cuda().requires_grad_(True)
    kw = dict(view_dir=view, light=light, light_intensity=inten, light_type="point", light_size=1.0)
    out = F.cook_torrance(d1["albedo"], d1["normal"], d1["roughness"], d1["metallic"],
                          blend=(d2["albedo"], d2["normal"], d2["roughness"], d2["metallic"], None, dm), **kw)
    assert out.requires_grad and (out.detach().cpu() - ref.detach().float()).abs().max().item() <= 1e-5
    (out * wt.cuda()).sum().backward()
    for name in m1:
        for got, want in ((d1[name], r1[name]), (d2[name], r2[name])):
            err = (got.grad.cpu().double() - want.grad).abs()
            assert (err <= 2e-5 * (1 + want.grad.abs())).all(), (name, float(err.max()))
    err = (dm.grad.cpu().double() - rm.grad).abs()
    assert (err <= 2e-5 * (1 + rm.grad.abs())).all(), float(err.max())
    with torch.no_grad():
        fused = F.cook_torrance(d1["albedo"], d1["normal"], d1["roughness"], d1["metallic"],
                                blend=(d2["albedo"], d2["normal"], d2["roughness"], d2["metallic"], None, dm), **kw)
    assert (fused - out.detach()).abs().max().item() <= 2e-6


def test_blend_sign_pass_reads_only_the_rows_of_its_band():
    """for an untiled row band the sign pass used to run over height_total rows from the band's start -- past
    the band, possibly past the tensor.  Here the rows BEHIND the band (same allocation) hold negative normals and the
    band itself none: the band's flag must stay 0, the full map's flag is 1, and a band without given flags is refused."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(77)
    H, W, cut = 64, 64, 24
    a = torch.rand(1, 3, H, W, generator=g).cuda(); r = (torch.rand(1, 1, H, W, generator=g) * 0.7 + 0.3).cuda()
    m = torch.rand(1, 1, H, W, generator=g).cuda(); mask = torch.rand(1, 1, H, W, generator=g).cuda()
    n = torch.zeros(1, 3, H, W); n[:, 0] = 0.3; n[:, 1] = 0.2; n[:, 2] = 0.9
    n[:, 0, cut:] = -0.6                                         # negative x only in the rows after the band
    n = (n / n.norm(dim=1, keepdim=True)).cuda()
    kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
    band = lambda t: t[:, :, :cut]
    second = (a.flip(-1), n, r, m, None, mask)
    flags = torch.zeros(1, dtype=torch.int32, device="cuda")
    plan = F.plan_cook_torrance(band(a), band(n), band(r), band(m), y_offset=0, height_total=H,
                                blend=tuple(None if t is None else band(t) for t in second), blend_flags=flags, **kw)
    assert plan.blend_normal_sign().item() == 0
    whole = F.plan_cook_torrance(a, n, r, m, blend=second, **kw)
    assert whole.blend_normal_sign().item() == 1
    full = whole.launch().clone()
    plan.use_blend_flags(torch.ones(1, dtype=torch.int32, device="cuda"))      # the whole-map answer, as the ranks would combine it
    assert torch.equal(plan.launch(), full[:, :, :cut])
    with pytest.raises(NotImplementedError):
        F.cook_torrance(band(a), band(n), band(r), band(m), y_offset=0, height_total=H,
                        blend=tuple(None if t is None else band(t) for t in second), **kw)
