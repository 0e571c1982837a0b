"""Round 6: the fused blend's backward over TILED maps (pbr_cook_torrance_blend_backward, ABI 8) -- `blend_with_mask(m1, m2, mask)` -> `tile(n)` ->
CookTorranceBRDF inside a rendering loss (/root/reference/examples/example_blend.py:14-31, pypbr/materials/base.py:524-537,
docs/source/tutorials/06_advanced.rst:73-107): MAP-sized gradients of both materials and of the mask from ONE kernel that blends once per
texel, walks the repeats and runs the folded gradients through the blend's chain rule; where the library does not serve a launch, the
differentiable pieces (which round 5 refused for tiled maps)."""
import pytest
import torch

import blend_oracle as BO
import torch_oracle as O
from test_gpu_blend_backward import _check, _material

pytestmark = pytest.mark.gpu


def _reference_tiled(m1, m2, mask, wt, view, lights, intens, light_type, light_size, tile, converted=False):
    """float64 autograd through the reference's ops: blend, re-assignment of the blended normal, repeat(1, ny, nx), the BRDF."""
    ny, nx = tile
    r1 = {k: v.double().requires_grad_(True) for k, v in m1.items()}
    r2 = {k: v.double().requires_grad_(True) for k, v in m2.items()}
    rm = mask.double().requires_grad_(True)
    bl = {k: v.repeat(1, ny, nx) for k, v in BO.blend_materials(r1, r2, rm).items()}
    kw = dict(view=view.double(), light_type=light_type, light_size=light_size)
    L, I = lights.double().reshape(-1, 3), intens.double().reshape(-1, 3)
    if converted:
        ref = O.cook_torrance_converted(bl["albedo"], bl["normal"], bl["roughness"], bl["metallic"], light=L[0], intensity=I[0], **kw)
    elif L.shape[0] > 1:
        ref = O.cook_torrance_multi(bl["albedo"], bl["normal"], bl["roughness"], bl.get("metallic"), bl.get("specular"), lights=L, intensities=I, **kw)
    else:
        ref = O.cook_torrance(bl["albedo"], bl["normal"], bl["roughness"], bl.get("metallic"), bl.get("specular"), light=L[0], intensity=I[0], **kw)
    (ref * wt.double()).sum().backward()
    return ref.detach(), r1, r2, rm


@pytest.mark.parametrize("workflow", ["metallic", "specular", "converted"])
@pytest.mark.parametrize("light_type", ["point", "directional"])
@pytest.mark.parametrize("flat", [False, True])
@pytest.mark.parametrize("hw", [(18, 44), (10, 38)])
def test_fused_blend_backward_over_tiled_maps_against_float64_autograd(workflow, light_type, flat, hw):
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(300 + 7 * ["metallic", "specular", "converted"].index(workflow) + (3 if flat else 0))
    (H, W), tile = hw, (2, 3)                                      # 44 = 11 groups of 4 texels; 38: ragged rows (the last lane of a row moves back)
    m1, m2 = _material(g, H, W, workflow, flat), _material(g, H, W, workflow, flat)
    mask, wt = torch.rand(1, H, W, generator=g), torch.rand(3, tile[0] * H, tile[1] * W, generator=g) - 0.4
    view = torch.tensor([0.0, 0.1, 1.0])
    light = torch.tensor([0.1, 0.1, 1.0]) if light_type == "point" else torch.tensor([0.3, -0.2, 1.0])
    inten = torch.tensor([1.0, 0.9, 0.8])
    size = 1.5 if light_type == "point" else None
    ref, r1, r2, rm = _reference_tiled(m1, m2, mask, wt, view, light, inten, light_type, size, tile, converted=(workflow == "converted"))
    d1 = {k: v.clone().cuda().requires_grad_(True) for k, v in m1.items()}
    d2 = {k: v.clone().cuda().requires_grad_(True) for k, v in m2.items()}
    dm = mask.clone().cuda().requires_grad_(True)
    kw = dict(view_dir=view, light=light, light_intensity=inten, light_type=light_type, light_size=size,
              convert_to_diffuse_specular=(workflow == "converted"), specular_is_srgb=True, tile=tile)
    second = (d2["albedo"], d2["normal"], d2["roughness"], d2.get("metallic"), d2.get("specular"), dm)
    out = F.cook_torrance(d1["albedo"], d1["normal"], d1["roughness"], d1.get("metallic"), d1.get("specular"), blend=second, **kw)
    assert out.shape == (3, tile[0] * H, tile[1] * W) and out.requires_grad
    assert type(out.grad_fn).__name__ == "_FusedBlendFnBackward"                              # ONE kernel forward, ONE kernel backward
    assert (out.detach().cpu().double() - ref).abs().max().item() <= 1e-5
    (out * wt.cuda()).sum().backward()
    for name in m1:
        assert d1[name].grad.shape == m1[name].shape                                           # MAP-sized: the sums over the repeats
        _check(d1[name].grad, r1[name].grad, (workflow, light_type, flat, "material 1", name))
        _check(d2[name].grad, r2[name].grad, (workflow, light_type, flat, "material 2", name))
    _check(dm.grad, rm.grad, "mask")


def test_tiled_blend_gradients_the_library_does_not_fuse_take_the_differentiable_pieces():
    """Several lights, and map rows shorter than a 4-texel lane: round 5 raised NotImplementedError for `tile` under a gradient through a
    blend; now the blend (map-sized), the re-decode and the tiled evaluation with its folded backward run as pieces."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(11)
    view, inten = torch.tensor([0.0, 0.1, 1.0]), torch.tensor([[0.6, 0.5, 0.4], [0.3, 0.3, 0.5]])
    lights = torch.tensor([[0.1, 0.1, 1.0], [-0.3, 0.2, 0.8]])
    for (H, W), L, I in (((12, 40), lights, inten), ((10, 3), lights[0], inten[0])):
        tile = (2, 2)
        m1, m2 = _material(g, H, W, "metallic"), _material(g, H, W, "metallic")
        mask, wt = torch.rand(1, H, W, generator=g), torch.rand(3, 2 * H, 2 * W, generator=g) - 0.4
        ref, r1, r2, rm = _reference_tiled(m1, m2, mask, wt, view, L, I, "point", 1.0, tile)
        d1 = {k: v.clone().cuda().requires_grad_(True) for k, v in m1.items()}
        d2 = {k: v.clone().cuda().requires_grad_(True) for k, v in m2.items()}
        dm = mask.clone().cuda().requires_grad_(True)
        second = (d2["albedo"], d2["normal"], d2["roughness"], d2["metallic"], None, dm)
        out = F.cook_torrance(d1["albedo"], d1["normal"], d1["roughness"], d1["metallic"], blend=second, view_dir=view, light=L, light_intensity=I,
                              light_type="point", light_size=1.0, tile=tile)
        assert type(out.grad_fn).__name__ != "_FusedBlendFnBackward"
        assert (out.detach().cpu().double() - ref).abs().max().item() <= 1e-5
        (out * wt.cuda()).sum().backward()
        for name in m1:
            _check(d1[name].grad, r1[name].grad, ((H, W), "material 1", name))
            _check(d2[name].grad, r2[name].grad, ((H, W), "material 2", name))
        _check(dm.grad, rm.grad, "mask")


def test_rendering_loss_on_a_blended_resized_tiled_material_reaches_both_materials():
    """The blend example's material inside the tutorial's loss: blender(m1, m2) -> tile(2) -> RenderingLoss; both source materials and the mask
    receive MAP-sized gradients, equal to the gradients through the materialised blend + repeat."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(5)
    H, W = 32, 64
    m1, m2 = _material(g, H, W, "metallic"), _material(g, H, W, "metallic")
    mask = torch.rand(1, H, W, generator=g).cuda()
    target = torch.rand(3, 2 * H, 2 * W, generator=g).cuda()
    kw = dict(view_dir=[0.0, 0.0, 1.0], light=[0.1, 0.1, 1.0], light_intensity=[1.0, 1.0, 1.0], light_type="point", light_size=1.0)

    def step(fused):
        d1 = {k: v.clone().cuda().requires_grad_(True) for k, v in m1.items()}
        d2 = {k: v.clone().cuda().requires_grad_(True) for k, v in m2.items()}
        dm = mask.clone().requires_grad_(True)
        second = (d2["albedo"], d2["normal"], d2["roughness"], d2["metallic"], None, dm)
        if fused:
            out = F.cook_torrance(d1["albedo"], d1["normal"], d1["roughness"], d1["metallic"], blend=second, tile=2, **kw)
        else:
            out = F._blend_then_render_with_grad(d1["albedo"], d1["normal"], d1["roughness"], d1["metallic"], None, blend=second, tile=2, **kw)
        loss = torch.nn.functional.mse_loss(out, target)
        loss.backward()
        return loss.detach(), d1, d2, dm
    lf, f1, f2, fm = step(True)
    lu, u1, u2, um = step(False)
    assert abs(float(lf) - float(lu)) <= 2e-6 * float(lu)
    for name in m1:
        for a, b in ((f1[name].grad, u1[name].grad), (f2[name].grad, u2[name].grad)):
            assert a.shape == b.shape and (a - b).abs().max().item() <= 2e-5 * (1 + float(b.abs().max())), name
    assert (fm.grad - um.grad).abs().max().item() <= 2e-5 * (1 + float(um.grad.abs().max()))


# ---------------------------------------------------------------- thin row bands and ragged map widths on the repeat-inner walk
def _tiled_maps(g, h, w, dtype=torch.float32):
    a = torch.rand(3, h, w, generator=g)
    n = torch.cat([torch.rand(2, h, w, generator=g) - 0.5, torch.ones(1, h, w)], 0) * (0.7 + torch.rand(1, h, w, generator=g))
    r = torch.rand(1, h, w, generator=g) * 0.7 + 0.25
    m = torch.rand(1, h, w, generator=g)
    return [t.to(dtype).cuda() for t in (a, n, r, m)]


def _knob(value):
    from pypbr_amd import _native as N
    N.lib().pbr_set_tuning(N.TUNE_TILE_REPEAT, value)


@pytest.mark.parametrize("light_type", ["point", "directional"])
@pytest.mark.parametrize("lights", [1, 3])
@pytest.mark.parametrize("hw,tile,dtype", [((16, 64), (3, 2), torch.float32), ((13, 40), (4, 1), torch.float32), ((16, 64), (2, 2), torch.float16),
                                           ((9, 38), (3, 2), torch.float32), ((8, 6), (3, 5), torch.float32), ((7, 13), (2, 3), torch.float16)])
def test_thin_bands_and_ragged_widths_take_the_repeat_inner_walk_and_equal_the_full_image(light_type, lights, hw, tile, dtype):
    """A multi-GPU shard of ONE tiled material is a row band of the tiled image, usually thinner than a period of the map's rows (tile(2) over 8
    ranks: a quarter period).  Every band -- any offset, any height, across a period boundary or not -- equals the rows of the whole image bit
    for bit, and so do maps whose width is not a whole number of 4-texel lanes; both forms name the repeat-inner kernel now, not the wrap-around one."""
    from pypbr_amd import functional as F
    (h, w), (ny, nx) = hw, tile
    g = torch.Generator().manual_seed(5 * h + w + lights)
    maps = _tiled_maps(g, h, w, dtype)
    L = [[0.1, 0.1, 1.0], [-0.4, 0.2, 0.7], [0.3, -0.3, 0.9]][:lights] if light_type == "point" else [[0.3, -0.2, 1.0], [0.1, 0.4, 0.8], [-0.2, 0.1, 1.0]][:lights]
    I = [[1.0, 0.9, 0.8], [0.4, 0.5, 0.6], [0.3, 0.3, 0.3]][:lights]
    kw = dict(view_dir=[0.05, 0.1, 0.9], light=L if lights > 1 else L[0], light_intensity=I if lights > 1 else I[0], light_type=light_type,
              light_size=1.5 if light_type == "point" else None)
    plan = F.plan_cook_torrance(*maps, tile=tile, **kw)
    assert plan.kernel_name.startswith("ctr_"), plan.kernel_name              # the repeat-inner walk serves ragged widths too
    full = plan.launch().clone()
    try:
        _knob(0)
        wrap = F.cook_torrance(*maps, tile=tile, **kw)                        # the wrap-around form: the same functions per pixel
    finally:
        _knob(-1)
    assert torch.equal(full, wrap)
    H = ny * h
    bands = [(0, 1), (h - 1, 2), (1, h - 1), (h // 2, h // 2 + 1), (H - 3, 3), (h + 2, max(1, h - 3)), (2, h), (0, H)]
    for y0, rows in bands:
        rows = min(rows, H - y0)
        p = F.plan_cook_torrance(*maps, tile=tile, y_offset=y0, rows=rows, **kw)
        assert p.kernel_name.startswith("ctr_"), (y0, rows, p.kernel_name)
        got = p.launch()
        assert got.shape == (3, rows, nx * w) and torch.equal(got, full[:, y0:y0 + rows]), (y0, rows)


@pytest.mark.parametrize("binding", ["torch_op", "ctypes"])
@pytest.mark.parametrize("light_type", ["point", "directional"])
@pytest.mark.parametrize("hw,tile", [((16, 64), (3, 2)), ((9, 38), (2, 3)), ((7, 13), (4, 2))])
def test_folded_gradients_of_thin_bands_add_up_to_the_whole_images_gradient(binding, light_type, hw, tile):
    """The ranks' partial sums: bands that partition the tiled image -- thinner than a period, one crossing a period boundary -- each give
    MAP-sized gradients with zeros where the band touches no repeat of a texel; their sum is the gradient of the whole image (to rounding), and
    each band equals float64 autograd through the materialised repeat cropped to it.  Ragged map widths included (9 x 38, 7 x 13)."""
    from pypbr_amd import functional as F
    (h, w), (ny, nx) = hw, tile
    g = torch.Generator().manual_seed(3 * h + w)
    maps = _tiled_maps(g, h, w)
    kw = dict(view_dir=[0.05, 0.1, 0.9], light=[0.1, 0.1, 1.0] if light_type == "point" else [0.3, -0.2, 1.0], light_intensity=[1.0, 0.9, 0.8],
              light_type=light_type, light_size=1.5 if light_type == "point" else None)
    H = ny * h
    gout = (torch.rand(3, H, nx * w, generator=g) - 0.3).cuda()
    before = F.USE_TORCH_OPS
    try:
        F.USE_TORCH_OPS = binding == "torch_op"
        whole = [t.clone().requires_grad_(True) for t in maps]
        (F.cook_torrance(*whole, tile=tile, **kw) * gout).sum().backward()
        cuts = sorted({0, h // 3, h - 1, h + 2, H - 2, H} | {min(H, k * (H // 5 + 1)) for k in range(6)})
        total = [torch.zeros_like(t) for t in maps]
        for y0, y1 in zip(cuts[:-1], cuts[1:]):
            leaves = [t.clone().requires_grad_(True) for t in maps]
            out = F.cook_torrance(*leaves, tile=tile, y_offset=y0, rows=y1 - y0, **kw)
            (out * gout[:, y0:y1]).sum().backward()
            # float64 autograd through the materialised repeat, cropped to the band
            ref = [t.detach().cpu().double().requires_grad_(True) for t in maps]
            okw = dict(view=torch.tensor(kw["view_dir"]).double(), light=torch.tensor(kw["light"]).double(), intensity=torch.tensor(kw["light_intensity"]).double(),
                       light_type=light_type, light_size=kw["light_size"])
            img = O.cook_torrance(*[t.repeat(1, ny, nx) for t in ref], None, **okw)
            (img[:, y0:y1] * gout[:, y0:y1].cpu().double()).sum().backward()
            for name, x, r, acc in zip(("albedo", "normal", "roughness", "metallic"), leaves, ref, total):
                err = (x.grad.cpu().double() - r.grad).abs()
                assert bool((err <= 2e-5 * (1 + r.grad.abs())).all()), (name, y0, y1, float(err.max()))
                acc += x.grad
        for name, acc, wl in zip(("albedo", "normal", "roughness", "metallic"), total, whole):
            assert (acc - wl.grad).abs().max().item() <= 1e-5 * (float(wl.grad.abs().max()) + 1e-12) + 1e-9, name
    finally:
        F.USE_TORCH_OPS = before


def test_ragged_widths_backward_equals_the_two_kernel_form_and_the_loss_step_falls_back():
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(9)
    maps = _tiled_maps(g, 11, 30)
    kw = dict(view_dir=[0.05, 0.1, 0.9], light=[0.1, 0.1, 1.0], light_intensity=[1.0, 0.9, 0.8], light_type="point", light_size=1.5)
    gout = (torch.rand(3, 22, 90, generator=g) - 0.3).cuda()

    def grads(knob):
        leaves = [t.clone().requires_grad_(True) for t in maps]
        try:
            _knob(knob)
            (F.cook_torrance(*leaves, tile=(2, 3), **kw) * gout).sum().backward()
        finally:
            _knob(-1)
        return [t.grad for t in leaves]
    for x, y in zip(grads(-1), grads(0)):
        assert (x - y).abs().max().item() <= 4e-6 * float(y.abs().max()) + 1e-9
    # the rendering-loss step over ragged tiled maps: not one pass (its sum must count every pixel once) -- the three steps, same values
    leaves = [t.clone().requires_grad_(True) for t in maps]
    target = torch.rand(3, 22, 90, generator=g).cuda()
    loss = F.rendering_loss_mse(*leaves, target=target, tile=(2, 3), **kw)
    loss.backward()
    ref = [t.clone().requires_grad_(True) for t in maps]
    want = torch.nn.functional.mse_loss(F.cook_torrance(*[t.repeat(1, 2, 3) for t in ref], **kw), target)
    want.backward()
    assert abs(float(loss) - float(want)) <= 2e-6 * float(want)
    for x, y in zip(leaves, ref):
        assert (x.grad - y.grad).abs().max().item() <= 1e-5 * (float(y.grad.abs().max()) + 1e-12) + 1e-9


def test_thin_bands_of_a_batch_of_tiled_materials_forward_and_folded_backward():
    """Batched tiled maps (B = 3, material-major arena: per-lane plane addresses) over row bands thinner than a period: the window walk with the
    material index in the row arithmetic -- forward bit-equal to the rows of the whole images, folded gradients adding up to the whole launch's."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(77)
    B, h, w, tile = 3, 10, 24, (3, 2)
    maps = [torch.stack([t for t in col]) for col in zip(*[_tiled_maps(g, h, w) for _ in range(B)])]
    for packed in (maps, list(F.pack_maps(*maps, material_major=True))):
        kw = dict(view_dir=[0.05, 0.1, 0.9], light=[0.1, 0.1, 1.0], light_intensity=[1.0, 0.9, 0.8], light_type="point", light_size=1.5)
        H = tile[0] * h
        full = F.cook_torrance(*packed, tile=tile, **kw)
        gout = (torch.rand(B, 3, H, tile[1] * w, generator=g) - 0.3).cuda()
        whole = [t.clone().requires_grad_(True) for t in packed]
        (F.cook_torrance(*whole, tile=tile, **kw) * gout).sum().backward()
        total = [torch.zeros_like(t) for t in packed]
        for y0, y1 in ((0, 4), (4, 13), (13, 19), (19, 22), (22, H)):
            p = F.plan_cook_torrance(*packed, tile=tile, y_offset=y0, rows=y1 - y0, **kw)
            assert p.kernel_name.startswith("ctr_") and torch.equal(p.launch(), full[:, :, y0:y1]), (y0, y1)
            leaves = [t.clone().requires_grad_(True) for t in packed]
            (F.cook_torrance(*leaves, tile=tile, y_offset=y0, rows=y1 - y0, **kw) * gout[:, :, y0:y1]).sum().backward()
            for acc, x in zip(total, leaves):
                assert x.grad.shape == x.shape
                acc += x.grad
        for acc, wl in zip(total, whole):
            assert (acc - wl.grad).abs().max().item() <= 1e-5 * (float(wl.grad.abs().max()) + 1e-12) + 1e-9


def test_fused_blend_backward_over_a_batch_of_tiled_materials():
    """B = 2 materials blended pairwise (each with its own planes and mask) under tile(2): the fused kernel's per-material plane arithmetic; against
    the per-material launches."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(31)
    B, H, W = 2, 12, 32
    mats1 = [_material(g, H, W, "metallic") for _ in range(B)]
    mats2 = [_material(g, H, W, "metallic") for _ in range(B)]
    masks = [torch.rand(1, H, W, generator=g) for _ in range(B)]
    wt = (torch.rand(B, 3, 2 * H, 2 * W, generator=g) - 0.4).cuda()
    kw = dict(view_dir=[0.0, 0.1, 1.0], light=[0.1, 0.1, 1.0], light_intensity=[1.0, 0.9, 0.8], light_type="point", light_size=1.5, tile=2)
    names = ("albedo", "normal", "roughness", "metallic")

    def leaves(ms):
        return [torch.stack([m[k] for m in ms]).cuda().requires_grad_(True) for k in names]
    a1, a2 = leaves(mats1), leaves(mats2)
    dm = torch.stack(masks).cuda().requires_grad_(True)
    out = F.cook_torrance(*a1, blend=(a2[0], a2[1], a2[2], a2[3], None, dm), **kw)
    assert type(out.grad_fn).__name__ == "_FusedBlendFnBackward" and out.shape == (B, 3, 2 * H, 2 * W)
    (out * wt).sum().backward()
    for b in range(B):
        s1 = [mats1[b][k].clone().cuda().requires_grad_(True) for k in names]
        s2 = [mats2[b][k].clone().cuda().requires_grad_(True) for k in names]
        sm = masks[b].clone().cuda().requires_grad_(True)
        one = F.cook_torrance(*s1, blend=(s2[0], s2[1], s2[2], s2[3], None, sm), **kw)
        assert torch.equal(one, out[b].detach())
        (one * wt[b]).sum().backward()
        for x, y in zip(a1 + a2 + [dm], s1 + s2 + [sm]):
            assert torch.equal(x.grad[b], y.grad), b


def test_folded_backward_guard_bands_thin_bands_and_ragged_widths():
    """pbr_cook_torrance_backward_folded through the C ABI with every buffer inside a larger one: NaN around the maps and the upstream gradient (a read
    outside would poison a gradient), a sentinel around the map-sized gradients (a write outside would damage it) -- random map shapes (ragged widths
    included), tiles, and bands of the tiled image of any height and offset (thin bands: the window walk and its zero-fill)."""
    import ctypes
    from pypbr_amd import functional as F, _native as N
    lib = N.lib()
    g = torch.Generator().manual_seed(606)
    G = 256

    def guarded(t, fill):
        flat = torch.full((t.numel() + 2 * G,), fill, dtype=t.dtype, device="cuda")
        flat[G:G + t.numel()] = t.reshape(-1).cuda()
        return flat, flat[G:G + t.numel()].view(t.shape)
    for trial in range(24):
        B = int(torch.randint(1, 3, (1,), generator=g))
        h = int(torch.randint(2, 24, (1,), generator=g))
        w = [4, 8, 12, 24, 40, 6, 14, 30, 38][int(torch.randint(0, 9, (1,), generator=g))]
        ny, nx = int(torch.randint(1, 4, (1,), generator=g)), int(torch.randint(1, 4, (1,), generator=g))
        if (ny, nx) == (1, 1):
            ny = 2
        H, W = ny * h, nx * w
        y0 = int(torch.randint(0, H, (1,), generator=g))
        rows = int(torch.randint(1, H - y0 + 1, (1,), generator=g))
        lights = 1 if trial % 3 else 2
        dtype = torch.float16 if trial % 4 == 3 else torch.float32
        a = torch.rand(B, 3, h, w, generator=g).to(dtype)
        n = torch.cat([torch.rand(B, 2, h, w, generator=g) - 0.5, torch.ones(B, 1, h, w)], 1).to(dtype)
        r = (torch.rand(B, 1, h, w, generator=g) * 0.7 + 0.25).to(dtype)
        m = torch.rand(B, 1, h, w, generator=g).to(dtype)
        kw = dict(view_dir=[0.1, 0, 1], light=[[0.2, -0.1, 0.9], [-0.3, 0.3, 0.7]][:lights], light_intensity=[[1, 0.9, 0.8]] * lights,
                  light_type="point" if trial % 5 else "directional", light_size=2.0, tile=(ny, nx), y_offset=y0, rows=rows)
        bufs, views = zip(*[guarded(t, float("nan")) for t in (a, n, r, m)])
        gbuf, gout = guarded(torch.rand(B, 3, rows, W, generator=g) - 0.3, float("nan"))
        plan = F.plan_cook_torrance(*views, **kw)
        assert lib.pbr_backward_folded_workspace_bytes(ctypes.byref(plan.desc)) == 0
        outs, grads = zip(*[guarded(torch.zeros_like(t), 1.0) for t in (a, n, r, m)])
        for o in outs:
            o[G:-G] = -5.0                                      # (the kernel must write every gradient value)
        st = torch.cuda.current_stream().cuda_stream
        N.check(lib.pbr_cook_torrance_backward_folded(ctypes.byref(plan.desc), gout.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(), grads[2].data_ptr(),
                                                      grads[3].data_ptr(), None, None, st))
        torch.cuda.synchronize()
        tag = (trial, B, h, w, ny, nx, y0, rows, lights, str(dtype))
        for o, gr in zip(outs, grads):
            assert bool((o[:G] == 1.0).all()) and bool((o[-G:] == 1.0).all()), tag
            assert bool(torch.isfinite(gr.float()).all()), tag
        # values: autograd through the materialised repeat, cropped to the band
        ref = [t.detach().clone().float().requires_grad_(True) for t in views]
        kw_ref = {k: v for k, v in kw.items() if k not in ("tile", "y_offset", "rows")}
        img = F.cook_torrance(*[t.repeat(1, 1, ny, nx) for t in ref], **kw_ref)
        (img[:, :, y0:y0 + rows] * gout).sum().backward()
        for gr, rf in zip(grads, ref):
            tol = (1e-5 if dtype == torch.float32 else 4e-3) * (float(rf.grad.abs().max()) + 1e-12) + 1e-9
            assert (gr.float() - rf.grad).abs().max().item() <= tol, tag


@pytest.mark.parametrize("light_type", ["point", "directional"])
@pytest.mark.parametrize("lights", [1, 3])
@pytest.mark.parametrize("workflow,hw,tile,B", [("metallic", (16, 64), (2, 2), None), ("specular", (9, 38), (3, 2), None), ("converted", (12, 40), (2, 3), 2),
                                                ("metallic", (7, 13), (3, 2), None)])
def test_fused_blend_over_tiled_maps_walks_the_source_and_equals_the_wrap_around_form(light_type, lights, workflow, hw, tile, B):
    """Round 6: `cook_torrance(blend=..., tile=n)` -- the blend example's material, lazily blended and tiled -- blends ONCE per texel and evaluates it at
    every repeat (cook_torrance_repeat_blend_kernel) instead of blending at every output pixel (2 x 2048^2 under tile(2): 154 -> 90 us).  Same functions per
    pixel: bit-identical to the wrap-around form (PBR_TUNE_TILE_REPEAT = 0) and to blend -> repeat -> render, whole images and row bands of any height."""
    from pypbr_amd import functional as F
    from pypbr_amd.blending import blend_maps
    (h, w), (ny, nx) = hw, tile
    g = torch.Generator().manual_seed(13 * h + w + lights)
    lead = () if B is None else (B,)

    def material():
        m = [_material(g, h, w, workflow) for _ in range(B or 1)]
        keys = ("albedo", "normal", "roughness", "metallic" if workflow != "specular" else "specular")
        cols = [torch.stack([x[k] for x in m]) if B else m[0][k] for k in keys]
        return [t.cuda() for t in cols]
    a1, n1, r1, x1 = material()
    a2, n2, r2, x2 = material()
    mask = torch.rand(*lead, 1, h, w, generator=g).cuda()
    spec = workflow == "specular"
    first = (a1, n1, r1, None if spec else x1, x1 if spec else None)
    second = (a2, n2, r2, None if spec else x2, x2 if spec else None, mask)
    L = [[0.1, 0.1, 1.0], [-0.4, 0.2, 0.7], [0.3, -0.3, 0.9]][:lights] if light_type == "point" else [[0.3, -0.2, 1.0], [0.1, 0.4, 0.8], [-0.2, 0.1, 1.0]][:lights]
    I = [[1.0, 0.9, 0.8], [0.4, 0.5, 0.6], [0.3, 0.3, 0.3]][:lights]
    kw = dict(view_dir=[0.05, 0.1, 0.9], light=L if lights > 1 else L[0], light_intensity=I if lights > 1 else I[0], light_type=light_type,
              light_size=1.5 if light_type == "point" else None, convert_to_diffuse_specular=(workflow == "converted"))
    full = F.cook_torrance(*first, blend=second, tile=tile, **kw)
    try:
        _knob(0)
        wrap = F.cook_torrance(*first, blend=second, tile=tile, **kw)
    finally:
        _knob(-1)
    assert torch.equal(full, wrap)
    H = ny * h
    for y0, rows in ((0, 1), (h - 1, 2), (1, h - 1), (H - 3, 3), (2, h), (h // 2 + 1, h + 1)):
        rows = min(rows, H - y0)
        band_kw = dict(kw, tile=tile, y_offset=y0, rows=rows)
        got = F.cook_torrance(*first, blend=second, **band_kw)
        assert got.shape[-2:] == (rows, nx * w) and torch.equal(got, full[..., y0:y0 + rows, :]), (y0, rows)


def test_row_walk_downscale_equals_the_strip_kernel_and_aten():
    """Round 6: antialiased down-scales that are NOT a whole factor -- MaterialBase.resize of a 4096^2 texture to 400^2 or 1365^2
    (/root/reference/pypbr/materials/base.py:490-504) -- as a walk down the INPUT rows (csrc/resize_stream.hpp): every row read once, added to the (at most three)
    output rows whose windows hold it, finished rows through LDS to a second wave for the width pass.  The strip kernel's taps in the strip kernel's order:
    BIT-IDENTICAL to it at every factor (knob PBR_TUNE_RESIZE_UP2 = 2 takes the walk wherever the shape allows, 0 the strip form), <= 2e-6 from ATen's
    antialiased interpolate.  The rule (knob 1) takes it from 7 x up; pbr_resize_form says which family serves a call, so the comparison cannot pass on
    one kernel compared with itself.  Shapes: factors 1.02 ... 16.4 that differ per axis, one and several strips / bands, widths that leave the last strip a few
    columns, several planes, windows clipped at all four edges."""
    from pypbr_amd import _native as N
    lib = N.lib()
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(86)

    def run(x, ho, wo, knob):
        planes, hi, wi = x.shape
        out = torch.full((planes, ho, wo), float("nan"), device="cuda")
        ws = torch.empty(max(1, lib.pbr_resize_workspace_bytes(planes, hi, wo) // 4), device="cuda")
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, knob)
        form = lib.pbr_resize_form(x.data_ptr(), out.data_ptr(), planes, hi, wi, ho, wo, 1, ws.data_ptr())
        N.check(lib.pbr_resize_bilinear(x.data_ptr(), out.data_ptr(), planes, hi, wi, ho, wo, 1, ws.data_ptr(), stream))
        torch.cuda.synchronize()
        return out, form

    shapes = [(1, 64, 64, 40, 40), (3, 512, 512, 341, 341), (2, 1024, 1024, 100, 100), (1, 1000, 1024, 333, 700), (3, 256, 2048, 77, 1365),
              (1, 2048, 2048, 1365, 1365), (2, 2048, 2048, 200, 200), (1, 4096, 4096, 400, 400), (1, 2048, 2048, 1500, 1500), (1, 1024, 1024, 1000, 1000),
              (2, 777, 1024, 123, 321), (1, 96, 128, 17, 16), (3, 600, 800, 37, 49), (1, 4096, 4096, 249, 249), (4, 300, 512, 20, 500),
              (1, 2048, 4096, 1990, 1366), (2, 128, 5000, 50, 1234)]
    try:
        for planes, hi, wi, ho, wo in shapes:
            x = torch.rand(planes, hi, wi, generator=g) * 2 - 0.5
            xd = x.cuda()
            walk, f2 = run(xd, ho, wo, 2)
            strip, f0 = run(xd, ho, wo, 0)
            rule, f1 = run(xd, ho, wo, 1)
            many_taps = max(hi / ho, wi / wo) >= 7.0
            assert f2 == N.RESIZE_ROW_WALK and f0 == N.RESIZE_STRIP, (planes, hi, wi, ho, wo, f2, f0)
            assert f1 == (N.RESIZE_ROW_WALK if many_taps else N.RESIZE_STRIP), (planes, hi, wi, ho, wo, f1)
            assert torch.equal(walk, strip) and torch.equal(rule, strip), (planes, hi, wi, ho, wo, float((walk - strip).abs().max()))
            ref = torch.nn.functional.interpolate(x[None], size=(ho, wo), mode="bilinear", align_corners=False, antialias=True)[0]
            assert (walk.cpu() - ref).abs().max().item() <= 2e-6, (planes, hi, wi, ho, wo)
        # an input beyond the 256 MB memory-side cache streams (non-temporal loads): the same sums
        big = torch.rand(5, 4096, 4096, device="cuda")
        for ho in (1365, 400):
            walk, f2 = run(big, ho, ho, 2)
            strip, f0 = run(big, ho, ho, 0)
            assert f2 == N.RESIZE_ROW_WALK and f0 == N.RESIZE_STRIP and torch.equal(walk, strip), ho
            assert run(big, ho, ho, 1)[1] == (N.RESIZE_ROW_WALK if ho == 400 else N.RESIZE_STRIP)      # the rule: the walk from 7 x up
        del big, walk, strip
        # infinities and NaNs poison exactly the outputs whose windows hold them: rows at weight 0 in front of a window's first tap, a slot's last row, do not leak
        x = torch.rand(1, 512, 512, generator=g)
        x[0, 100, 200] = float("inf"); x[0, 300, 17] = float("nan"); x[0, 511, 511] = float("-inf"); x[0, 0, 0] = float("inf")
        for ho in (150, 40):
            walk, f2 = run(x.cuda(), ho, ho, 2)
            strip, _ = run(x.cuda(), ho, ho, 0)
            ref = torch.nn.functional.interpolate(x[None], size=(ho, ho), mode="bilinear", align_corners=False, antialias=True)[0]
            assert f2 == N.RESIZE_ROW_WALK
            assert torch.equal(torch.isfinite(walk), torch.isfinite(strip)) and torch.equal(torch.isfinite(walk).cpu(), torch.isfinite(ref)), ho
            assert torch.equal(walk[torch.isfinite(walk)], strip[torch.isfinite(strip)]), ho
        # what the walk does not take stays with the other families whatever the knob says: rows that are not whole 16-byte pieces, a view off a 16-byte
        # boundary, no antialiasing, an up-scale on one axis, a whole factor (the band walk), a factor within 1 % of 1, more than 36 taps
        flat = torch.rand(3 * 64 * 64 + 1, generator=g).cuda()
        for x, ho, wo, aa, want in ((torch.rand(2, 100, 1022, generator=g).cuda(), 30, 300, 1, N.RESIZE_STRIP), (flat[1:].view(3, 64, 64), 20, 20, 1, N.RESIZE_STRIP),
                                    (torch.rand(1, 512, 512, generator=g).cuda(), 100, 100, 0, N.RESIZE_STRIP), (torch.rand(1, 512, 512, generator=g).cuda(), 100, 600, 1, N.RESIZE_STRIP),
                                    (torch.rand(1, 512, 512, generator=g).cuda(), 128, 128, 1, N.RESIZE_BAND_WALK), (torch.rand(1, 2048, 512, generator=g).cuda(), 2047, 100, 1, N.RESIZE_STRIP), (torch.rand(1, 2048, 2048, generator=g).cuda(), 64, 64, 1, N.RESIZE_TWO_PASS)):
            planes, hi, wi = x.shape
            out = torch.empty(planes, ho, wo, device="cuda")
            ws = torch.empty(max(1, lib.pbr_resize_workspace_bytes(planes, hi, wo) // 4), device="cuda")
            lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 2)
            assert lib.pbr_resize_form(x.data_ptr(), out.data_ptr(), planes, hi, wi, ho, wo, aa, ws.data_ptr()) == want, (hi, wi, ho, wo, aa)
    finally:
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, -1)


def test_row_walk_fuzz():
    """tools/resize_walk_fuzz.py: 120 random shapes (factors 1.02 ... 16.9 that differ per axis, 1-5 planes, ragged bands and strips) through the row walk
    (knob value 2) and the strip kernel (knob value 0), bit for bit; pbr_resize_form tells them apart.  (A walk whose two waves disagreed about a barrier
    would hang: the driver's per-test timeout is the guard.)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("resize_walk_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "resize_walk_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(120, seed=3, verbose=False) >= 80
