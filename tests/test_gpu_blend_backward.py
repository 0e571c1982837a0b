"""The fused blend's own backward kernel (pbr_cook_torrance_blend_backward, round 3): the gradients of BOTH materials' maps and
of the mask of `cook_torrance(blend=...)` in one pass, against float64 autograd through the oracles (blend_oracle: the reference's
blend_with_mask + the re-assignment of the blended normal; torch_oracle: CookTorranceBRDF.forward) and against the unfused
differentiable pieces it replaces.  Covers both settings of the map-global "already signed?" decision (base.py:212), both
workflows, both light types, several lights, batches with shared maps / mask (the sum over the batch), ragged widths, and row
bands with exchanged flags."""
import numpy as np
import pytest
import torch

import blend_oracle as BO
import torch_oracle as O

pytestmark = pytest.mark.gpu


def _material(g, H, W, workflow, flat=False):
    if flat:        # every component of the blended normal positive: the re-assignment reads it as [0,1]-encoded (base.py:214-216)
        n = torch.cat([torch.rand(2, H, W, generator=g) * 0.3 + 0.1, torch.ones(1, H, W)], 0)
    else:
        n = torch.cat([(torch.rand(2, H, W, generator=g) - 0.5), torch.ones(1, H, W)], 0)
    m = {"albedo": torch.rand(3, H, W, generator=g), "normal": n * (0.6 + torch.rand(1, H, W, generator=g)),   # not unit length
         "roughness": torch.rand(1, H, W, generator=g) * 0.6 + 0.35}
    m["metallic" if workflow != "specular" else "specular"] = torch.rand(1 if workflow != "specular" else 3, H, W, generator=g)
    return m


def _reference_grads(m1, m2, mask, wt, view, lights, intens, light_type, light_size, converted=False, y_offset=0, H_total=None,
                     whole=None):
    r1 = {k: v.double().requires_grad_(True) for k, v in m1.items()}
    r2 = {k: v.double().requires_grad_(True) for k, v in m2.items()}
    rm = mask.double().requires_grad_(True)
    bl = BO.blend_materials(r1, r2, rm)
    if whole is not None:            # a row band: the "already signed?" decision is the WHOLE map's
        nb = BO.blend_normals(r1["normal"], r2["normal"], rm)
        bl["normal"] = nb if whole else torch.nn.functional.normalize(nb * 2.0 - 1.0, dim=0)
    kw = dict(view=view.double(), light_type=light_type, light_size=light_size, y_offset=y_offset, H_total=H_total)
    L = lights.double().reshape(-1, 3)
    I = intens.double().reshape(-1, 3)
    if converted:
        ref = O.cook_torrance_converted(bl["albedo"], bl["normal"], bl["roughness"], bl["metallic"], light=L[0], intensity=I[0], **kw)
    elif L.shape[0] > 1:
        ref = O.cook_torrance_multi(bl["albedo"], bl["normal"], bl["roughness"], bl.get("metallic"), bl.get("specular"), lights=L, intensities=I, **kw)
    else:
        ref = O.cook_torrance(bl["albedo"], bl["normal"], bl["roughness"], bl.get("metallic"), bl.get("specular"), light=L[0], intensity=I[0], **kw)
    (ref * wt.double()).sum().backward()
    return ref.detach(), r1, r2, rm


def _check(got, want, what):
    err = (got.cpu().double() - want).abs()
    assert got.shape == want.shape and bool((err <= 2e-5 * (1 + want.abs())).all()), (what, float(err.max()))


@pytest.mark.parametrize("workflow", ["metallic", "specular", "converted"])
@pytest.mark.parametrize("light_type", ["point", "directional"])
@pytest.mark.parametrize("flat", [False, True])
def test_fused_blend_backward_against_float64_autograd(workflow, light_type, flat):
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(100 + 7 * ["metallic", "specular", "converted"].index(workflow) + (3 if flat else 0))
    H, W = 24, 46                                                  # W = 46: two pixels per lane, the last lane of a row complete
    m1, m2 = _material(g, H, W, workflow, flat), _material(g, H, W, workflow, flat)
    mask, wt = torch.rand(1, H, W, generator=g), torch.rand(3, H, W, generator=g) - 0.4
    view = torch.tensor([0.0, 0.1, 1.0])
    light = torch.tensor([0.1, 0.1, 1.0]) if light_type == "point" else torch.tensor([0.3, -0.2, 1.0])
    inten = torch.tensor([1.0, 0.9, 0.8])
    ref, r1, r2, rm = _reference_grads(m1, m2, mask, wt, view, light, inten, light_type, 1.5 if light_type == "point" else None,
                                       converted=(workflow == "converted"))
    assert bool((BO.blend_normals(r1["normal"], r2["normal"], rm).detach().min() < 0)) == (not flat)   # both settings of base.py:212
    d1 = {k: v.clone().cuda().requires_grad_(True) for k, v in m1.items()}
    d2 = {k: v.clone().cuda().requires_grad_(True) for k, v in m2.items()}
    dm = mask.clone().cuda().requires_grad_(True)
    kw = dict(view_dir=view, light=light, light_intensity=inten, light_type=light_type, light_size=1.5 if light_type == "point" else None,
              convert_to_diffuse_specular=(workflow == "converted"), specular_is_srgb=True)
    second = (d2["albedo"], d2["normal"], d2["roughness"], d2.get("metallic"), d2.get("specular"), dm)
    out = F.cook_torrance(d1["albedo"], d1["normal"], d1["roughness"], d1.get("metallic"), d1.get("specular"), blend=second, **kw)
    assert out.requires_grad and type(out.grad_fn).__name__ == "_FusedBlendFnBackward"      # the fused kernels, forward and backward
    assert (out.detach().cpu().double() - ref).abs().max().item() <= 1e-5
    (out * wt.cuda()).sum().backward()
    for name in m1:
        _check(d1[name].grad, r1[name].grad, (workflow, light_type, flat, "material 1", name))
        _check(d2[name].grad, r2[name].grad, (workflow, light_type, flat, "material 2", name))
    _check(dm.grad, rm.grad, "mask")


def test_fused_blend_backward_equals_the_unfused_differentiable_pieces_and_skips_unwanted_gradients():
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(7)
    H, W = 20, 37                                                  # odd width: one pixel per lane
    m1, m2 = _material(g, H, W, "metallic"), _material(g, H, W, "metallic")
    mask, wt = torch.rand(1, H, W, generator=g).cuda(), (torch.rand(3, H, W, generator=g) - 0.4).cuda()
    kw = dict(view_dir=[0.0, 0.1, 1.0], light=[[0.1, 0.1, 1.0], [-0.3, 0.2, 0.8]], light_intensity=[[0.6, 0.5, 0.4], [0.3, 0.3, 0.5]],
              light_type="point", light_size=1.0)

    def run(fused):
        d1 = {k: v.clone().cuda().requires_grad_(True) for k, v in m1.items()}
        d2 = {k: v.clone().cuda().requires_grad_(k != "albedo") for k, v in m2.items()}        # material 2's albedo: no gradient wanted
        dm = mask.clone().requires_grad_(True)
        second = (d2["albedo"], d2["normal"], d2["roughness"], d2["metallic"], None, dm)
        if fused:
            out = F.cook_torrance(d1["albedo"], d1["normal"], d1["roughness"], d1["metallic"], blend=second, **kw)
        else:
            out = F._blend_then_render_with_grad(d1["albedo"], d1["normal"], d1["roughness"], d1["metallic"], None, blend=second, **kw)
        (out * wt).sum().backward()
        return out.detach(), d1, d2, dm
    fo, f1, f2, fm = run(True)
    uo, u1, u2, um = run(False)
    assert (fo - uo).abs().max().item() <= 2e-6
    assert f2["albedo"].grad is None
    for name in m1:
        for a, b in ((f1[name].grad, u1[name].grad), (f2[name].grad, u2[name].grad)):
            if a is None and b is None:
                continue
            assert (a - b).abs().max().item() <= 2e-5 * (1 + float(b.abs().max())), name
    assert (fm.grad - um.grad).abs().max().item() <= 2e-5 * (1 + float(um.grad.abs().max()))


def test_fused_blend_backward_batch_with_shared_second_material_and_mask():
    """[B,C,H,W] first materials blended with ONE second material and ONE mask: those own the sum over the batch."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(9)
    B, H, W = 3, 16, 32
    firsts = [_material(g, H, W, "metallic") for _ in range(B)]
    m2 = _material(g, H, W, "metallic")
    mask, wt = torch.rand(1, H, W, generator=g), torch.rand(B, 3, H, W, generator=g) - 0.4
    view, light, inten = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0])
    sum2 = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in m2.items()}
    summ = torch.zeros(1, H, W, dtype=torch.float64)
    want1 = []
    for b in range(B):
        _, r1, r2, rm = _reference_grads(firsts[b], m2, mask, wt[b], view, light, inten, "point", 1.0)
        want1.append(r1)
        for k in sum2:
            sum2[k] += r2[k].grad
        summ += rm.grad
    d1 = {k: torch.stack([f[k] for f in firsts]).cuda().requires_grad_(True) for k in firsts[0]}
    d2 = {k: v.clone().cuda().requires_grad_(True) for k, v in m2.items()}          # [C,H,W]: shared by the batch
    dm = mask.clone().cuda().requires_grad_(True)
    out = F.cook_torrance(d1["albedo"], d1["normal"], d1["roughness"], d1["metallic"], view_dir=view, light=light, light_intensity=inten,
                          light_type="point", light_size=1.0, blend=(d2["albedo"], d2["normal"], d2["roughness"], d2["metallic"], None, dm))
    (out * wt.cuda()).sum().backward()
    for k in firsts[0]:
        _check(d1[k].grad, torch.stack([w[k].grad for w in want1]), ("batched material 1", k))
        _check(d2[k].grad, sum2[k], ("shared material 2", k))
    _check(dm.grad, summ, "shared mask")


def test_fused_blend_backward_over_a_row_band_with_given_flags():
    """A band of a taller blended map (multi-GPU sharding of one material): with the whole map's flag given, the band's
    gradients equal the rows of the whole map's gradients -- for both values of the flag."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(11)
    H, W, y0, y1 = 40, 32, 12, 29
    for flat in (False, True):
        m1, m2 = _material(g, H, W, "specular", flat), _material(g, H, W, "specular", flat)
        mask, wt = torch.rand(1, H, W, generator=g).cuda(), (torch.rand(3, H, W, generator=g) - 0.4).cuda()
        kw = dict(view_dir=[0.0, 0.1, 1.0], light=[0.1, 0.1, 1.0], light_intensity=[1.0, 0.9, 0.8], light_type="point", light_size=1.0)

        def run(rows):
            sl = slice(None) if rows is None else slice(*rows)
            d1 = {k: v[:, sl].clone().cuda().requires_grad_(True) for k, v in m1.items()}
            d2 = {k: v[:, sl].clone().cuda().requires_grad_(True) for k, v in m2.items()}
            dm = mask[:, sl].clone().requires_grad_(True)
            extra = {} if rows is None else dict(y_offset=rows[0], height_total=H,
                                                 blend_flags=torch.tensor([0 if flat else 1], dtype=torch.int32, device="cuda"))
            out = F.cook_torrance(d1["albedo"], d1["normal"], d1["roughness"], None, d1["specular"],
                                  blend=(d2["albedo"], d2["normal"], d2["roughness"], None, d2["specular"], dm), **kw, **extra)
            (out * wt[:, sl]).sum().backward()
            return out.detach(), d1, d2, dm
        fo, f1, f2, fm = run(None)
        bo, b1, b2, bm = run((y0, y1))
        assert torch.equal(bo, fo[:, y0:y1])
        for k in m1:
            assert torch.equal(b1[k].grad, f1[k].grad[:, y0:y1]) and torch.equal(b2[k].grad, f2[k].grad[:, y0:y1]), (flat, k)
        assert torch.equal(bm.grad, fm.grad[:, y0:y1])


def test_fused_blend_backward_full_size_agrees_with_float64_on_a_crop():
    """One 2048^2 blended material: finite everywhere, deterministic, and a crop against float64 autograd."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(13)
    H = W = 2048
    m1, m2 = _material(g, H, W, "metallic"), _material(g, H, W, "metallic")
    mask, wt = torch.rand(1, H, W, generator=g), torch.rand(3, H, W, generator=g) - 0.4
    view, light, inten = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.3, -0.2, 1.0]), torch.tensor([1.0, 1.0, 1.0])

    def run():
        d1 = {k: v.clone().cuda().requires_grad_(True) for k, v in m1.items()}
        d2 = {k: v.clone().cuda().requires_grad_(True) for k, v in m2.items()}
        dm = mask.clone().cuda().requires_grad_(True)
        out = F.cook_torrance(d1["albedo"], d1["normal"], d1["roughness"], d1["metallic"], view_dir=view, light=light, light_intensity=inten,
                              light_type="directional", blend=(d2["albedo"], d2["normal"], d2["roughness"], d2["metallic"], None, dm))
        (out * wt.cuda()).sum().backward()
        return d1, d2, dm
    d1, d2, dm = run()
    e1, e2, em = run()
    for k in m1:
        assert bool(torch.isfinite(d1[k].grad).all()) and torch.equal(d1[k].grad, e1[k].grad) and torch.equal(d2[k].grad, e2[k].grad)
    assert torch.equal(dm.grad, em.grad)
    ys, xs = slice(H - 24, H), slice(W - 40, W)                     # directional light: a crop is self-contained
    crop = lambda m: {k: v[:, ys, xs].contiguous() for k, v in m.items()}
    _, r1, r2, rm = _reference_grads(crop(m1), crop(m2), mask[:, ys, xs].contiguous(), wt[:, ys, xs].contiguous(), view, light, inten,
                                     "directional", None, whole=True)
    for k in m1:
        _check(d1[k].grad[:, ys, xs], r1[k].grad, ("2048^2 crop, material 1", k))
        _check(d2[k].grad[:, ys, xs], r2[k].grad, ("2048^2 crop, material 2", k))
    _check(dm.grad[:, ys, xs], rm.grad, "2048^2 crop, mask")
