"""Round 5 on the GPU box: ADVICE r4 fixes on the device side; the repeat-inner backward / loss step; the repeat-inner walk with
several lights (added as they are built)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ARGS = (torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0]), 1.0)


def _maps(H, W, seed=0, device="cuda", dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    n = torch.cat([(torch.rand(2, H, W, generator=g) - 0.5), torch.ones(1, H, W)], 0)
    n = n / n.norm(dim=0, keepdim=True)
    maps = dict(albedo=torch.rand(3, H, W, generator=g), normal=n, roughness=torch.rand(1, H, W, generator=g) * 0.8 + 0.2,
                metallic=torch.rand(1, H, W, generator=g))
    return {k: v.to(device=device, dtype=dtype) for k, v in maps.items()}


def test_material_moves_between_gpus_by_device_copy():
    """to('cuda:1') of a material on cuda:0 must not take the host staging memcpy."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two ROCm devices")
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    m = BasecolorMetallicMaterial(**_maps(40, 64, seed=4, device="cuda:0"))
    want = CookTorranceBRDF("point")(m, *ARGS).cpu()
    m.to("cuda:1")
    assert all(t.device == torch.device("cuda", 1) for t in m._raw.values())
    assert torch.equal(CookTorranceBRDF("point")(m, *ARGS).cpu(), want)


def test_map_assigned_after_tile_on_a_device_material_keeps_its_size():
    """ADVICE r4 (medium), device side: load -> to(cuda) -> tile(2) records the repeat; a map assigned afterwards is not repeated."""
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    host = _maps(16, 24, seed=2, device="cpu")
    m = BasecolorMetallicMaterial(**host)
    m.resize((16, 24))                                 # first whole-material operation: the maps now wait on the device
    m.tile(2)
    assert m.lazy_tile == (2, 2)
    r = torch.full((1, 32, 48), 0.5)
    m.roughness = r
    assert m.lazy_tile == (1, 1) and m.size == (32, 48)
    ref = BasecolorMetallicMaterial(albedo=host["albedo"].repeat(1, 2, 2), normal=None, roughness=r, metallic=host["metallic"].repeat(1, 2, 2))
    ref._maps["normal"] = host["normal"].repeat(1, 2, 2)
    brdf = CookTorranceBRDF("point")
    got, want = brdf(m, *ARGS), brdf(ref, *ARGS)
    assert got.shape == (3, 32, 48) and torch.allclose(got, want, atol=2e-6)


# ---------------------------------------------------------------- the repeat-inner backward
import torch.nn.functional as TF

import torch_oracle as O


def _leaf_maps(g, H, W, workflow, dtype=torch.float32, B=None):
    lead = () if B is None else (B,)
    a = torch.rand(*lead, 3, H, W, generator=g)
    n = torch.cat([(torch.rand(*lead, 2, H, W, generator=g) - 0.5) * 1.4, torch.ones(*lead, 1, H, W)], -3)      # un-normalised on purpose
    r = torch.rand(*lead, 1, H, W, generator=g) * 0.7 + 0.25
    m = torch.rand(*lead, 1, H, W, generator=g) if workflow != "specular" else None
    s = torch.rand(*lead, 3, H, W, generator=g) * 0.6 if workflow == "specular" else None
    return [None if t is None else t.to(dtype) for t in (a, n, r, m, s)]


REPEAT_CASES = [
    # workflow, light_type, (h, w), (ny, nx), dtype, batch
    ("metallic", "point", (24, 48), (2, 2), torch.float32, None),
    ("metallic", "directional", (16, 64), (3, 2), torch.float32, None),
    ("specular", "point", (20, 40), (2, 3), torch.float32, None),
    ("converted", "directional", (12, 36), (2, 2), torch.float32, None),
    ("converted", "point", (10, 128), (4, 1), torch.float32, None),
    ("metallic", "point", (7, 260), (1, 2), torch.float32, 2),           # ragged tiles (130 lanes a row), a batch
    ("specular", "directional", (9, 24), (2, 2), torch.float32, 3),
    ("metallic", "point", (16, 64), (2, 2), torch.float16, None),
    ("specular", "point", (12, 48), (3, 1), torch.float16, 2),
]


def _tiled_grads(F, maps, kw, tile, gout, knob, binding="torch_op"):
    from pypbr_amd import _native as N
    leaves = [None if t is None else t.detach().clone().cuda().requires_grad_(True) for t in maps]
    before = F.USE_TORCH_OPS
    try:
        F.USE_TORCH_OPS = binding == "torch_op"          # the two bindings of the same C ABI: torch.ops.pbr_hip.* | ctypes
        N.lib().pbr_set_tuning(N.TUNE_TILE_REPEAT, knob)
        out = F.cook_torrance(*leaves, tile=tile, **kw)
        (out * gout).sum().backward()
    finally:
        N.lib().pbr_set_tuning(N.TUNE_TILE_REPEAT, -1)
        F.USE_TORCH_OPS = before
    return out.detach(), [None if t is None else t.grad for t in leaves]


def _equal_to_rounding(x, y, what=""):
    """Round 6: the one-kernel folded backward sums the ADJOINTS of a texel's positions (fused multiply-adds) and runs the light-independent
    tail of the chain rule once -- the tail is linear in them, with coefficients that do not depend on the position -- instead of adding up
    per-position gradients.  Equal to the two-kernel form in real arithmetic, to fp32 rounding in practice (never to the bit)."""
    err = (x.float() - y.float()).abs().max().item()
    assert err <= 4e-6 * (float(y.float().abs().max()) + 1e-12) + 1e-9, (what, err, float(y.float().abs().max()))


@pytest.mark.parametrize("binding", ["torch_op", "ctypes"])
@pytest.mark.parametrize("workflow,light_type,hw,tile,dtype,B", REPEAT_CASES)
def test_repeat_inner_backward_equals_backward_plus_fold_and_float64_autograd(workflow, light_type, hw, tile, dtype, B, binding):
    """pbr_cook_torrance_backward_folded: ONE kernel walks the maps and accumulates every texel's gradient over its repeats in
    registers.  fp32 maps: equal to pbr_cook_torrance_backward + pbr_fold_gradient (PBR_TUNE_TILE_REPEAT = 0 is that form) to fp32
    rounding -- point light: the adjoints of a texel's positions are summed and the light-independent tail of the chain rule runs once
    (round 6); directional light: the upstream values of its repeats are summed first and the texel differentiated once; fp16 maps: the
    sum is rounded once instead of per repeat.  And against float64 autograd of the reference's ops through
    map.repeat(1, ny, nx) (MaterialBase.tile, base.py:524-537)."""
    from pypbr_amd import functional as F
    (h, w), (ny, nx) = hw, tile
    g = torch.Generator().manual_seed(31 * h + w + ny)
    maps = _leaf_maps(g, h, w, workflow, dtype, B)
    view = torch.tensor([0.05, 0.1, 0.9])
    light = torch.tensor([0.1, 0.1, 1.0]) if light_type == "point" else torch.tensor([0.3, -0.2, 1.0])
    inten = torch.tensor([1.0, 0.9, 0.8])
    size = 1.5 if light_type == "point" else None
    kw = dict(view_dir=view, light=light, light_intensity=inten, light_type=light_type, light_size=size,
              convert_to_diffuse_specular=(workflow == "converted"))
    lead = () if B is None else (B,)
    gout = (torch.rand(*lead, 3, ny * h, nx * w, generator=g) - 0.3).cuda()
    out1, one = _tiled_grads(F, maps, kw, tile, gout, -1, binding)
    out0, two = _tiled_grads(F, maps, kw, tile, gout, 0, binding)
    assert torch.equal(out1, out0)
    # the one-kernel form really is what ran: its launches need no workspace, the two-kernel form's do
    import ctypes
    from pypbr_amd import _native as N
    plan = F.plan_cook_torrance(*[None if t is None else t.cuda() for t in maps], tile=tile, **kw)
    assert N.lib().pbr_backward_folded_workspace_bytes(ctypes.byref(plan.desc)) == 0
    wrap = F.plan_cook_torrance(*[None if t is None else t.cuda() for t in maps], tile=tile, tuning=dict(tile_repeat=0), **kw)      # (kept alive: the descriptor points into it)
    assert N.lib().pbr_backward_folded_workspace_bytes(ctypes.byref(wrap.desc)) > 0
    for name, x, y in zip(("albedo", "normal", "roughness", "metallic", "specular"), one, two):
        if x is None:
            assert y is None
            continue
        assert x.shape == y.shape and x.dtype == dtype and bool(torch.isfinite(x.float()).all()), name
        if dtype == torch.float32:            # point: the positions' adjoints are summed before the tail; directional: their upstream values before the chain rule
            _equal_to_rounding(x, y, name)
        else:
            assert (x.float() - y.float()).abs().max().item() <= 2e-3 * (float(y.float().abs().max()) + 1e-12) + 1e-6, name
    # float64 autograd through repeat()
    rep = (1,) * (len(lead) + 1) + (ny, nx)
    leaves = [None if t is None else t.float().double().requires_grad_(True) for t in maps]
    outs = []
    for b in range(B or 1):
        args = [None if t is None else (t if B is None else t[b]).repeat(1, ny, nx) for t in leaves]
        okw = dict(view=view.double(), light=light.double(), intensity=inten.double(), light_type=light_type, light_size=size)
        if workflow == "converted":
            outs.append(O.cook_torrance_converted(args[0], args[1], args[2], args[3], **okw))
        else:
            outs.append(O.cook_torrance(*args, **okw))
    ref = torch.stack(outs) if B is not None else outs[0]
    (ref * gout.cpu().double()).sum().backward()
    for name, x, y in zip(("albedo", "normal", "roughness", "metallic", "specular"), one, leaves):
        if x is None:
            continue
        err = (x.float().cpu().double() - y.grad).abs()
        tol = (2e-5 if dtype == torch.float32 else 2e-3) * (1 + y.grad.abs())
        assert bool((err <= tol).all()), (name, float(err.max()))
    del rep


def test_repeat_inner_backward_dispatch_and_fallbacks():
    """Which launches the one-kernel form serves is the library's decision (pbr_backward_folded_workspace_bytes == 0): map rows that hold a
    4-texel lane, one or several lights -- ragged widths included since round 6 (the last lane of a row moves back).  Map rows SHORTER than 4
    texels go through the workspace -- same gradients as autograd of the materialised repeat."""
    from pypbr_amd import functional as F, _native as N
    import ctypes
    g = torch.Generator().manual_seed(9)
    lib = N.lib()
    for (h, w), lights, served in (((8, 32), 1, True), ((8, 30), 1, True), ((8, 3), 1, False), ((8, 32), 2, True), ((8, 30), 2, True), ((8, 3), 2, False)):
        a, n, r, m, _ = [None if t is None else t.cuda() for t in _leaf_maps(g, h, w, "metallic")]
        L = [[0.1, 0.1, 1.0], [-0.3, 0.2, 0.8]][:lights]
        I = [[1.0, 0.9, 0.8], [0.5, 0.5, 0.5]][:lights]
        kw = dict(view_dir=[0.0, 0.1, 1.0], light=L if lights > 1 else L[0], light_intensity=I if lights > 1 else I[0], light_type="point", light_size=1.0)
        plan = F.plan_cook_torrance(a, n, r, m, tile=2, **kw)
        ws = lib.pbr_backward_folded_workspace_bytes(ctypes.byref(plan.desc))
        assert (ws == 0) == served, ((h, w), lights, ws)
        if not served:
            assert ws == 8 * 4 * (2 * h) * (2 * w)              # output-sized gradients of 8 planes, fp32
        leaves = [t.clone().requires_grad_(True) for t in (a, n, r, m)]
        gout = torch.rand(3, 2 * h, 2 * w, generator=g).cuda()
        (F.cook_torrance(*leaves, tile=2, **kw) * gout).sum().backward()
        mats = [t.clone().requires_grad_(True) for t in (a, n, r, m)]
        (F.cook_torrance(*[t.repeat(1, 2, 2) for t in mats], **kw) * gout).sum().backward()
        for x, y in zip(leaves, mats):
            assert x.grad.shape == x.shape
            assert (x.grad - y.grad).abs().max().item() <= 1e-5 * (float(y.grad.abs().max()) + 1e-12) + 1e-9


@pytest.mark.parametrize("workflow,light_type,hw,tile,dtype", [("metallic", "point", (24, 48), (2, 2), torch.float32),
                                                             ("specular", "directional", (16, 40), (2, 3), torch.float32),
                                                             ("converted", "point", (10, 64), (3, 1), torch.float32),
                                                             ("metallic", "point", (7, 260), (2, 2), torch.float32),
                                                             ("metallic", "point", (16, 64), (2, 2), torch.float16)])
def test_loss_step_over_tiled_maps_is_one_pass(workflow, light_type, hw, tile, dtype):
    """pbr_cook_torrance_mse_step with tiled maps (ABI 7): loss and MAP-sized gradients from one pass over the maps and the target --
    against float64 autograd of MSELoss(brdf(material.tile(n)), target) (06_advanced.rst:73-107 over examples/example_brdf.py:11's
    material) and against the three-step path (evaluate, torch's MSE, folded backward)."""
    from pypbr_amd import functional as F
    (h, w), (ny, nx) = hw, tile
    g = torch.Generator().manual_seed(77 + h + w)
    maps = _leaf_maps(g, h, w, workflow, dtype)
    target = torch.rand(3, ny * h, nx * w, generator=g)
    view = torch.tensor([0.05, 0.1, 0.9])
    light = torch.tensor([0.1, 0.1, 1.0]) if light_type == "point" else torch.tensor([0.3, -0.2, 1.0])
    inten = torch.tensor([1.0, 0.9, 0.8])
    size = 1.5 if light_type == "point" else None
    kw = dict(view_dir=view, light=light, light_intensity=inten, light_type=light_type, light_size=size,
              convert_to_diffuse_specular=(workflow == "converted"))
    leaves = [None if t is None else t.cuda().requires_grad_(True) for t in maps]
    loss = F.rendering_loss_mse(*leaves, target=target.cuda(), tile=tile, **kw)
    assert loss.shape == () and type(loss.grad_fn).__name__ == "_MseStepFnBackward"
    loss.backward()
    ref_leaves = [None if t is None else t.float().double().requires_grad_(True) for t in maps]
    args = [None if t is None else t.repeat(1, ny, nx) for t in ref_leaves]
    okw = dict(view=view.double(), light=light.double(), intensity=inten.double(), light_type=light_type, light_size=size)
    ref = O.cook_torrance_converted(args[0], args[1], args[2], args[3], **okw) if workflow == "converted" else O.cook_torrance(*args, **okw)
    want = TF.mse_loss(ref, target.double())
    want.backward()
    assert abs(loss.item() - want.item()) <= 1e-6 * (1 + want.item())
    for name, x, y in zip(("albedo", "normal", "roughness", "metallic", "specular"), leaves, ref_leaves):
        if x is None:
            continue
        assert x.grad.dtype == dtype and x.grad.shape == x.shape
        err = (x.grad.float().cpu().double() - y.grad).abs()
        scale = float(y.grad.abs().max())
        assert float(err.max()) <= (2e-5 if dtype == torch.float32 else 2e-3) * (scale + 1e-12) + 1e-9, (name, float(err.max()), scale)
    again = [None if t is None else t.detach().clone().requires_grad_(True) for t in leaves]
    unfused = TF.mse_loss(F.cook_torrance(*again, tile=tile, **kw), target.cuda())
    unfused.backward()
    assert abs(unfused.item() - loss.item()) <= 2e-6 * (1 + loss.item())
    for x, y in zip(leaves, again):
        if x is not None:
            d = (x.grad.float() - y.grad.float()).abs().max().item()
            assert d <= (2e-5 if dtype == torch.float32 else 2e-3) * (float(y.grad.float().abs().max()) + 1e-12) + 1e-9
    # several lights: not one pass, same numbers through the three steps
    multi = dict(kw, light=torch.stack([light, light * 0.8 + 0.1]), light_intensity=torch.stack([inten, inten * 0.5]))
    more = [None if t is None else t.detach().clone().requires_grad_(True) for t in leaves]
    l2 = F.rendering_loss_mse(*more, target=target.cuda(), tile=tile, **multi)
    assert type(l2.grad_fn).__name__ != "_MseStepFnBackward"
    l2.backward()
    assert all(t is None or bool(torch.isfinite(t.grad.float()).all()) for t in more)


def test_rendering_loss_on_the_examples_material_takes_the_one_pass_path():
    """examples/example_brdf.py:11's material -- resize(512).tile(2) -- inside docs/source/tutorials/06_advanced.rst:73-107's loss: the
    recorded tile reaches the loss step, whose gradients are map-sized and equal float64 autograd through the materialised repeat."""
    from pypbr_amd.losses import RenderingLoss
    from pypbr_amd.materials import BasecolorMetallicMaterial
    g = torch.Generator().manual_seed(3)
    h = w = 64
    a, n, r, m, _ = _leaf_maps(g, h, w, "metallic")
    leaves = [t.cuda().requires_grad_(True) for t in (a, n, r, m)]
    pred = BasecolorMetallicMaterial(albedo=leaves[0], roughness=leaves[2], metallic=leaves[3], device="cuda")
    pred._raw["normal"] = leaves[1]
    pred.tile(2)
    assert pred.lazy_tile == (2, 2) and pred.size == (128, 128)
    target = torch.rand(3, 2 * h, 2 * w, generator=g).cuda()
    crit = RenderingLoss(light_type="point", light_size=1.0)
    loss = crit(pred, target)
    assert type(loss.grad_fn).__name__ == "_MseStepFnBackward"
    loss.backward()
    ref_leaves = [t.double().requires_grad_(True) for t in (a, n, r, m)]
    ref = O.cook_torrance(*[t.repeat(1, 2, 2) for t in ref_leaves], None, view=torch.tensor([0.0, 0.0, 1.0]).double(),
                          light=torch.tensor([0.1, 0.1, 1.0]).double(), intensity=torch.ones(3).double(), light_type="point", light_size=1.0)
    want = TF.mse_loss(ref, target.cpu().double())
    want.backward()
    assert abs(loss.item() - want.item()) <= 1e-6 * (1 + want.item())
    for x, y in zip(leaves, ref_leaves):
        assert x.grad.shape == x.shape
        assert (x.grad.cpu().double() - y.grad).abs().max().item() <= 2e-5 * (float(y.grad.abs().max()) + 1e-12) + 1e-9


# ---------------------------------------------------------------- the repeat-inner walk with several lights
@pytest.mark.parametrize("workflow,light_type,hw,tile,dtype,out_dtype", [("metallic", "point", (24, 64), (2, 2), torch.float32, torch.float32),
                                                                       ("specular", "directional", (16, 40), (3, 2), torch.float32, torch.float32),
                                                                       ("converted", "point", (10, 128), (2, 3), torch.float16, torch.float32),
                                                                       ("metallic", "point", (12, 48), (2, 2), torch.float16, torch.float16)])
def test_repeat_inner_walk_serves_several_lights_bit_identically(workflow, light_type, hw, tile, dtype, out_dtype):
    """Several lights over tiled maps used to take the wrap-around form (every texel read and decoded once per repeat, the second
    read past L2: 1.40 x the maps from HBM).  The repeat-inner kernel now loops the lights inside each position: same bits as the
    wrap-around form (PBR_TUNE_TILE_REPEAT = 0) and as the materialised repeat (MaterialBase.tile, base.py:524-537)."""
    from pypbr_amd import functional as F
    (h, w), (ny, nx) = hw, tile
    g = torch.Generator().manual_seed(5 * h + w)
    maps = [None if t is None else t.cuda() for t in _leaf_maps(g, h, w, workflow, dtype)]
    L = torch.tensor([[0.1, 0.1, 1.0], [-0.4, 0.2, 0.7], [0.3, -0.3, 0.9]])
    I = torch.tensor([[1.0, 0.9, 0.8], [0.4, 0.5, 0.6], [0.3, 0.3, 0.3]])
    kw = dict(view_dir=[0.05, 0.1, 0.9], light=L, light_intensity=I, light_type=light_type, light_size=1.5 if light_type == "point" else None,
              convert_to_diffuse_specular=(workflow == "converted"), out_dtype=out_dtype)
    plan = F.plan_cook_torrance(*maps, tile=tile, **kw)
    assert plan.kernel_name.startswith("ctr_") and plan.kernel_name.endswith("_multi")
    got = plan.launch().clone()
    wrap = F.plan_cook_torrance(*maps, tile=tile, tuning=dict(tile_repeat=0), **kw)
    assert not wrap.kernel_name.startswith("ctr_")
    assert torch.equal(got, wrap.launch())
    full = F.cook_torrance(*[None if t is None else t.repeat(1, ny, nx) for t in maps], **kw)
    assert torch.equal(got.reshape(full.shape), full)


# ---------------------------------------------------------------- plan reuse in CookTorranceBRDF.__call__
def test_plan_reuse_is_invisible_except_for_its_speed():
    """A device-resident material evaluated again reuses the filled descriptor of its last call (models.CookTorranceBRDF._reuse_plan).
    Whatever changes between two calls must be seen: light / view values edited in place, maps edited in place, maps replaced, flags,
    another light count; every result a fresh tensor; gradients never take the shortcut."""
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    maps = _maps(40, 64, seed=11)
    mat = BasecolorMetallicMaterial(**maps)
    brdf = CookTorranceBRDF("point")
    view, light, inten = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0])

    def reference(**over):
        CookTorranceBRDF.PLAN_REUSE = False
        try:
            return brdf(mat, over.get("view", view), over.get("light", light), over.get("inten", inten), over.get("size", 1.0), over.get("srgb", True)).clone()
        finally:
            CookTorranceBRDF.PLAN_REUSE = True
    want = reference()
    outs = [brdf(mat, view, light, inten, 1.0) for _ in range(4)]           # 1st: seen, 2nd: plan built, 3rd and 4th: reused
    assert "_plan_cache" in mat.__dict__ and mat.__dict__["_plan_cache"][1].out is None
    assert all(torch.equal(o, want) for o in outs) and len({o.data_ptr() for o in outs}) == 4
    light[0] = -0.3                                                             # a CPU tensor edited in place
    assert torch.equal(brdf(mat, view, light, inten, 1.0), reference())
    lst = [0.2, 0.0, 1.0]
    assert torch.equal(brdf(mat, lst, light, inten, 1.0), reference(view=lst))
    lst[0] = -0.2                                                               # a Python list edited in place
    assert torch.equal(brdf(mat, lst, light, inten, 1.0), reference(view=lst))
    mat._raw["roughness"].mul_(0.5)                                             # a map edited in place: the kernel reads the memory as it is
    assert torch.equal(brdf(mat, view, light, inten, 1.0), reference())
    mat.roughness = torch.rand(1, 40, 64, generator=torch.Generator().manual_seed(1)).cuda() * 0.5 + 0.3        # a map replaced
    assert torch.equal(brdf(mat, view, light, inten, 1.0), reference())
    assert torch.equal(brdf(mat, view, light, inten, 1.0), reference())
    assert torch.equal(brdf(mat, view, light, inten, 2.0), reference(size=2.0))                                   # another light size
    assert torch.equal(brdf(mat, view, light, inten, 1.0, False), reference(srgb=False))
    two = torch.tensor([[0.1, 0.1, 1.0], [-0.3, 0.2, 0.8]])
    for _ in range(3):
        assert torch.equal(brdf(mat, view, two, inten, 1.0), reference(light=two))
    for _ in range(3):
        assert torch.equal(brdf(mat, view, light, inten, 1.0), reference())
    mat.to_linear()
    for _ in range(3):
        assert torch.equal(brdf(mat, view, light, inten, 1.0), reference())
    tiled = BasecolorMetallicMaterial(**_maps(16, 32, seed=12)).tile(2)
    ref_t = None
    for _ in range(3):
        got = brdf(tiled, view, light, inten, 1.0)
        ref_t = got if ref_t is None else ref_t
        assert got.shape == (3, 32, 64) and torch.equal(got, ref_t)
    # gradients: the general path (autograd needs its graph)
    leaf = mat._raw["albedo"].clone().requires_grad_(True)
    mat.albedo = leaf
    for _ in range(3):
        out = brdf(mat, view, light, inten, 1.0)
        assert out.requires_grad
    out.sum().backward()
    assert leaf.grad is not None and bool(torch.isfinite(leaf.grad).all())
    # a map whose rows are strided is evaluated through a copy: no plan is kept, edits of the original are seen
    wide = torch.rand(1, 40, 128, generator=torch.Generator().manual_seed(2)).cuda() * 0.5 + 0.3
    s_mat = BasecolorMetallicMaterial(**_maps(40, 64, seed=13))
    s_mat._raw["roughness"] = wide[:, :, ::2]
    first = [brdf(s_mat, view, light, inten, 1.0) for _ in range(3)][-1]
    assert "_plan_cache" not in s_mat.__dict__
    wide.mul_(0.5)
    assert not torch.equal(brdf(s_mat, view, light, inten, 1.0), first)
    # a clone does not share the plan
    c = BasecolorMetallicMaterial(**maps)
    for _ in range(3):
        brdf(c, view, light, inten, 1.0)
    assert "_plan_cache" not in c.clone().__dict__


def test_eager_call_equals_its_graph_capture_and_the_host_overhead_is_recorded():
    """BASELINE configs[0] (one 256^2 material through CookTorranceBRDF.__call__): the call of a device-resident material can be captured
    in a HIP graph and the replay writes the same values.  The eager / replay times per call (13.3 against 9.8 us on an idle box,
    bench.py --config 1) are WRITTEN to gpurun_out/host_overhead_256.json, not asserted: no wall-clock figure may turn this suite red."""
    import statistics
    import time
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    mat = BasecolorMetallicMaterial(**_maps(256, 256, seed=1))
    brdf = CookTorranceBRDF("point")
    view, light, inten = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0])
    call = lambda: brdf(mat, view, light, inten, 1.0)      # noqa: E731

    def per_call(fn):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        times = []
        for _ in range(7):
            t0 = time.perf_counter()
            for _ in range(200):
                fn()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / 200 * 1e6)
        return statistics.median(times)
    eager = per_call(call)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = call()
    replay = per_call(graph.replay)
    assert torch.equal(captured, call())
    print(f"\n[256^2 call] eager {eager:.1f} us, graph replay {replay:.1f} us")
    import json
    import os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "host_overhead_256.json"), "w") as f:
            json.dump({"eager_us_per_call": round(eager, 2), "graph_replay_us_per_call": round(replay, 2)}, f)
    except OSError:
        pass


@pytest.mark.parametrize("binding", ["torch_op", "ctypes"])
@pytest.mark.parametrize("light_type,hw,tile,band", [("point", (16, 64), (3, 2), (5, 30)), ("point", (8, 32), (4, 1), (8, 16)),
                                                     ("directional", (12, 48), (2, 2), (3, 20)), ("point", (16, 64), (3, 2), (20, 10))])
def test_folded_gradients_of_a_row_band_of_a_tiled_image(binding, light_type, hw, tile, band):
    """A row band of the tiled image (a multi-GPU shard, `y_offset` / `rows`): the folded gradient of a texel is the sum over its repeats
    INSIDE the band (the ranks' partial sums add up to the whole).  Bands that hold a full period of the map's rows take the repeat-inner
    kernel with its per-lane band test; thinner bands (round 6) walk the cyclic window of source rows they touch, the other texels'
    gradients zeroed -- here: against autograd through the materialised repeat, cropped to the band."""
    from pypbr_amd import functional as F
    (h, w), (ny, nx), (y0, rows) = hw, tile, band
    g = torch.Generator().manual_seed(7 * h + y0)
    maps = [t.cuda() for t in _leaf_maps(g, h, w, "metallic")[:4]]
    kw = dict(view_dir=[0.05, 0.1, 0.9], light=[0.1, 0.1, 1.0] if light_type == "point" else [0.3, -0.2, 1.0], light_intensity=[1.0, 0.9, 0.8],
              light_type=light_type, light_size=1.5 if light_type == "point" else None)
    gout = (torch.rand(3, rows, nx * w, generator=g) - 0.3).cuda()
    ref_leaves = [t.clone().requires_grad_(True) for t in maps]
    full = F.cook_torrance(*[t.repeat(1, ny, nx) for t in ref_leaves], **kw)
    (full[:, y0:y0 + rows] * gout).sum().backward()
    leaves = [t.clone().requires_grad_(True) for t in maps]
    before = F.USE_TORCH_OPS
    try:
        F.USE_TORCH_OPS = binding == "torch_op"
        out = F.cook_torrance(*leaves, tile=tile, y_offset=y0, rows=rows, **kw)      # (rows < h: round 6 serves thin bands too -- the windowed walk)
        assert torch.equal(out, full[:, y0:y0 + rows].detach())
        (out * gout).sum().backward()
    finally:
        F.USE_TORCH_OPS = before
    for name, x, y in zip(("albedo", "normal", "roughness", "metallic"), leaves, ref_leaves):
        assert x.grad.shape == x.shape
        assert (x.grad - y.grad).abs().max().item() <= 1e-5 * (float(y.grad.abs().max()) + 1e-12) + 1e-9, name


def test_repeat_inner_backward_partial_requests_no_normal_and_strided_batches():
    """Corners of the one-kernel folded backward: only some gradients wanted (the others are not written), a material without a normal
    map (+Z, cooktorrance.py:147-152), a batch whose materials sit material-major in one arena (per-lane plane addresses), odd map
    heights -- each against the two-kernel form (equal to fp32 rounding: _equal_to_rounding)."""
    from pypbr_amd import functional as F, _native as N
    g = torch.Generator().manual_seed(21)
    kw = dict(view_dir=[0.05, 0.1, 0.9], light=[0.1, 0.1, 1.0], light_intensity=[1.0, 0.9, 0.8], light_type="point", light_size=1.5)

    def grads(maps, wanted, tile, knob, **extra):
        leaves = [None if t is None else t.detach().clone().requires_grad_(w) for t, w in zip(maps, wanted)]
        try:
            N.lib().pbr_set_tuning(N.TUNE_TILE_REPEAT, knob)
            out = F.cook_torrance(*leaves, tile=tile, **dict(kw, **extra))
            gout = torch.rand(out.shape, generator=torch.Generator().manual_seed(3)).cuda() - 0.3
            (out * gout).sum().backward()
        finally:
            N.lib().pbr_set_tuning(N.TUNE_TILE_REPEAT, -1)
        return [None if t is None else t.grad for t in leaves]

    def same(x, y):
        for a, b in zip(x, y):
            if a is None or b is None:
                if a is not b:
                    return False
                continue
            _equal_to_rounding(a, b)
        return True
    a, n, r, m, _ = [None if t is None else t.cuda() for t in _leaf_maps(g, 13, 40, "metallic")]
    for wanted in ((True, False, False, False), (False, True, True, False), (False, False, False, True)):
        one, two = grads((a, n, r, m), wanted, (2, 3), -1), grads((a, n, r, m), wanted, (2, 3), 0)
        assert same(one, two) and [x is not None for x in one] == list(wanted), wanted
    one, two = grads((a, None, r, m), (True, False, True, True), 2, -1), grads((a, None, r, m), (True, False, True, True), 2, 0)
    assert same(one, two) and one[1] is None
    B = 3
    batched = [t.cuda() for t in _leaf_maps(g, 9, 24, "metallic", B=B)[:4]]
    packed = F.pack_maps(*batched, material_major=True)    # material-major: materials one pitch apart, no scalar plane addresses
    assert packed[0].stride(0) != packed[0][0].numel()
    one, two = grads(packed, (True,) * 4, (2, 2), -1), grads(packed, (True,) * 4, (2, 2), 0)
    assert same(one, two) and one[0].shape == (B, 3, 9, 24)
    lin = grads((a, n, r, m), (True,) * 4, 2, -1, albedo_is_srgb=False, return_srgb=False)
    assert same(lin, grads((a, n, r, m), (True,) * 4, 2, 0, albedo_is_srgb=False, return_srgb=False))


@pytest.mark.parametrize("binding", ["torch_op", "ctypes"])
@pytest.mark.parametrize("workflow,light_type,hw,tile,dtype", [("metallic", "point", (12, 48), (2, 2), torch.float32),
                                                             ("specular", "directional", (9, 40), (3, 2), torch.float32),
                                                             ("converted", "point", (8, 64), (2, 3), torch.float16)])
def test_repeat_inner_backward_with_several_lights(binding, workflow, light_type, hw, tile, dtype):
    """Several lights over tiled maps: the one-kernel folded backward runs backward_body_to's two passes over the lights per position
    (the summed colour decides the outer clamp and the encode's slope, then every light's chain rule): fp32 equal to the
    wrap-around backward + fold to rounding, and against float64 autograd of the reference's ops through repeat()."""
    from pypbr_amd import functional as F
    (h, w), (ny, nx) = hw, tile
    g = torch.Generator().manual_seed(17 * h + w)
    maps = _leaf_maps(g, h, w, workflow, dtype)
    view = torch.tensor([0.05, 0.1, 0.9])
    L = torch.tensor([[0.1, 0.1, 1.0], [-0.4, 0.2, 0.7], [0.3, -0.3, 0.9]])
    I = torch.tensor([[1.0, 0.9, 0.8], [0.4, 0.5, 0.6], [0.3, 0.3, 0.3]])
    size = 1.5 if light_type == "point" else None
    kw = dict(view_dir=view, light=L, light_intensity=I, light_type=light_type, light_size=size, convert_to_diffuse_specular=(workflow == "converted"))
    gout = (torch.rand(3, ny * h, nx * w, generator=g) - 0.3).cuda()
    out1, one = _tiled_grads(F, maps, kw, tile, gout, -1, binding)
    out0, two = _tiled_grads(F, maps, kw, tile, gout, 0, binding)
    assert torch.equal(out1, out0)
    for name, x, y in zip(("albedo", "normal", "roughness", "metallic", "specular"), one, two):
        if x is None:
            continue
        if dtype == torch.float32:
            _equal_to_rounding(x, y, name)
        else:
            assert (x.float() - y.float()).abs().max().item() <= 2e-3 * (float(y.float().abs().max()) + 1e-12) + 1e-6, name
    leaves = [None if t is None else t.float().double().requires_grad_(True) for t in maps]
    args = [None if t is None else t.repeat(1, ny, nx) for t in leaves]
    okw = dict(view=view.double(), lights=L.double(), intensities=I.double(), light_type=light_type, light_size=size)
    if workflow == "converted":
        pytest.skip("the float64 oracle has no several-lights form of the converted workflow; the two-kernel comparison above stands")
    ref = O.cook_torrance_multi(*args, **okw)
    (ref * gout.cpu().double()).sum().backward()
    for name, x, y in zip(("albedo", "normal", "roughness", "metallic", "specular"), one, leaves):
        if x is None:
            continue
        err = (x.float().cpu().double() - y.grad).abs()
        assert bool((err <= (2e-5 if dtype == torch.float32 else 2e-3) * (1 + y.grad.abs())).all()), (name, float(err.max()))


def test_loss_step_keeps_its_descriptor_across_training_steps_and_sees_every_change():
    """A training loop hands the loss step the same leaf tensors every iteration: the filled descriptor of the last step is kept
    (functional._MseStepFn._PLANS: pointers only, maps held weakly).  In-place updates of the maps (an optimiser step), edited light
    values, another target, tiled and untiled calls: every step equals the step computed without the kept descriptor."""
    import gc
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(8)
    leaves = [t.cuda().requires_grad_(True) for t in _leaf_maps(g, 24, 48, "metallic")[:4]]
    light = torch.tensor([0.1, 0.1, 1.0])
    kw = dict(view_dir=[0.0, 0.1, 1.0], light=light, light_intensity=[1.0, 0.9, 0.8], light_type="point", light_size=1.5)

    def step(keep, target, **extra):
        for t in leaves:
            t.grad = None
        before = F._MseStepFn._PLANS_MAX
        F._MseStepFn._PLANS_MAX = 8 if keep else 0
        if not keep:
            F._MseStepFn._PLANS.clear()
        try:
            loss = F.rendering_loss_mse(*leaves, target=target, **dict(kw, **extra))
            loss.backward()
        finally:
            F._MseStepFn._PLANS_MAX = before
        return loss.detach().clone(), [t.grad.clone() for t in leaves]
    F._MseStepFn._PLANS.clear()
    for it in range(6):
        tiled = it % 2 == 1
        target = torch.rand(3, 48 if tiled else 24, 96 if tiled else 48, generator=g).cuda()
        extra = dict(tile=2) if tiled else {}
        got = step(True, target, **extra)
        want = step(False, target, **extra)
        assert torch.equal(got[0], want[0]) and all(torch.equal(x, y) for x, y in zip(got[1], want[1])), it
        with torch.no_grad():                               # an optimiser step, in place
            for t, gr in zip(leaves, got[1]):
                t.sub_(0.1 * gr)
            leaves[2].clamp_(0.2, 1.0)
        light[0] += 0.05                                    # a light edited in place between steps
    step(True, torch.rand(3, 24, 48, generator=g).cuda())
    assert 1 <= len(F._MseStepFn._PLANS) <= 2
    # the kept descriptors do not keep the maps alive
    probe = weakref_of = __import__("weakref").ref(leaves[0])
    del leaves, got, want, t, gr
    gc.collect()
    assert probe() is None and weakref_of() is None


def test_whole_factor_downscale_register_kernel_equals_the_strip_kernel_and_aten():
    """Round 5: an antialiased down-scale by a whole factor 2 ... 8 | 16 on both axes -- MaterialBase.resize of a 1024^2 ... 4096^2 texture to 512^2
    (/root/reference/pypbr/materials/base.py:490-504) -- runs the register-only band-walking kernel (csrc/resize_down.hpp): the strip
    kernel's taps in the strip kernel's order, BIT-IDENTICAL (knob PBR_TUNE_RESIZE_UP2 = 0 selects the strip form; for 7 x, 8 x and 16 x -- 17 ... 35 taps -- its wide
    instantiation, which round 5 added too: they had fallen to two passes through a workspace).  All: <= 2e-6 from ATen's antialiased interpolate.  Shapes: the smallest the kernel takes, widths that
    leave lanes and whole workgroups idle, bands of ragged height, several planes, more than 1 536 / 8 column strips (one band)."""
    from pypbr_amd import _native as N, functional as F
    lib = N.lib()
    g = torch.Generator().manual_seed(85)
    try:
        for S in (2, 3, 4, 5, 6, 7, 8, 16):
            for planes, ho, wo in ((1, 2, 8), (3, 8, 256), (2, 13, 260), (1, 64, 1028), (3, 37, 12), (1, 512, 512), (4, 301, 2048 // S // 4 * 4), (1, 3, 8192 // S // 4 * 4))[:6 if S == 16 else 8]:
                x = (torch.rand(planes, S * ho, S * wo, generator=g) * 2 - 0.5)
                xd = x.cuda()
                lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 1)
                fast = F.resize(xd, (ho, wo))
                lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 0)
                other = F.resize(xd, (ho, wo))
                assert torch.equal(fast, other), (S, planes, ho, wo, float((fast - other).abs().max()))
                ref = torch.nn.functional.interpolate(x[None], size=(ho, wo), mode="bilinear", align_corners=False, antialias=True)[0]
                assert (fast.cpu() - ref).abs().max().item() <= 2e-6, (S, planes, ho, wo)
        # a large side beyond the 256 MB memory-side cache: lanes of 16 bytes per row, non-temporal loads (S = 2, 4) -- the same sums
        big = torch.rand(5, 4096, 4096, device="cuda")
        for S in (2, 4, 8):
            lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 1)
            fast = F.resize(big, (4096 // S, 4096 // S))
            lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 0)
            other = F.resize(big, (4096 // S, 4096 // S))
            assert torch.equal(fast, other), S
        del big, fast, other
        # an input of infinities and NaNs stays where it is: taps outside a clipped window are never multiplied (0 x inf)
        x = torch.rand(1, 64, 64, generator=g)
        x[0, 0, 0] = float("inf"); x[0, 63, 63] = float("-inf"); x[0, 31, 0] = float("nan")
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 1)
        fast = F.resize(x.cuda(), (32, 32))
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 0)
        other = F.resize(x.cuda(), (32, 32))
        assert torch.equal(torch.isnan(fast), torch.isnan(other)) and torch.equal(torch.nan_to_num(fast), torch.nan_to_num(other))
        ref = torch.nn.functional.interpolate(x[None], size=(32, 32), mode="bilinear", align_corners=False, antialias=True)[0]
        assert torch.equal(torch.isfinite(fast).cpu(), torch.isfinite(ref))
        # a view that starts off a 16-byte boundary, an aspect ratio that differs per axis: the strip form, whatever the knob says
        flat = torch.rand(3 * 64 * 64 + 1, generator=g).cuda()
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 1)
        a, b = F.resize(flat[1:].view(3, 64, 64), (32, 32)), F.resize(flat[:-1].view(3, 64, 64), (32, 16))
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, 0)
        assert torch.equal(a, F.resize(flat[1:].view(3, 64, 64), (32, 32))) and torch.equal(b, F.resize(flat[:-1].view(3, 64, 64), (32, 16)))
    finally:
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, -1)
