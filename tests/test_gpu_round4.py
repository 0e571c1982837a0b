"""Round 4: where a CPU-resident material's maps live between operations (one upload, results left on the device until somebody
reads them: examples/example_brdf.py is one H2D and one D2H), the repeat-inner kernel behind tile(n), the rendering loss's
ground-truth branch under autograd (docs/source/tutorials/06_advanced.rst:101-105), and the advisor's findings on the fused loss step."""
import os
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

import torch_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
VIEW, LIGHT, INTEN = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0])


def _load(folder="tiles"):
    from pypbr_amd.io import load_material_from_folder
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return load_material_from_folder(os.path.join(GOLDEN, folder), preferred_workflow="metallic")


def test_example_brdf_statements_are_one_upload_and_one_download(golden, monkeypatch):
    """/root/reference/examples/example_brdf.py:8-26 on the CPU-resident material the loader returns: the maps go up ONCE (all of
    them in one transfer, the normal map decoded on arrival), resize is one launch over all planes, tile(2) is only recorded (the
    kernel evaluates every texel at its four positions), and the image is the one thing that comes back."""
    from pypbr_amd import functional as F
    from pypbr_amd.models import CookTorranceBRDF
    calls = {"up": 0, "down": 0, "resize": 0}
    up, down, rz = F.upload_packed, F.to_host, F._resize_raw
    monkeypatch.setattr(F, "upload_packed", lambda *a, **k: (calls.__setitem__("up", calls["up"] + 1), up(*a, **k))[1])
    monkeypatch.setattr(F, "to_host", lambda *a, **k: (calls.__setitem__("down", calls["down"] + 1), down(*a, **k))[1])
    monkeypatch.setattr(F, "_resize_raw", lambda *a, **k: (calls.__setitem__("resize", calls["resize"] + 1), rz(*a, **k))[1])
    material = _load()
    assert material.device.type == "cpu" and material.__dict__["_raw_normal"] and all(t.device.type == "cpu" for t in material._raw.values())
    assert calls == {"up": 0, "down": 0, "resize": 0}                    # loading moved nothing
    material.resize((512, 512)).tile(2)
    raw = material._raw
    assert calls == {"up": 1, "down": 0, "resize": 1}
    assert all(t.is_cuda and t.shape[-2:] == (512, 512) for t in raw.values()) and not material.__dict__["_raw_normal"]
    assert len({t.untyped_storage().data_ptr() for t in raw.values()}) == 1         # one block of planes
    assert material.lazy_tile == (2, 2) and material.size == (1024, 1024) and material.device.type == "cpu"
    color = CookTorranceBRDF(light_type="point")(material, VIEW, LIGHT, INTEN, 1.0)
    assert calls == {"up": 1, "down": 1, "resize": 1}
    assert color.device.type == "cpu" and color.shape == (3, 1024, 1024)
    z = golden("example")
    assert np.abs(color[:, 448:576, 448:576].numpy() - z["example_crop"]).max() <= 1e-5
    assert abs(float(color.double().mean()) - float(z["example_mean"])) <= 1e-6
    # a second evaluation moves nothing but its image; only now does somebody look at the maps: each comes home once, repeated
    CookTorranceBRDF(light_type="point")(material, VIEW, LIGHT, INTEN, 1.0)
    assert calls == {"up": 1, "down": 2, "resize": 1}
    maps = material._maps
    assert all(t.device.type == "cpu" and t.shape[-2:] == (1024, 1024) for t in maps.values()) and material.lazy_tile == (1, 1)
    for k, v in maps.items():
        assert np.abs(v[:, 480:544, 480:544].numpy() - z[f"resized_crop_{k}"]).max() <= 1e-5, k
    assert material._maps["albedo"] is maps["albedo"]                     # ... and stays there


def test_loaded_material_without_any_operation_renders_with_one_upload(monkeypatch):
    """load -> render: the maps nobody has seen yet (an image's samples, the normal map undecoded) ride on the render's own upload and
    their float / decoded forms stay on the device; host maps somebody may have edited are re-read on every call, as the reference
    re-reads them."""
    import pypbr_amd.materials as M
    from pypbr_amd import functional as F
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    ups = []
    up = F.upload_packed
    monkeypatch.setattr(F, "upload_packed", lambda ts, *a, **k: (ups.append(len(ts)), up(ts, *a, **k))[1])
    material = _load("rocks")
    assert material._has_pending()
    brdf = CookTorranceBRDF("point")
    first = brdf(material, VIEW, LIGHT, INTEN, 1.0)
    assert ups == [len(material._raw)] and all(t.is_cuda and t.dtype == torch.float32 for t in material._raw.values()) and not material._has_pending()
    second = brdf(material, VIEW, LIGHT, INTEN, 1.0)
    assert len(ups) == 1 and torch.equal(first, second)     # read in place
    # the same maps assigned as tensors (decoded at assignment, as upstream) give the same image, and are uploaded on every call:
    # albedo, normal, roughness, metallic -- what the evaluation reads
    eager = BasecolorMetallicMaterial(albedo=material.albedo, normal=material.normal, roughness=material.roughness, metallic=material.metallic)
    assert (brdf(eager, VIEW, LIGHT, INTEN, 1.0) - first).abs().max().item() <= 2e-6
    assert (brdf(eager, VIEW, LIGHT, INTEN, 1.0) - first).abs().max().item() <= 2e-6
    assert ups[1:] == [4, 4] and all(t.device.type == "cpu" for t in eager._raw.values())
    assert material.normal.device.type == "cpu" and abs(float(material.normal.norm(dim=0).mean()) - 1.0) < 1e-5
    # a loaded material whose images were converted at assignment (no deferral): its float normal map was decoded then
    monkeypatch.setattr(M, "DEFER_IMAGE_DECODE", False)
    plain = _load("rocks")
    assert not plain._has_pending() and all(t.device.type == "cpu" and t.dtype == torch.float32 for t in plain._raw.values())
    assert torch.equal(brdf(plain, VIEW, LIGHT, INTEN, 1.0), first)


def test_results_of_material_operations_stay_on_the_device_until_read():
    """to_linear / workflow conversions / blends of a CPU-resident material: computed on the device, handed out on the CPU."""
    import pypbr_amd.blending as B
    from pypbr_amd import utils
    m1, m2 = _load("tiles"), _load("rocks")
    want_lin = utils.srgb_to_linear(m1.albedo)
    m1.to_linear()
    assert m1._raw["albedo"].is_cuda and not m1.albedo_is_srgb
    assert torch.equal(m1.albedo, want_lin) and m1._raw["albedo"].device.type == "cpu"
    ds = m1.to_diffuse_specular_material()
    assert ds.device.type == "cpu" and all(t.is_cuda for t in ds._raw.values() if t is not None)
    assert ds.specular.device.type == "cpu" and ds.specular.shape == (3, 1024, 1024)
    blended, mask = B.HeightBlend(blend_width=0.1, shift=-0.5)(_load("tiles"), m2)
    assert mask.device.type == "cpu" and blended.device.type == "cpu" and all(t.is_cuda for t in blended._raw.values() if t is not None)
    assert blended.albedo.device.type == "cpu"
    moved = _load("tiles").to("cuda")
    assert moved.device == torch.device("cuda", torch.cuda.current_device()) and all(t.device == moved.device for t in moved._raw.values())
    assert len({t.untyped_storage().data_ptr() for t in moved._raw.values()}) == 1
    back = moved.to("cpu")
    assert all(t.device.type == "cpu" for t in back._raw.values())


def test_tile_zero_and_negative_follow_torch_repeat():
    """base.py:534-536 is `map.repeat(1, n, n)`: n = 0 gives empty maps, a negative count torch's RuntimeError."""
    from pypbr_amd.materials import BasecolorMetallicMaterial
    m = BasecolorMetallicMaterial(albedo=torch.rand(3, 8, 8), roughness=torch.rand(1, 8, 8), metallic=torch.rand(1, 8, 8))
    assert m.tile(0) is m and m.albedo.shape == (3, 0, 0) and m.roughness.shape == (1, 0, 0)
    with pytest.raises(RuntimeError):
        BasecolorMetallicMaterial(albedo=torch.rand(3, 8, 8)).tile(-1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("light_type", ["point", "directional"])
def test_repeat_inner_kernel_equals_the_materialised_repeat_and_the_wrap_around_form(light_type, dtype):
    """tile(n) over the whole output runs cook_torrance_repeat_kernel (texels loaded and decoded once, evaluated at every repeat);
    bit-identical to evaluating the repeated maps and to the wrap-around kernel (PBR_TUNE_TILE_REPEAT = 0); bands of any height and several
    lights take it too (rounds 5 and 6)."""
    from pypbr_amd import _native as N, functional as F
    g = torch.Generator().manual_seed(11)
    lib = N.lib()
    for (B, h, w, ny, nx, wf) in ((1, 24, 64, 2, 2, "metallic"), (2, 17, 40, 3, 1, "specular"), (1, 8, 260, 1, 3, "metallic"), (3, 5, 12, 2, 4, "converted")):
        a = torch.rand(B, 3, h, w, generator=g).cuda().to(dtype)
        n = TF.normalize(torch.cat([torch.rand(B, 2, h, w, generator=g) - 0.5, torch.ones(B, 1, h, w)], 1), dim=1).cuda().to(dtype)
        r = (torch.rand(B, 1, h, w, generator=g) * 0.8 + 0.2).cuda().to(dtype)
        m = torch.rand(B, 1, h, w, generator=g).cuda().to(dtype) if wf != "specular" else None
        s = torch.rand(B, 3, h, w, generator=g).cuda().to(dtype) if wf == "specular" else None
        kw = dict(view_dir=[0.1, -0.2, 1.0], light=[0.3, -0.2, 0.8], light_intensity=[1.0, 0.9, 0.8], light_type=light_type, light_size=1.5,
                  convert_to_diffuse_specular=(wf == "converted"))
        rep = lambda t: None if t is None else t.repeat(1, 1, ny, nx)
        want = F.cook_torrance(rep(a), rep(n), rep(r), rep(m), rep(s), **kw)
        plan = F.plan_cook_torrance(a, n, r, m, s, tile=(ny, nx), **kw)
        assert plan.kernel_name.startswith("ctr_"), plan.kernel_name
        got = plan.launch().clone()
        assert torch.equal(got, want), (B, h, w, ny, nx, wf)
        lib.pbr_set_tuning(N.TUNE_TILE_REPEAT, 0)
        try:
            wrap = F.plan_cook_torrance(a, n, r, m, s, tile=(ny, nx), **kw)
            assert not wrap.kernel_name.startswith("ctr_")
            assert torch.equal(wrap.launch(), want)
        finally:
            lib.pbr_set_tuning(N.TUNE_TILE_REPEAT, -1)
        if ny * h > 4:                             # a row band of the tiled output (a multi-GPU shard): the repeat-inner kernel whatever its height
            rows = ny * h - 4                      # (round 6: a band thinner than a period walks the window of source rows it touches); same values
            band = F.plan_cook_torrance(a, n, r, m, s, tile=(ny, nx), y_offset=3, rows=rows, **kw)
            assert band.kernel_name.startswith("ctr_") and torch.equal(band.launch(), want[:, :, 3:ny * h - 1])
            thin = F.plan_cook_torrance(a, n, r, m, s, tile=(ny, nx), y_offset=ny * h - 3, rows=2, **kw)
            assert thin.kernel_name.startswith("ctr_") and torch.equal(thin.launch(), want[:, :, ny * h - 3:ny * h - 1])
            if ny >= 2:                            # exactly one period, straddling a seam
                mid = F.plan_cook_torrance(a, n, r, m, s, tile=(ny, nx), y_offset=h // 2 + 1, rows=h, **kw)
                assert mid.kernel_name.startswith("ctr_") and torch.equal(mid.launch(), want[:, :, h // 2 + 1:h // 2 + 1 + h])
        # strided result planes (out= inside a larger allocation)
        big = torch.zeros(B, 3, ny * h + 2, nx * w, device="cuda")
        out = big[:, :, 1:ny * h + 1]
        if dtype == torch.float32:
            F.plan_cook_torrance(a, n, r, m, s, tile=(ny, nx), out=out, **kw).launch()
            assert torch.equal(out, want) and not big[:, :, 0].any() and not big[:, :, -1].any()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_repeat_inner_kernel_at_full_size(dtype):
    """2048^2 maps, tile(2) -> the 4096^2 image of the evidence set, bit-identical to evaluating the materialised repeat; a (3, 2) repeat of
    a two-material batch as well (strided result planes of the second material)."""
    from pypbr_amd import functional as F
    import bench
    maps = [t.to(dtype) for t in bench.synth_material(2048, torch.device("cuda", 0), 77)]
    kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
    plan = F.plan_cook_torrance(*maps, tile=2, **kw)
    assert plan.kernel_name.startswith("ctr_point_metallic")
    got = plan.launch()
    want = F.cook_torrance(*[t.repeat(1, 2, 2) for t in maps], **kw)
    assert got.shape == (3, 4096, 4096) and torch.equal(got, want)
    del got, want, plan
    two = [torch.stack([t[:, :1024, :1536], t[:, 1024:, :1536]]).contiguous() for t in maps]
    got = F.cook_torrance(*two, tile=(3, 2), **kw)
    want = F.cook_torrance(*[t.repeat(1, 1, 3, 2) for t in two], **kw)
    assert got.shape == (2, 3, 3072, 3072) and torch.equal(got, want)


def test_rendering_loss_differentiates_the_ground_truth_branch():
    """06_advanced.rst:101-105 renders BOTH materials under autograd: a light that is being fitted (requires grad, shared by both
    renderings) receives the gradient of both branches; ground-truth maps that require grad receive theirs.  Against float64
    autograd of the tutorial's three lines through the ATen restatement of the reference."""
    from pypbr_amd.losses import RenderingLoss
    from pypbr_amd.materials import BasecolorMetallicMaterial
    g = torch.Generator().manual_seed(21)
    H, W = 24, 40

    def maps():
        a = torch.rand(3, H, W, generator=g)
        n = TF.normalize(torch.cat([(torch.rand(2, H, W, generator=g) - 0.5), torch.ones(1, H, W)], 0), dim=0)
        return a, n, torch.rand(1, H, W, generator=g) * 0.7 + 0.25, torch.rand(1, H, W, generator=g)
    pm, gm = maps(), maps()

    def oracle(light, gt_albedo):
        kw = dict(view=VIEW.double(), light=light, intensity=INTEN.double(), light_type="point", light_size=1.0)
        pred = O.cook_torrance(*[t.double() for t in pm], None, **kw)
        gt = O.cook_torrance(gt_albedo, *[t.double() for t in gm[1:]], None, **kw)
        return TF.mse_loss(pred, gt)
    l64, a64 = LIGHT.double().requires_grad_(True), gm[0].double().requires_grad_(True)
    ref = oracle(l64, a64)
    ref.backward()

    def material(m, albedo=None):
        mat = BasecolorMetallicMaterial(albedo=m[0].cuda() if albedo is None else albedo, normal=None, roughness=m[2].cuda(), metallic=m[3].cuda(),
                                        device=torch.device("cuda"))
        mat._raw["normal"] = m[1].cuda()
        return mat
    light = LIGHT.clone().cuda().requires_grad_(True)
    gt_albedo = gm[0].cuda().requires_grad_(True)
    loss = RenderingLoss(light_dir=light, light_size=1.0)(material(pm), material(gm, gt_albedo))
    loss.backward()
    assert abs(loss.item() - ref.item()) <= 2e-6 * (1 + ref.item())
    assert (light.grad.cpu().double() - l64.grad).abs().max().item() <= 2e-5 * float(l64.grad.abs().max()) + 1e-9, (light.grad, l64.grad)
    assert (gt_albedo.grad.cpu().double() - a64.grad).abs().max().item() <= 2e-5 * float(a64.grad.abs().max()) + 1e-10
    # the predicted branch alone gives a different light gradient: the ground-truth branch really contributes
    l2 = LIGHT.double().requires_grad_(True)
    kw = dict(view=VIEW.double(), intensity=INTEN.double(), light_type="point", light_size=1.0)
    with torch.no_grad():
        gt_img = O.cook_torrance(*[t.double() for t in gm], None, light=LIGHT.double(), **kw)
    TF.mse_loss(O.cook_torrance(*[t.double() for t in pm], None, light=l2, **kw), gt_img).backward()
    assert (l2.grad - l64.grad).abs().max().item() > 1e-3 * float(l64.grad.abs().max())
    # nothing requires grad on the ground-truth side: it is rendered without a graph, the fused step serves the predicted maps
    pa = pm[0].cuda().requires_grad_(True)
    plain = RenderingLoss(light_size=1.0)(material(pm, pa), material(gm))
    assert type(plain.grad_fn).__name__ == "_MseStepFnBackward"


def test_fused_loss_step_takes_a_cpu_target_and_survives_a_retained_graph():
    """rendering_loss_mse(target=cpu tensor) computes on the maps' device; a second backward through the same node
    (retain_graph=True) gives the same gradients again instead of failing on buffers that were handed over."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(5)
    H, W = 16, 32
    a = torch.rand(3, H, W, generator=g).cuda().requires_grad_(True)
    n = TF.normalize(torch.cat([torch.rand(2, H, W, generator=g) - 0.5, torch.ones(1, H, W)], 0), dim=0).cuda()
    r = (torch.rand(1, H, W, generator=g) * 0.7 + 0.25).cuda().requires_grad_(True)
    m = torch.rand(1, H, W, generator=g).cuda()
    kw = dict(view_dir=VIEW, light=LIGHT, light_intensity=INTEN, light_type="point", light_size=1.0)
    target = torch.rand(3, H, W, generator=g)                          # on the CPU
    loss = F.rendering_loss_mse(a, n, r, m, target=target, **kw)
    assert type(loss.grad_fn).__name__ == "_MseStepFnBackward" and loss.is_cuda
    with_gpu_target = F.rendering_loss_mse(a, n, r, m, target=target.cuda(), **kw)
    assert loss.item() == with_gpu_target.item()
    ga1, gr1 = torch.autograd.grad(loss, (a, r), retain_graph=True)
    ga1, gr1 = ga1.clone(), gr1.clone()
    ga2, gr2 = torch.autograd.grad(loss, (a, r), retain_graph=True)     # the node runs again
    assert torch.equal(ga1, ga2) and torch.equal(gr1, gr2)
    (2.0 * loss).backward()                                             # and once more, scaled, releasing the graph
    assert torch.allclose(a.grad, 2.0 * ga1, rtol=1e-6, atol=0) and torch.allclose(r.grad, 2.0 * gr1, rtol=1e-6, atol=0)
    with pytest.raises(RuntimeError):                                   # autograd's own error: the graph is gone
        loss.backward()


def test_plan_launched_on_another_stream_folds_its_device_parameters_there():
    """a plan whose light lives on the device, launched on an explicit stream, must not read the parameter block
    unordered; and prepare_device_parameters() on a plan without device parameters is harmless."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(2)
    H, W = 16, 64
    a, r, m = torch.rand(3, H, W, generator=g).cuda(), (torch.rand(1, H, W, generator=g) * 0.7 + 0.2).cuda(), torch.rand(1, H, W, generator=g).cuda()
    n = TF.normalize(torch.cat([torch.rand(2, H, W, generator=g) - 0.5, torch.ones(1, H, W)], 0), dim=0).cuda()
    light = LIGHT.clone().cuda()
    kw = dict(view_dir=VIEW, light_intensity=INTEN, light_type="point", light_size=1.0)
    plan = F.plan_cook_torrance(a, n, r, m, light=light, **kw)
    want = F.plan_cook_torrance(a, n, r, m, light=[0.3, 0.2, 0.9], **kw).launch().clone()
    side = torch.cuda.Stream()
    light.copy_(torch.tensor([0.3, 0.2, 0.9]))
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got = plan.launch(side.cuda_stream)
    side.synchronize()
    assert torch.equal(got, want)
    host_plan = F.plan_cook_torrance(a, n, r, m, light=[0.3, 0.2, 0.9], **kw)
    assert host_plan._param_tensors == (None, None, None) and host_plan.desc.device_params is None


def test_per_call_tuning_changes_the_schedule_not_the_bits():
    """ABI 6: knobs passed with the plan (pbr_render_desc.tuning) pick other schedules -- no occupancy
    governor, larger workgroups, one-pixel lanes, the wrap-around form of a tiled launch -- with bit-identical results, and leave no trace for the next call."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(4)
    H, W = 48, 512
    a, r, m = torch.rand(3, H, W, generator=g).cuda(), (torch.rand(1, H, W, generator=g) * 0.7 + 0.2).cuda(), torch.rand(1, H, W, generator=g).cuda()
    n = TF.normalize(torch.cat([torch.rand(2, H, W, generator=g) - 0.5, torch.ones(1, H, W)], 0), dim=0).cuda()
    kw = dict(view_dir=VIEW, light=LIGHT, light_intensity=INTEN, light_type="point", light_size=1.0)
    want = F.plan_cook_torrance(a, n, r, m, **kw).launch().clone()
    for knobs in (dict(lds_bytes=0), dict(block_log2=8, scalar_base=0), dict(max_vec=1)):
        plan = F.plan_cook_torrance(a, n, r, m, tuning=knobs, **kw)
        assert torch.equal(plan.launch(), want), knobs
    one_pixel = F.plan_cook_torrance(a, n, r, m, tuning=dict(max_vec=1), **kw)
    assert one_pixel.kernel_name.endswith("_v1") and F.plan_cook_torrance(a, n, r, m, **kw).kernel_name.endswith("_v4")      # nothing stuck
    tiled = F.plan_cook_torrance(a, n, r, m, tile=2, **kw)
    wrap = F.plan_cook_torrance(a, n, r, m, tile=2, tuning=dict(tile_repeat=0), **kw)
    assert tiled.kernel_name.startswith("ctr_") and wrap.kernel_name.startswith("ct_") and torch.equal(tiled.launch(), wrap.launch())
    assert wrap.set_tuning().kernel_name.startswith("ctr_")               # back to the rules
