"""The rendering-loss step as one kernel (pbr_cook_torrance_mse_step, round 3): loss = MSELoss(brdf(predicted), target) of
docs/source/tutorials/06_advanced.rst:73-107 -- value AND the gradients of the predicted maps from a single pass -- against
float64 autograd of the same loss through the ATen restatement of the reference (oracle/torch_oracle.py), and against the
unfused path (fused evaluation, torch's MSE, backward kernel) it replaces."""
import pytest
import torch
import torch.nn.functional as TF

import torch_oracle as O

pytestmark = pytest.mark.gpu


def _maps(g, B, H, W, workflow):
    a = torch.rand(B, 3, H, W, generator=g)
    n = torch.cat([(torch.rand(B, 2, H, W, generator=g) - 0.5) * 1.4, torch.ones(B, 1, H, W)], 1)      # un-normalised on purpose
    r = torch.rand(B, 1, H, W, generator=g) * 0.7 + 0.25
    m = torch.rand(B, 1, H, W, generator=g) if workflow != "specular" else None
    s = torch.rand(B, 3, H, W, generator=g) * 0.6 if workflow == "specular" else None
    return a, n, r, m, s


def _oracle_loss(maps, target, view, lights, intens, light_type, light_size, converted=False):
    leaves = [None if t is None else t.double().requires_grad_(True) for t in maps]
    outs = []
    L, I = lights.double().reshape(-1, 3), intens.double().reshape(-1, 3)
    kw = dict(view=view.double(), light_type=light_type, light_size=light_size)
    for b in range(maps[0].shape[0]):
        args = [None if t is None else t[b] for t in leaves]
        if converted:
            outs.append(O.cook_torrance_converted(args[0], args[1], args[2], args[3], light=L[0], intensity=I[0], **kw))
        elif L.shape[0] > 1:
            outs.append(O.cook_torrance_multi(*args, lights=L, intensities=I, **kw))
        else:
            outs.append(O.cook_torrance(*args, light=L[0], intensity=I[0], **kw))
    loss = TF.mse_loss(torch.stack(outs), target.double())
    loss.backward()
    return loss.detach(), leaves


CASES = [
    # workflow, light_type, lights, B, H, W, dtype
    ("metallic", "point", 1, 1, 24, 48, torch.float32),          # two pixels per lane (point light, fp32)
    ("metallic", "directional", 1, 2, 20, 64, torch.float32),    # four pixels per lane
    ("specular", "point", 1, 2, 18, 40, torch.float32),
    ("converted", "directional", 1, 1, 16, 36, torch.float32),
    ("metallic", "point", 3, 2, 12, 32, torch.float32),          # several lights: the packed pair, two passes over the lights
    ("specular", "directional", 2, 1, 14, 30, torch.float32),
    ("metallic", "point", 1, 1, 15, 37, torch.float32),          # odd width: one pixel per lane
    ("metallic", "point", 1, 2, 16, 64, torch.float16),          # fp16 maps: gradients come back in fp16
    ("specular", "point", 2, 1, 12, 48, torch.float16),
]


@pytest.mark.parametrize("workflow,light_type,n_lights,B,H,W,dtype", CASES)
def test_loss_and_gradients_from_one_kernel_against_float64_autograd(workflow, light_type, n_lights, B, H, W, dtype):
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(1000 + H * W + n_lights)
    maps = [None if t is None else t.to(dtype).float() for t in _maps(g, B, H, W, workflow)]        # the values the device sees
    target = torch.rand(B, 3, H, W, generator=g)
    view = torch.tensor([0.05, 0.1, 0.9])
    base = torch.tensor([[0.1, 0.1, 1.0], [-0.4, 0.2, 0.7], [0.3, -0.3, 0.9]])[:n_lights]
    lights = base if light_type == "point" else base * 1.3
    intens = torch.tensor([[1.0, 0.9, 0.8], [0.4, 0.5, 0.6], [0.3, 0.3, 0.3]])[:n_lights]
    size = 1.5 if light_type == "point" else None
    want_loss, want = _oracle_loss(maps, target, view, lights, intens, light_type, size, converted=(workflow == "converted"))
    leaves = [None if t is None else t.to(dtype).cuda().requires_grad_(True) for t in maps]
    kw = dict(view_dir=view, light=lights, light_intensity=intens, light_type=light_type, light_size=size,
              convert_to_diffuse_specular=(workflow == "converted"))
    loss = F.rendering_loss_mse(*leaves, target=target.cuda(), **kw)
    assert loss.shape == () and type(loss.grad_fn).__name__ == "_MseStepFnBackward"
    assert abs(loss.item() - want_loss.item()) <= 1e-6 * (1 + want_loss.item())
    loss.backward()
    for name, x, y in zip(("albedo", "normal", "roughness", "metallic", "specular"), leaves, want):
        if x is None:
            continue
        assert x.grad.dtype == dtype and x.grad.shape == x.shape
        err = (x.grad.float().cpu().double() - y.grad).abs()
        scale = float(y.grad.abs().max())
        tol = (2e-5 if dtype == torch.float32 else 2e-3) * (scale + 1e-12) + 1e-9
        assert float(err.max()) <= tol, (name, float(err.max()), scale)
    # the unfused path: same loss to fp32 rounding, same gradients
    again = [None if t is None else t.detach().clone().requires_grad_(True) for t in leaves]
    unfused = TF.mse_loss(F.cook_torrance(*again, **kw), target.cuda())
    unfused.backward()
    assert abs(unfused.item() - loss.item()) <= 2e-6 * (1 + loss.item())
    for x, y in zip(leaves, again):
        if x is not None:
            d = (x.grad.float() - y.grad.float()).abs().max().item()
            assert d <= (2e-5 if dtype == torch.float32 else 2e-3) * (float(y.grad.float().abs().max()) + 1e-12) + 1e-9


def test_upstream_gradient_partial_gradients_and_determinism():
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(5)
    a, n, r, m, _ = _maps(g, 2, 20, 48, "metallic")
    target = torch.rand(2, 3, 20, 48, generator=g).cuda()
    kw = dict(view_dir=[0.0, 0.0, 1.0], light=[0.1, 0.1, 1.0], light_intensity=[1.0, 1.0, 1.0], light_type="point")

    def run(k, want=(True, True, True, True)):
        leaves = [t.clone().cuda().requires_grad_(w) for t, w in zip((a, n, r, m), want)]
        loss = F.rendering_loss_mse(*leaves, target=target, **kw)
        (loss * k).backward()
        return loss.detach(), [t.grad for t in leaves]
    l1, g1 = run(1.0)
    l2, g2 = run(1.0)
    assert torch.equal(l1, l2) and all(torch.equal(x, y) for x, y in zip(g1, g2))             # fixed summation order
    _, g3 = run(3.0)
    for x, y in zip(g1, g3):
        assert (y - 3.0 * x).abs().max().item() <= 1e-6 * (3.0 * x.abs().max().item() + 1e-12)
    _, gp = run(1.0, want=(True, False, False, True))
    assert gp[1] is None and gp[2] is None and torch.equal(gp[0], g1[0]) and torch.equal(gp[3], g1[3])
    # no gradient wanted at all: the plain differentiable composition (a number, no graph)
    with torch.no_grad():
        plain = F.rendering_loss_mse(a.cuda(), n.cuda(), r.cuda(), m.cuda(), target=target, **kw)
    assert not plain.requires_grad and abs(plain.item() - l1.item()) <= 2e-6 * (1 + l1.item())


def test_rendering_loss_module_follows_the_tutorial():
    """pypbr_amd.losses.RenderingLoss against the tutorial's own composition (two BRDF calls + nn.MSELoss) on the same maps."""
    from pypbr_amd.losses import RenderingLoss
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    g = torch.Generator().manual_seed(9)
    H, W = 32, 48

    def material(grad):
        a, n, r, m, _ = [None if t is None else t[0] for t in _maps(g, 1, H, W, "metallic")]
        n = TF.normalize(n, dim=0)
        leaves = {"albedo": a.cuda().requires_grad_(grad), "roughness": r.cuda().requires_grad_(grad), "metallic": m.cuda().requires_grad_(grad)}
        mat = BasecolorMetallicMaterial(albedo=leaves["albedo"], normal=None, roughness=leaves["roughness"], metallic=leaves["metallic"],
                                        device=torch.device("cuda"))
        mat._maps["normal"] = n.cuda()
        return mat, leaves
    gt, _ = material(False)
    pred, leaves = material(True)
    loss = RenderingLoss()(pred, gt)
    assert type(loss.grad_fn).__name__ == "_MseStepFnBackward"
    loss.backward()
    fused = {k: v.grad.clone() for k, v in leaves.items()}
    for v in leaves.values():
        v.grad = None
    brdf = CookTorranceBRDF("point")
    vd, ld, li = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0])
    ref = torch.nn.MSELoss()(brdf(pred, vd, ld, li), brdf(gt, vd, ld, li))
    ref.backward()
    assert abs(ref.item() - loss.item()) <= 2e-6 * (1 + ref.item())
    for k, v in leaves.items():
        assert (fused[k] - v.grad).abs().max().item() <= 2e-5 * (float(v.grad.abs().max()) + 1e-12) + 1e-9, k
    # a reference rendering given directly, and a CPU-resident predicted material (falls back to the plain composition)
    img = brdf(gt, vd, ld, li)
    assert abs(RenderingLoss()(pred, img).item() - loss.item()) <= 1e-7
    cpu_pred = BasecolorMetallicMaterial(albedo=leaves["albedo"].detach().cpu(), normal=None, roughness=leaves["roughness"].detach().cpu(),
                                         metallic=leaves["metallic"].detach().cpu())
    cpu_pred._maps["normal"] = pred._maps["normal"].cpu()
    assert abs(RenderingLoss()(cpu_pred, gt).item() - loss.item()) <= 2e-6 * (1 + loss.item())


@pytest.mark.gpu
def test_fused_blend_and_loss_step_on_random_shapes():
    """tools/fused_fuzz.py: 40 random cases (1-2 materials, extents 1 ... 257, all workflows / light types / flags, shared and per-material
    masks, fp16 maps for the loss step, an upstream scale): blend + render forward and one-pass backward, and the one-kernel loss step,
    against the unfused differentiable pieces."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fused_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fused_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.run(40, 9, verbose=False)


@pytest.mark.gpu
@pytest.mark.parametrize("size,dtype", [(64, torch.float32), (128, torch.float16)])
def test_whole_training_step_is_capturable_into_a_hip_graph(size, dtype):
    """tools/graph_step_probe.py: rendering loss through autograd (the one-kernel step; fp16 maps of 128-pixel rows: its streamed form) + an SGD
    update of the maps, captured with torch.cuda.graph and replayed -- every library call only enqueues.  Five replays leave the maps exactly where
    five eager steps leave them (the kernels are deterministic)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("graph_step_probe", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "graph_step_probe.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    leaves_a, target = mod.make(size, dtype)
    leaves_b = [t.detach().clone().requires_grad_(True) for t in leaves_a]
    graph, loss = mod.capture(leaves_a, target)              # three warm-up steps + the captured one (capturing does not run it)
    for _ in range(5):
        graph.replay()
    for _ in range(3 + 5):
        eager_loss = mod.train_step(leaves_b, target)
    torch.cuda.synchronize()
    for x, y in zip(leaves_a, leaves_b):
        assert torch.equal(x.detach(), y.detach())
    assert float(loss.detach()) == float(eager_loss.detach())


@pytest.mark.gpu
@pytest.mark.parametrize("workflow", ["metallic", "specular", "converted"])
@pytest.mark.parametrize("light_type", ["point", "directional"])
@pytest.mark.parametrize("flags", [dict(), dict(return_srgb=False), dict(albedo_is_srgb=False)])
def test_streamed_loss_step_equals_the_one_tile_kernels(workflow, light_type, flags):
    """fp16 maps with rows of whole 128-pixel tiles take cook_torrance_mse_stream_kernel; PBR_TUNE_MSE_STREAM = 0 takes the one-tile kernels.
    Same chain rule on the same pixels: gradients bit for bit, the loss to fp32 rounding (another summation order) -- for every flag
    combination, several materials, one and several tiles per wave.  (The flag-free instantiation failed exactly this: ct_loss.hip.)"""
    from pypbr_amd import _native as N, functional as F
    lib = N.lib()
    g = torch.Generator(device="cuda").manual_seed(77)
    for B, H, W in ((1, 2, 256), (2, 3, 128), (2, 160, 384)):
        rnd = lambda *s: torch.rand(*s, device="cuda", generator=g)
        a = rnd(B, 3, H, W).half()
        n = torch.nn.functional.normalize(torch.cat([rnd(B, 2, H, W) - 0.5, torch.ones(B, 1, H, W, device="cuda")], 1), dim=1).half()
        r = (rnd(B, 1, H, W) * 0.6 + 0.3).half()
        m = rnd(B, 1, H, W).half() if workflow != "specular" else None
        s = rnd(B, 3, H, W).half() if workflow == "specular" else None
        kw = dict(view_dir=[0.0, 0.1, 1.0], light=[0.1, 0.1, 1.0] if light_type == "point" else [0.3, 0.2, 1.0], light_intensity=[1.0, 0.9, 0.8],
                  light_type=light_type, light_size=1.5 if light_type == "point" else None, **flags)
        if workflow == "converted":
            kw.update(convert_to_diffuse_specular=True)
        target = rnd(B, 3, H, W)
        got = {}
        try:
            for knob in (1, 0):
                lib.pbr_set_tuning(N.TUNE_MSE_STREAM, knob)
                leaves = [None if t is None else t.clone().requires_grad_() for t in (a, n, r, m, s)]
                loss = F.rendering_loss_mse(*leaves, target=target, **kw) * 64.0            # keep the fp16 gradients out of the subnormal range
                loss.backward()
                got[knob] = (float(loss.detach()), [None if t is None else t.grad for t in leaves])
        finally:
            lib.pbr_set_tuning(N.TUNE_MSE_STREAM, 1)
        assert abs(got[1][0] - got[0][0]) <= 2e-6 * abs(got[0][0]), (B, H, W, got[1][0], got[0][0])
        for x, y in zip(got[1][1], got[0][1]):
            assert (x is None and y is None) or torch.equal(x, y), (B, H, W)
        ref = torch.nn.functional.mse_loss(F.cook_torrance(a, n, r, m, s, **kw).float(), target) * 64.0
        assert abs(got[1][0] - float(ref)) <= 2e-6 * float(ref)
