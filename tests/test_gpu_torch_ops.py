"""torch.ops.pbr_hip.* on the GPU: the registered operators against the ctypes binding of the same C ABI (bit-equal:
both end in the same kernels), torch.library.opcheck (schema, fake kernel, autograd registration, AOT dispatch), and
tracing through torch.compile's AOT path (no Triton involved: backend "aot_eager")."""
import os

import pytest
import torch

# PBR_NO_TORCH_OPS=1 is the explicit opt-out that runs everything through the ctypes binding (A/B of the two bindings);
# without it a missing operator library is a failure, not a skip.
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("PBR_NO_TORCH_OPS") == "1", reason="operators switched off by PBR_NO_TORCH_OPS=1")]


def _maps(B, H, W, seed, dtype=torch.float32, specular=False, grad=False):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(B, 3, H, W, generator=g)
    n = torch.cat([(torch.rand(B, 2, H, W, generator=g) - 0.5), torch.ones(B, 1, H, W)], 1)
    r = torch.rand(B, 1, H, W, generator=g) * 0.7 + 0.3
    m = None if specular else torch.rand(B, 1, H, W, generator=g)
    s = torch.rand(B, 3, H, W, generator=g) if specular else None
    out = [None if t is None else t.to(dtype).cuda() for t in (a, n, r, m, s)]
    return [None if t is None else t.requires_grad_(grad) for t in out]


KW = dict(view_dir=[0.0, 0.1, 1.0], light=[[0.1, 0.1, 1.0], [-0.3, 0.2, 0.8]], light_intensity=[[0.6, 0.5, 0.4], [0.3, 0.3, 0.5]],
          light_type="point", light_size=1.5)


def _both(fn):
    from pypbr_amd import functional as F
    assert F.USE_TORCH_OPS
    via_op = fn()
    F.USE_TORCH_OPS = False
    try:
        via_ctypes = fn()
    finally:
        F.USE_TORCH_OPS = True
    return via_op, via_ctypes


@pytest.mark.parametrize("case", ["batch", "single", "fp16", "specular_dir", "converted", "tile", "band", "half_out"])
def test_operator_equals_the_ctypes_binding(case):
    from pypbr_amd import functional as F
    from pypbr_amd import torch_ops
    assert torch_ops.available()
    maps = _maps(3, 40, 72, 1)
    kw = dict(KW)
    if case == "single":
        maps = [None if t is None else t[0] for t in maps]
    elif case == "fp16":
        maps = _maps(2, 32, 64, 2, torch.float16)
    elif case == "specular_dir":
        maps = _maps(2, 33, 50, 3, specular=True)
        kw.update(light_type="directional", light=[0.3, -0.2, 1.0], light_intensity=[1, 1, 1], light_size=None, specular_is_srgb=False, return_srgb=False)
    elif case == "converted":
        kw.update(convert_to_diffuse_specular=True)
    elif case == "tile":
        kw.update(tile=(2, 3), y_offset=16, rows=40)
    elif case == "band":
        maps = [None if t is None else t[:, :, 8:24] for t in maps]
        kw.update(y_offset=8, height_total=40)
    elif case == "half_out":
        kw.update(out_dtype=torch.float16)
    a, b = _both(lambda: F.cook_torrance(*maps, **kw))
    assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b)


def test_operator_gradients_equal_the_ctypes_autograd_function():
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(5)
    wt = (torch.rand(2, 3, 24, 40, generator=g) - 0.4).cuda()

    def run():
        maps = _maps(2, 24, 40, 7, grad=True)
        maps[2] = maps[2].detach()[:1].clone().requires_grad_(True)                   # ONE roughness map shared by the batch
        params = [torch.tensor(KW[k], device="cuda" if k != "view_dir" else "cpu", requires_grad=True) for k in ("view_dir", "light", "light_intensity")]
        out = F.cook_torrance(*maps, view_dir=params[0], light=params[1], light_intensity=params[2], light_type="point", light_size=1.5)
        (out * wt).sum().backward()
        return [t.grad for t in maps if t is not None] + [p.grad for p in params]
    via_op, via_ctypes = _both(run)
    for x, y in zip(via_op, via_ctypes):
        assert x is not None and x.shape == y.shape and x.device == y.device and torch.equal(x, y)
    # tiled maps: the gradient of a texel is the sum over its repeats
    def tiled():
        maps = _maps(1, 16, 24, 8, grad=True)
        out = F.cook_torrance(*maps, tile=2, **KW)
        out.sum().backward()
        return [t.grad for t in maps if t is not None]
    for x, y in zip(*_both(tiled)):
        assert x.shape == y.shape and torch.equal(x, y)


def test_opcheck():
    from pypbr_amd import torch_ops
    assert torch_ops.available()
    v, l, i = torch.tensor([0.0, 0.1, 1.0]), torch.tensor(KW["light"]), torch.tensor(KW["light_intensity"])
    a, n, r, m, _ = _maps(2, 16, 24, 11, grad=True)
    args = (a, n, r, m, None, v, l, i, 1.5, 1, True, True, False, True, 0, 0, 1, 1, 0, False)
    torch.library.opcheck(torch.ops.pbr_hip.cook_torrance.default, args)
    # parameters that require grad (device-resident lights), specular workflow, no normal map
    a, n, r, _, s = _maps(1, 16, 24, 12, specular=True, grad=True)
    args = (a, None, r, None, s, v.clone().requires_grad_(True), l.cuda().requires_grad_(True), i.cuda().requires_grad_(True),
            0.0, 0, True, False, False, False, 0, 0, 1, 1, 0, False)
    torch.library.opcheck(torch.ops.pbr_hip.cook_torrance.default, args)
    x = torch.rand(3, 16, 24, device="cuda")
    for op in (torch.ops.pbr_hip.srgb_to_linear, torch.ops.pbr_hip.linear_to_srgb):
        torch.library.opcheck(op.default, (x,))
    torch.library.opcheck(torch.ops.pbr_hip.metallic_to_diffuse_specular.default, (x, torch.rand(1, 16, 24, device="cuda"), True))
    torch.library.opcheck(torch.ops.pbr_hip.diffuse_specular_to_basecolor_metallic.default, (x, torch.rand(3, 16, 24, device="cuda"), False))
    torch.library.opcheck(torch.ops.pbr_hip.fold_gradient.default, (torch.rand(2, 3, 32, 48, device="cuda"), 16, 24, True))
    # round 3: the map ops carry an autograd formula (their own backward operators), and resize is an operator too
    xg = torch.rand(3, 16, 24, device="cuda", requires_grad=True)
    for op in (torch.ops.pbr_hip.srgb_to_linear, torch.ops.pbr_hip.linear_to_srgb):
        torch.library.opcheck(op.default, (xg,))
    torch.library.opcheck(torch.ops.pbr_hip.metallic_to_diffuse_specular.default, (xg, torch.rand(1, 16, 24, device="cuda", requires_grad=True), True))
    torch.library.opcheck(torch.ops.pbr_hip.diffuse_specular_to_basecolor_metallic.default, (xg, torch.rand(3, 16, 24, device="cuda", requires_grad=True), False))
    torch.library.opcheck(torch.ops.pbr_hip.resize.default, (xg, 9, 31, True))
    torch.library.opcheck(torch.ops.pbr_hip.colour_backward.default, (x, torch.rand(3, 16, 24, device="cuda"), True))
    torch.library.opcheck(torch.ops.pbr_hip.resize_backward.default, (torch.rand(3, 9, 31, device="cuda"), 16, 24, True))


def test_traces_through_aot_autograd():
    """A rendering loss (docs/source/tutorials/06_advanced.rst:73-107) captured by torch.compile with the AOT-eager
    backend: fake kernels for tracing, the registered autograd formula for the backward graph."""
    from pypbr_amd import functional as F
    a, n, r, m, _ = _maps(1, 32, 48, 21, grad=True)
    target = torch.rand(1, 3, 32, 48, device="cuda")

    def loss_fn(a, n, r, m):
        out = F.cook_torrance(a, n, r, m, **KW)
        return torch.nn.functional.mse_loss(out, target)
    eager = loss_fn(a, n, r, m)
    eager.backward()
    want = [t.grad.clone() for t in (a, n, r, m)]
    for t in (a, n, r, m):
        t.grad = None
    compiled = torch.compile(loss_fn, backend="aot_eager", fullgraph=True)
    got = compiled(a, n, r, m)
    got.backward()
    assert torch.equal(got, eager)
    for t, w in zip((a, n, r, m), want):
        assert torch.equal(t.grad, w)


def test_operator_errors_match_the_reference_shaped_exceptions():
    """Same exception types through the operator as through the ctypes plan (and as the reference raises where it has the
    check): ValueError for a missing workflow map / bad shapes / bad light type, TypeError for mixed dtypes, RuntimeError
    for maps that are not on a device."""
    from pypbr_amd import functional as F
    a, n, r, m, _ = _maps(1, 16, 24, 31)
    kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1])
    with pytest.raises(ValueError, match="either 'metallic' or 'specular'"):
        F.cook_torrance(a, n, r, None, None, **kw)
    with pytest.raises(ValueError, match="Unsupported light_type"):
        F.cook_torrance(a, n, r, m, light_type="spot", **kw)
    with pytest.raises(ValueError):
        F.cook_torrance(a, n, r[..., :8], m, **kw)
    with pytest.raises(TypeError):
        F.cook_torrance(a, n.half(), r, m, **kw)
    with pytest.raises(ValueError):
        F.cook_torrance(a, n, r, m, view_dir=[0, 0, 1], light=[[0, 0, 1]] * 17, light_intensity=[1, 1, 1])
    with pytest.raises(RuntimeError, match="no CPU path"):
        F.cook_torrance(a.cpu(), n.cpu(), r.cpu(), m.cpu(), **kw)
    with pytest.raises(TypeError):
        F.cook_torrance(a, n, r, m, bogus_keyword=1, **kw)
    # directly at the operator: the same checks live in C++ (TORCH_CHECK_VALUE / _TYPE)
    v, l, i = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([[0.1, 0.1, 1.0]]), torch.ones(1, 3)
    with pytest.raises(ValueError):
        torch.ops.pbr_hip.cook_torrance(a, n, r, None, None, v, l, i, 1.0, 1, True, True, False, True)
    with pytest.raises(ValueError):
        torch.ops.pbr_hip.cook_torrance(a, n, r, m, None, v, l, i, 1.0, 7, True, True, False, True)
    with pytest.raises(TypeError):
        torch.ops.pbr_hip.cook_torrance(a, n, r.double(), m, None, v, l, i, 1.0, 1, True, True, False, True)
