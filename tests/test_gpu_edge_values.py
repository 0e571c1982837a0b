"""Degenerate texel values through the forward and backward kernels: zero normals, roughness 0 and 1, metallic 0 and 1,
albedo at and beyond the [0,1] clamp, a light sitting exactly on a pixel, a view along the surface.  Results and
gradients must stay finite, and the forward must still agree with the oracle (which evaluates the reference's formulas)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _maps():
    H, W = 8, 16
    a = torch.zeros(3, H, W)
    a[:, 0] = 1.0; a[:, 1] = -0.25; a[:, 2] = 1.5; a[:, 3:] = torch.linspace(0, 1, W).expand(3, H - 3, W)
    n = torch.zeros(3, H, W)
    n[2, :, 4:] = 1.0                                   # columns 0-3: the zero vector (F.normalize gives 0)
    n[0, 4:, 8:] = 0.7; n[1, 5:, 8:] = -0.7
    r = torch.zeros(1, H, W)
    r[0, :, 1::4] = 1.0; r[0, :, 2::4] = 0.5; r[0, :, 3::4] = 1e-3
    m = torch.zeros(1, H, W)
    m[0, ::2] = 1.0; m[0, 1, :] = 0.5
    return a, n, r, m


CASES = [("point", [0.0, 0.0, 1.0], [0.1, 0.1, 1.0], 1.0), ("point", [0.0, 0.0, 1.0], [-0.5, 0.5, 0.0], 1.0),   # light ON the corner pixel
         ("point", [1.0, 0.0, 0.0], [0.2, 0.0, 0.3], 2.0), ("directional", [0.0, 0.0, 1.0], [0.0, 0.0, -1.0], None),  # L = -V
         ("directional", [0.3, 0.1, 1.0], [0.3, -0.2, 1.0], None)]


@pytest.mark.parametrize("light_type,view,light,size", CASES)
@pytest.mark.parametrize("srgb", [True, False])
def test_degenerate_values_stay_finite_and_match_the_oracle(light_type, view, light, size, srgb):
    import torch_oracle as O
    from pypbr_amd import functional as F
    a, n, r, m = _maps()
    kw = dict(view_dir=view, light=light, light_intensity=[1.0, 0.8, 0.6], light_type=light_type, light_size=size,
              albedo_is_srgb=srgb, return_srgb=srgb)
    out = F.cook_torrance(a.cuda(), n.cuda(), r.cuda(), m.cuda(), **kw)
    assert bool(torch.isfinite(out).all()) and float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    ref = O.cook_torrance(a, n, r, m, None, view=torch.tensor(view), light=torch.tensor(light), intensity=torch.tensor([1.0, 0.8, 0.6]),
                          light_type=light_type, light_size=size, albedo_is_srgb=srgb, return_srgb=srgb)
    # Audit (round 3): the reference's epsilons (1e-7 in every denominator, F.normalize's 1e-12) keep ALL of these degenerate
    # points finite -- zero normals, roughness 0, a light on a pixel, L = -V: the set of positions where it returns NaN / inf
    # is EMPTY for every case here, so nothing is excluded from the comparison on that account.
    n_bad = int((~torch.isfinite(ref)).sum())
    print(f"\n[edge values/{light_type}/{light}] reference non-finite values: {n_bad} of {ref.numel()}")
    assert n_bad == 0
    rough_ok = (r >= 0.2).expand_as(ref)                # below that the reference's own fp32 rounding exceeds 1e-5
    assert (out.cpu()[rough_ok] - ref[rough_ok]).abs().max().item() <= 1e-5
    # ... and where roughness is below that (0 and 1e-3 here) the build stays inside the float64 evaluation's neighbourhood
    ref64 = O.cook_torrance(a.double(), n.double(), r.double(), m.double(), None, view=torch.tensor(view, dtype=torch.float64),
                            light=torch.tensor(light, dtype=torch.float64), intensity=torch.tensor([1.0, 0.8, 0.6], dtype=torch.float64),
                            light_type=light_type, light_size=size, albedo_is_srgb=srgb, return_srgb=srgb)
    assert bool(torch.isfinite(ref64).all())
    assert (out.cpu().double() - ref64).abs().max().item() <= 1e-5
    leaves = [t.clone().cuda().requires_grad_(True) for t in (a, n, r, m)]
    F.cook_torrance(*leaves, **kw).sum().backward()
    for name, t in zip(("albedo", "normal", "roughness", "metallic"), leaves):
        assert bool(torch.isfinite(t.grad).all()), name


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(0, 5), (4, 0), (0, 0)])
def test_zero_sized_maps_give_an_empty_image_like_the_reference(shape):
    """The reference's forward is whole-map torch ops (cooktorrance.py:92-182): on maps with no pixels it returns an empty
    (3, H, W) image for both light types (checked against the ATen restatement); so does the build, without a launch."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import torch_oracle as O
    from pypbr_amd import functional as F
    h, w = shape
    a, r, m = torch.rand(3, h, w), torch.rand(1, h, w), torch.rand(1, h, w)
    for lt in ("point", "directional"):
        want = O.cook_torrance(a, None, r, m, None, view=torch.tensor([0.0, 0, 1]), light=torch.tensor([0.1, 0.1, 1.0]),
                               intensity=torch.tensor([1.0, 1, 1]), light_type=lt)
        got = F.cook_torrance(a.cuda(), None, r.cuda(), m.cuda(), view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1],
                              light_type=lt)
        assert tuple(got.shape) == tuple(want.shape) == (3, h, w) and got.is_cuda
    with pytest.raises(ValueError):           # the workflow check still applies (cooktorrance.py:113-114)
        F.cook_torrance(a.cuda(), None, r.cuda(), None, None, view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1])
