"""tests/golden/edge.npz (outputs of the REAL reference, oracle/gen_golden.py --only edge) through the HIP path:
  * `light_size or 1.0` (cooktorrance.py:130) is Python truthiness -- a negative size mirrors the point-light grid, 0.0 / -0.0 /
    None mean 1.0, NaN makes every value of the result NaN;
  * NaN texels: what the build returns where the reference returns NaN (stated, not hidden);
  * BASELINE.json configs[0] at the size it names: a CPU-resident 256x256 BasecolorMetallicMaterial loaded from the reference's
    PNG fixtures, resize((256, 256)), point light, through CookTorranceBRDF (SURVEY.md 8c anchors 0.492009 / 0.256256)."""

import numpy as np
import pytest
import torch

from test_gpu_parity import TOL, parity_report

pytestmark = pytest.mark.gpu
T = torch.from_numpy
SIZES = {"neg1": -1.0, "neg2p5": -2.5, "zero": 0.0, "negzero": -0.0, "none": None}


def _maps(z, kind, prefix="in_"):
    a, n, r = (T(z[prefix + k]).cuda() for k in ("albedo", "normal", "roughness"))
    m = T(z[prefix + "metallic"]).cuda() if kind == "metallic" else None
    s = T(z["in_specular"]).cuda() if kind == "specular" else None
    return a, n, r, m, s


@pytest.mark.parametrize("binding", ["torch_op", "ctypes"])
def test_light_size_follows_python_truthiness(golden, manifest, binding):
    import torch_oracle as O
    from pypbr_amd import functional as F
    z = golden("edge")
    F.USE_TORCH_OPS = binding == "torch_op"
    try:
        worst = 0.0
        for kind in ("metallic", "specular"):
            a, n, r, m, s = _maps(z, kind)
            cpu = [None if t is None else t.cpu() for t in (a, n, r, m, s)]
            for tag, size in SIZES.items():
                for cs in ("srgb", "lin"):
                    got = F.cook_torrance(a, n, r, m, s, view_dir=manifest["view1"], light=[0.1, 0.1, 1.0],
                                          light_intensity=manifest["intensity1"], light_type="point", light_size=size,
                                          return_srgb=(cs == "srgb")).cpu().numpy()
                    ref64 = O.cook_torrance(*[None if t is None else t.double() for t in cpu],
                                            view=torch.tensor(manifest["view1"], dtype=torch.float64),
                                            light=torch.tensor([0.1, 0.1, 1.0], dtype=torch.float64),
                                            intensity=torch.tensor(manifest["intensity1"], dtype=torch.float64), light_type="point",
                                            light_size=size, return_srgb=(cs == "srgb")).numpy()
                    rep = parity_report(got, z[f"out_{kind}_{tag}_{cs}"], ref64, z["in_roughness"], what=(kind, tag, cs), set_name="edge_golden")
                    worst = max(worst, rep["max32"])
            # the mirrored grid is not the default grid: the build must not fold a negative size into "not given"
            neg = F.cook_torrance(a, n, r, m, s, view_dir=manifest["view1"], light=[0.1, 0.1, 1.0], light_intensity=manifest["intensity1"],
                                  light_type="point", light_size=-1.0).cpu().numpy()
            assert np.abs(neg - z[f"out_{kind}_none_srgb"]).max() > 0.05
        print(f"\n[edge/{binding}] light_size in {list(SIZES.values())}: max|hip-ref32| = {worst:.2e}")
    finally:
        F.USE_TORCH_OPS = True


def test_nan_light_size_gives_nan_everywhere_like_the_reference(golden, manifest):
    """`nan or 1.0` is nan: the reference's grid, hence every value it returns, is NaN.  Directional lights never read the size."""
    from pypbr_amd import functional as F
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    z = golden("edge")
    a, n, r, m, _ = _maps(z, "metallic")
    kw = dict(view_dir=manifest["view1"], light=[0.1, 0.1, 1.0], light_intensity=manifest["intensity1"], light_size=float("nan"))
    assert np.isnan(z["out_metallic_nan_srgb"]).all()
    for use_op in (True, False):
        F.USE_TORCH_OPS = use_op
        try:
            out = F.cook_torrance(a, n, r, m, light_type="point", **kw)
            assert out.dtype == torch.float32 and bool(torch.isnan(out).all())
            dirn = F.cook_torrance(a, n, r, m, light_type="directional", **kw)
            assert bool(torch.isfinite(dirn).all())
        finally:
            F.USE_TORCH_OPS = True
    half = F.cook_torrance(a.half(), n.half(), r.half(), m.half(), light_type="point", out_dtype=torch.float16, **kw)
    assert half.dtype == torch.float16 and bool(torch.isnan(half).all())
    # batch, result placed in a strided arena view, several lights: every plane of every material is filled, nothing else
    B = 3
    arena = torch.full((B, 5, 33, 48), 7.0, device="cuda")
    out = arena[:, 1:4]
    F.cook_torrance(a.expand(B, -1, -1, -1).contiguous(), n, r, m, light_type="point", out=out, view_dir=[0, 0, 1],
                    light=[[0.1, 0.1, 1.0], [0.3, 0.0, 1.0]], light_intensity=[[1, 1, 1], [0.5, 0.5, 0.5]], light_size=float("nan"))
    assert bool(torch.isnan(arena[:, 1:4]).all()) and bool((arena[:, 0] == 7.0).all()) and bool((arena[:, 4] == 7.0).all())
    mat = BasecolorMetallicMaterial(albedo=a.cpu(), normal=None, roughness=r.cpu(), metallic=m.cpu())
    mat._maps["normal"] = n.cpu()
    img = CookTorranceBRDF("point")(mat, torch.tensor(manifest["view1"]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor(manifest["intensity1"]),
                                    float("nan"))
    assert img.device.type == "cpu" and bool(torch.isnan(img).all())


def test_negative_light_size_through_the_material_api_and_row_bands(golden, manifest):
    from pypbr_amd import functional as F
    from pypbr_amd.materials import DiffuseSpecularMaterial
    from pypbr_amd.models import CookTorranceBRDF
    z = golden("edge")
    a, n, r, _, s = _maps(z, "specular")
    mat = DiffuseSpecularMaterial(albedo=a, normal=None, roughness=r, specular=s)
    mat._maps["normal"] = n
    view, light, inten = torch.tensor(manifest["view1"]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor(manifest["intensity1"])
    img = CookTorranceBRDF("point")(mat, view, light, inten, -2.5)
    well = np.broadcast_to(z["in_roughness"] >= 0.185, (3, 33, 48))
    assert np.abs(img.cpu().numpy() - z["out_specular_neg2p5_srgb"])[well].max() <= TOL
    # a row band of the mirrored grid (multi-GPU split of one material) equals the rows of the full evaluation, bit for bit
    band = F.cook_torrance(a[:, 10:21], n[:, 10:21], r[:, 10:21], None, s[:, 10:21], view_dir=view, light=light, light_intensity=inten,
                           light_type="point", light_size=-2.5, y_offset=10, height_total=33)
    assert torch.equal(band, img[:, 10:21])


def test_nan_texels_what_the_build_returns_where_the_reference_returns_nan(golden, manifest):
    """A NaN texel is not a valid map value.  The reference propagates it (torch.clamp, pow and the masked colour transfers keep
    NaN): a NaN albedo value makes its own colour channel NaN at that pixel, a NaN normal / roughness / metallic value all
    three (pinned in tests/test_oracle_pin.py).  The kernels' clamps (v_med3 / the VOP3P clamp modifier: IEEE minNum/maxNum)
    return the non-NaN operand, so the build returns a FINITE value in [0, 1] at exactly those positions and the reference's
    values everywhere else.  Counted and printed; DESIGN.md 4 states it."""
    from pypbr_amd import functional as F
    z = golden("edge")
    a, n, r, m = (T(z["in_nan_" + k]).cuda() for k in ("albedo", "normal", "roughness", "metallic"))
    for lk in ("pt1", "dir"):
        ltype, lvec, lsize = manifest["lights"][lk]
        want = z[f"out_nantexel_{lk}"]
        got = F.cook_torrance(a, n, r, m, view_dir=manifest["view0"], light=lvec, light_intensity=manifest["intensity0"],
                              light_type=ltype, light_size=lsize).cpu().numpy()
        ref_nan = np.isnan(want)
        assert int(ref_nan.sum()) == 10                               # 1 (albedo channel) + 3 x 3
        well = np.broadcast_to(z["in_roughness"] >= 0.185, want.shape) & ~ref_nan
        assert np.abs(got - want)[well].max() <= TOL                  # a NaN texel does not leak into any other pixel
        there = got[ref_nan]
        print(f"\n[edge/nan texels/{lk}] reference NaN at {int(ref_nan.sum())} values; build there: "
              f"{int(np.isnan(there).sum())} NaN, {int(np.isfinite(there).sum())} finite in [{np.nanmin(there):.3f}, {np.nanmax(there):.3f}]")
        assert np.isfinite(there).all() and there.min() >= 0.0 and there.max() <= 1.0
