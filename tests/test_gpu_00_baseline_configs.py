"""ALL FIVE BASELINE.json configurations against the oracles, FIRST in the suite (tests/conftest.py orders by purpose).

  configs[0]  single 256x256 BasecolorMetallicMaterial, point light, the examples/example_brdf.py path -- the reference's own PNG fixtures
              (golden outputs of the real reference, tests/golden/edge.npz) and SURVEY.md 8d's seed-0 synthetic material;
  configs[1]  B=1 4096x4096, point light, fp32: the WHOLE output against the C oracle (fp32 and float64 builds), bands against the ATen
              restatement of the reference (pinned bit-equal to it), plus the size-independent properties;
  configs[2]  B=64 2048x2048, directional light, sRGB decode + metallic -> diffuse/specular conversion fused, both F6 settings;
  configs[3]  B=512 1024x1024 over 8 GPUs: the per-GPU share, 64 materials, point light;
  configs[4]  B=32 4096x4096 over 8 GPUs, 16 point lights, fp16 maps, fp32 accumulate: the per-GPU share, 4 materials.

One launch over the whole batch each -- the XCD-run workgroup order, 8-pixel lanes and rows = B * H arithmetic at sizes the small tests
never reach -- then sampled parity: crops / row bands (first and LAST material, first and LAST rows included) against the ATen
restatement of the reference (fp32) and against float64 (the plain-C oracle), under the criterion of tests/test_gpu_parity.py.
Nothing in this file starts a process, reads a clock or depends on the host's speed."""
import math
import os
import warnings

import numpy as np
import pytest
import torch

import c_oracle as C
import torch_oracle as O
from test_gpu_parity import TOL, parity_report

pytestmark = pytest.mark.gpu


def _maps(B, H, W, seed, dtype=torch.float32):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.rand(B, 3, H, W, device="cuda", generator=g)
    nxy = torch.rand(B, 2, H, W, device="cuda", generator=g) - 0.5
    n = torch.cat([nxy, torch.ones(B, 1, H, W, device="cuda")], 1)
    n = n / n.norm(dim=1, keepdim=True)
    r = torch.rand(B, 1, H, W, device="cuda", generator=g) * 0.95 + 0.05
    m = torch.rand(B, 1, H, W, device="cuda", generator=g)
    return [t.to(dtype) for t in (a, n, r, m)]


def _windows(B, H, W, h, w, count, seed):
    """(b, y0, x0) of `count` windows: the four corners of the batch (first/last material, first/last rows and columns)
    plus seeded random ones."""
    rng = np.random.default_rng(seed)
    fixed = [(0, 0, 0), (B - 1, H - h, W - w), (B - 1, 0, W - w), (0, H - h, 0)]
    rand = [(int(rng.integers(B)), int(rng.integers(0, H - h + 1)), int(rng.integers(0, (W - w) // 8 + 1)) * 8) for _ in range(count - len(fixed))]
    return fixed + rand


def _check_properties(F, out, maps, kw):
    assert bool(torch.isfinite(out).all()) and float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    again = F.cook_torrance(*maps, **kw)
    assert torch.equal(out, again)
    del again


@pytest.mark.parametrize("folder,mean", [("tiles", 0.492009), ("rocks", 0.256256)])
def test_config0_256x256_cpu_material_from_the_reference_png_fixtures(golden, folder, mean):
    """BASELINE.json configs[0]: single 256x256 BasecolorMetallicMaterial, point light, the examples/example_brdf.py path with
    the material left on the CPU (uploaded, evaluated on the device, returned on the CPU)."""
    from pypbr_amd.io import load_material_from_folder
    from pypbr_amd.models import CookTorranceBRDF
    z = golden("edge")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        material = load_material_from_folder(os.path.join(os.path.dirname(__file__), "golden", folder), preferred_workflow="metallic")
    assert material.device.type == "cpu"
    material.resize((256, 256))
    assert material.size == (256, 256) and material.albedo.device.type == "cpu"
    out = CookTorranceBRDF(light_type="point")(material, torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]),
                                               torch.tensor([1.0, 1.0, 1.0]), 1.0)
    assert out.shape == (3, 256, 256) and out.device.type == "cpu"
    want = z[f"out_{folder}256"]
    err = np.abs(out.numpy() - want)
    print(f"\n[config0/{folder}256] max|hip-ref32| = {err.max():.2e}, mean {float(out.double().mean()):.6f} (reference {float(z[f'mean_{folder}256']):.6f})")
    assert err.max() <= TOL
    assert abs(float(out.double().mean()) - mean) <= 1e-6 and abs(float(out.double().mean()) - float(z[f"mean_{folder}256"])) <= 1e-6


def test_config0_256x256_seed0_synthetic_material_against_the_oracle_and_its_known_answer():
    """`torch.manual_seed(0)`, a, n, r, m drawn with the global CPU generator, metallic / point, size 1.0 --
    the reference's mean is 0.06412173807621002.  The material is built and rendered through the reference's surface."""
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    torch.manual_seed(0)
    a, n, r, m = torch.rand(3, 256, 256), torch.rand(3, 256, 256) * 2 - 1, torch.rand(1, 256, 256), torch.rand(1, 256, 256)
    view, light, inten = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0])
    ref32 = O.cook_torrance(a, n, r, m, None, view=view, light=light, intensity=inten, light_type="point", light_size=1.0)
    # the oracle IS the reference here too: SURVEY.md 8c quotes the fp32 mean (0.06412173807621002; ATen's fp32 summation order may differ
    # by an ulp with the thread count) and the max; the float64 mean of the same fp32 values is 0.06412173707398151
    assert abs(float(ref32.mean()) - 0.06412173807621002) <= 1e-8 and float(ref32.max()) == 0.9999999403953552
    assert abs(float(ref32.double().mean()) - 0.06412173707398151) <= 1e-9      # one ATen thread moves a few values by an ulp (seen: 1.1e-12 on the mean)
    ref64 = O.cook_torrance(a.double(), n.double(), r.double(), m.double(), None, view=view.double(), light=light.double(),
                            intensity=inten.double(), light_type="point", light_size=1.0)
    mat = BasecolorMetallicMaterial(albedo=a, normal=None, roughness=r, metallic=m)
    mat._maps["normal"] = n                                                          # as drawn: signed, not unit (the kernel normalises)
    got = CookTorranceBRDF("point")(mat, view, light, inten, 1.0)
    assert got.shape == (3, 256, 256) and got.device.type == "cpu"
    rep = parity_report(got.numpy(), ref32.numpy(), ref64.numpy(), r.numpy(), what=("baseline_cfg0_seed0",))
    assert abs(float(got.double().mean()) - 0.06412173807621002) <= 1e-6
    print(f"\n[config0/seed0 256^2] max|hip-ref32| {rep['max32']:.2e}, max|hip-ref64| {rep['max64']:.2e}, values > 1e-5: {rep['n_hip']} "
          f"(reference vs its own float64: {rep['n_ref']}); criterion (i) from roughness {rep['rough_needed']:.4f}")


def test_config1_1x4096x4096_point_fp32_whole_output_and_properties():
    """BASELINE.json config 2 size (1 x 4096 x 4096, point light).  The ATen oracle needs ~12 s per 4K map,
    the plain-C oracle ~1 s: full-map comparison against the C oracle, plus size-independent properties."""
    from pypbr_amd import functional as F
    H = W = 4096
    g = torch.Generator(device="cuda").manual_seed(1234)
    a = torch.rand(3, H, W, device="cuda", generator=g)
    nxy = torch.rand(2, H, W, device="cuda", generator=g) - 0.5
    n = torch.cat([nxy, torch.ones(1, H, W, device="cuda")], 0)
    n = n / n.norm(dim=0, keepdim=True)
    r = torch.rand(1, H, W, device="cuda", generator=g) * 0.95 + 0.05        # bench.py's roughness range
    m = torch.rand(1, H, W, device="cuda", generator=g)
    kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
    out = F.cook_torrance(a, n, r, m, **kw)
    assert out.shape == (3, H, W) and bool(torch.isfinite(out).all()) and float(out.min()) >= 0 and float(out.max()) <= 1
    # determinism / idempotence: a second launch is bit-identical
    assert torch.equal(out, F.cook_torrance(a, n, r, m, **kw))
    # tiling property: any aligned crop rendered as a band/window equals the crop of the full render (rows)
    y0 = 1234
    band = F.cook_torrance(a[:, y0:y0 + 64], n[:, y0:y0 + 64], r[:, y0:y0 + 64], m[:, y0:y0 + 64], y_offset=y0, height_total=H, **kw)
    assert torch.equal(band, out[:, y0:y0 + 64])
    # linear output then stand-alone encode == fused encode
    lin = F.cook_torrance(a, n, r, m, return_srgb=False, **kw)
    assert torch.equal(F.linear_to_srgb(lin), out)
    host = [t.cpu().numpy() for t in (a, n, r, m)]
    ckw = dict(view=[0, 0, 1], lights=[0.1, 0.1, 1.0], intensities=[1, 1, 1], light_type="point", light_size=1.0)
    ref32 = C.render(*host, None, **ckw)
    ref64 = C.render(*host, None, dtype=np.float64, **ckw)
    got = out.cpu().numpy()
    rep = parity_report(got, ref32, ref64, host[2], what="4096x4096 vs the C oracle")
    worst = 0.0
    for y0 in (0, 2040, H - 8):                     # first, middle, LAST rows against the ATen restatement of the reference
        crop = [t[:, y0:y0 + 8].cpu() for t in (a, n, r, m)]
        aten = O.cook_torrance(*crop, None, view=torch.tensor([0.0, 0.0, 1.0]), light=torch.tensor([0.1, 0.1, 1.0]), intensity=torch.ones(3),
                               light_type="point", light_size=1.0, y_offset=y0, H_total=H).numpy()
        band = parity_report(got[:, y0:y0 + 8], aten, ref64[:, y0:y0 + 8], host[2][:, y0:y0 + 8], what=("baseline_cfg1_aten_bands", y0))
        worst = max(worst, band["max32"])
    print(f"[4096x4096] 3 bands of 8 rows vs the ATen restatement: max|hip-ref32| {worst:.2e}")
    print(f"\n[4096x4096] max|hip - C oracle fp32| = {rep['max32']:.2e} ({rep['n_hip']} of {rep['n']} values > 1e-5; the fp32 C oracle "
          f"against its own fp64 build: {rep['n_ref']}); max|hip - C oracle fp64| = {rep['max64']:.2e}; "
          f"criterion (i) holds from roughness {rep['rough_needed']:.3f} up")


@pytest.mark.parametrize("quirk", [True, False])
def test_config3_full_shape_64x2048_converted_directional(quirk):
    """B=64 2048^2, directional light, sRGB decode + metallic -> diffuse/specular conversion fused, both settings of
    the upstream specular_is_srgb quirk."""
    from pypbr_amd import functional as F
    B, H, W = 64, 2048, 2048
    maps = _maps(B, H, W, seed=3)
    view, light, inten = [0.0, 0.0, 1.0], [0.3, -0.2, 1.0], [1.0, 1.0, 1.0]
    kw = dict(view_dir=view, light=light, light_intensity=inten, light_type="directional",
              convert_to_diffuse_specular=True, specular_is_srgb=quirk)
    plan = F.plan_cook_torrance(*maps, **kw)
    assert plan.kernel_name == "ct_directional_converted_f32_f32_v4"
    out = plan.launch()
    _check_properties(F, out, maps, kw)
    worst32 = worst64 = 0.0
    n_hip = n_ref = 0
    for b, y0, x0 in _windows(B, H, W, 128, 128, 10, seed=33):
        crop = [t[b, :, y0:y0 + 128, x0:x0 + 128].cpu() for t in maps]
        got = out[b, :, y0:y0 + 128, x0:x0 + 128].cpu().numpy()
        okw = dict(view=torch.tensor(view), light=torch.tensor(light), intensity=torch.tensor(inten), light_type="directional")
        ref32 = O.cook_torrance_converted(*crop, quirk_specular_srgb=quirk, **okw).numpy()
        ref64 = O.cook_torrance_converted(*[t.double() for t in crop], quirk_specular_srgb=quirk,
                                          **{k: (v.double() if isinstance(v, torch.Tensor) else v) for k, v in okw.items()}).numpy()
        rep = parity_report(got, ref32, ref64, crop[2].numpy(), what=("cfg3", quirk, b, y0, x0))
        worst32, worst64 = max(worst32, rep["max32"]), max(worst64, rep["max64"])
        n_hip, n_ref = n_hip + rep["n_hip"], n_ref + rep["n_ref"]
    print(f"\n[cfg3 64x2048^2 converted directional quirk={quirk}] 10 crops of 128^2: max|hip-ref32| {worst32:.2e}, "
          f"max|hip-ref64| {worst64:.2e}, values > 1e-5 vs ref32: {n_hip} (reference vs its own float64: {n_ref})")


def test_config4_share_full_shape_64x1024_point():
    """Per-GPU share of config 4: B=64 1024^2, point light.  Row bands span the full width (the point-light grid)."""
    from pypbr_amd import functional as F
    B, H, W = 64, 1024, 1024
    maps = _maps(B, H, W, seed=4)
    view, light, inten = [0.0, 0.0, 1.0], [0.1, 0.1, 1.0], [1.0, 1.0, 1.0]
    kw = dict(view_dir=view, light=light, light_intensity=inten, light_type="point", light_size=1.0)
    out = F.cook_torrance(*maps, **kw)
    _check_properties(F, out, maps, kw)
    band = F.cook_torrance(*[t[B - 1:, :, H - 24:] for t in maps], y_offset=H - 24, height_total=H, **kw)
    assert torch.equal(band, out[B - 1:, :, H - 24:])
    worst32 = worst64 = 0.0
    n_hip = n_ref = 0
    for b, y0, _ in _windows(B, H, W, 16, W, 10, seed=44):
        crop = [t[b, :, y0:y0 + 16].cpu() for t in maps]
        got = out[b, :, y0:y0 + 16].cpu().numpy()
        ref32 = O.cook_torrance(*crop, None, view=torch.tensor(view), light=torch.tensor(light), intensity=torch.tensor(inten),
                                light_type="point", light_size=1.0, y_offset=y0, H_total=H).numpy()
        ref64 = C.render(*[t.numpy() for t in crop], None, view=view, lights=light, intensities=inten, light_type="point",
                         light_size=1.0, y_offset=y0, H_total=H, dtype=np.float64)
        rep = parity_report(got, ref32, ref64, crop[2].numpy(), what=("cfg4", b, y0))
        worst32, worst64 = max(worst32, rep["max32"]), max(worst64, rep["max64"])
        n_hip, n_ref = n_hip + rep["n_hip"], n_ref + rep["n_ref"]
    print(f"\n[cfg4 share 64x1024^2 point] 10 bands of 16 rows: max|hip-ref32| {worst32:.2e}, max|hip-ref64| {worst64:.2e}, "
          f"values > 1e-5 vs ref32: {n_hip} (reference vs its own float64: {n_ref})")


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.float16])
def test_config5_share_full_shape_4x4096_16_lights_fp16(out_dtype):
    """Per-GPU share of config 5: B=4 4096^2, 16 point lights on a ring, fp16 maps, fp32 accumulate.  The oracle is fed
    the exact fp32 up-casts of the fp16 maps (SURVEY.md 8c iii); an fp16 result adds its own rounding (<= 4.9e-4)."""
    from pypbr_amd import functional as F
    B, H, W = 4, 4096, 4096
    maps = _maps(B, H, W, seed=5, dtype=torch.float16)
    lights = [[math.cos(2 * math.pi * i / 16), math.sin(2 * math.pi * i / 16), 1.0] for i in range(16)]
    inten = [[1.0 / 16] * 3] * 16
    view = [0.0, 0.0, 1.0]
    kw = dict(view_dir=view, light=lights, light_intensity=inten, light_type="point", light_size=1.0, out_dtype=out_dtype)
    out = F.cook_torrance(*maps, **kw)
    assert out.dtype == out_dtype
    _check_properties(F, out, maps, kw)
    worst32 = worst64 = 0.0
    for b, y0, _ in _windows(B, H, W, 4, W, 8, seed=55):
        crop = [t[b, :, y0:y0 + 4].float().cpu() for t in maps]
        got = out[b, :, y0:y0 + 4].float().cpu().numpy()
        ref32 = O.cook_torrance_multi(*crop, None, lights=torch.tensor(lights), intensities=torch.tensor(inten), view=torch.tensor(view),
                                      light_type="point", light_size=1.0, y_offset=y0, H_total=H).numpy()
        ref64 = C.render(*[t.numpy() for t in crop], None, view=view, lights=lights, intensities=inten, light_type="point",
                         light_size=1.0, y_offset=y0, H_total=H, dtype=np.float64)
        if out_dtype == torch.float32:
            rep = parity_report(got, ref32, ref64, crop[2].numpy(), what=("cfg5", b, y0))
            worst32, worst64 = max(worst32, rep["max32"]), max(worst64, rep["max64"])
        else:       # fp16 storage of the result: half an fp16 ulp below 1 on top of the fp32 criterion
            e64 = np.abs(got.astype(np.float64) - ref64)
            worst64 = max(worst64, float(e64.max()))
            assert e64.max() <= 4.9e-4 + 2e-6, (b, y0, float(e64.max()))
    print(f"\n[cfg5 share 4x4096^2 16 lights fp16 -> {out_dtype}] 8 bands of 4 rows: max|hip-ref32| {worst32:.2e}, max|hip-ref64| {worst64:.2e}")
