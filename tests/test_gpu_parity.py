"""GPU parity: the HIP path (through the C ABI) against the committed golden vectors
(outputs of the real reference) and against the oracles on seeded inputs.

Criterion (DESIGN.md "Parity criterion").  TOL = 1e-5 absolute on fp32 outputs is the
tolerance BASELINE.json's north_star states.  ref32 = the reference's own fp32 output (golden vectors; on seeded
inputs the ATen restatement, pinned bit-equal to it), ref64 = the same reference code run in float64.
  (i)   |hip - ref32| <= TOL on EVERY value of: crops of the reference's own PNG fixtures (tiles, rocks), both example
        scripts, and synthetic maps wherever roughness lies above the set's threshold -- a MEASURED output of the suite, recorded per
        set in tests/golden/parity_thresholds.json and not allowed to rise (tests/test_gpu_zz_parity_table.py).
  (ii)  Below that roughness the reference's OWN fp32 output is not reproducible to 1e-5 by anything that is not
        bit-identical to ATen: its GGX denominator NdotH^2 (a^2-1) + 1 cancels, and its fp32 run is up to 5.6e-5 away
        from its own float64 run on these very fixtures.  There, on every value of every variant:
        (a) |hip - ref64| <= TOL (measured <= 6e-7: TRACK asserts <= 2e-6) -- the build is far closer to the exact value
            of the reference's formula than the reference's fp32 run is;
        (b) |hip - ref32| <= |ref32 - ref64| + TOL -- every difference above TOL lies inside the reference's own fp32
            rounding envelope;
        (c) count(|hip - ref32| > TOL) <= count(|ref32 - ref64| > TOL - TRACK): the build exceeds TOL against ref32 on
            no more values than the reference exceeds it against its own float64 run.  SURVEY.md's absolute bound
            count <= 2e-5 * N is ASSERTED per set wherever the reference's own count meets it (the closing test of the suite);
            the reference itself misses it on low-roughness sets: 3.4e-5 * N on rand64, 1.4e-4 * N on real48.
        No blanket bound: variants whose float64 twin is not in the fixtures get it from the pinned oracle's float64
        mode (tests/test_oracle_pin.py checks that mode bit-equal to the committed float64 runs).
"""
import numpy as np
import pytest
import torch

from conftest import RANDOM_SETS, oracle_render, parse_case, render_keys

pytestmark = pytest.mark.gpu

from conftest import PARITY_SETS, parity_threshold

TOL = 1e-5
TRACK = 2e-6          # |hip - ref64|: what the cancellation-free kernel is held to (measured 5.4e-7)
ROUGH_OK = 0.185      # where seeded test inputs start their roughness range when they want criterion (i) on every value; the threshold the
                      # suite ASSERTS is per set and measured: tests/golden/parity_thresholds.json (conftest.parity_threshold)


def parity_report(got, ref32, ref64, rough=None, what="", set_name=None):
    """Asserts criterion (i) where `rough` > the set's recorded threshold and (ii a-c) everywhere; returns the numbers for the log and
    adds them to the set's row of the suite's parity table (conftest.PARITY_SETS; closed by tests/test_gpu_zz_parity_table.py: SURVEY 8c's
    count bound 2e-5 N wherever the reference itself meets it, and the measured threshold must not have risen)."""
    set_name = set_name or (str(what[0]) if isinstance(what, tuple) and what else str(what))
    thr = parity_threshold(set_name)
    got64 = got.astype(np.float64)
    err, e64 = np.abs(got64 - ref32), np.abs(got64 - ref64)
    env = np.abs(ref32.astype(np.float64) - ref64)
    n_hip, n_ref = int((err > TOL).sum()), int((env > TOL - TRACK).sum())
    assert e64.max() <= TRACK, (what, "vs float64 reference", float(e64.max()))
    assert (err <= env + TOL).all(), (what, "outside the reference's own envelope", float((err - env).max()))
    assert n_hip <= n_ref, (what, "more values over TOL than the reference has against its own float64 run", n_hip, n_ref)
    need = 0.0
    if rough is not None:
        well = np.broadcast_to(rough > thr, err.shape)
        if well.any():
            assert err[well].max() <= TOL, (what, "roughness > %.4f (the set's recorded threshold)" % thr, float(err[well].max()))
        bad = err > TOL
        if bad.any():
            need = float(np.broadcast_to(rough, err.shape)[bad].max())
    row = PARITY_SETS.setdefault(set_name, dict(n=0, n_hip=0, n_ref=0, rough_needed=0.0, max32=0.0, max64=0.0, with_roughness=False))
    row["n"] += err.size; row["n_hip"] += n_hip; row["n_ref"] += n_ref
    row["rough_needed"] = max(row["rough_needed"], need); row["max32"] = max(row["max32"], float(err.max())); row["max64"] = max(row["max64"], float(e64.max()))
    row["with_roughness"] = row["with_roughness"] or rough is not None
    return dict(max32=float(err.max()), max64=float(e64.max()), n_hip=n_hip, n_ref=n_ref, n=err.size, rough_needed=need)


def _dev(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _hip_render(z, case, prefix="in_"):
    from pypbr_amd import functional as F
    kind = case["kind"]
    a = _dev(z[prefix + "albedo"])
    n = None if case["no_normal"] else _dev(z[prefix + "normal"])
    r = _dev(z[prefix + "roughness"])
    m = _dev(z[prefix + "metallic"]) if kind in ("metallic", "converted") else None
    s = _dev(z[prefix + "specular"]) if kind == "specular" else None
    lin = case["linear_maps"]
    out = F.cook_torrance(
        a, n, r, m, s, view_dir=case["view"], light=case["light"], light_intensity=case["intensity"],
        light_type=case["light_type"], light_size=case["light_size"],
        albedo_is_srgb=not lin,
        specular_is_srgb=(case["extra"] == "quirk") if kind == "converted" else (not lin),
        return_srgb=case["return_srgb"], convert_to_diffuse_specular=(kind == "converted"))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("name", RANDOM_SETS)
def test_golden_random_sets(name, golden, manifest):
    z = golden(name)
    rough = z["in_roughness"]                       # (1,H,W)
    tot = dict(max32=0.0, max64=0.0, n_hip=0, n_ref=0, n=0, rough_needed=0.0)
    for key in render_keys(z):
        case = parse_case(key, manifest)
        got, ref = _hip_render(z, case), z[key]
        assert got.shape == ref.shape and got.dtype == np.float32
        k64 = "f64_" + key[4:]
        ref64 = z[k64] if k64 in z else oracle_render(z, case, dtype=torch.float64).numpy()
        rep = parity_report(got, ref, ref64, rough, what=(name, key))
        for k in ("max32", "max64", "rough_needed"):
            tot[k] = max(tot[k], rep[k])
        for k in ("n_hip", "n_ref", "n"):
            tot[k] += rep[k]
    print(f"\n[{name}] {tot['n']} values: max|hip-ref32| {tot['max32']:.2e}, max|hip-ref64| {tot['max64']:.2e}; values > {TOL:g} vs ref32: "
          f"{tot['n_hip']} (reference vs its own float64 run: {tot['n_ref']}; 2e-5*N = {2e-5 * tot['n']:.1f}); "
          f"criterion (i) holds above roughness {tot['rough_needed']:.4f} (asserted above {parity_threshold(name):.4f})")


@pytest.mark.parametrize("name", ["tiles96", "rocks96"])
def test_golden_reference_fixtures(name, golden, manifest):
    """Crops of the PNG fixtures the reference's own tests hold: criterion (i), every pixel."""
    z = golden(name)
    worst = 0.0
    for kind in ("metallic", "specular"):
        for lk in ("pt1", "dir"):
            for cs in ("srgb", "lin"):
                key = f"out_{kind}_{lk}_{cs}"
                case = parse_case(key, manifest)
                got = _hip_render(z, case, prefix=f"in_{kind}_")
                err = np.abs(got - z[key])
                worst = max(worst, float(err.max()))
                assert err.max() <= TOL, (name, key, float(err.max()))
        got = _hip_render(z, parse_case(f"out_{kind}_pt1_srgb", manifest), prefix=f"in_{kind}_")
        assert np.abs(got.astype(np.float64) - z[f"f64_{kind}_pt1_srgb"]).max() <= TOL
    print(f"\n[{name}] max|hip-ref32| = {worst:.2e}")


def test_known_answer_means(golden, manifest):
    """SURVEY.md 8c known answers (means of the reference output, seed 1234, 64x64)."""
    z = golden("rand64")
    known = {"out_metallic_pt1_srgb": 0.064043984, "out_metallic_pt5_srgb": 0.000869892,
             "out_metallic_dir_srgb": 0.070766144, "out_specular_pt1_srgb": 0.094788788,
             "out_specular_pt5_srgb": 0.001569378, "out_specular_dir_srgb": 0.104157447}
    for key, mean in known.items():
        assert abs(float(z[key].astype(np.float64).mean()) - mean) < 5e-9          # the fixture is the survey's
        got = _hip_render(z, parse_case(key, manifest))
        assert abs(float(got.astype(np.float64).mean()) - mean) < 2e-7, key


def test_decode_normal_one_pass_equals_two_pass():
    """pbr_decode_normal reads a 3-channel map ONCE when source and destination are disjoint (a 4096-sample probe, then a
    speculative decode that records negatives exactly, then a fix-up that only runs if one turned up); in place it keeps
    the flag pass first.  Both must give the same bits: for encoded maps, for maps whose only negative value is one the
    probe does not look at (the last element; index 1 of a large map), for signed maps, and for odd sizes / fp16."""
    from pypbr_amd import _native as N, functional as F
    lib = N.lib()
    g = torch.Generator(device="cuda").manual_seed(5)
    for dtype in (torch.float32, torch.float16):
        for h, w in ((64, 64), (37, 53), (1, 1), (513, 1027)):
            for case in ("encoded", "last_negative", "second_negative", "first_negative", "signed"):
                x = torch.rand(3, h, w, device="cuda", generator=g).to(dtype)
                if case == "last_negative":
                    x[2, -1, -1] = -0.25
                elif case == "second_negative":
                    x.view(-1)[min(1, x.numel() - 1)] = -0.25
                elif case == "first_negative":
                    x[0, 0, 0] = -0.25
                elif case == "signed":
                    x = x * 2 - 1
                    x[0, 0, 0] = -0.5                                           # 1 x 1 maps too
                one_pass = F.decode_normal(x)
                two_pass, flag = x.clone(), torch.empty(1, dtype=torch.int32, device="cuda")
                N.check(lib.pbr_decode_normal(two_pass.data_ptr(), two_pass.data_ptr(), 3, h * w, F._DTYPES[dtype], flag.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream))
                assert torch.equal(one_pass, two_pass), (dtype, h, w, case)
                if case == "encoded":
                    assert flag.item() == 0 and not torch.equal(one_pass, x)
                    n = one_pass.float().norm(dim=0)
                    assert (n - 1).abs().max().item() <= (2e-6 if dtype == torch.float32 else 2e-3)
                else:
                    assert flag.item() == 1 and torch.equal(one_pass, x)          # kept as it is (base.py:212-213)


def test_conversions_and_colour(golden):
    from pypbr_amd import functional as F
    z = golden("misc")
    for nm in ("ramp", "rand", "knees"):
        x = _dev(z[f"in_colour_{nm}"])
        assert np.abs(F.srgb_to_linear(x).cpu().numpy() - z[f"out_s2l_{nm}"]).max() <= 2e-6
        assert np.abs(F.linear_to_srgb(x).cpu().numpy() - z[f"out_l2s_{nm}"]).max() <= 2e-6
    for nm in ("rgb01", "signed", "xy"):
        got = F.decode_normal(_dev(z[f"in_normal_{nm}"])).cpu().numpy()
        assert got.shape == z[f"out_normal_{nm}"].shape
        assert np.abs(got - z[f"out_normal_{nm}"]).max() <= 2e-6, nm
    assert np.array_equal(F.decode_normal(_dev(z["in_normal_signed"])).cpu().numpy(), z["in_normal_signed"])
    with pytest.raises(ValueError, match="2 or 3 channels"):
        F.decode_normal(torch.rand(4, 8, 8, device="cuda"))
    for name in ("rand64", "rand37x53"):
        g = golden(name)
        d, s = F.metallic_to_diffuse_specular(_dev(g["in_albedo"]), _dev(g["in_metallic"]), albedo_is_srgb=True)
        assert np.abs(d.cpu().numpy() - g["out_conv_diffuse"]).max() <= 2e-6
        assert np.abs(s.cpu().numpy() - g["out_conv_specular"]).max() <= 2e-6
        b, m = F.diffuse_specular_to_basecolor_metallic(_dev(g["in_albedo"]), _dev(g["in_specular"]), albedo_is_srgb=True)
        eb, em = np.abs(b.cpu().numpy() - g["out_back_basecolor"]), np.abs(m.cpu().numpy() - g["out_back_metallic"])
        # diffuse.py:128-147 is discontinuous and, near diffuse = 0.04, ill-conditioned: metallic = (s - 0.04) / (den + 1e-6)
        # with den = d - 0.04 + 1e-6, zeroed where den < 1e-6; basecolor switches to the specular colour at metallic >= 0.95.
        # Every value that is off by more than 1e-4 must be explained by the float64 evaluation of the reference's own
        # formula: (a) den within 1e-4 of 0 -- one ulp of d (3.7e-9) then moves the quotient by > 1e-5 relative, and the
        # den < eps branch sits there too -- or (b) metallic within 1e-5 of the 0.95 threshold (a tie).  No allowance by count.
        lin = O_srgb_to_linear64(g["in_albedo"])
        den64 = lin - 0.04 + 1e-6
        m64 = np.clip((g["in_specular"].astype(np.float64) - 0.04) / (den64 + 1e-6), 0.0, 1.0)
        m64 = np.where(den64 < 1e-6, 0.0, m64)
        explained = (np.abs(den64) <= 1e-4) | (np.abs(m64 - 0.95) <= 1e-5)
        bad = (em > 1e-4) | (eb > 1e-4)
        assert not (bad & ~explained).any(), (name, int((bad & ~explained).sum()), float(em[bad & ~explained].max(initial=0)), float(eb[bad & ~explained].max(initial=0)))
        print(f"\n[{name}] to_basecolor_metallic: {int(bad.sum())} of {bad.size} values off by > 1e-4, all at den ~ 0 or metallic ~ 0.95 "
              f"({int(explained.sum())} such values in the set); elsewhere max {max(float(em[~explained].max()), float(eb[~explained].max())):.2e}")
        assert em[~explained].max() <= 1e-4 and eb[~explained].max() <= 1e-4
        assert np.median(em) <= 1e-7 and np.median(eb) <= 1e-7


def O_srgb_to_linear64(x):
    """float64 evaluation of utils.srgb_to_linear (functions.py:31-47) for the conditioning analysis above."""
    t = np.clip(x.astype(np.float64), 0.0, 1.0)
    return np.clip(np.where(t <= 0.04045, t / 12.92, ((t + 0.055) / 1.055) ** 2.4), 0.0, 1.0)


def test_multilight_matches_composition_of_reference_calls(golden):
    from pypbr_amd import functional as F
    z = golden("misc")
    args = [_dev(z[f"in_ml_{k}"]) for k in ("albedo", "normal", "roughness", "metallic")]
    for srgb, key in ((False, "out_ml_lin"), (True, "out_ml_srgb")):
        out = F.cook_torrance(*args, view_dir=[0, 0, 1], light=z["in_ml_lights"], light_intensity=z["in_ml_intensities"],
                              light_type="point", light_size=1.0, return_srgb=srgb)
        assert np.abs(out.cpu().numpy() - z[key]).max() <= TOL, key


def _seeded(seed, B, H, W, rough_lo=ROUGH_OK):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(B, 3, H, W, generator=g)
    nxy = torch.rand(B, 2, H, W, generator=g) - 0.5
    n = torch.cat([nxy, torch.ones(B, 1, H, W)], 1)
    n = n / n.norm(dim=1, keepdim=True)
    r = torch.rand(B, 1, H, W, generator=g) * (1 - rough_lo) + rough_lo
    m = torch.rand(B, 1, H, W, generator=g)
    s = torch.rand(B, 3, H, W, generator=g)
    return a, n, r, m, s


@pytest.mark.parametrize("light_type,light,size", [("point", [0.1, 0.1, 1.0], 1.0), ("directional", [0.3, -0.2, 1.0], None)])
@pytest.mark.parametrize("workflow", ["metallic", "specular", "converted"])
def test_batched_equals_loop_of_oracle_calls(workflow, light_type, light, size):
    """[B,C,H,W] batches against a loop of per-material ATen-oracle evaluations."""
    import torch_oracle as O
    from pypbr_amd import functional as F
    B, H, W = 3, 40, 72
    a, n, r, m, s = _seeded(11, B, H, W)
    kw = dict(view=torch.tensor([0.1, 0.05, 1.0]), light=torch.tensor(light), intensity=torch.tensor([1.0, 0.9, 0.8]),
              light_type=light_type, light_size=size)
    if workflow == "converted":
        ref = O.cook_torrance_batched(a, n, r, m, None, converted=True, quirk_specular_srgb=True, **kw)
    else:
        ref = O.cook_torrance_batched(a, n, r, m if workflow == "metallic" else None,
                                      s if workflow == "specular" else None, **kw)
    out = F.cook_torrance(a.cuda(), n.cuda(), r.cuda(), None if workflow == "specular" else m.cuda(),
                          s.cuda() if workflow == "specular" else None, view_dir=kw["view"], light=kw["light"],
                          light_intensity=kw["intensity"], light_type=light_type, light_size=size,
                          convert_to_diffuse_specular=(workflow == "converted"))
    assert out.shape == (B, 3, H, W)
    assert (out.cpu() - ref).abs().max().item() <= TOL


def test_row_bands_tile_the_full_map():
    """(y_offset, height_total): bands of a taller map reproduce the full evaluation exactly
    (multi-GPU row sharding of one material)."""
    from pypbr_amd import functional as F
    a, n, r, m, _ = _seeded(5, 1, 50, 64)
    a, n, r, m = a[0].cuda(), n[0].cuda(), r[0].cuda(), m[0].cuda()
    kw = dict(view_dir=[0, 0, 1], light=[0.2, -0.1, 0.8], light_intensity=[1, 1, 1], light_type="point", light_size=2.0)
    full = F.cook_torrance(a, n, r, m, **kw)
    bands = []
    for y0, y1 in ((0, 17), (17, 18), (18, 50)):
        bands.append(F.cook_torrance(a[:, y0:y1], n[:, y0:y1], r[:, y0:y1], m[:, y0:y1], y_offset=y0, height_total=50, **kw))
    assert torch.equal(torch.cat(bands, dim=1), full)


def test_ragged_and_unaligned_views_use_the_scalar_kernel():
    import torch_oracle as O
    from pypbr_amd import functional as F
    a, n, r, m, _ = _seeded(6, 1, 33, 64)
    kw = dict(view=torch.tensor([0.0, 0.0, 1.0]), light=torch.tensor([0.1, 0.1, 1.0]), intensity=torch.tensor([1.0, 1.0, 1.0]),
              light_type="point", light_size=1.0)
    # width 61: not a multiple of 4; column slice: unaligned row starts
    for sl in (slice(0, 61), slice(3, 64)):
        aa, nn, rr, mm = [t[0][:, :, sl].contiguous() for t in (a, n, r, m)]
        ref = O.cook_torrance(aa, nn, rr, mm, None, **kw)
        out = F.cook_torrance(aa.cuda(), nn.cuda(), rr.cuda(), mm.cuda(), view_dir=kw["view"], light=kw["light"],
                              light_intensity=kw["intensity"], light_type="point", light_size=1.0)
        assert (out.cpu() - ref).abs().max().item() <= TOL


def test_fp16_maps_fp32_accumulate():
    """Config 5 storage: maps quantised to fp16, oracle fed their exact fp32 up-casts (SURVEY.md 8c iii)."""
    import torch_oracle as O
    from pypbr_amd import functional as F
    a, n, r, m, _ = [t[0].half() for t in _seeded(9, 1, 48, 96, rough_lo=0.15)]
    lights = torch.tensor([[np.cos(t), np.sin(t), 1.0] for t in np.linspace(0, 2 * np.pi, 16, endpoint=False)], dtype=torch.float32)
    inten = torch.full((16, 3), 1.0 / 16)
    ref = O.cook_torrance_multi(a.float(), n.float(), r.float(), m.float(), None, lights=lights, intensities=inten,
                                view=torch.tensor([0.0, 0.0, 1.0]), light_type="point", light_size=1.0)
    out = F.cook_torrance(a.cuda(), n.cuda(), r.cuda(), m.cuda(), view_dir=[0, 0, 1], light=lights, light_intensity=inten,
                          light_type="point", light_size=1.0)
    assert out.dtype == torch.float32
    assert (out.cpu() - ref).abs().max().item() <= TOL
    out16 = F.cook_torrance(a.cuda(), n.cuda(), r.cuda(), m.cuda(), view_dir=[0, 0, 1], light=lights, light_intensity=inten,
                            light_type="point", light_size=1.0, out_dtype=torch.float16)
    assert out16.dtype == torch.float16
    assert (out16.float().cpu() - ref).abs().max().item() <= 4.9e-4 + TOL      # fp16 rounding of values <= 1


def test_material_api_end_to_end(golden, manifest):
    """The reference-shaped callable: CPU-resident material (example_brdf.py style) and .to('cuda')."""
    from pypbr_amd.materials import BasecolorMetallicMaterial, DiffuseSpecularMaterial
    from pypbr_amd.models import CookTorranceBRDF
    z = golden("tiles96")
    view, light, inten = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0])
    mat = BasecolorMetallicMaterial(albedo=torch.from_numpy(z["in_metallic_albedo"]), normal=torch.from_numpy(z["in_metallic_normal"]),
                                    roughness=torch.from_numpy(z["in_metallic_roughness"]), metallic=torch.from_numpy(z["in_metallic_metallic"]))
    brdf = CookTorranceBRDF(light_type="point")
    out_cpu = brdf(mat, view, light, inten, 1.0)
    assert out_cpu.device.type == "cpu" and out_cpu.shape == (3, 96, 96)
    assert np.abs(out_cpu.numpy() - z["out_metallic_pt1_srgb"]).max() <= TOL
    out_gpu = brdf(mat.to("cuda"), view, light, inten, 1.0)
    assert out_gpu.device.type == "cuda" and torch.equal(out_gpu.cpu(), out_cpu)
    # specular workflow + directional + linear output
    ds = DiffuseSpecularMaterial(albedo=torch.from_numpy(z["in_specular_albedo"]), normal=torch.from_numpy(z["in_specular_normal"]),
                                 roughness=torch.from_numpy(z["in_specular_roughness"]), specular=torch.from_numpy(z["in_specular_specular"]),
                                 device=torch.device("cuda"))
    out = CookTorranceBRDF("directional")(ds, view, torch.tensor([0.3, -0.2, 1.0]), inten, return_srgb=False)
    assert np.abs(out.cpu().numpy() - z["out_specular_dir_lin"]).max() <= TOL
    # F6: converted material, upstream default flag vs decoded-once
    g = golden("rand64")
    m2 = BasecolorMetallicMaterial(albedo=torch.from_numpy(g["in_albedo"]), normal=torch.from_numpy(g["in_normal"]),
                                   roughness=torch.from_numpy(g["in_roughness"]), metallic=torch.from_numpy(g["in_metallic"])).to("cuda")
    conv = m2.to_diffuse_specular_material()
    assert conv.albedo_is_srgb is False and conv.specular_is_srgb is True
    assert np.abs(conv.albedo.cpu().numpy() - g["out_conv_diffuse"]).max() <= 2e-6
    d = CookTorranceBRDF("directional")
    q = d(conv, view, torch.tensor([0.3, -0.2, 1.0]), inten).cpu().numpy()
    conv.specular_is_srgb = False
    f = d(conv, view, torch.tensor([0.3, -0.2, 1.0]), inten).cpu().numpy()
    well = np.broadcast_to(g["in_roughness"] >= ROUGH_OK, q.shape)
    assert np.abs(q - g["out_converted_dir_srgb_quirk"])[well].max() <= TOL
    assert np.abs(f - g["out_converted_dir_srgb_fixed"])[well].max() <= TOL
    assert np.abs(q - f).max() > 0.1                                            # the quirk is large
    # F7: +Z only after an explicit `mat.normal = None`; a missing entry raises AttributeError
    bare = BasecolorMetallicMaterial(albedo=torch.rand(3, 8, 8), roughness=torch.rand(1, 8, 8), metallic=torch.rand(1, 8, 8))
    with pytest.raises(AttributeError):
        brdf(bare, view, light, inten, 1.0)
    bare.normal = None
    assert brdf(bare, view, light, inten, 1.0).shape == (3, 8, 8)
    from pypbr_amd.materials import MaterialBase
    plain = MaterialBase(albedo=torch.rand(3, 8, 8), roughness=torch.rand(1, 8, 8))
    plain.normal = None
    with pytest.raises(ValueError, match="either 'metallic' or 'specular'"):
        brdf(plain, view, light, inten)


def test_launch_is_stream_ordered_and_graph_capturable():
    """The C-ABI contract: a call only enqueues on the given stream -- no host sync, no allocation, no
    state.  So it can run on a side stream and be captured into a HIP graph and replayed."""
    from pypbr_amd import functional as F
    a, n, r, m, _ = [t[0].cuda() for t in _seeded(8, 1, 64, 128)]
    kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
    ref = F.cook_torrance(a, n, r, m, **kw).clone()
    plan = F.plan_cook_torrance(a, n, r, m, **kw)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        plan.out.zero_()
        plan.launch()                                   # picks up torch's current (side) stream
    side.synchronize()
    assert torch.equal(plan.result, ref)
    graph = torch.cuda.CUDAGraph()
    plan.out.zero_()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        plan.launch()
    plan.out.zero_()
    a.mul_(0.5)                                         # the graph reads the maps at replay time
    graph.replay()
    torch.cuda.synchronize()
    a.mul_(2.0)
    half = plan.result.clone()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(plan.result, ref) and not torch.equal(half, ref)


def test_shape_fuzz_against_c_oracle():
    """Ragged shapes, tile edges and batch/row arithmetic: 60 seeded (B, H, W) draws x workflows x
    light types, fp32 and fp16 maps, against the plain-C oracle (float64 build = truth)."""
    import c_oracle as C
    from pypbr_amd import functional as F
    rng = np.random.default_rng(2025)
    shapes = [(1, 1, 1), (1, 1, 4), (2, 3, 8), (1, 2, 256), (1, 1, 260), (3, 5, 12), (1, 64, 4), (2, 65, 20), (1, 7, 1028)]
    while len(shapes) < 60:
        shapes.append((int(rng.integers(1, 4)), int(rng.integers(1, 40)), int(rng.integers(1, 300))))
    worst = 0.0
    for i, (B, H, W) in enumerate(shapes):
        a = rng.random((B, 3, H, W), dtype=np.float32)
        n = np.concatenate([rng.random((B, 2, H, W), dtype=np.float32) - 0.5, np.ones((B, 1, H, W), np.float32)], 1)
        r = rng.random((B, 1, H, W), dtype=np.float32) * 0.7 + 0.3
        m = rng.random((B, 1, H, W), dtype=np.float32)
        s = rng.random((B, 3, H, W), dtype=np.float32) * 0.3
        workflow = ("metallic", "specular", "converted")[i % 3]
        ltype = ("point", "directional")[(i // 3) % 2]
        nl = (1, 3)[(i // 6) % 2]
        half = (i // 12) % 2 == 1
        if half:
            a, n, r, m, s = [x.astype(np.float16).astype(np.float32) for x in (a, n, r, m, s)]
        lights = rng.random((nl, 3)) * [0.8, 0.8, 0.5] + [-0.4, -0.4, 0.6]
        inten = rng.random((nl, 3)) * 0.8 + 0.2
        view = [0.1, -0.05, 1.0]
        kw = dict(view=view, lights=lights, intensities=inten, light_type=ltype, light_size=1.3, workflow=workflow,
                  specular_is_srgb=True)
        ref = C.render(a, n, r, None if workflow == "specular" else m, s if workflow == "specular" else None,
                       dtype=np.float64, **kw)
        dt = torch.float16 if half else torch.float32
        dev = [None if x is None else torch.from_numpy(x).to(dt).cuda() for x in
               (a, n, r, None if workflow == "specular" else m, s if workflow == "specular" else None)]
        out = F.cook_torrance(*dev, view_dir=view, light=lights, light_intensity=inten, light_type=ltype, light_size=1.3,
                              convert_to_diffuse_specular=(workflow == "converted"))
        assert out.shape == (B, 3, H, W) and out.dtype == torch.float32
        err = float(np.abs(out.cpu().numpy().astype(np.float64) - ref).max())
        worst = max(worst, err)
        assert err <= TOL, ((B, H, W), workflow, ltype, nl, half, err)
    print(f"\n[shape fuzz] worst |hip - C oracle fp64| over {len(shapes)} cases: {worst:.2e}")


def test_resize_matches_reference_vectors(golden):
    """MaterialBase.resize (N1): antialiased bilinear, tuple and int sizes, up and down."""
    from pypbr_amd import functional as F
    z = golden("example")
    x = _dev(z["in_resize"])
    for name, size, aa in (("down", (20, 31), True), ("up", (80, 97), True), ("down_noaa", (20, 31), False),
                           ("int", 24, True), ("same", (37, 53), True), ("half", (18, 26), True)):
        got = F.resize(x, size, antialias=aa).cpu().numpy()
        assert got.shape == z[f"out_resize_{name}"].shape, name
        assert np.abs(got - z[f"out_resize_{name}"]).max() <= 2e-6, (name, float(np.abs(got - z[f"out_resize_{name}"]).max()))


def test_example_brdf_script_path(golden):
    """examples/example_brdf.py statement by statement, through the `pypbr` alias: load the tiles folder,
    resize((512, 512)).tile(2), point-light render -- against the output of the real reference."""
    import os
    import sys
    import warnings
    from pypbr_amd import compat
    saved = {k: v for k, v in sys.modules.items() if k == "pypbr" or k.startswith("pypbr.")}
    for k in saved:
        del sys.modules[k]
    compat.install()
    try:
        from pypbr.io import load_material_from_folder
        from pypbr.models import CookTorranceBRDF
        z = golden("example")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            material = load_material_from_folder(os.path.join(os.path.dirname(__file__), "golden", "tiles"),
                                                 preferred_workflow="metallic")
        assert list(material._maps) == list(z["meta_map_order"])
        for k, v in material._maps.items():
            assert np.abs(v[:, 500:532, 700:732].numpy() - z[f"loaded_crop_{k}"]).max() <= 2e-6, k
        H, W = 512, 512
        material.resize((H, W)).tile(2)
        assert material.size == (1024, 1024)
        for k, v in material._maps.items():
            assert np.abs(v[:, 480:544, 480:544].numpy() - z[f"resized_crop_{k}"]).max() <= 1e-5, k
            assert abs(float(v.double().mean()) - float(z[f"resized_mean_{k}"])) <= 1e-6, k
        brdf = CookTorranceBRDF(light_type="point")
        view_dir = torch.tensor([0.0, 0.0, 1.0])
        light_dir = torch.tensor([0.1, 0.1, 1.0])
        light_intensity = torch.tensor([1.0, 1.0, 1.0])
        light_size = 1.0
        reflected_color = brdf(material, view_dir, light_dir, light_intensity, light_size)
        assert reflected_color.shape == (3, 1024, 1024) and reflected_color.device.type == "cpu"
        assert np.abs(reflected_color[:, 448:576, 448:576].numpy() - z["example_crop"]).max() <= TOL
        assert abs(float(reflected_color.double().mean()) - float(z["example_mean"])) <= 1e-6     # 0.493188
        assert np.abs(reflected_color.double().sum(dim=(0, 2)).numpy() - z["example_rowsum"]).max() <= 3e-3   # 3072 values per row
    finally:
        compat.uninstall()
        sys.modules.update(saved)


def test_ragged_rows_take_the_vector_kernels_and_match_the_one_pixel_kernels():
    """Widths that 4 does not divide, and maps that start off a 16-byte boundary, run on the 4-pixel (2-pixel) kernels:
    element-aligned vector accesses, the last two lanes of a row overlapping (ct_kernel.hpp lane_pos).  The one-pixel
    kernels (PBR_TUNE_MAX_VEC = 1) evaluate the same functions per pixel, so everything must agree bit for bit: forward,
    map gradients, light gradients (odd widths fall back to one pixel there), fp16, several lights (batch-inner, two
    pixels per lane), fused blend."""
    from pypbr_amd import _native as N, functional as F
    import pypbr_amd.blending as B
    lib = N.lib()
    g = torch.Generator(device="cuda").manual_seed(21)

    def off16(t, k=1):
        """the same values in a tensor whose storage starts k elements past a 16-byte boundary"""
        flat = torch.empty(t.numel() + 8, dtype=t.dtype, device=t.device)
        v = flat[k:k + t.numel()].view(t.shape)
        v.copy_(t)
        assert v.data_ptr() % 16 != 0 and v.is_contiguous()
        return v

    def maps(b, h, w, dtype=torch.float32, shift=False):
        a = torch.rand(b, 3, h, w, device="cuda", generator=g)
        n = torch.nn.functional.normalize(torch.rand(b, 3, h, w, device="cuda", generator=g) * 2 - 1, dim=1)
        r = torch.rand(b, 1, h, w, device="cuda", generator=g) * 0.9 + 0.1
        m = torch.rand(b, 1, h, w, device="cuda", generator=g)
        out = [t.to(dtype) for t in (a, n, r, m)]
        return [off16(t) for t in out] if shift else out
    point = dict(view_dir=[0.1, -0.2, 1.0], light=[0.1, 0.1, 1.0], light_intensity=[1.0, 0.9, 0.8], light_type="point", light_size=1.0)
    ring = dict(view_dir=[0, 0, 1], light=[[1, 0, 1], [0, 1, 1], [-1, 0, 1], [0, -1, 1]], light_intensity=[[0.25] * 3] * 4, light_type="point")
    cases = [("w % 4 = 1", maps(1, 9, 53), point), ("w % 4 = 2, batch", maps(3, 7, 130), point), ("w % 4 = 3", maps(2, 5, 7), point),
             ("w = 4", maps(1, 3, 4), point), ("w = 5", maps(1, 3, 5), point), ("aligned width, shifted storage", maps(2, 6, 64, shift=True), point),
             ("ragged and shifted", maps(1, 6, 67, shift=True), point), ("fp16 ragged", maps(2, 5, 37, torch.float16), point),
             ("fp16 shifted", maps(1, 4, 64, torch.float16, shift=True), point),
             ("4 lights, batch-inner, odd width", maps(4, 6, 45, torch.float16), ring), ("4 lights, single, ragged", maps(1, 6, 46), ring)]
    try:
        for name, mp, light in cases:
            outs, grads, lgrads = [], [], []
            for max_vec in (8, 1):
                lib.pbr_set_tuning(N.TUNE_MAX_VEC, max_vec)
                outs.append(F.cook_torrance(*mp, **light).clone())
                if mp[0].dtype == torch.float32:
                    leaves = [t.clone().requires_grad_() for t in mp]
                    lt = torch.tensor(light["light"], dtype=torch.float32, device="cuda", requires_grad=True)
                    F.cook_torrance(*leaves, **{**light, "light": lt}).square().sum().backward()
                    grads.append([t.grad.clone() for t in leaves])
                    lgrads.append(lt.grad.clone())
            assert torch.equal(outs[0], outs[1]), name
            for ga, gb in zip(*grads) if grads else ():
                assert torch.equal(ga, gb), name
            if lgrads:                                            # sums over pixels: same values, another order of addition
                assert torch.allclose(lgrads[0], lgrads[1], rtol=2e-5, atol=1e-7), (name, lgrads)
        m1, m2 = maps(1, 9, 53), maps(1, 9, 53)
        mask = torch.rand(1, 1, 9, 53, device="cuda", generator=g)
        fused = []
        for max_vec in (8, 1):
            lib.pbr_set_tuning(N.TUNE_MAX_VEC, max_vec)
            fused.append(F.cook_torrance(*m1, **point, blend=(*m2, None, mask)).clone())
        assert torch.equal(fused[0], fused[1])
    finally:
        lib.pbr_set_tuning(N.TUNE_MAX_VEC, 8)


def test_scalar_plane_addresses_are_bit_identical():
    """KArgs::sbase (ct_kernel.hpp, plane_at): plane addresses formed with scalar instructions when a workgroup stays inside
    one material.  Same loads, same arithmetic, same stores -- so knob 0 (never), 1 (rule: single materials) and 2
    (whenever the launch allows it) must agree bit for bit, forward and backward, for single materials, batches whose
    height the tile rows divide, batches where they do not (the general path whatever the knob), fp16, several lights,
    tiled maps and strided (packed) maps."""
    from pypbr_amd import _native as N, functional as F
    lib = N.lib()
    g = torch.Generator(device="cuda").manual_seed(11)

    def maps(b, h, w, dtype=torch.float32):
        a = torch.rand(b, 3, h, w, device="cuda", generator=g)
        n = torch.nn.functional.normalize(torch.rand(b, 3, h, w, device="cuda", generator=g) * 2 - 1, dim=1)
        r = torch.rand(b, 1, h, w, device="cuda", generator=g) * 0.9 + 0.1
        m = torch.rand(b, 1, h, w, device="cuda", generator=g)
        return [t.to(dtype) for t in (a, n, r, m)]
    point = dict(view_dir=[0.1, -0.2, 1.0], light=[0.1, 0.1, 1.0], light_intensity=[1.0, 0.9, 0.8], light_type="point", light_size=1.0)
    ring = dict(view_dir=[0, 0, 1], light=[[1, 0, 1], [0, 1, 1], [-1, 0, 1], [0, -1, 1]], light_intensity=[[0.25] * 3] * 4, light_type="point")
    cases = [("single", maps(1, 64, 256), point, {}), ("batch, rows divide", maps(4, 32, 64), point, {}),
             ("batch, ragged", maps(3, 37, 52), point, {}), ("fp16 x8", maps(2, 16, 512, torch.float16), point, {}),
             ("fp16 -> fp16", maps(1, 16, 512, torch.float16), point, dict(out_dtype=torch.float16)),
             ("4 lights, batch-inner", maps(4, 16, 128, torch.float16), ring, {}), ("4 lights, single", maps(1, 16, 128), ring, {})]
    try:
        for name, mp, light, extra in cases:
            outs, grads = [], []
            for knob in (0, 1, 2):
                lib.pbr_set_tuning(N.TUNE_SCALAR_BASE, knob)
                outs.append(F.cook_torrance(*mp, **light, **extra).clone())
                if mp[0].dtype == torch.float32:
                    leaves = [t.clone().requires_grad_() for t in mp]
                    F.cook_torrance(*leaves, **light).square().sum().backward()
                    grads.append([t.grad.clone() for t in leaves])
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), name
            for ga, gb, gc in zip(*grads) if grads else ():
                assert torch.equal(ga, gb) and torch.equal(ga, gc), name
        a, n, r, m = (t[0] for t in maps(1, 32, 64))                              # fused tile(2): (32, 64) maps, (64, 128) image
        tiled = []
        for knob in (0, 1, 2):
            lib.pbr_set_tuning(N.TUNE_SCALAR_BASE, knob)
            tiled.append(F.cook_torrance(a, n, r, m, **point, tile=(2, 2)).clone())
        assert torch.equal(tiled[0], tiled[1]) and torch.equal(tiled[0], tiled[2])
    finally:
        lib.pbr_set_tuning(N.TUNE_SCALAR_BASE, 1)


def test_schedules_are_bit_identical_and_autotune_picks_one():
    """The workgroup -> tile order (descriptor field `schedule`) only changes WHERE a tile runs: every order must
    write bit-identical results, on shapes whose tile count is not a multiple of the run length too."""
    from pypbr_amd import _native as N, functional as F
    g = torch.Generator().manual_seed(41)
    for B, H, W in ((1, 96, 1000), (3, 70, 256), (2, 33, 52)):
        a = torch.rand(B, 3, H, W, generator=g).cuda()
        n = torch.cat([torch.rand(B, 2, H, W, generator=g) - 0.5, torch.ones(B, 1, H, W)], 1).cuda()
        r = (torch.rand(B, 1, H, W, generator=g) * 0.8 + 0.2).cuda()
        m = torch.rand(B, 1, H, W, generator=g).cuda()
        kw = dict(view_dir=[0, 0, 1], light=[0.1, -0.2, 1.0], light_intensity=[1, 0.9, 0.8], light_type="point", light_size=1.0)
        ref = F.plan_cook_torrance(a, n, r, m, schedule=N.SCHEDULE_LINEAR, **kw).launch().clone()
        for sched in (N.SCHEDULE_AUTO, N.schedule_xcd(1), N.schedule_xcd(3), N.schedule_xcd(6), N.schedule_xcd(12)):
            out = F.plan_cook_torrance(a, n, r, m, schedule=sched, **kw).launch()
            assert torch.equal(out, ref), (B, H, W, sched)
        plan = F.plan_cook_torrance(a, n, r, m, autotune=True, **kw)
        assert plan.desc.schedule in (N.SCHEDULE_LINEAR, N.schedule_xcd(6))
        assert torch.equal(plan.launch(), ref)
        # gradients go through the same tile order
        leaves = [t.clone().requires_grad_(True) for t in (a, n, r, m)]
        F.cook_torrance(*leaves, schedule=N.schedule_xcd(2), **kw).sum().backward()
        leaves2 = [t.clone().requires_grad_(True) for t in (a, n, r, m)]
        F.cook_torrance(*leaves2, schedule=N.SCHEDULE_LINEAR, **kw).sum().backward()
        for x, y in zip(leaves, leaves2):
            assert torch.equal(x.grad, y.grad)


@pytest.mark.parametrize("h,w,tile,dtype,lights", [(48, 64, 2, torch.float32, 1), (33, 52, 3, torch.float32, 1),
                                                   (20, 30, (2, 3), torch.float32, 1), (24, 64, (3, 2), torch.float16, 1),
                                                   (16, 32, 2, torch.float32, 3)])
def test_fused_tile_equals_materialised_repeat(h, w, tile, dtype, lights):
    """MaterialBase.tile (base.py:524-537) fused as wrap-around addressing.  Same texels, same
    pixel positions, same arithmetic -> bit-identical to evaluating the repeated maps; row bands included."""
    from pypbr_amd import functional as F
    ny, nx = (tile, tile) if isinstance(tile, int) else tile
    g = torch.Generator().manual_seed(7)
    B = 2
    a = torch.rand(B, 3, h, w, generator=g).cuda().to(dtype)
    n = torch.cat([torch.rand(B, 2, h, w, generator=g) - 0.5, torch.ones(B, 1, h, w)], 1).cuda().to(dtype)
    r = (torch.rand(1, 1, h, w, generator=g) * 0.8 + 0.2).cuda().to(dtype)         # shared by the batch
    m = torch.rand(B, 1, h, w, generator=g).cuda().to(dtype)
    light = [[0.2, -0.1, 0.9], [-0.3, 0.3, 0.7], [0.0, 0.4, 1.1]][:lights]
    kw = dict(view_dir=[0.1, 0, 1], light=light, light_intensity=[[1, 0.9, 0.8]] * lights, light_type="point", light_size=2.0)
    rep = lambda t: t.repeat(1, 1, ny, nx)
    ref = F.cook_torrance(rep(a), rep(n), rep(r), rep(m), **kw)
    out = F.cook_torrance(a, n, r, m, tile=tile, **kw)
    assert out.shape == (B, 3, ny * h, nx * w) and torch.equal(out, ref)
    y0, rows = h // 2 + 1, h                                                           # a band that crosses a seam
    band = F.cook_torrance(a, n, r, m, tile=tile, y_offset=y0, rows=rows, **kw)
    assert torch.equal(band, ref[:, :, y0:y0 + rows])
    plan = F.plan_cook_torrance(a, n, r, m, tile=tile, **kw)
    full = F.plan_cook_torrance(rep(a), rep(n), rep(r), rep(m), **kw)
    assert plan.bytes_per_pixel < full.bytes_per_pixel
    with pytest.raises(ValueError):
        F.cook_torrance(a, n, r, m, rows=4, **kw)


@pytest.mark.parametrize("dtype,h,w,tile", [(torch.float16, 256, 512, (3, 2)), (torch.float32, 64, 256, (2, 2)), (torch.float16, 48, 1024, (4, 1))])
def test_fused_tile_fold_order_is_bit_identical(dtype, h, w, tile):
    """The fold order of a tiled launch (ct_kernel.hpp tile_of_workgroup: all vertical repeats of a band of source rows
    back to back) is a permutation of the tiles: whatever the band and the XCD order under it, the image equals the
    one evaluated on the materialised repeat (MaterialBase.tile, base.py:524-537), forward and backward."""
    from pypbr_amd import functional as F, _native as N
    ny, nx = tile
    g = torch.Generator().manual_seed(11)
    a = torch.rand(3, h, w, generator=g).cuda().to(dtype)
    n = torch.cat([torch.rand(2, h, w, generator=g) - 0.5, torch.ones(1, h, w)], 0).cuda().to(dtype)
    r = (torch.rand(1, h, w, generator=g) * 0.8 + 0.2).cuda().to(dtype)
    m = torch.rand(1, h, w, generator=g).cuda().to(dtype)
    kw = dict(view_dir=[0.1, 0, 1], light=[0.2, -0.1, 0.9], light_intensity=[1, 0.9, 0.8], light_type="point", light_size=2.0)
    rep = lambda t: t.repeat(1, ny, nx)
    ref = F.cook_torrance(rep(a), rep(n), rep(r), rep(m), **kw)
    gout = torch.rand(ref.shape, generator=g).cuda()
    leaves0 = [t.clone().requires_grad_(True) for t in (a, n, r, m)]
    try:
        # the wrap-around form (PBR_TUNE_TILE_REPEAT = 0: fp16 maps take its fold order by rule) under every workgroup order
        N.lib().pbr_set_tuning(N.TUNE_TILE_REPEAT, 0)
        (F.cook_torrance(*leaves0, tile=tile, **kw) * gout).sum().backward()
        for sched in (N.SCHEDULE_AUTO, N.SCHEDULE_LINEAR, N.schedule_xcd(1), N.schedule_xcd(3)):
            out = F.cook_torrance(a, n, r, m, tile=tile, schedule=sched, **kw)
            assert torch.equal(out, ref), sched
        leaves = [t.clone().requires_grad_(True) for t in (a, n, r, m)]
        (F.cook_torrance(*leaves, tile=tile, schedule=N.SCHEDULE_LINEAR, **kw) * gout).sum().backward()
        for x, y in zip(leaves, leaves0):
            assert torch.equal(x.grad, y.grad)
    finally:
        N.lib().pbr_set_tuning(N.TUNE_TILE_REPEAT, -1)


def test_fused_tile_gradients_and_material_api():
    from pypbr_amd import functional as F
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    g = torch.Generator().manual_seed(8)
    h, w = 24, 40
    a = torch.rand(3, h, w, generator=g)
    n = torch.cat([torch.rand(2, h, w, generator=g) - 0.5, torch.ones(1, h, w)], 0)
    r = torch.rand(1, h, w, generator=g) * 0.7 + 0.3
    m = torch.rand(1, h, w, generator=g)
    wt = (torch.rand(3, 2 * h, 2 * w, generator=g) - 0.4).cuda()
    kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
    lazy = [t.clone().cuda().requires_grad_(True) for t in (a, n, r, m)]
    (F.cook_torrance(*lazy, tile=2, **kw) * wt).sum().backward()
    eager = [t.clone().cuda().requires_grad_(True) for t in (a, n, r, m)]
    (F.cook_torrance(*[t.repeat(1, 2, 2) for t in eager], **kw) * wt).sum().backward()     # torch sums the repeats
    for x, y in zip(lazy, eager):
        assert x.grad.shape == y.grad.shape
        assert (x.grad - y.grad).abs().max().item() <= 1e-6 * (1 + y.grad.abs().max().item())
    # material API: tile(n, lazy=True) renders like tile(n) and reports the tiled size
    dev = torch.device("cuda")
    mk = lambda: BasecolorMetallicMaterial(albedo=a.clone(), normal=None, roughness=r.clone(), metallic=m.clone(), device=dev)
    m1, m2 = mk(), mk()
    m1._maps["normal"], m2._maps["normal"] = n.cuda(), n.cuda()
    brdf = CookTorranceBRDF("point")
    args = (torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0]), 1.0)
    assert m1.tile(2, lazy=True) is m1 and m1.size == (2 * h, 2 * w) and m1.__dict__["_store"]["albedo"].shape == (3, h, w)
    assert torch.equal(brdf(m1, *args), brdf(m2.tile(2), *args))
    assert m1.lazy_tile == (2, 2) and m1.__dict__["_store"]["albedo"].shape == (3, h, w)      # the BRDF materialised nothing
    assert m1.albedo.shape == (3, 2 * h, 2 * w) and m1.lazy_tile == (1, 1)                    # reading a map does
    assert torch.equal(m1.albedo, m2.albedo)


@pytest.mark.parametrize("shape,size", [((3, 300, 500), (37, 41)), ((2, 257, 130), (100, 64)), ((1, 40, 60), (90, 100)),
                                        ((2, 3, 96, 200), (48, 100)), ((1, 2000, 70), (9, 70)), ((1, 33, 1000), (33, 130)),
                                        ((1, 1024, 1024), (64, 64)), ((2, 700, 1100), (64, 100)), ((1, 640, 1024), (37, 64)), ((3, 512, 2048), (40, 128))])
@pytest.mark.parametrize("antialias", [True, False])
def test_resize_fused_and_two_pass_forms_against_aten(shape, size, antialias):
    """The schedules of pbr_resize_bilinear (LDS-fused tile kernel, its instantiation for 17 ... 36 taps -- down-scales of 7x ... 17x, round 5 --, two
    passes beyond) against the op the reference ends up in: torch.nn.functional.interpolate(bilinear, align_corners=False, antialias) on CPU."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(sum(shape) + size[0])
    x = torch.rand(*shape, generator=g)
    ref = torch.nn.functional.interpolate(x if x.dim() == 4 else x[None], size=size, mode="bilinear", align_corners=False,
                                          antialias=antialias)
    ref = ref if x.dim() == 4 else ref[0]
    got = F.resize(x.cuda(), size, antialias=antialias).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 2e-6


@pytest.mark.parametrize("light_type,size", [("directional", None), ("point", 1.5)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("workflow", ["metallic", "specular"])
def test_several_lights_every_light_type_and_storage(light_type, size, dtype, workflow):
    """The packed-math light loop (two pixels per instruction, clamp riding on v_pk_mul/fma) for both light types,
    both workflows and both map storages against the oracle's composition of single-light reference evaluations
    (H12: per-light clamp, sum, clamp, encode once).  Includes lights behind the surface and a grazing view."""
    import torch_oracle as O
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(31)
    B, H, W = 2, 18, 44
    a = torch.rand(B, 3, H, W, generator=g)
    n = torch.cat([(torch.rand(B, 2, H, W, generator=g) - 0.5) * 1.6, torch.ones(B, 1, H, W)], 1)
    r = torch.rand(B, 1, H, W, generator=g) * 0.8 + 0.2
    m = torch.rand(B, 1, H, W, generator=g)
    s = torch.rand(B, 3, H, W, generator=g)
    q = lambda t: t.to(dtype).float()                                  # the oracle sees exactly the stored values
    lights = torch.tensor([[0.3, 0.2, 0.8], [-0.4, 0.1, 0.6], [0.0, -0.5, -0.3], [0.9, 0.9, 0.05]])
    inten = torch.tensor([[0.6, 0.5, 0.4], [0.3, 0.3, 0.5], [0.4, 0.4, 0.4], [0.2, 0.1, 0.3]])
    view = torch.tensor([0.3, -0.1, 0.6])
    second = dict(metallic=q(m)) if workflow == "metallic" else dict(specular=q(s))
    ref = O.cook_torrance_batched(q(a), q(n), q(r), lights=lights, intensities=inten, view=view, light_type=light_type,
                                  light_size=size, **second)
    dev = lambda t: t.to(dtype).cuda()
    out = F.cook_torrance(dev(a), dev(n), dev(r), dev(m) if workflow == "metallic" else None, dev(s) if workflow == "specular" else None,
                          view_dir=view, light=lights, light_intensity=inten, light_type=light_type, light_size=size)
    assert out.dtype == torch.float32 and (out.cpu() - ref).abs().max().item() <= TOL


def test_material_to_device_packs_maps_into_one_allocation(golden):
    from pypbr_amd import functional as F
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    z = golden("rand64")
    a, n, r, m = (torch.from_numpy(z[k]) for k in ("in_albedo", "in_normal", "in_roughness", "in_metallic"))
    mat = BasecolorMetallicMaterial(albedo=a, roughness=r, metallic=m)
    mat._maps["normal"] = n
    mat.to("cuda")
    maps = [mat._maps[k] for k in ("albedo", "roughness", "metallic", "normal")]
    assert all(t.is_cuda for t in maps) and len({t.untyped_storage().data_ptr() for t in maps}) == 1
    assert torch.equal(mat.albedo.cpu(), a) and torch.equal(mat.normal.cpu(), n)
    out = CookTorranceBRDF("point")(mat, torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0]), 1.0)
    assert np.abs(out.cpu().numpy() - z["out_metallic_pt1_srgb"]).max() <= TOL
    # the result can live in the same allocation
    pa, pn, pr, pm, res = F.pack_maps(a, n, r, m, device="cuda", reserve_output=True)
    got = F.cook_torrance(pa, pn, pr, pm, view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_size=1.0,
                          out=res.unsqueeze(0))
    assert got.data_ptr() == res.data_ptr() and torch.equal(got.cpu(), out.cpu())


def test_strided_result_and_material_major_batches():
    """The result may be strided (descriptor out_batch_stride / out_channel_stride): pack_maps(material_major=True) puts
    material b's maps and its result next to each other.  Same values as free-standing contiguous tensors; gradients too."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(51)
    B, H, W = 3, 20, 48
    a = torch.rand(B, 3, H, W, generator=g).cuda()
    n = torch.cat([torch.rand(B, 2, H, W, generator=g) - 0.5, torch.ones(B, 1, H, W)], 1).cuda()
    r = (torch.rand(1, 1, H, W, generator=g) * 0.8 + 0.2).cuda()            # shared by the batch
    m = torch.rand(B, 1, H, W, generator=g).cuda()
    kw = dict(view_dir=[0, 0.1, 1], light=[0.2, -0.1, 0.9], light_intensity=[1, 0.9, 0.8], light_type="point", light_size=1.2)
    ref = F.cook_torrance(a, n, r, m, **kw)
    pa, pn, pr, pm, res = F.pack_maps(a, n, r, m, reserve_output=True, material_major=True)
    assert not pa.is_contiguous() and not res.is_contiguous() and pa.untyped_storage().data_ptr() == res.untyped_storage().data_ptr()
    got = F.cook_torrance(pa, pn, pr, pm, out=res, **kw)
    assert got.data_ptr() == res.data_ptr() and torch.equal(got, ref)
    # a channel-strided result of a single material
    big = torch.empty(3, H + 5, W, device="cuda")
    view = big[:, :H]
    one = F.cook_torrance(a[0], n[0], r[0], m[0], out=view, **kw)
    assert torch.equal(one, ref[0]) and one.data_ptr() == big.data_ptr()
    with pytest.raises(ValueError):
        F.cook_torrance(a[0], n[0], r[0], m[0], out=torch.empty(3, H, W + 4, device="cuda")[:, :, :W], **kw)   # rows not contiguous
    leaves = [t.clone().requires_grad_(True) for t in (pa, pn, pr, pm)]
    F.cook_torrance(*leaves, **kw).sum().backward()
    plain = [t.clone().requires_grad_(True) for t in (a, n, r, m)]
    F.cook_torrance(*plain, **kw).sum().backward()
    for x, y in zip(leaves, plain):
        assert torch.equal(x.grad, y.grad)


def test_guard_bands_no_out_of_bounds_reads_or_writes():
    """Every map and the result sit inside larger buffers whose margins are NaN (inputs) / a sentinel (output).  A read
    outside a map would poison the result, a write outside the result would damage the sentinel.  Random shapes that
    hit the 4-pixel, 8-pixel (fp16) and 1-pixel instantiations, fused tiles, row bands, strided results, several lights."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(2024)
    def guarded(t, fill):
        flat = torch.full((t.numel() + 2 * G,), fill, dtype=t.dtype, device="cuda")
        flat[G:G + t.numel()] = t.reshape(-1).cuda()
        return flat, flat[G:G + t.numel()].view(t.shape)

    for trial in range(36):
        # guard elements on each side: 256 keeps the maps 16-byte aligned (vector instantiations), 257 moves them off
        # (1-pixel-per-lane instantiation)
        G = 256 if trial % 3 else 257
        B = int(torch.randint(1, 4, (1,), generator=g))
        h = int(torch.randint(1, 40, (1,), generator=g))
        w = [4, 8, 12, 16, 24, 40, 64, 7, 30, 1][int(torch.randint(0, 10, (1,), generator=g))]
        dtype = torch.float16 if trial % 3 == 2 else torch.float32
        ny, nx = (1, 1) if trial % 2 else (int(torch.randint(1, 4, (1,), generator=g)), int(torch.randint(1, 4, (1,), generator=g)))
        lights = 1 if trial % 4 else 3
        a = torch.rand(B, 3, h, w, generator=g).to(dtype)
        n = torch.cat([torch.rand(B, 2, h, w, generator=g) - 0.5, torch.ones(B, 1, h, w)], 1).to(dtype)
        r = (torch.rand(B, 1, h, w, generator=g) * 0.8 + 0.2).to(dtype)
        m = torch.rand(B, 1, h, w, generator=g).to(dtype)
        H, W = ny * h, nx * w
        y0 = int(torch.randint(0, H, (1,), generator=g)) if (ny, nx) != (1, 1) else 0
        rows = int(torch.randint(1, H - y0 + 1, (1,), generator=g)) if (ny, nx) != (1, 1) else H
        kw = dict(view_dir=[0.1, 0, 1], light=[[0.2, -0.1, 0.9], [-0.3, 0.3, 0.7], [0.0, 0.4, 1.1]][:lights],
                  light_intensity=[[1, 0.9, 0.8]] * lights, light_type="point" if trial % 5 else "directional", light_size=2.0)
        if (ny, nx) != (1, 1):
            kw.update(tile=(ny, nx), y_offset=y0, rows=rows)
        ref = F.cook_torrance(a.cuda(), n.cuda(), r.cuda(), m.cuda(), **kw)
        bufs, views = zip(*[guarded(t, float("nan")) for t in (a, n, r, m)])
        obuf = torch.full((B * 3 * rows * W + 2 * G,), -7.0, device="cuda")
        out = obuf[G:G + B * 3 * rows * W].view(B, 3, rows, W)
        got = F.cook_torrance(*views, out=out, **kw)
        torch.cuda.synchronize()
        tag = (trial, B, h, w, str(dtype), ny, nx, y0, rows, lights)
        assert bool(torch.isfinite(got).all()), tag
        # the odd guard width moves the maps off 16-byte alignment, i.e. onto the 1-pixel-per-lane instantiation: same
        # arithmetic, different fma contraction -> equal to the aligned run to a few ulp, not bit for bit
        assert (got - ref).abs().max().item() <= 2e-6, tag
        assert bool((obuf[:G] == -7.0).all()) and bool((obuf[-G:] == -7.0).all()), tag
        for bf in bufs:
            assert bool(torch.isnan(bf[:G]).all()) and bool(torch.isnan(bf[-G:]).all()), tag
        if dtype == torch.float32 and lights == 1 and (ny, nx) == (1, 1):       # the backward kernel under the same guards
            gout, gv = guarded(torch.rand(B, 3, h, w, generator=g), float("nan"))
            leaves = [v.detach().clone().requires_grad_(True) for v in views]
            F.cook_torrance(*leaves, **kw).backward(gv)
            assert all(bool(torch.isfinite(t.grad).all()) for t in leaves), tag


def test_resize_guard_bands_and_random_shapes():
    """Random in/out sizes (up- and down-scales up to ~40x, both schedules of pbr_resize_bilinear) inside NaN margins, against
    ATen's interpolate on the CPU: a read outside the source would poison the result."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(404)
    for trial in range(30):
        c = int(torch.randint(1, 4, (1,), generator=g))
        hi, wi = int(torch.randint(1, 200, (1,), generator=g)), int(torch.randint(1, 300, (1,), generator=g))
        ho, wo = int(torch.randint(1, 150, (1,), generator=g)), int(torch.randint(1, 260, (1,), generator=g))
        aa = bool(trial % 2)
        x = torch.rand(c, hi, wi, generator=g)
        G = 300
        flat = torch.full((x.numel() + 2 * G,), float("nan"), device="cuda")
        flat[G:G + x.numel()] = x.reshape(-1).cuda()
        got = F.resize(flat[G:G + x.numel()].view(c, hi, wi), (ho, wo), antialias=aa).cpu()
        ref = torch.nn.functional.interpolate(x[None], size=(ho, wo), mode="bilinear", align_corners=False, antialias=aa)[0]
        assert got.shape == ref.shape and bool(torch.isfinite(got).all()), (trial, c, hi, wi, ho, wo, aa)
        # the source coordinate scale * (i + 0.5) of a large map carries ~1e-5 of fp32 rounding (ulp of 100 is 7.6e-6), and
        # ATen and this kernel round it at different points: weights, hence values in [0,1], agree to ~1e-5, not to the ulp
        assert (got - ref).abs().max().item() <= 1e-5, (trial, c, hi, wi, ho, wo, aa, float((got - ref).abs().max()))


def test_resize_in_xcd_contiguous_chunks_matches_aten_on_ragged_tile_counts():
    """pbr_resize_bilinear walks its tiles in XCD-contiguous chunks of 64 (every XCD takes its own chunk of each block of 8 chunks, so
    the tiles that share halo rows and boundary lines meet in one L2; the other orders of round 3's A/B went with their knob).  Tile
    counts that are not a multiple of a block, down- and up-scaling, against ATen."""
    from pypbr_amd import functional as F
    g = torch.Generator().manual_seed(77)
    for shape, size, aa in (((3, 1000, 1400), (700, 900), True), ((2, 520, 1096), (1210, 1500), False), ((5, 777, 640), (300, 333), True)):
        x = torch.rand(*shape, generator=g)
        ref = torch.nn.functional.interpolate(x[None], size=size, mode="bilinear", align_corners=False, antialias=aa)[0]
        out = F.resize(x.cuda(), size, antialias=aa)
        # source coordinates of a 1 500-pixel axis carry ~1e-5 of fp32 rounding, and ATen rounds them at other points (see the test above)
        assert (out.cpu() - ref).abs().max().item() <= 3e-5, (shape, size)


def test_random_shapes_workflows_flags_and_lights_against_the_oracle():
    """tools/render_fuzz.py: 80 random cases (1-3 materials, extents 1 ... 260, all three workflows, both light types, 1-3 lights, fp32 /
    fp16 maps, optional normal map, every flag, three light sizes; gradients on 40 % of the fp32 cases) against the ATen restatement of
    the reference: every value within 1e-5 (roughness >= 0.2: criterion (i)), every gradient within 5e-5 (1 + |g64|)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("render_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "render_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.run(80, 5, verbose=False)


def test_fused_tile_on_random_shapes_bands_fold_bands_and_orders():
    """tools/tile_fuzz.py: 60 random cases (maps 1 ... 128 rows x 4 ... 1536 columns, repeats 1-4 x 1-3, fp32 / fp16, 1-2 lights, 1-2 materials,
    every fold band and workgroup order, a random row band each): bit-identical to the evaluation of the materialised repeat."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("tile_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "tile_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.run(60, 17, verbose=False)
