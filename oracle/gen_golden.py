#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (/root/reference).

Development-container tool (the reference tree does not travel to the GPU box;
the vectors it produces do).  Re-run with:  python oracle/gen_golden.py
Every expected output in the fixtures is produced by the reference's own classes
(pypbr.models.CookTorranceBRDF, pypbr.materials.*, pypbr.utils.*); inputs are
seeded draws (recipe of SURVEY.md section 8c) or crops of the PNG fixtures the
reference's own tests hold (tests/data/{tiles,rocks}).

Output keys:  in_*  = inputs,  out_* = fp32 reference outputs,  f64_* = the same
reference code evaluated in float64 (maps poked into material._maps)
used only for the conditioning-aware criterion.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_import import import_reference, REFERENCE_ROOT  # noqa: E402

pypbr = import_reference()
from pypbr.materials import BasecolorMetallicMaterial, DiffuseSpecularMaterial, MaterialBase  # noqa: E402
from pypbr.models import CookTorranceBRDF  # noqa: E402
from pypbr.utils import linear_to_srgb, srgb_to_linear  # noqa: E402
from pypbr.io import load_material_from_folder  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")

LIGHTS = {
    # name: (light_type, light vector, light_size)
    "pt1": ("point", [0.1, 0.1, 1.0], 1.0),
    "pt5": ("point", [0.0, 10.0, 10.0], 5.0),
    "dir": ("directional", [0.3, -0.2, 1.0], None),
}
VIEW0 = [0.0, 0.0, 1.0]
INT0 = [1.0, 1.0, 1.0]
VIEW1 = [0.2, -0.1, 1.0]
INT1 = [2.0, 1.6, 1.2]


def draw(seed, H, W):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(3, H, W, generator=g)
    n = torch.rand(3, H, W, generator=g) * 2 - 1
    r = torch.rand(1, H, W, generator=g)
    m = torch.rand(1, H, W, generator=g)
    s = torch.rand(3, H, W, generator=g)
    return a, n, r, m, s


def draw_realistic(seed, H, W):
    """Criterion (i) inputs: roughness in [0.05,1], normals with z>0."""
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(3, H, W, generator=g)
    nxy = torch.rand(2, H, W, generator=g) - 0.5
    n = torch.cat([nxy, torch.ones(1, H, W)], dim=0)
    n = n / n.norm(dim=0, keepdim=True)
    r = torch.rand(1, H, W, generator=g) * 0.95 + 0.05
    m = torch.rand(1, H, W, generator=g)
    s = torch.rand(3, H, W, generator=g)
    return a, n, r, m, s


def make_material(kind, a, n, r, m, s, dtype=torch.float32, albedo_is_srgb=True, specular_is_srgb=True):
    """Builds the reference material.  fp64: poke maps past the FloatTensor gate."""
    if dtype == torch.float32:
        if kind == "metallic":
            mat = BasecolorMetallicMaterial(albedo=a, normal=n, roughness=r, metallic=m,
                                            albedo_is_srgb=albedo_is_srgb)
        else:
            mat = DiffuseSpecularMaterial(albedo=a, normal=n, roughness=r, specular=s,
                                          albedo_is_srgb=albedo_is_srgb, specular_is_srgb=specular_is_srgb)
        if n is None:
            mat.normal = None
        return mat
    if kind == "metallic":
        mat = BasecolorMetallicMaterial(albedo_is_srgb=albedo_is_srgb)
        mat._maps["metallic"] = m.double()
    else:
        mat = DiffuseSpecularMaterial(albedo_is_srgb=albedo_is_srgb, specular_is_srgb=specular_is_srgb)
        mat._maps["specular"] = s.double()
    mat._maps["albedo"] = a.double()
    mat._maps["normal"] = None if n is None else n.double()
    mat._maps["roughness"] = r.double()
    return mat


def render(mat, light_key, view=VIEW0, inten=INT0, return_srgb=True, dtype=torch.float32):
    ltype, lvec, lsize = LIGHTS[light_key]
    brdf = CookTorranceBRDF(light_type=ltype)
    out = brdf(mat, torch.tensor(view, dtype=dtype), torch.tensor(lvec, dtype=dtype),
               torch.tensor(inten, dtype=dtype), lsize, return_srgb=return_srgb)
    assert out.dtype == dtype
    return out


def sha12(t):
    return hashlib.sha1(np.ascontiguousarray(t).tobytes()).hexdigest()[:12]


def random_set(name, a, n, r, m, s, manifest, with_f64=False, full=True):
    d = {"in_albedo": a, "in_normal": n, "in_roughness": r, "in_metallic": m, "in_specular": s}
    outs = {}
    for kind in ("metallic", "specular"):
        mat = make_material(kind, a, n, r, m, s)
        for lk in LIGHTS:
            for srgb in (True, False):
                if not full and not srgb and lk == "pt5":
                    continue
                key = f"{kind}_{lk}_{'srgb' if srgb else 'lin'}"
                outs[key] = render(mat, lk, return_srgb=srgb)
        # non-trivial view direction + coloured, >1 intensity
        outs[f"{kind}_pt1_srgb_view1"] = render(mat, "pt1", VIEW1, INT1)
        outs[f"{kind}_dir_srgb_view1"] = render(mat, "dir", VIEW1, INT1)
        # +Z default normal branch (cooktorrance.py:147-152; reachable after mat.normal = None)
        mat0 = make_material(kind, a, None, r, m, s)
        outs[f"{kind}_pt1_srgb_nonormal"] = render(mat0, "pt1")
        outs[f"{kind}_dir_srgb_nonormal"] = render(mat0, "dir")
        # stored maps already linear
        matl = make_material(kind, a, n, r, m, s, albedo_is_srgb=False, specular_is_srgb=False)
        outs[f"{kind}_pt1_srgb_linmaps"] = render(matl, "pt1")
        if with_f64:
            mat64 = make_material(kind, a, n, r, m, s, dtype=torch.float64)
            for lk in ("pt1", "dir"):
                d[f"f64_{kind}_{lk}_srgb"] = render(mat64, lk, dtype=torch.float64).numpy()
                d[f"f64_{kind}_{lk}_lin"] = render(mat64, lk, return_srgb=False, dtype=torch.float64).numpy()
    # metallic -> diffuse/specular conversion, then render (H13, quirk F6)
    matm = make_material("metallic", a, n, r, m, s)
    conv = matm.to_diffuse_specular_material()
    assert conv.albedo_is_srgb is False and conv.specular_is_srgb is True
    d["out_conv_diffuse"] = conv.albedo.numpy()
    d["out_conv_specular"] = conv.specular.numpy()
    outs["converted_dir_srgb_quirk"] = render(conv, "dir")
    outs["converted_pt1_srgb_quirk"] = render(conv, "pt1")
    conv.specular_is_srgb = False
    outs["converted_dir_srgb_fixed"] = render(conv, "dir")
    outs["converted_pt1_srgb_fixed"] = render(conv, "pt1")
    # diffuse/specular -> basecolor/metallic (H14): raw specular, 3-channel metallic
    mats = make_material("specular", a, n, r, m, s)
    back = mats.to_basecolor_metallic_material()
    d["out_back_basecolor"] = back.albedo.numpy()
    d["out_back_metallic"] = back.metallic.numpy()
    assert back.metallic.shape[0] == 3

    for k, v in outs.items():
        d["out_" + k] = v.numpy()
    d = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}
    np.savez(os.path.join(GOLDEN, name + ".npz"), **d)
    manifest["sets"][name] = {
        k: {"sha1_12": sha12(v), "mean": float(np.asarray(v, dtype=np.float64).mean()),
            "shape": list(v.shape), "dtype": str(v.dtype)}
        for k, v in d.items()}


def fixture_set(name, folder, crop, manifest):
    """96x96 crops of the maps the reference loads from its own PNG fixtures."""
    y0, x0, hh, ww = crop
    d = {}
    for kind in ("metallic", "specular"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mat = load_material_from_folder(os.path.join(REFERENCE_ROOT, "tests", "data", folder),
                                            preferred_workflow=kind)
        maps = {k: v[:, y0:y0 + hh, x0:x0 + ww].contiguous() for k, v in mat._maps.items()
                if v is not None and k in ("albedo", "normal", "roughness", "metallic", "specular")}
        if kind == "metallic":
            m2 = BasecolorMetallicMaterial(albedo=maps["albedo"], normal=None, roughness=maps["roughness"],
                                           metallic=maps["metallic"])
        else:
            m2 = DiffuseSpecularMaterial(albedo=maps["albedo"], normal=None, roughness=maps["roughness"],
                                         specular=maps["specular"])
        m2._maps["normal"] = maps["normal"]  # keep the decoded map exactly as loaded
        for k, v in maps.items():
            d[f"in_{kind}_{k}"] = v.numpy()
        for lk in ("pt1", "dir"):
            d[f"out_{kind}_{lk}_srgb"] = render(m2, lk).numpy()
            d[f"out_{kind}_{lk}_lin"] = render(m2, lk, return_srgb=False).numpy()
        m64 = make_material(kind, maps["albedo"], maps["normal"], maps["roughness"],
                            maps.get("metallic"), maps.get("specular"), dtype=torch.float64)
        d[f"f64_{kind}_pt1_srgb"] = render(m64, "pt1", dtype=torch.float64).numpy()
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **d)
    manifest["sets"][name] = {
        k: {"sha1_12": sha12(v), "mean": float(np.asarray(v, dtype=np.float64).mean()),
            "shape": list(v.shape), "dtype": str(v.dtype)} for k, v in d.items()}


def misc_set(manifest):
    d = {}
    x = torch.linspace(-0.25, 1.25, 1537)
    g = torch.Generator().manual_seed(5)
    xr = torch.rand(3, 33, 31, generator=g)
    knees = torch.tensor([0.0, 0.0031307, 0.0031308, 0.0031309, 0.04044, 0.04045, 0.04046, 1.0])
    for nm, t in (("ramp", x), ("rand", xr), ("knees", knees)):
        d[f"in_colour_{nm}"] = t.numpy()
        d[f"out_s2l_{nm}"] = srgb_to_linear(t).numpy()
        d[f"out_l2s_{nm}"] = linear_to_srgb(t).numpy()
    # normal decode (base.py:191-242)
    n01 = torch.rand(3, 19, 23, generator=g)
    nneg = torch.rand(3, 19, 23, generator=g) * 2 - 1
    n2 = torch.rand(2, 19, 23, generator=g)
    for nm, t in (("rgb01", n01), ("signed", nneg), ("xy", n2)):
        mat = MaterialBase(normal=t)
        d[f"in_normal_{nm}"] = t.numpy()
        d[f"out_normal_{nm}"] = mat.normal.numpy()
    # multi-light composition (H12) out of single reference calls
    a, n, r, m, s = draw_realistic(21, 24, 40)
    mat = make_material("metallic", a, n, r, m, s)
    pos = torch.tensor([[np.cos(t), np.sin(t), 1.0] for t in np.linspace(0, 2 * np.pi, 4, endpoint=False)],
                       dtype=torch.float32)
    inten = torch.tensor([[0.9, 0.8, 0.7], [0.5, 0.6, 0.7], [0.3, 0.3, 0.3], [0.6, 0.2, 0.4]])
    acc = torch.zeros(3, 24, 40)
    for l in range(4):
        acc = acc + CookTorranceBRDF("point")(mat, torch.tensor(VIEW0), pos[l], inten[l], 1.0, return_srgb=False)
    acc = acc.clamp(0, 1)
    for k, v in (("albedo", a), ("normal", n), ("roughness", r), ("metallic", m)):
        d[f"in_ml_{k}"] = v.numpy()
    d["in_ml_lights"] = pos.numpy()
    d["in_ml_intensities"] = inten.numpy()
    d["out_ml_lin"] = acc.numpy()
    d["out_ml_srgb"] = linear_to_srgb(acc).numpy()
    np.savez(os.path.join(GOLDEN, "misc.npz"), **d)
    manifest["sets"]["misc"] = {
        k: {"sha1_12": sha12(v), "mean": float(np.asarray(v, dtype=np.float64).mean()),
            "shape": list(v.shape), "dtype": str(v.dtype)} for k, v in d.items()}


def _entry(v):
    return {"sha1_12": sha12(v), "mean": float(np.asarray(v, dtype=np.float64).mean()),
            "shape": list(v.shape), "dtype": str(v.dtype)}


def example_set(manifest):
    """examples/example_brdf.py, literally: load the `tiles` folder (the PNGs are committed as data
    fixtures under tests/golden/tiles/), resize((512,512)), tile(2), point-light render.  Also
    resize-only vectors (antialiased down / up-scaling, tuple and int sizes) on small maps."""
    import shutil
    import warnings
    src = os.path.join(REFERENCE_ROOT, "tests", "data", "tiles")
    dst = os.path.join(GOLDEN, "tiles")
    os.makedirs(dst, exist_ok=True)
    for f in sorted(os.listdir(src)):
        if f.endswith(".png"):
            shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
    d = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mat = load_material_from_folder(dst, preferred_workflow="metallic")
    d["meta_map_order"] = np.array(list(mat._maps.keys()))
    for k, v in mat._maps.items():
        d[f"loaded_mean_{k}"] = np.array(float(v.double().mean()))
        d[f"loaded_crop_{k}"] = v[:, 500:532, 700:732].numpy()
    mat.resize((512, 512)).tile(2)
    assert mat.size == (1024, 1024)
    for k, v in mat._maps.items():
        d[f"resized_crop_{k}"] = v[:, 480:544, 480:544].numpy()          # straddles the tile seam at 512
        d[f"resized_mean_{k}"] = np.array(float(v.double().mean()))
    out = CookTorranceBRDF(light_type="point")(mat, torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]),
                                               torch.tensor([1.0, 1.0, 1.0]), 1.0)
    d["example_mean"] = np.array(float(out.double().mean()))
    d["example_crop"] = out[:, 448:576, 448:576].numpy()
    d["example_rowsum"] = out.double().sum(dim=(0, 2)).numpy()           # 1024 row sums: a checksum of the whole image
    # resize-only vectors on a small seeded map
    g = torch.Generator().manual_seed(31)
    x = torch.rand(3, 37, 53, generator=g)
    d["in_resize"] = x.numpy()
    for name, size, aa in (("down", (20, 31), True), ("up", (80, 97), True), ("down_noaa", (20, 31), False),
                           ("int", 24, True), ("same", (37, 53), True), ("half", (18, 26), True)):
        m = MaterialBase(albedo=x.clone(), normal=None)
        m.resize(size, antialias=aa)
        d[f"out_resize_{name}"] = m.albedo.numpy()
    np.savez_compressed(os.path.join(GOLDEN, "example.npz"), **d)
    manifest["sets"]["example"] = {k: _entry(v) for k, v in d.items() if v.dtype.kind == "f"}
    print("example path mean", float(d["example_mean"]))


def grad_set(manifest):
    """Gradients of the reference (torch.autograd through CookTorranceBRDF.forward) of
    loss = sum(out * W) w.r.t. the maps: the rendering-loss use of docs/.../06_advanced.rst:73-107."""
    H, W = 24, 40
    g = torch.Generator().manual_seed(515)
    a = torch.rand(3, H, W, generator=g)
    nxy = (torch.rand(2, H, W, generator=g) - 0.5) * 1.6
    n = torch.cat([nxy, torch.ones(1, H, W)], 0)                      # un-normalised on purpose (F.normalize backward)
    r = torch.rand(1, H, W, generator=g) * 0.8 + 0.2
    m = torch.rand(1, H, W, generator=g)
    s = torch.rand(3, H, W, generator=g) * 0.5
    wt = torch.rand(3, H, W, generator=g) - 0.3
    d = {"in_albedo": a, "in_normal": n, "in_roughness": r, "in_metallic": m, "in_specular": s, "in_weight": wt}

    def run(kind, lk, srgb, dtype):
        leaves = {k: v.clone().to(dtype).requires_grad_(True) for k, v in (("albedo", a), ("normal", n), ("roughness", r),
                                                                             ("metallic", m), ("specular", s))}
        if kind == "metallic":
            mat = BasecolorMetallicMaterial()
            mat._maps["metallic"] = leaves["metallic"]
        else:
            mat = DiffuseSpecularMaterial()
            mat._maps["specular"] = leaves["specular"]
        for k in ("albedo", "normal", "roughness"):
            mat._maps[k] = leaves[k]
        out = render(mat, lk, return_srgb=srgb, dtype=dtype)
        (out * wt.to(dtype)).sum().backward()
        names = ("albedo", "normal", "roughness", "metallic" if kind == "metallic" else "specular")
        return {k: leaves[k].grad.numpy() for k in names}

    for kind in ("metallic", "specular"):
        for lk in ("pt1", "dir"):
            for srgb in (True, False):
                tag = f"{kind}_{lk}_{'srgb' if srgb else 'lin'}"
                for k, v in run(kind, lk, srgb, torch.float32).items():
                    d[f"grad_{tag}_{k}"] = v
                for k, v in run(kind, lk, srgb, torch.float64).items():
                    d[f"g64_{tag}_{k}"] = v
    d = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}
    np.savez(os.path.join(GOLDEN, "grad.npz"), **d)
    manifest["sets"]["grad"] = {k: _entry(v) for k, v in d.items()}


def grad_param_set(manifest):
    """Gradients of the REAL reference w.r.t. view_dir / light_dir_or_position / light_intensity (its forward is plain torch
    ops on them, cooktorrance.py:95-96, :126-140): loss = sum(out * W), fp32 and float64 runs, the maps of grad.npz."""
    z = dict(np.load(os.path.join(GOLDEN, "grad.npz")))
    T = torch.from_numpy
    a, n, r, m, s, wt = (T(z["in_" + k]) for k in ("albedo", "normal", "roughness", "metallic", "specular", "weight"))
    view, inten = [0.05, 0.1, 0.9], [0.9, 0.8, 0.7]               # not unit length, not grey
    d = {"in_view": np.array(view, np.float32), "in_intensity": np.array(inten, np.float32)}
    for kind in ("metallic", "specular"):
        for lk in ("pt1", "dir"):
            ltype, lvec, lsize = LIGHTS[lk]
            for dtype, pre in ((torch.float32, "grad"), (torch.float64, "g64")):
                mat = make_material(kind, a, n, r, m, s, dtype=torch.float64) if dtype == torch.float64 else make_material(kind, a, n, r, m, s)
                V = torch.tensor(view, dtype=dtype, requires_grad=True)
                L = torch.tensor(lvec, dtype=dtype, requires_grad=True)
                I = torch.tensor(inten, dtype=dtype, requires_grad=True)
                out = CookTorranceBRDF(light_type=ltype)(mat, V, L, I, lsize, return_srgb=True)
                assert out.dtype == dtype
                (out * wt.to(dtype)).sum().backward()
                if dtype == torch.float32:
                    d[f"out_{kind}_{lk}"] = out.detach().numpy()
                for name, leaf in (("view", V), ("light", L), ("intensity", I)):
                    d[f"{pre}_{kind}_{lk}_{name}"] = leaf.grad.numpy()
    np.savez(os.path.join(GOLDEN, "grad_params.npz"), **d)
    manifest["sets"]["grad_params"] = {k: _entry(v) for k, v in d.items()}


def example_blend_set(manifest):
    """examples/example_blend.py, literally: load `tiles` and `rocks` (PNG data fixtures under tests/golden/), HeightBlend(0.1,
    -0.5), resize((512,512)), tile(2), point-light render.  Only the maps the metallic workflow loads are copied for `rocks`."""
    import shutil
    import warnings
    import pypbr.blending as B
    src, dst = os.path.join(REFERENCE_ROOT, "tests", "data", "rocks"), os.path.join(GOLDEN, "rocks")
    os.makedirs(dst, exist_ok=True)
    for f in ("basecolor.png", "height.png", "metallic.png", "normal.png", "roughness.png", "opacity.png"):
        shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m1 = load_material_from_folder(os.path.join(GOLDEN, "tiles"), preferred_workflow="metallic")
        m2 = load_material_from_folder(dst, preferred_workflow="metallic")
    d = {"meta_m2_map_order": np.array(list(m2._maps.keys()))}
    material, mask = B.HeightBlend(blend_width=0.1, shift=-0.5)(m1, m2)
    d["mask_mean"] = np.array(float(mask.double().mean()))
    d["mask_crop"] = mask[:, 500:532, 700:732].numpy()
    material.resize((512, 512)).tile(2)
    for k, v in material._maps.items():
        d[f"blended_crop_{k}"] = v[:, 480:544, 480:544].numpy()
    out = CookTorranceBRDF(light_type="point")(material, torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]),
                                               torch.tensor([1.0, 1.0, 1.0]), 1.0)
    d["example_mean"] = np.array(float(out.double().mean()))
    d["example_crop"] = out[:, 448:576, 448:576].numpy()
    d["example_rowsum"] = out.double().sum(dim=(0, 2)).numpy()
    np.savez_compressed(os.path.join(GOLDEN, "example_blend.npz"), **d)
    manifest["sets"]["example_blend"] = {k: _entry(v) for k, v in d.items() if v.dtype.kind == "f"}


def blend_set(manifest):
    """pypbr.blending on 96x96 crops of the reference's two PNG materials (examples/example_blend.py uses
    HeightBlend(blend_width=0.1, shift=-0.5) on the full maps): every blend kind, blended maps + masks."""
    import warnings
    import pypbr.blending as B
    crop = (slice(None), slice(300, 396), slice(420, 516))
    mats = []
    for folder in ("tiles", "rocks"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            full = load_material_from_folder(os.path.join(REFERENCE_ROOT, "tests", "data", folder), preferred_workflow="metallic")
        m = BasecolorMetallicMaterial()
        for k, v in full._maps.items():
            m._maps[k] = v[crop].contiguous()
        mats.append(m)
    d = {}
    for i, m in enumerate(mats, 1):
        for k, v in m._maps.items():
            d[f"in_m{i}_{k}"] = v.numpy()
    g = torch.Generator().manual_seed(77)
    rnd_mask = torch.rand(96, 96, generator=g)
    d["in_mask"] = rnd_mask.numpy()
    blends = {"height": B.HeightBlend(blend_width=0.1, shift=-0.5), "mask": B.MaskBlend(rnd_mask),
              "prop": B.PropertyBlend(property_name="roughness", blend_width=0.1),
              "gradh": B.GradientBlend("horizontal"), "gradv": B.GradientBlend("vertical")}
    for name, blender in blends.items():
        out, mask = blender(mats[0], mats[1])
        d[f"out_{name}_mask"] = mask.numpy()
        for k, v in out._maps.items():
            d[f"out_{name}_{k}"] = v.numpy()
        # example_blend.py:22-32: the blended material goes straight into the BRDF
        for lk, (ltype, lvec, lsize) in {"pt1": ("point", [0.1, 0.1, 1.0], 1.0), "dir": ("directional", [0.3, -0.2, 1.0], None)}.items():
            brdf = CookTorranceBRDF(light_type=ltype)
            d[f"render_{name}_{lk}"] = brdf(out, torch.tensor(VIEW0), torch.tensor(lvec), torch.tensor(INT0), lsize).numpy()
    np.savez_compressed(os.path.join(GOLDEN, "blend.npz"), **d)
    manifest["sets"]["blend"] = {k: _entry(v) for k, v in d.items()}


def edge_set(manifest):
    """Round 3: the edges of the reference's argument handling, from the REAL reference.
    (a) `light_size or 1.0` (cooktorrance.py:130) is Python truthiness: negative sizes are truthy and mirror the grid, 0.0 and
        -0.0 mean 1.0, NaN is truthy (every value of the result is NaN);
    (b) NaN texels: which output values the reference turns into NaN;
    (c) BASELINE.json configs[0] at the size it names: `tiles` / `rocks` loaded from the PNG data fixtures, resize((256, 256)),
        point light (SURVEY.md 8c anchors: means 0.492009 / 0.256256)."""
    import warnings
    a, n, r, m, s = draw_realistic(41, 33, 48)
    d = {"in_albedo": a, "in_normal": n, "in_roughness": r, "in_metallic": m, "in_specular": s}
    sizes = {"neg1": -1.0, "neg2p5": -2.5, "zero": 0.0, "negzero": -0.0, "nan": float("nan"), "none": None}
    d["meta_sizes"] = np.array([[k, repr(v)] for k, v in sizes.items()])
    for kind in ("metallic", "specular"):
        mat = make_material(kind, a, n, r, m, s)
        for tag, size in sizes.items():
            for srgb in (True, False):
                out = CookTorranceBRDF("point")(mat, torch.tensor(VIEW1), torch.tensor([0.1, 0.1, 1.0]), torch.tensor(INT1), size,
                                                return_srgb=srgb)
                d[f"out_{kind}_{tag}_{'srgb' if srgb else 'lin'}"] = out
    assert torch.equal(d["out_metallic_zero_srgb"], d["out_metallic_none_srgb"]) and torch.equal(d["out_metallic_negzero_srgb"], d["out_metallic_none_srgb"])
    assert bool(torch.isnan(d["out_metallic_nan_srgb"]).all()) and bool(torch.isnan(d["out_specular_nan_lin"]).all())
    assert (d["out_metallic_neg1_srgb"] - d["out_metallic_none_srgb"]).abs().max() > 0.05
    # (b) one NaN per map, at distinct pixels
    an, nn, rn, mn = a.clone(), n.clone(), r.clone(), m.clone()
    an[1, 3, 5] = float("nan"); nn[0, 7, 9] = float("nan"); rn[0, 11, 13] = float("nan"); mn[0, 15, 17] = float("nan")
    matn = make_material("metallic", an, None, rn, mn, s)
    matn._maps["normal"] = nn                      # past the decode: the NaN stays where it was put
    for lk in ("pt1", "dir"):
        d[f"out_nantexel_{lk}"] = render(matn, lk)
    d.update({"in_nan_albedo": an, "in_nan_normal": nn, "in_nan_roughness": rn, "in_nan_metallic": mn})
    # (c) configs[0]
    for folder in ("tiles", "rocks"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mat = load_material_from_folder(os.path.join(GOLDEN, folder), preferred_workflow="metallic")
        mat.resize((256, 256))
        out = CookTorranceBRDF(light_type="point")(mat, torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]),
                                                   torch.tensor([1.0, 1.0, 1.0]), 1.0)
        d[f"out_{folder}256"] = out
        d[f"mean_{folder}256"] = np.array(float(out.double().mean()))
        print(folder, "256^2 point: mean %.6f min %.4f max %.4f" % (float(out.double().mean()), float(out.min()), float(out.max())))
    d = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}
    np.savez_compressed(os.path.join(GOLDEN, "edge.npz"), **d)
    manifest["sets"]["edge"] = {k: _entry(v) for k, v in d.items() if v.dtype.kind == "f"}


def only(name, fn):
    """`python oracle/gen_golden.py --only <set>`: adds one fixture file and its MANIFEST entries without touching the others
    (they are byte-for-byte what earlier rounds committed)."""
    path = os.path.join(GOLDEN, "MANIFEST.json")
    with open(path) as f:
        manifest = json.load(f)
    assert manifest["torch"] == torch.__version__, "regenerate everything on a new torch build"
    torch.set_num_threads(manifest["aten_threads"])
    fn(manifest)
    with open(path, "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(name + ".npz written:", len(manifest["sets"][name]), "entries")


def only_grad_params():
    """`python oracle/gen_golden.py --only grad_params`: adds tests/golden/grad_params.npz and its MANIFEST entries without
    touching the other fixtures (they are byte-for-byte what earlier rounds committed)."""
    path = os.path.join(GOLDEN, "MANIFEST.json")
    with open(path) as f:
        manifest = json.load(f)
    assert manifest["torch"] == torch.__version__, "regenerate everything on a new torch build"
    torch.set_num_threads(manifest["aten_threads"])
    grad_param_set(manifest)
    with open(path, "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("grad_params.npz written:", sorted(manifest["sets"]["grad_params"])[:6], "...")


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    torch.set_num_threads(8)
    if sys.argv[1:2] == ["--only"] and sys.argv[2:] in (["blend"], ["example_blend"]):   # refresh one set, keep the rest
        with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
            manifest = json.load(f)
        (blend_set if sys.argv[2] == "blend" else example_blend_set)(manifest)
        with open(os.path.join(GOLDEN, "MANIFEST.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    manifest = {
        "generator": "oracle/gen_golden.py",
        "reference": "giuvecchio/PyPBR at /root/reference (imported, unmodified)",
        "torch": torch.__version__,
        "aten_threads": torch.get_num_threads(),
        "lights": LIGHTS, "view0": VIEW0, "intensity0": INT0, "view1": VIEW1, "intensity1": INT1,
        "sets": {},
    }
    random_set("rand64", *draw(1234, 64, 64), manifest, with_f64=True)
    random_set("rand37x53", *draw(1234, 37, 53), manifest, full=False)
    random_set("rand1x1", *draw(7, 1, 1), manifest, full=False)
    random_set("rand1x17", *draw(7, 1, 17), manifest, full=False)
    random_set("rand5x1", *draw(8, 5, 1), manifest, full=False)
    random_set("real48", *draw_realistic(99, 48, 48), manifest, with_f64=True)
    fixture_set("tiles96", "tiles", (300, 420, 96, 96), manifest)
    fixture_set("rocks96", "rocks", (512, 100, 96, 96), manifest)
    misc_set(manifest)
    example_set(manifest)
    grad_set(manifest)
    blend_set(manifest)
    example_blend_set(manifest)
    grad_param_set(manifest)
    edge_set(manifest)
    with open(os.path.join(GOLDEN, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    tot = sum(os.path.getsize(os.path.join(GOLDEN, f)) for f in os.listdir(GOLDEN))
    print("golden fixtures written: %.1f KiB" % (tot / 1024))
    # SURVEY.md 8c known answers
    ka = manifest["sets"]["rand64"]
    print("rand64 metallic_pt1_srgb", ka["out_metallic_pt1_srgb"]["sha1_12"], ka["out_metallic_pt1_srgb"]["mean"])


if __name__ == "__main__":
    if sys.argv[1:3] == ["--only", "grad_params"]:
        only_grad_params()
    elif sys.argv[1:3] == ["--only", "edge"]:
        only("edge", edge_set)
    else:
        main()
