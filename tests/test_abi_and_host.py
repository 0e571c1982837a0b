"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/pbr_hip.h declares, the ctypes struct layout matches, descriptor validation returns the
documented codes (no launch, no GPU), and the Python mirror of the reference surface behaves like
the reference for everything that does not compute."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from pypbr_amd import _native as N
from pypbr_amd import functional as F
from pypbr_amd.materials import BasecolorMetallicMaterial, DiffuseSpecularMaterial, MaterialBase
from pypbr_amd.models import BRDFModel, CookTorranceBRDF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAS_GPU = torch.cuda.is_available()


def _header_functions():
    text = open(os.path.join(ROOT, "include", "pbr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pbr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    declared = _header_functions()
    assert declared and set(declared) == set(N.EXPORTS)
    lib = N.lib()
    for sym in declared:
        assert getattr(lib, sym) is not None
    dyn = subprocess.run(["nm", "-D", "--defined-only", N.LIB_PATH], capture_output=True, text=True).stdout
    for sym in declared:
        assert re.search(r"\bT %s$" % sym, dyn, flags=re.M), sym
    assert lib.pbr_abi_version() == N.ABI_VERSION
    assert lib.pbr_render_desc_size() == ctypes.sizeof(N.RenderDesc)


def test_no_torch_types_in_the_abi():
    text = open(os.path.join(ROOT, "include", "pbr_hip.h")).read()
    assert "torch" not in re.sub(r"/\*.*?\*/", "", text, flags=re.S).lower() and "at::" not in text


def _desc(**over):
    B, H, W = 2, 8, 16
    a, n, r, m = torch.rand(B, 3, H, W), torch.rand(B, 3, H, W), torch.rand(B, 1, H, W), torch.rand(B, 1, H, W)
    o = torch.empty(B, 3, H, W)
    kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=None,
              albedo_is_srgb=True, specular_is_srgb=True, return_srgb=True, convert_to_diffuse_specular=False,
              y_offset=0, height_total=None)
    o = over.pop("out", o)
    kw.update(over)
    d = F.build_descriptor(a, n, r, m, None, o, **kw)
    d._keep = (a, n, r, m, o)
    return d


def test_descriptor_contents_and_kernel_selection():
    lib = N.lib()
    d = _desc()
    assert (d.batch, d.height, d.width, d.height_total, d.y_offset) == (2, 8, 16, 8, 0)
    assert d.workflow == N.WORKFLOW_METALLIC and d.light_type == N.LIGHT_POINT and d.n_lights == 1
    assert d.albedo.batch_stride == 3 * 8 * 16 and d.albedo.channel_stride == 8 * 16 and d.roughness.batch_stride == 8 * 16
    assert d.light_size == 0.0                                   # falsy -> 1.0 inside the library (cooktorrance.py:130)
    assert lib.pbr_kernel_name(ctypes.byref(d)) == b"ct_point_metallic_f32_f32_v4"
    assert lib.pbr_bytes_per_pixel(ctypes.byref(d)) == 44
    d2 = _desc(light_type="Directional", light=[[0, 0, 1], [1, 0, 1]], light_intensity=[[1, 1, 1], [0.5, 0.5, 0.5]],
               convert_to_diffuse_specular=True)
    assert d2.light_type == N.LIGHT_DIRECTIONAL and d2.n_lights == 2 and d2.workflow == N.WORKFLOW_CONVERTED
    # several lights, an even batch: the batch-inner kernel (2 materials per lane share the light geometry) ...
    assert lib.pbr_kernel_name(ctypes.byref(d2)) == b"ctb_directional_converted_f32_f32_v2_b2"
    lib.pbr_set_tuning(N.TUNE_BATCH_INNER, 0)                     # ... unless switched off: one material per lane
    assert lib.pbr_kernel_name(ctypes.byref(d2)) == b"ct_directional_converted_f32_f32_v4_multi"
    lib.pbr_set_tuning(N.TUNE_BATCH_INNER, -1)
    assert [d2.lights[1][c] for c in range(3)] == [1.0, 0.0, 1.0] and d2.intensities[1][0] == 0.5
    # one intensity for several lights is broadcast; mismatched counts are rejected
    assert _desc(light=[[0, 0, 1], [1, 0, 1]], light_intensity=[1, 1, 1]).n_lights == 2
    with pytest.raises(ValueError):
        _desc(light=[[0, 0, 1]] * 3, light_intensity=[[1, 1, 1]] * 2)
    with pytest.raises(ValueError):
        _desc(light=[[0, 0, 1]] * (N.MAX_LIGHTS + 1), light_intensity=[1, 1, 1])
    with pytest.raises(ValueError, match="Unsupported light_type"):
        _desc(light_type="spot")


def test_specular_workflow_bytes_and_batch_broadcast():
    lib = N.lib()
    B, H, W = 3, 4, 8
    a, s, r = torch.rand(B, 3, H, W), torch.rand(B, 3, H, W), torch.rand(1, 1, H, W)     # roughness shared by the batch
    o = torch.empty(B, 3, H, W)
    d = F.build_descriptor(a, None, r, None, s, o, view_dir=[0, 0, 1], light=[0, 0, 1], light_intensity=[1, 1, 1],
                           light_type="directional", light_size=None, albedo_is_srgb=True, specular_is_srgb=False,
                           return_srgb=False, convert_to_diffuse_specular=False, y_offset=0, height_total=None)
    assert d.workflow == N.WORKFLOW_SPECULAR and d.roughness.batch_stride == 0 and not d.normal.data
    assert lib.pbr_bytes_per_pixel(ctypes.byref(d)) == 40        # 52 B minus the absent normal map
    assert lib.pbr_kernel_name(ctypes.byref(d)) == b"ct_directional_specular_f32_f32_v4"
    with pytest.raises(ValueError, match="either 'metallic' or 'specular'"):
        F.build_descriptor(a, None, r, None, None, o, view_dir=[0, 0, 1], light=[0, 0, 1], light_intensity=[1, 1, 1],
                           light_type="point", light_size=None, albedo_is_srgb=True, specular_is_srgb=True,
                           return_srgb=True, convert_to_diffuse_specular=False, y_offset=0, height_total=None)


def test_validation_codes_without_a_device():
    """pbr_cook_torrance validates before it launches, so bad descriptors are testable on CPU."""
    lib = N.lib()
    d = _desc(); d.workflow = 7
    assert lib.pbr_cook_torrance(ctypes.byref(d), None) == N.ERR_WORKFLOW
    d = _desc(); d.metallic.data = None
    assert lib.pbr_cook_torrance(ctypes.byref(d), None) == N.ERR_WORKFLOW
    d = _desc(); d.light_type = 2
    assert lib.pbr_cook_torrance(ctypes.byref(d), None) == N.ERR_LIGHT_TYPE
    d = _desc(); d.n_lights = 0
    assert lib.pbr_cook_torrance(ctypes.byref(d), None) == N.ERR_SHAPE
    d = _desc(); d.y_offset, d.height_total = 4, 8               # band sticks out of the map
    assert lib.pbr_cook_torrance(ctypes.byref(d), None) == N.ERR_SHAPE
    d = _desc(); d.albedo.data = None
    assert lib.pbr_cook_torrance(ctypes.byref(d), None) == N.ERR_NULL_MAP
    d = _desc(); d.map_dtype = 5
    assert lib.pbr_cook_torrance(ctypes.byref(d), None) == N.ERR_DTYPE
    d = _desc(); d.abi_version = 99
    assert lib.pbr_cook_torrance(ctypes.byref(d), None) == N.ERR_SHAPE
    d = _desc(); d.schedule = N.schedule_xcd(13)                # runs of more than 4096 tiles are not a schedule
    assert lib.pbr_cook_torrance(ctypes.byref(d), None) == N.ERR_SHAPE
    assert _desc().schedule == N.SCHEDULE_AUTO and _desc(schedule=N.schedule_xcd(6)).schedule == 7
    d = _desc(); d.map_height, d.map_width = 3, 16              # tiled maps: whole repeats only (8 rows are not 3k)
    assert lib.pbr_cook_torrance(ctypes.byref(d), None) == N.ERR_SHAPE
    o = torch.empty(2, 3, 16, 48)
    d = _desc(out=o, tile=(2, 3))                                # 8x16 maps over a 16x48 output
    assert (d.height, d.width, d.height_total, d.map_height, d.map_width) == (16, 48, 16, 8, 16)
    assert lib.pbr_bytes_per_pixel(ctypes.byref(d)) == 12 + 6    # 32 B of texels shared by 6 repeats, rounded up
    with pytest.raises(ValueError):
        _desc(out=torch.empty(2, 3, 16, 40), tile=(2, 3))
    strided = torch.empty(2, 3, 8 + 2, 16)[:, :, :8]                # a result whose planes are 10 rows apart
    d = _desc(out=strided)
    assert (d.out_batch_stride, d.out_channel_stride) == (3 * 10 * 16, 10 * 16) and lib.pbr_kernel_name(ctypes.byref(d)) == b"ct_point_metallic_f32_f32_v4"
    d.out_channel_stride = 8 * 16 - 1                            # planes may not overlap
    assert lib.pbr_cook_torrance(ctypes.byref(d), None) == N.ERR_SHAPE
    assert (_desc().out_batch_stride, _desc().out_channel_stride) == (0, 0)
    best = ctypes.c_int32(-1)
    d = _desc(); d.workflow = 7                                  # autotune validates like a launch and needs a result slot
    assert lib.pbr_cook_torrance_autotune(ctypes.byref(d), None, ctypes.byref(best)) == N.ERR_WORKFLOW
    assert lib.pbr_cook_torrance_autotune(ctypes.byref(_desc()), None, None) == N.ERR_NULL_MAP and best.value == -1
    assert lib.pbr_decode_normal(ctypes.c_void_p(16), ctypes.c_void_p(16), 4, 10, N.F32, ctypes.c_void_p(16), None) == N.ERR_CHANNELS
    assert lib.pbr_srgb_to_linear(None, None, 10, N.F32, None) == N.ERR_NULL_MAP
    assert b"metallic" in lib.pbr_error_string(N.ERR_WORKFLOW) and b"2 or 3 channels" in lib.pbr_error_string(N.ERR_CHANNELS)
    with pytest.raises(ValueError, match="either 'metallic' or 'specular'"):
        N.check(N.ERR_WORKFLOW)
    with pytest.raises(TypeError):
        N.check(N.ERR_DTYPE)


def test_kernel_selection_for_ragged_and_narrow_maps():
    """Widths that 4 does not divide still take the 4-pixel kernels (the last two lanes of a row overlap, ct_kernel.hpp
    lane_pos); rows shorter than 4 pixels, and the PBR_TUNE_MAX_VEC = 1 test knob, select the one-pixel kernels."""
    lib = N.lib()
    kw = dict(view_dir=[0, 0, 1], light=[0, 0, 1], light_intensity=[1, 1, 1], light_type="point", light_size=1.0,
              albedo_is_srgb=True, specular_is_srgb=True, return_srgb=True, convert_to_diffuse_specular=False,
              y_offset=0, height_total=None)

    def name(w):
        a, n, r, m = torch.rand(1, 3, 5, w), torch.rand(1, 3, 5, w), torch.rand(1, 1, 5, w), torch.rand(1, 1, 5, w)
        return lib.pbr_kernel_name(ctypes.byref(F.build_descriptor(a, n, r, m, None, torch.empty(1, 3, 5, w), **kw)))
    assert name(13).endswith(b"_v4") and name(16).endswith(b"_v4") and name(4).endswith(b"_v4") and name(3).endswith(b"_v1")
    assert lib.pbr_set_tuning(N.TUNE_MAX_VEC, 1) == 8
    try:
        assert name(13).endswith(b"_v1") and name(16).endswith(b"_v1")
    finally:
        lib.pbr_set_tuning(N.TUNE_MAX_VEC, 8)
    big = torch.rand(1, 3, 5, 20)
    view = big[:, :, :, 2:18]                                     # width 16 but rows start 8 bytes off alignment
    view_c = F._as_batched(view, (3,), "albedo")
    assert view_c.is_contiguous()                                 # non-contiguous rows are repacked host-side


def test_fp16_descriptor():
    lib = N.lib()
    a, n, r, m = (torch.rand(1, c, 4, 8).half() for c in (3, 3, 1, 1))
    kw = dict(view_dir=[0, 0, 1], light=[0, 0, 1], light_intensity=[1, 1, 1], light_type="point", light_size=1.0,
              albedo_is_srgb=True, specular_is_srgb=True, return_srgb=True, convert_to_diffuse_specular=False,
              y_offset=0, height_total=None)
    d = F.build_descriptor(a, n, r, m, None, torch.empty(1, 3, 4, 8), **kw)
    assert d.map_dtype == N.F16 and d.out_dtype == N.F32 and lib.pbr_bytes_per_pixel(ctypes.byref(d)) == 28
    assert lib.pbr_kernel_name(ctypes.byref(d)) == b"ct_point_metallic_f16_f32_v8"        # W % 8 == 0: 16-byte fp16 loads
    a6, n6, r6, m6 = (torch.rand(1, c, 4, 12).half() for c in (3, 3, 1, 1))
    d6 = F.build_descriptor(a6, n6, r6, m6, None, torch.empty(1, 3, 4, 12), **kw)
    assert lib.pbr_kernel_name(ctypes.byref(d6)) == b"ct_point_metallic_f16_f32_v4"       # W % 8 != 0
    with pytest.raises(TypeError):
        F.build_descriptor(a.double(), n.double(), r.double(), m.double(), None, torch.empty(1, 3, 4, 8), **kw)
    with pytest.raises(TypeError):
        F.build_descriptor(a, n.float(), r, m, None, torch.empty(1, 3, 4, 8), **kw)


# ---------------------------------------------------------------- reference-shaped Python surface
def test_brdf_constructor_mirrors_reference():
    assert issubclass(CookTorranceBRDF, BRDFModel) and issubclass(BRDFModel, torch.nn.Module)
    b = CookTorranceBRDF(light_type="POINT")
    assert b.light_type == "point" and b.override_device is None
    assert list(b.parameters()) == [] and list(b.buffers()) == []
    assert CookTorranceBRDF().light_type == "point"
    with pytest.raises(ValueError, match="Unsupported light_type: spot"):
        CookTorranceBRDF(light_type="spot")


def test_material_container_semantics_without_compute():
    a, r, m = torch.rand(3, 6, 9), torch.rand(1, 6, 9), torch.rand(1, 6, 9)
    signed = torch.rand(3, 6, 9) * 2 - 1
    mat = BasecolorMetallicMaterial(albedo=a, roughness=r, metallic=m, height=np.zeros((1, 6, 9), np.float32))
    assert list(mat._maps) == ["albedo", "roughness", "height", "metallic"]        # insertion order (base.py:71-84)
    assert mat.albedo is mat.basecolor and mat.size == (6, 9) and mat.albedo_is_srgb is True
    assert mat.height.dtype == torch.float32 and isinstance(mat.as_dict(), dict)
    with pytest.raises(AttributeError):
        mat.normal
    mat.normal = None
    assert "normal" in mat._maps and mat.normal is None
    mat.tag = "plain attribute"                                                     # non-map values are not filed in _maps
    assert "tag" not in mat._maps and mat.tag == "plain attribute"
    ds = DiffuseSpecularMaterial(albedo=a, roughness=r, specular=a, specular_is_srgb=False, albedo_is_srgb=False)
    assert ds.diffuse is ds.albedo and ds.specular_is_srgb is False
    assert ds.linear_albedo is ds.albedo and ds.linear_specular is ds.specular      # already linear: no compute, same tensor
    assert MaterialBase().size is None and MaterialBase().linear_albedo is None
    t = mat.clone()
    assert t.albedo is not mat.albedo and torch.equal(t.albedo, mat.albedo) and t.albedo_is_srgb == mat.albedo_is_srgb
    mat.tile(2)
    assert mat.size == (12, 18) and torch.equal(mat.albedo[:, 6:, 9:], a)
    with pytest.raises(ValueError, match="2 or 3 channels"):
        MaterialBase(normal=torch.rand(4, 6, 9))
    with pytest.raises(ValueError, match="albedo and metallic"):
        BasecolorMetallicMaterial(albedo=a).to_diffuse_specular_material()
    with pytest.raises(ValueError, match="specular maps are required"):
        DiffuseSpecularMaterial(albedo=a).to_basecolor_metallic_material()
    if not HAS_GPU:                # everything that computes needs the device: loud failure, no CPU fallback
        with pytest.raises(RuntimeError, match="no CPU"):
            MaterialBase(normal=signed)
        with pytest.raises(RuntimeError, match="no CPU"):
            BasecolorMetallicMaterial(albedo=a, roughness=r, metallic=m).linear_albedo
        full = BasecolorMetallicMaterial(albedo=a, roughness=r, metallic=m)
        full.normal = None
        with pytest.raises(RuntimeError, match="no CPU"):
            CookTorranceBRDF()(full, torch.tensor([0, 0, 1.0]), torch.tensor([0, 0, 1.0]), torch.tensor([1, 1, 1.0]))
        with pytest.raises(RuntimeError, match="no CPU"):
            F.cook_torrance(a, None, r, m, view_dir=[0, 0, 1], light=[0, 0, 1], light_intensity=[1, 1, 1])


def test_pil_and_numpy_ingest():
    from PIL import Image
    rgb = (np.random.default_rng(0).random((5, 7, 3)) * 255).astype(np.uint8)
    mat = MaterialBase(albedo=Image.fromarray(rgb, "RGB"), roughness=Image.fromarray(rgb[:, :, 0], "L"))
    assert mat.albedo.shape == (3, 5, 7) and mat.roughness.shape == (1, 5, 7)
    assert torch.equal(mat.albedo, torch.from_numpy(rgb.transpose(2, 0, 1).astype(np.float32) / 255))
    h16 = Image.fromarray((np.arange(35, dtype=np.uint16) * 1000).reshape(5, 7))
    mat.height = h16
    assert mat.height.shape == (1, 5, 7) and abs(float(mat.height.max()) - 34000 / 65535) < 1e-6
    mat.metallic = np.ones((1, 5, 7))
    assert mat.metallic.dtype == torch.float32


def test_product_never_imports_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "pypbr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "torch_oracle" not in text and "c_oracle" not in text and "ct_oracle" not in text, f
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f


# ---------------------------------------------------------------- N1: loader + alias (host side)
def _write_png(path, arr, mode):
    from PIL import Image
    img = Image.fromarray(arr)               # mode inferred from dtype / shape ("mode=" goes away in Pillow 13)
    assert img.mode == mode, (img.mode, mode)
    img.save(path)


def test_loader_workflow_selection_and_alias(tmp_path):
    import warnings
    from pypbr_amd import compat
    from pypbr_amd.io import load_material_from_folder, select_material_class
    rng = np.random.default_rng(3)
    rgb = (rng.random((6, 8, 3)) * 255).astype(np.uint8)
    gray = (rng.random((6, 8)) * 255).astype(np.uint8)
    _write_png(tmp_path / "albedo.png", rgb, "RGB")                  # "albedo" is an accepted stem for basecolor
    _write_png(tmp_path / "diffuse.png", rgb, "RGB")
    _write_png(tmp_path / "roughness.png", gray, "L")
    _write_png(tmp_path / "metalness.png", gray, "L")
    _write_png(tmp_path / "specular.png", rgb, "RGB")
    _write_png(tmp_path / "height.png", (rng.random((6, 8)) * 65535).astype(np.uint16), "I;16")
    with pytest.warns(UserWarning, match="Using metallic workflow as preferred"):
        mat = load_material_from_folder(str(tmp_path), preferred_workflow="metallic")
    assert isinstance(mat, BasecolorMetallicMaterial)
    assert list(mat._maps) == ["albedo", "roughness", "height", "metallic"]      # no normal file -> no entry (F7)
    assert torch.equal(mat.albedo, torch.from_numpy(rgb.transpose(2, 0, 1).astype(np.float32) / 255))
    assert mat.height.shape == (1, 6, 8) and float(mat.height.max()) <= 1.0 and mat.albedo_is_srgb is True
    with pytest.warns(UserWarning, match="Using specular workflow as preferred"):
        ds = load_material_from_folder(str(tmp_path), preferred_workflow="specular", is_srgb=False)
    assert isinstance(ds, DiffuseSpecularMaterial) and "metallic" not in ds._maps and ds.specular_is_srgb is False
    with pytest.warns(UserWarning, match="Defaulting to metallic workflow"):
        assert select_material_class({"metallic": 1, "specular": 2}) is BasecolorMetallicMaterial
    assert select_material_class({"specular": 1}) is DiffuseSpecularMaterial
    assert select_material_class({"diffuse": 1}) is DiffuseSpecularMaterial
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert select_material_class({}) is BasecolorMetallicMaterial
    # alias: the reference's import lines resolve to this package
    import sys
    saved = {k: v for k, v in sys.modules.items() if k == "pypbr" or k.startswith("pypbr.")}
    for k in saved:
        del sys.modules[k]
    try:
        compat.install()
        from pypbr.io import load_material_from_folder as aliased_loader
        from pypbr.models import CookTorranceBRDF as AliasedBRDF
        assert AliasedBRDF is CookTorranceBRDF and aliased_loader is load_material_from_folder
        import pypbr
        assert pypbr.materials.MaterialBase is MaterialBase
    finally:
        compat.uninstall()
        sys.modules.update(saved)


def test_fused_blend_entry_point_validates_without_a_device():
    lib = N.lib()
    d = _desc()
    bd = N.BlendDesc()
    ws = ctypes.c_void_p(64)
    assert lib.pbr_cook_torrance_blend(ctypes.byref(d), None, ws, None) == N.ERR_NULL_MAP          # no second material
    assert lib.pbr_cook_torrance_blend(ctypes.byref(d), ctypes.byref(bd), None, None) == N.ERR_NULL_MAP   # no flag workspace
    assert lib.pbr_cook_torrance_blend(ctypes.byref(d), ctypes.byref(bd), ws, None) == N.ERR_NULL_MAP      # empty second material
    bd.albedo = bd.normal = bd.roughness = bd.mask = N.PbrMap(64, 0, 0)
    assert lib.pbr_cook_torrance_blend(ctypes.byref(d), ctypes.byref(bd), ws, None) == N.ERR_WORKFLOW      # metallic workflow, no metallic2
    d16 = _desc(); d16.map_dtype = N.F16
    assert lib.pbr_cook_torrance_blend(ctypes.byref(d16), ctypes.byref(bd), ws, None) == N.ERR_DTYPE       # fp32 only


def test_lazy_blend_and_lazy_tile_protocol_on_the_host():
    """Lazy blends / tiles only RECORD work: creating and inspecting them needs no device; turning them into maps
    does, and without a device that raises (no CPU arithmetic anywhere in the package)."""
    import pypbr_amd.blending as B
    from pypbr_amd.materials import BasecolorMetallicMaterial
    g = torch.Generator().manual_seed(0)
    H, W = 8, 12

    def mat():
        m = BasecolorMetallicMaterial()
        for name, c in (("albedo", 3), ("normal", 3), ("roughness", 1), ("metallic", 1), ("height", 1)):
            m._maps[name] = torch.rand(c, H, W, generator=g)          # poked in: assignment of a normal map needs the device
        return m
    m1, m2 = mat(), mat()
    mask = torch.rand(H, W, generator=g)
    with B.lazy_blending():
        lazy, mk = B.MaskBlend(mask)(m1, m2)
    assert mk.shape == (1, H, W) and type(lazy) is type(m1) and lazy.albedo_is_srgb == m1.albedo_is_srgb
    pending = lazy.__dict__["_lazy_blend"]
    assert pending is not None and pending[0]["albedo"] is m2._maps["albedo"] and lazy.__dict__["_store"]["albedo"] is m1._maps["albedo"]
    assert lazy.size == (H, W) or lazy.__dict__["_lazy_blend"] is None        # `size` may look at the maps ...
    # ... and looking at the maps is what triggers the real blend, which has no CPU path
    lazy2, _ = B.blend_with_mask(m1, m2, mask, lazy=True)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU"):
            lazy2.albedo
    # materials that cannot be fused fall back to the eager blend (here: different workflows -> needs the device)
    assert not B._fusable({k: v for k, v in m1._maps.items() if k != "normal"}, m2._maps, mk)
    assert not B._fusable(m1._maps, m2._maps, mk.double())
    # lazy tile: only the reported size changes
    t = mat().tile(3, lazy=True)
    assert t.lazy_tile == (3, 3) and t.size == (3 * H, 3 * W) and t._raw["albedo"].shape == (3, H, W)
    assert t.tile(2, lazy=True).lazy_tile == (6, 6)
    assert t.materialize_tile().albedo.shape == (3, 6 * H, 6 * W) and t.lazy_tile == (1, 1)     # torch.repeat: indexing only


def test_pack_maps_layout_on_the_host():
    """functional.pack_maps is data movement only, so its layout logic runs on the CPU too: same values, one
    allocation, 256-byte aligned maps, optional room for the result right behind them."""
    g = torch.Generator().manual_seed(1)
    a, r, m = torch.rand(3, 7, 9, generator=g), torch.rand(1, 7, 9, generator=g), torch.rand(1, 7, 9, generator=g).half()
    pa, pn, pr, pm, out = F.pack_maps(a, None, r, m, reserve_output=True)
    assert pn is None and torch.equal(pa, a) and torch.equal(pr, r) and torch.equal(pm, m) and pm.dtype == torch.float16
    assert out.shape == (3, 7, 9) and out.dtype == torch.float32 and out.is_contiguous()
    base = pa.untyped_storage().data_ptr()
    assert all(t.untyped_storage().data_ptr() == base for t in (pr, pm, out))            # ONE allocation
    assert all(t.data_ptr() % 256 == 0 for t in (pa, pr, pm, out))                       # in memory, whatever the allocator's base
    assert pa.data_ptr() < pr.data_ptr() < pm.data_ptr() < out.data_ptr()
    b = torch.rand(2, 3, 4, 8, generator=g)
    pb, ob = F.pack_maps(b, reserve_output=True)
    assert ob.shape == (2, 3, 4, 8) and torch.equal(pb, b)
    assert F.pack_maps(None, None) == (None, None)


def test_light_parameters_that_require_grad_are_read_detached():
    """The reference's autograd differentiates w.r.t. light / view tensors (cooktorrance.py:95-96, :126-140); here the
    descriptor carries their values and functional._CookTorranceFn returns their gradients from the backward kernel
    (tests/test_gpu_backward.py).  Building a descriptor from them must not fail or detach the caller's tensor."""
    light = torch.tensor([0.1, 0.1, 1.0], requires_grad=True)
    d = _desc(light=light)
    assert d.n_lights == 1 and abs(d.lights[0][2] - 1.0) < 1e-7 and light.requires_grad
    assert _desc(light=light.detach()).n_lights == 1


def test_integration_md_binding_stub_matches_the_abi():
    """INTEGRATION.md shows the ctypes stub a PyPBR maintainer would add: its struct must be the library's."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = text[text.index("# pypbr/models/_pbr_hip.py"):]
    block = block[:block.index("_lib = ctypes.CDLL")]
    scope = {}
    exec(block, scope)                                            # class _Map, class _Desc
    lib = N.lib()
    assert ctypes.sizeof(scope["_Desc"]) == lib.pbr_render_desc_size() == ctypes.sizeof(N.RenderDesc)
    assert [f[0] for f in scope["_Desc"]._fields_] == [f[0] for f in N.RenderDesc._fields_]
    assert f"pbr_abi_version() == {N.ABI_VERSION}" in text and f"abi_version={N.ABI_VERSION}" in text


def test_torch_ops_are_registered_with_fake_kernels():
    """torch.ops.pbr_hip.* (SURVEY.md 8b): the extension loads, every operator has a schema, and the fake kernels give
    the result's shape / dtype / device without a GPU (FakeTensor and meta tensors) -- what torch.compile traces with."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from pypbr_amd import torch_ops
    assert torch_ops.available() and os.path.exists(torch_ops.LIB_PATH)
    for op in ("cook_torrance", "cook_torrance_backward", "fold_gradient", "srgb_to_linear", "linear_to_srgb",
               "metallic_to_diffuse_specular", "diffuse_specular_to_basecolor_metallic"):
        assert getattr(torch.ops.pbr_hip, op).default._schema.name == "pbr_hip::" + op
    v, l, i = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([[0.1, 0.1, 1.0], [0.0, 0.5, 1.0]]), torch.ones(1, 3)
    with FakeTensorMode(allow_non_fake_inputs=True):
        a, n, r, m = (torch.empty(2, c, 8, 16, device="cuda") for c in (3, 3, 1, 1))
        out = torch.ops.pbr_hip.cook_torrance(a, n, r, m, None, v, l, i, 1.0, 1, True, True, False, True)
        assert out.shape == (2, 3, 8, 16) and out.dtype == torch.float32 and out.device.type == "cuda"
        out = torch.ops.pbr_hip.cook_torrance(a.half(), None, r.half(), m.half(), None, v, l, i, 0.0, 0, True, True, True, False, 4, 0, 2, 3, 10, True)
        assert out.shape == (2, 3, 10, 48) and out.dtype == torch.float16
        grads = torch.ops.pbr_hip.cook_torrance_backward(torch.empty(2, 3, 8, 16, device="cuda"), a, n, r, m, None, v, l, i, 1.0, 1, True, True,
                                                         False, True, 0, 0, 1, 1, 0, True, False, True, True, False, True)
        assert [tuple(g.shape) for g in grads] == [(2, 3, 8, 16), (0,), (2, 1, 8, 16), (2, 1, 8, 16), (0,), (15,)]
        assert torch.ops.pbr_hip.fold_gradient(torch.empty(2, 3, 16, 48, device="cuda"), 8, 16, True).shape == (1, 3, 8, 16)
        d, s = torch.ops.pbr_hip.metallic_to_diffuse_specular(a, m, True)
        assert d.shape == s.shape == a.shape
    with pytest.raises(NotImplementedError):          # no CPU kernels: the product has no CPU path
        torch.ops.pbr_hip.srgb_to_linear(torch.rand(3, 4, 4))


def test_pack_maps_layouts_on_the_host():
    """pack_maps is pure data movement, so its layouts are checkable without a GPU: values preserved, every map 256-byte
    aligned inside ONE allocation, rows dense; the optional plane skew (off by default) only changes the plane pitch."""
    g = torch.Generator().manual_seed(0)
    a, r = torch.rand(2, 3, 8, 16, generator=g), torch.rand(2, 1, 8, 16, generator=g)
    pa, pn, pr, out = F.pack_maps(a, None, r, reserve_output=True)
    assert pn is None and torch.equal(pa, a) and torch.equal(pr, r) and out.shape == (2, 3, 8, 16) and out.dtype == torch.float32
    base = pa.untyped_storage().data_ptr()
    assert pr.untyped_storage().data_ptr() == base == out.untyped_storage().data_ptr()
    assert all(t.data_ptr() % 256 == 0 and t.stride(-1) == 1 and t.stride(-2) == 16 for t in (pa, pr, out)) and pa.is_contiguous()
    big = torch.rand(3, 2048, 1024, generator=g)                      # 8 MiB planes: the size the skew option applies to
    old = F.PLANE_SKEW_BYTES
    try:
        F.PLANE_SKEW_BYTES = 4352
        (skewed,) = F.pack_maps(big)
        assert torch.equal(skewed, big) and skewed.stride() == (2048 * 1024 + 1088, 1024, 1) and not skewed.is_contiguous()
        (small,) = F.pack_maps(a)                                       # other plane sizes stay dense
        assert small.is_contiguous()
    finally:
        F.PLANE_SKEW_BYTES = old
    (dense,) = F.pack_maps(big)
    assert dense.is_contiguous() and torch.equal(dense, big)
    pm = F.pack_maps(a, None, r, reserve_output=True, material_major=True)
    assert torch.equal(pm[0], a) and torch.equal(pm[2], r) and pm[0].stride(0) == pm[2].stride(0) == pm[3].stride(0)


def test_bench_parent_process_refuses_more_ranks_than_devices_without_touching_the_gpu():
    """`python bench.py --gpus N` (N > 1) starts its own ranks; with fewer devices than ranks the PARENT says so and exits
    (it only counts devices, which does not initialise HIP) instead of printing a usage message for torchrun."""
    if torch.cuda.device_count() >= 4:
        pytest.skip("box has 4+ devices")
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PBR_BENCH_SHARE_GPU")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode != 0 and "only %d ROCm device(s) visible" % torch.cuda.device_count() in (run.stderr + run.stdout)


def test_tuning_knob_numbers_match_the_header():
    """The PBR_TUNE_* enumerators of include/pbr_hip.h, the TUNE_* constants of the ctypes binding and pbr_set_tuning agree:
    every knob of the header is known to the binding under the same number, the library accepts it (and hands back the
    previous value), and an unknown number is refused."""
    import re
    from pypbr_amd import _native as N
    header = open(os.path.join(ROOT, "include", "pbr_hip.h")).read()
    knobs = {m.group(1): int(m.group(2)) for m in re.finditer(r"PBR_TUNE_(\w+)\s*=\s*(\d+)", header)}
    count = knobs.pop("COUNT")
    assert sorted(knobs.values()) == list(range(len(knobs))) and count == len(knobs) == N.TUNE_COUNT
    # ABI 8: the boundary lists no closed experiment: two more slots are retired -- kept as numbered
    # RESERVED entries so that the others keep their numbers -- and the binding names only the ten in use
    live = {k: v for k, v in knobs.items() if not k.startswith("RESERVED_")}
    assert len(live) == 10 and {k.upper() for k in N.TUNE_NAMES} == set(live) and all(N.TUNE_NAMES[k.lower()] == v for k, v in live.items())
    lib = N.lib()
    for name, number in knobs.items():
        if name in live:
            assert getattr(N, "TUNE_" + name) == number, name
        old = lib.pbr_set_tuning(number, 0)
        assert lib.pbr_set_tuning(number, old) == 0, name              # the value just set comes back; the old one is restored
    assert lib.pbr_set_tuning(len(knobs), 0) == -1
    # PBR_TUNE_UNSET through the hook restores the rule
    rule = lib.pbr_set_tuning(N.TUNE_BATCH_INNER, 0)
    assert lib.pbr_set_tuning(N.TUNE_BATCH_INNER, N.TUNE_UNSET) == 0 and lib.pbr_set_tuning(N.TUNE_BATCH_INNER, rule) == rule


def test_per_call_tuning_travels_with_the_descriptor_and_touches_no_shared_state():
    """ABI 6: schedule knobs a caller sets go into a pbr_tuning referenced from ITS descriptor; other descriptors,
    other threads and later calls see the rules.  Checked without a GPU through pbr_kernel_name (the dispatch decision, no launch)."""
    import threading
    from pypbr_amd import _native as N
    lib = N.lib()
    multi = dict(light_type="Directional", light=[[0, 0, 1], [1, 0, 1]], light_intensity=[[1, 1, 1], [0.5, 0.5, 0.5]])
    plain, tuned = _desc(**multi), _desc(**multi)
    t = N.Tuning()
    lib.pbr_tuning_init(ctypes.byref(t))
    assert all(t.knob[i] == N.TUNE_UNSET for i in range(N.TUNE_SLOTS)) and ctypes.sizeof(N.Tuning) == 4 * N.TUNE_SLOTS
    t.knob[N.TUNE_BATCH_INNER] = 0
    tuned.tuning = ctypes.pointer(t)
    assert lib.pbr_kernel_name(ctypes.byref(plain)) == b"ctb_directional_metallic_f32_f32_v2_b2"
    assert lib.pbr_kernel_name(ctypes.byref(tuned)) == b"ct_directional_metallic_f32_f32_v4_multi"      # this descriptor only
    assert lib.pbr_kernel_name(ctypes.byref(plain)) == b"ctb_directional_metallic_f32_f32_v2_b2"        # ... nothing stuck
    # Tuning.of names the knobs; unknown names are refused
    t2 = N.Tuning.of(max_vec=1)
    assert t2.knob[N.TUNE_MAX_VEC] == 1 and t2.knob[N.TUNE_BATCH_INNER] == N.TUNE_UNSET
    with pytest.raises(KeyError):
        N.Tuning.of(warp_size=32)
    one = _desc()
    one.tuning = ctypes.pointer(t2)
    assert lib.pbr_kernel_name(ctypes.byref(one)) == b"ct_point_metallic_f32_f32_v1" and lib.pbr_kernel_name(ctypes.byref(_desc())) == b"ct_point_metallic_f32_f32_v4"
    # two threads, each with its own settings, hammering the dispatch decision: neither ever sees the other's
    wrong = []

    def worker(desc, want):
        for _ in range(2000):
            if lib.pbr_kernel_name(ctypes.byref(desc)) != want:
                wrong.append(want)
    threads = [threading.Thread(target=worker, args=(plain, b"ctb_directional_metallic_f32_f32_v2_b2")),
               threading.Thread(target=worker, args=(tuned, b"ct_directional_metallic_f32_f32_v4_multi"))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not wrong


# ---------------------------------------------------------------- build-time ISA assertions
def test_isa_assumptions_of_the_hand_scheduled_kernels_hold_and_the_checker_can_fail():
    """tools/check_isa.py on the objects of this build: the streamed backward kernel's hand-counted `s_waitcnt vmcnt(N)` needs
    exactly N stores per tile, the source's LDS-DMA loads and nothing else in vector memory, no scratch, no compiler-emitted
    access to the DMA buffer; the fp16 forward's piece exchange needs its LDS writes before its reads.  And the checker is not
    vacuous: doctored instruction streams (one spill, one extra store, an interleaved load, a compiler wait) are rejected."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa as C
    for name in ("ct_backward", "ct_loss", "cook_torrance"):
        if not os.path.exists(os.path.join(C.CSRC, name + ".o")):
            subprocess.check_call(["make", "-s", "-j4", "-C", C.CSRC])
            break
    report = C.check()
    # 12 + 12 streamed kernels, 6 piece-exchange kernels (one per light type x workflow since the streaming hint became a rule), the hazard scan, the resource scan
    assert len(report) == 34 and sum("backward_stream<" in r for r in report) == 12 and sum("mse_stream<" in r for r in report) == 12
    assert sum(r.startswith("row_walk<1, 4>") and "ring loads 8  counted waits 4" in r for r in report) == 2      # round 6: csrc/resize_stream.hpp's hand-counted loads
    assert any(r.startswith("resources:") and "none with scratch" in r for r in report)

    import tempfile
    tmp = tempfile.mkdtemp()
    co = C._code_object(os.path.join(C.CSRC, "ct_backward.o"), tmp)
    fns, meta = C._functions(co), C._metadata(co)
    sym = next(s for s in fns if "cook_torrance_backward_stream_kernelILi1ELi0ELb1E" in s)       # point, metallic, FULL
    good = fns[sym]
    assert C.check_stream_kernel(sym, good, meta[sym])[1] == []
    first_store = next(i for i, (m, _) in enumerate(good) if m.startswith("global_store"))
    first_loop_load = [i for i, (m, _) in enumerate(good) if m == "global_load_lds_dword"][14]
    doctored = {
        "spill": good[:first_store] + [("scratch_store_dword", "off, v1, s32")] + good[first_store:],
        "extra store": good[:first_store] + [("global_store_dword", "v78, v1, s[2:3] nt")] + good[first_store:],
        "plain load": good[:first_store] + [("global_load_dword", "v1, v78, s[2:3]")] + good[first_store:],
        "interleaved": good[:first_loop_load + 3] + [("global_store_dword", "v78, v1, s[2:3] nt")] + good[first_loop_load + 3:first_store] + good[first_store + 1:],
        "compiler wait": good[:first_store] + [("s_waitcnt", "vmcnt(3)")] + good[first_store:],
        "compiler lds read": good[:first_store] + [("ds_read_b32", "v9, v2 offset:2048")] + good[first_store:],
        "lds write": good[:first_store] + [("ds_write_b32", "v2, v9")] + good[first_store:],
    }
    for what, insts in doctored.items():
        assert C.check_stream_kernel(sym, insts, meta[sym])[1], what
    assert C.check_stream_kernel(sym, good, dict(meta[sym], private_segment_fixed_size=16))[1]
    co = C._code_object(os.path.join(C.CSRC, "cook_torrance.o"), tmp)
    fns, meta = C._functions(co), C._metadata(co)
    sym = next(s for s in fns if "cook_torrance_kernelILi1ELi0E6__halffLi8ELb0ELb1E" in s)
    good = fns[sym]
    assert C.check_xpose_kernel(sym, good, meta[sym])[1] == []
    w = [i for i, (m, _) in enumerate(good) if m == "ds_write_b128"]
    r = [i for i, (m, _) in enumerate(good) if m == "ds_read_b128"]
    swapped = list(good)
    swapped[w[-1]], swapped[r[0]] = swapped[r[0]], swapped[w[-1]]
    assert C.check_xpose_kernel(sym, swapped, meta[sym])[1]
    # the trans-forwarding hazard scan (every kernel of the build is clean, see `report`): the sequences it must flag and must not
    hazard = {"k": [("v_exp_f32_e32", "v91, v91"), ("v_pk_fma_f32", "v[140:141], v[6:7], v[90:91], v[8:9] clamp")]}     # the round-3 bug, verbatim
    assert C.trans_forwarding_violations(hazard)
    assert C.trans_forwarding_violations({"k": [("v_rsq_f32_e32", "v5, v4"), ("v_mul_f32_e32", "v6, v5, v7")]})
    assert not C.trans_forwarding_violations({"k": [("v_exp_f32_e32", "v91, v91"), ("s_nop", "0"), ("v_pk_fma_f32", "v[140:141], v[6:7], v[90:91], v[8:9] clamp")]})
    assert not C.trans_forwarding_violations({"k": [("v_exp_f32_e32", "v91, v91"), ("v_exp_f32_e32", "v92, v91")]})            # trans -> trans: no hazard
    assert not C.trans_forwarding_violations({"k": [("v_exp_f32_e32", "v91, v91"), ("v_mul_f32_e32", "v91, v6, v7")]})         # written, not read
    assert any("trans-forwarding hazard" in r for r in report)
    assert C.store_data_violations({"k": [("global_store_dwordx4", "v[26:27], v[18:21], off nt"), ("v_mov_b32_e32", "v19, 0")]})
    assert not C.store_data_violations({"k": [("global_store_dwordx4", "v[26:27], v[18:21], off nt"), ("s_nop", "0"), ("v_mov_b32_e32", "v19, 0")]})
    assert not C.store_data_violations({"k": [("global_store_dwordx2", "v[26:27], v[18:19], off"), ("v_mov_b32_e32", "v19, 0")]})


def test_upload_packed_layout_on_the_host():
    """functional.upload_packed is data movement only: its layout runs against a CPU 'device' too.  Maps of one dtype and (H, W) become
    ONE dense block of planes in the order given (+ free tail planes for the decoded normal map); anything else starts 256-byte aligned."""
    g = torch.Generator().manual_seed(3)
    n_raw, a, r = torch.rand(2, 6, 10, generator=g), torch.rand(3, 6, 10, generator=g), torch.rand(1, 6, 10, generator=g)
    views, block = F.upload_packed([n_raw, a, r], "cpu", tail_planes=3)
    assert [tuple(v.shape) for v in views] == [(2, 6, 10), (3, 6, 10), (1, 6, 10)] and all(torch.equal(v, t) for v, t in zip(views, (n_raw, a, r)))
    assert block.shape == (9, 6, 10) and torch.equal(block[:2], n_raw) and torch.equal(block[2:5], a) and torch.equal(block[5:6], r)
    assert views[1].data_ptr() == views[0].data_ptr() + n_raw.numel() * 4 and views[2].data_ptr() == views[1].data_ptr() + a.numel() * 4
    block[-3:].fill_(7.0)                                       # the tail is the block's own memory, behind the maps
    assert torch.equal(views[2], r)
    mixed, none = F.upload_packed([a, torch.rand(1, 3, 5, generator=g).half()], "cpu")
    assert none is None and mixed[1].dtype == torch.float16 and mixed[1].data_ptr() % 256 == 0 and torch.equal(mixed[0], a)
    assert F.upload_packed([], "cpu") == ([], None)


def test_image_maps_kept_as_samples_decode_on_the_host_exactly_like_upstream(monkeypatch):
    """materials.DEFER_IMAGE_DECODE: a map that comes out of an image keeps the image's own samples (a (C,H,W) view of PIL's (H,W,C) array)
    until somebody needs floats; read on the host, it is base.py:143-164's arithmetic, bit for bit -- all 256 / 65 536 sample values."""
    from PIL import Image
    import pypbr_amd.materials as M
    rng = np.random.default_rng(11)
    rgb = np.arange(256, dtype=np.uint8).repeat(3 * 12).reshape(-1)[rng.permutation(256 * 36)].reshape(32, 96, 3)
    deep = np.arange(65536, dtype=np.uint16).reshape(256, 256)
    grey = rng.integers(0, 256, size=(32, 96), dtype=np.uint8)
    images = dict(albedo=Image.fromarray(rgb, "RGB"), roughness=Image.fromarray(grey, "L"), height=Image.fromarray(deep))
    want = {k: M._image_to_tensor(v) for k, v in images.items()}
    assert torch.equal(want["albedo"], torch.from_numpy(rgb.transpose(2, 0, 1).copy()).float() / 255)
    assert torch.equal(want["height"], (torch.from_numpy(deep.astype(np.float32)) / 65535.0).unsqueeze(0))
    monkeypatch.setattr(M, "DEFER_IMAGE_DECODE", True)
    m = M.BasecolorMetallicMaterial(albedo=images["albedo"], roughness=images["roughness"])
    m.height = M.ImageMap(M._image_to_tensor(images["height"], defer=True))            # what the loader's worker threads hand over
    assert m._has_pending() and {k: v.dtype for k, v in m._raw.items()} == {"albedo": torch.uint8, "roughness": torch.uint8, "height": torch.uint16}
    assert m.size == (32, 96) and m._raw["albedo"].shape == (3, 32, 96) and not m._is_away()
    m.tile(2)
    assert m.lazy_tile == (2, 2) and m._has_pending()          # nobody has seen the maps: the repeat waits with them
    maps = m._maps
    assert not m._has_pending() and m.lazy_tile == (1, 1)
    assert torch.equal(maps["albedo"], want["albedo"].repeat(1, 2, 2)) and torch.equal(maps["roughness"], want["roughness"].repeat(1, 2, 2))
    assert all(t.dtype == torch.float32 and t.is_contiguous() for t in maps.values())
    c = M.BasecolorMetallicMaterial(albedo=images["albedo"], height=images["height"]).clone()
    assert not c._has_pending() and torch.equal(c._raw["height"], want["height"]) and torch.equal(c._raw["albedo"], want["albedo"])
    monkeypatch.setattr(M, "DEFER_IMAGE_DECODE", False)        # no deferral: floats at assignment, as before
    e = M.BasecolorMetallicMaterial(albedo=images["albedo"], height=M.ImageMap(M._image_to_tensor(images["height"], defer=True)))
    assert not e._has_pending() and torch.equal(e._raw["albedo"], want["albedo"]) and torch.equal(e._raw["height"], want["height"])


def test_loader_keeps_samples_and_decodes_only_the_chosen_workflows_maps(tmp_path, monkeypatch):
    """io.load_material_from_folder with materials.DEFER_IMAGE_DECODE: the workflow is decided from the file names before anything is
    decoded, the maps of the other workflow are opened (a file that is no image still raises, as upstream) but not decoded, and the maps
    read back on the host are what the eager loader gives."""
    import warnings
    import pypbr_amd.io as IO
    import pypbr_amd.materials as M
    rng = np.random.default_rng(8)
    rgb = lambda: (rng.random((8, 12, 3)) * 255).astype(np.uint8)      # noqa: E731
    gray = lambda: (rng.random((8, 12)) * 255).astype(np.uint8)        # noqa: E731
    for name, arr, mode in (("basecolor", rgb(), "RGB"), ("diffuse", rgb(), "RGB"), ("roughness", gray(), "L"), ("metallic", gray(), "L"),
                            ("specular", rgb(), "RGB"), ("height", (rng.random((8, 12)) * 65535).astype(np.uint16), "I;16")):
        _write_png(tmp_path / (name + ".png"), arr, mode)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        eager = IO.load_material_from_folder(str(tmp_path), preferred_workflow="specular")
        decoded = []
        real = IO._decoded
        monkeypatch.setattr(IO, "_decoded", lambda image, map_type, *a, **k: (decoded.append(map_type), real(image, map_type, *a, **k))[1])
        monkeypatch.setattr(M, "DEFER_IMAGE_DECODE", True)
        lazy = IO.load_material_from_folder(str(tmp_path), preferred_workflow="specular")
    assert sorted(decoded) == ["diffuse", "height", "roughness", "specular"]          # basecolor.png and metallic.png: not this workflow's
    assert type(lazy) is type(eager) and list(lazy._raw) == list(eager._raw) and lazy._has_pending()
    assert {k: v.dtype for k, v in lazy._raw.items()} == {"albedo": torch.uint8, "roughness": torch.uint8, "height": torch.uint16, "specular": torch.uint8}
    for k, v in eager._maps.items():
        assert torch.equal(lazy._maps[k], v), k
    (tmp_path / "metallic.png").write_bytes(b"not an image")
    with pytest.raises(Exception):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            IO.load_material_from_folder(str(tmp_path), preferred_workflow="specular")


def test_resize_form_names_the_family_that_serves_a_shape():
    """pbr_resize_form (ABI 8) decides like pbr_resize_bilinear and launches nothing -- it only looks at the pointers' alignment -- so the dispatch rule of
    csrc/resize.hip is host logic a box without a GPU can pin: up-scales -> the two-tap kernel, antialiased whole factors 2 ... 8 | 16 -> the band walk,
    other antialiased down-scales from 7 x up -> the row walk (round 6), everything else -> the strip kernel, beyond 36 taps -> two passes; knob value 0
    keeps the strip kernel, 2 takes the row walk at every factor it can; rows that are not whole 16-byte pieces, unaligned views and outputs too small for a
    strip never reach the row walk.  MaterialBase.resize, /root/reference/pypbr/materials/base.py:490-504."""
    from pypbr_amd import _native as N
    lib = N.lib()
    A, U = 0x7000000000, 0x7000000004            # a 16-byte aligned address and one that is not (never dereferenced)

    def form(planes, hi, wi, ho, wo, aa=1, src=A, ws=A, knob=-1):
        lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, knob)
        try:
            return lib.pbr_resize_form(src, A, planes, hi, wi, ho, wo, aa, ws)
        finally:
            lib.pbr_set_tuning(N.TUNE_RESIZE_UP2, -1)

    assert form(3, 4096, 4096, 6144, 6144, aa=0) == N.RESIZE_TWO_TAP and form(3, 512, 512, 512, 512) == N.RESIZE_TWO_TAP
    for S in (2, 3, 4, 5, 6, 7, 8, 16):
        out = 4096 // S // 4 * 4                  # (3, 5, 6, 7 do not divide 4096: 1364, 816, 680, 584 -- factors 3.003 ... 7.01)
        want = N.RESIZE_BAND_WALK if 4096 % S == 0 else (N.RESIZE_ROW_WALK if int(2.0 * 4096 / out) + 3 > 16 else N.RESIZE_STRIP)
        assert form(3, 4096, 4096, out, out) == want, S
    assert form(3, 3 * 1024, 3 * 1024, 1024, 1024) == N.RESIZE_BAND_WALK and form(3, 7 * 512, 7 * 512, 512, 512) == N.RESIZE_BAND_WALK
    assert form(8, 4096, 4096, 400, 400) == N.RESIZE_ROW_WALK and form(3, 4096, 4096, 300, 300) == N.RESIZE_ROW_WALK      # 10.24 x, 13.65 x
    assert form(3, 4096, 4096, 580, 2000) == N.RESIZE_ROW_WALK and form(3, 4096, 4096, 590, 2000) == N.RESIZE_STRIP      # 7.06 x | 6.94 x down the rows, 2.05 x across: one axis with 17 taps is enough
    assert form(8, 4096, 4096, 1365, 1365) == N.RESIZE_STRIP and form(3, 4096, 4096, 700, 700) == N.RESIZE_STRIP           # below 7 x: the strip kernel is the rule
    assert form(8, 4096, 4096, 1365, 1365, knob=2) == N.RESIZE_ROW_WALK and form(3, 4096, 4096, 3000, 3000, knob=2) == N.RESIZE_ROW_WALK
    assert form(8, 4096, 4096, 400, 400, knob=0) == N.RESIZE_STRIP and form(3, 4096, 4096, 2048, 2048, knob=0) == N.RESIZE_STRIP
    assert form(1, 4096, 4096, 100, 100) == N.RESIZE_TWO_PASS                       # 41 x: more than 36 taps
    assert form(3, 4096, 4096, 400, 400, aa=0) == N.RESIZE_STRIP                    # no antialiasing: two taps per output, no walk
    assert form(3, 4096, 4094, 400, 400) == N.RESIZE_STRIP                          # rows that are not whole 16-byte pieces
    assert form(3, 4096, 4096, 400, 400, src=U) == N.RESIZE_STRIP and form(3, 4096, 4096, 400, 400, ws=U) == N.RESIZE_STRIP
    assert form(3, 4096, 4096, 400, 12) == N.RESIZE_TWO_PASS and form(3, 128, 128, 3, 12, knob=2) == N.RESIZE_TWO_PASS     # (341 x across; 42 x down the rows)
    assert form(3, 200, 200, 20, 12, knob=2) == N.RESIZE_STRIP                      # a result narrower than a strip's minimum
    assert form(3, 4096, 4096, 4000, 4090, knob=2) == N.RESIZE_STRIP                # a factor within 1 % of 1
    assert form(0, 4096, 4096, 400, 400) == -1 and form(3, 4096, 4096, 0, 400) == -1 and lib.pbr_resize_form(0, A, 3, 64, 64, 32, 32, 1, A) == -1
