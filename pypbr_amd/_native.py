"""ctypes binding of libpbr_hip.so (the C ABI declared in include/pbr_hip.h).

There is no Python/ATen fallback behind these calls: if the shared library is
missing, or no HIP device is present when a kernel has to run, the call raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PBR_HIP_LIB") or os.path.join(_HERE, "libpbr_hip.so")   # env override: A/B of two builds

ABI_VERSION = 8
RESIZE_TWO_PASS, RESIZE_STRIP, RESIZE_TWO_TAP, RESIZE_BAND_WALK, RESIZE_ROW_WALK = range(5)      # pbr_resize_form
MAX_LIGHTS = 16

F32, F16 = 0, 1
LIGHT_DIRECTIONAL, LIGHT_POINT = 0, 1
WORKFLOW_METALLIC, WORKFLOW_SPECULAR, WORKFLOW_CONVERTED = 0, 1, 2
# (slots 0 and 4 are retired since ABI 8 -- PBR_TUNE_NONTEMPORAL, PBR_TUNE_BWD_VEC: their experiments are closed, the library ignores them)
(_TUNE_RESERVED_0, TUNE_BLOCK_LOG2, TUNE_F16_VEC, TUNE_LDS_BYTES, _TUNE_RESERVED_4, TUNE_BATCH_INNER, TUNE_SCALAR_BASE, TUNE_MAX_VEC, TUNE_BWD_RUN,
 TUNE_MSE_STREAM, TUNE_TILE_REPEAT, TUNE_RESIZE_UP2) = range(12)

TUNE_COUNT, TUNE_SLOTS, TUNE_UNSET = 12, 32, -2 ** 31
TUNE_NAMES = {"block_log2": TUNE_BLOCK_LOG2, "f16_vec": TUNE_F16_VEC, "lds_bytes": TUNE_LDS_BYTES, "batch_inner": TUNE_BATCH_INNER, "scalar_base": TUNE_SCALAR_BASE, "max_vec": TUNE_MAX_VEC,
              "bwd_run": TUNE_BWD_RUN, "mse_stream": TUNE_MSE_STREAM, "tile_repeat": TUNE_TILE_REPEAT, "resize_up2": TUNE_RESIZE_UP2}

OK = 0
ERR_NULL_MAP, ERR_WORKFLOW, ERR_LIGHT_TYPE, ERR_SHAPE, ERR_DTYPE, ERR_CHANNELS, ERR_NO_DEVICE = -1, -2, -3, -4, -5, -6, -7
ERR_UNSUPPORTED = -8

# every symbol include/pbr_hip.h declares (tests/test_abi.py checks the export table against the header)
EXPORTS = (
    "pbr_cook_torrance", "pbr_srgb_to_linear", "pbr_linear_to_srgb", "pbr_metallic_to_specular",
    "pbr_specular_to_metallic", "pbr_decode_normal", "pbr_abi_version", "pbr_error_string",
    "pbr_kernel_name", "pbr_bytes_per_pixel", "pbr_set_tuning", "pbr_render_desc_size",
    "pbr_resize_workspace_bytes", "pbr_resize_bilinear", "pbr_resize_form", "pbr_cook_torrance_backward",
    "pbr_blend_maps", "pbr_blend_sigmoid_mask", "pbr_blend_gradient_mask", "pbr_cook_torrance_autotune",
    "pbr_cook_torrance_blend", "pbr_fold_gradient", "pbr_decode_normal_backward",
    "pbr_blend_normal_sign", "pbr_blend_backward_serves", "pbr_blend_maps_backward", "pbr_param_grad_workspace_bytes", "pbr_cook_torrance_backward_params",
    "pbr_srgb_to_linear_backward", "pbr_linear_to_srgb_backward", "pbr_metallic_to_specular_backward",
    "pbr_specular_to_metallic_backward", "pbr_resize_backward_workspace_bytes", "pbr_resize_bilinear_backward",
    "pbr_blend_sigmoid_mask_backward", "pbr_cook_torrance_blend_backward", "pbr_fold_gradient_typed",
    "pbr_mse_step_workspace_bytes", "pbr_cook_torrance_mse_step", "pbr_scale_by_device_scalar", "pbr_scale_list_by_device_scalar", "pbr_device_params_bytes", "pbr_prepare_device_params", "pbr_tuning_init", "pbr_build_id", "pbr_unpack_image",
    "pbr_backward_folded_workspace_bytes", "pbr_cook_torrance_backward_folded",
)


class PbrMap(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("batch_stride", ctypes.c_int64), ("channel_stride", ctypes.c_int64)]


class Tuning(ctypes.Structure):
    """pbr_tuning: per-call schedule knobs (PBR_TUNE_* index; TUNE_UNSET = the rule).  `Tuning.of(lds_bytes=0, ...)` builds one."""
    _fields_ = [("knob", ctypes.c_int32 * TUNE_SLOTS)]

    @classmethod
    def of(cls, **knobs):
        t = cls()
        for i in range(TUNE_SLOTS):
            t.knob[i] = TUNE_UNSET
        for name, value in knobs.items():
            t.knob[TUNE_NAMES[name]] = int(value)
        return t


class RenderDesc(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32), ("batch", ctypes.c_int32), ("height", ctypes.c_int32),
        ("width", ctypes.c_int32), ("height_total", ctypes.c_int32), ("y_offset", ctypes.c_int32),
        ("map_dtype", ctypes.c_int32), ("out_dtype", ctypes.c_int32), ("workflow", ctypes.c_int32),
        ("light_type", ctypes.c_int32), ("n_lights", ctypes.c_int32), ("albedo_is_srgb", ctypes.c_int32),
        ("specular_is_srgb", ctypes.c_int32), ("return_srgb", ctypes.c_int32),
        ("albedo", PbrMap), ("normal", PbrMap), ("roughness", PbrMap), ("metallic", PbrMap), ("specular", PbrMap),
        ("out", ctypes.c_void_p),
        ("view_dir", ctypes.c_float * 3), ("light_size", ctypes.c_float),
        ("lights", (ctypes.c_float * 3) * MAX_LIGHTS), ("intensities", (ctypes.c_float * 3) * MAX_LIGHTS),
        ("schedule", ctypes.c_int32), ("map_height", ctypes.c_int32), ("map_width", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("out_batch_stride", ctypes.c_int64), ("out_channel_stride", ctypes.c_int64),
        ("device_params", ctypes.c_void_p),
        ("tuning", ctypes.POINTER(Tuning)),
    ]


class BlendDesc(ctypes.Structure):
    """pbr_blend_desc: material 2 of a fused blend + the weights of material 1."""
    _fields_ = [("albedo", PbrMap), ("normal", PbrMap), ("roughness", PbrMap), ("metallic", PbrMap), ("specular", PbrMap),
                ("mask", PbrMap), ("sign_mode", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class MapGrads(ctypes.Structure):
    """pbr_map_grads: where the gradients of one material's maps go (NULL = not wanted)."""
    _fields_ = [("albedo", ctypes.c_void_p), ("normal", ctypes.c_void_p), ("roughness", ctypes.c_void_p), ("metallic", ctypes.c_void_p),
                ("specular", ctypes.c_void_p)]


BLEND_SIGN_COMPUTE, BLEND_SIGN_GIVEN = 0, 1


SCHEDULE_AUTO, SCHEDULE_LINEAR = 0, 1


def schedule_xcd(c: int) -> int:
    """PBR_SCHEDULE_XCD(c): every XCD takes runs of 1 << c consecutive tiles."""
    return 1 + c


class NativeLibraryError(RuntimeError):
    pass


_lib = None


def lib():
    """Loads libpbr_hip.so once.  torch is imported first so that the process keeps ONE
    HIP runtime (torch's bundled libamdhip64.so.7; ours has the same soname)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads libamdhip64 before our library asks for it)
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C pypbr_amd/csrc`.  pypbr_amd has no CPU fallback." % LIB_PATH)
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise NativeLibraryError("cannot load %s: %s" % (LIB_PATH, e)) from e
    vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t
    L.pbr_cook_torrance.argtypes = [ctypes.POINTER(RenderDesc), vp]
    L.pbr_cook_torrance.restype = ctypes.c_int
    L.pbr_cook_torrance_autotune.argtypes = [ctypes.POINTER(RenderDesc), vp, ctypes.POINTER(ctypes.c_int32)]
    L.pbr_cook_torrance_autotune.restype = ctypes.c_int
    L.pbr_cook_torrance_blend.argtypes = [ctypes.POINTER(RenderDesc), ctypes.POINTER(BlendDesc), vp, vp]
    L.pbr_cook_torrance_blend.restype = ctypes.c_int
    L.pbr_cook_torrance_blend_backward.argtypes = [ctypes.POINTER(RenderDesc), ctypes.POINTER(BlendDesc), vp, vp, ctypes.POINTER(MapGrads),
                                                   ctypes.POINTER(MapGrads), vp, vp]
    L.pbr_cook_torrance_blend_backward.restype = ctypes.c_int
    L.pbr_blend_normal_sign.argtypes = [ctypes.POINTER(RenderDesc), ctypes.POINTER(BlendDesc), vp, vp]
    L.pbr_blend_normal_sign.restype = ctypes.c_int
    L.pbr_blend_backward_serves.argtypes = [ctypes.POINTER(RenderDesc)]
    L.pbr_blend_backward_serves.restype = ctypes.c_int
    L.pbr_param_grad_workspace_bytes.argtypes = [ctypes.POINTER(RenderDesc)]
    L.pbr_param_grad_workspace_bytes.restype = ctypes.c_size_t
    L.pbr_cook_torrance_backward_params.argtypes = [ctypes.POINTER(RenderDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pbr_cook_torrance_backward_params.restype = ctypes.c_int
    L.pbr_fold_gradient.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, ctypes.c_int, vp]
    L.pbr_fold_gradient.restype = ctypes.c_int
    L.pbr_mse_step_workspace_bytes.argtypes = [ctypes.POINTER(RenderDesc)]
    L.pbr_mse_step_workspace_bytes.restype = ctypes.c_size_t
    L.pbr_cook_torrance_mse_step.argtypes = [ctypes.POINTER(RenderDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pbr_cook_torrance_mse_step.restype = ctypes.c_int
    L.pbr_scale_by_device_scalar.argtypes = [vp, sz, ctypes.c_int, vp, vp]
    L.pbr_scale_by_device_scalar.restype = ctypes.c_int
    L.pbr_scale_list_by_device_scalar.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.c_int, ctypes.c_int, vp, vp]
    L.pbr_scale_list_by_device_scalar.restype = ctypes.c_int
    L.pbr_device_params_bytes.argtypes = []
    L.pbr_device_params_bytes.restype = ctypes.c_size_t
    L.pbr_prepare_device_params.argtypes = [ctypes.POINTER(RenderDesc), vp, vp, vp, i32, vp, vp]
    L.pbr_prepare_device_params.restype = ctypes.c_int
    L.pbr_fold_gradient_typed.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, ctypes.c_int, ctypes.c_int, vp]
    L.pbr_fold_gradient_typed.restype = ctypes.c_int
    L.pbr_cook_torrance_backward.argtypes = [ctypes.POINTER(RenderDesc), vp, vp, vp, vp, vp, vp, vp]
    L.pbr_cook_torrance_backward.restype = ctypes.c_int
    L.pbr_backward_folded_workspace_bytes.argtypes = [ctypes.POINTER(RenderDesc)]
    L.pbr_backward_folded_workspace_bytes.restype = ctypes.c_size_t
    L.pbr_cook_torrance_backward_folded.argtypes = [ctypes.POINTER(RenderDesc), vp, vp, vp, vp, vp, vp, vp, vp]
    L.pbr_cook_torrance_backward_folded.restype = ctypes.c_int
    L.pbr_srgb_to_linear.argtypes = [vp, vp, sz, ctypes.c_int, vp]
    L.pbr_linear_to_srgb.argtypes = [vp, vp, sz, ctypes.c_int, vp]
    L.pbr_metallic_to_specular.argtypes = [vp, vp, vp, vp, i32, i64, ctypes.c_int, ctypes.c_int, vp]
    L.pbr_specular_to_metallic.argtypes = [vp, vp, vp, vp, sz, ctypes.c_int, ctypes.c_int, vp]
    L.pbr_srgb_to_linear_backward.argtypes = [vp, vp, vp, sz, ctypes.c_int, vp]
    L.pbr_linear_to_srgb_backward.argtypes = [vp, vp, vp, sz, ctypes.c_int, vp]
    L.pbr_metallic_to_specular_backward.argtypes = [vp, vp, vp, vp, vp, vp, i32, i64, ctypes.c_int, ctypes.c_int, vp]
    L.pbr_specular_to_metallic_backward.argtypes = [vp, vp, vp, vp, vp, vp, sz, ctypes.c_int, ctypes.c_int, vp]
    L.pbr_resize_backward_workspace_bytes.argtypes = [i64, i32, i32, i32, i32]
    L.pbr_resize_backward_workspace_bytes.restype = ctypes.c_size_t
    L.pbr_resize_bilinear_backward.argtypes = [vp, vp, i64, i32, i32, i32, i32, ctypes.c_int, vp, vp]
    for name in ("pbr_srgb_to_linear_backward", "pbr_linear_to_srgb_backward", "pbr_metallic_to_specular_backward",
                 "pbr_specular_to_metallic_backward", "pbr_resize_bilinear_backward"):
        getattr(L, name).restype = ctypes.c_int
    L.pbr_decode_normal.argtypes = [vp, vp, i32, i64, ctypes.c_int, vp, vp]
    L.pbr_decode_normal_backward.argtypes = [vp, vp, vp, i32, i64, vp, vp]
    L.pbr_unpack_image.argtypes = [vp, i32, i32, i32, i32, i64, i64, i64, vp, i32, vp]
    L.pbr_unpack_image.restype = ctypes.c_int
    L.pbr_decode_normal_backward.restype = ctypes.c_int
    for name in ("pbr_srgb_to_linear", "pbr_linear_to_srgb", "pbr_metallic_to_specular",
                 "pbr_specular_to_metallic", "pbr_decode_normal", "pbr_abi_version", "pbr_set_tuning",
                 "pbr_bytes_per_pixel"):
        getattr(L, name).restype = ctypes.c_int
    L.pbr_error_string.argtypes = [ctypes.c_int]
    L.pbr_error_string.restype = ctypes.c_char_p
    L.pbr_kernel_name.argtypes = [ctypes.POINTER(RenderDesc)]
    L.pbr_kernel_name.restype = ctypes.c_char_p
    L.pbr_bytes_per_pixel.argtypes = [ctypes.POINTER(RenderDesc)]
    L.pbr_set_tuning.argtypes = [ctypes.c_int, ctypes.c_int]
    L.pbr_resize_workspace_bytes.argtypes = [i64, i32, i32]
    L.pbr_resize_workspace_bytes.restype = ctypes.c_size_t
    L.pbr_resize_bilinear.argtypes = [vp, vp, i64, i32, i32, i32, i32, ctypes.c_int, vp, vp]
    L.pbr_resize_bilinear.restype = ctypes.c_int
    L.pbr_resize_form.argtypes = [vp, vp, i64, i32, i32, i32, i32, ctypes.c_int, vp]
    L.pbr_resize_form.restype = ctypes.c_int
    L.pbr_blend_maps.argtypes = [vp, vp, vp, vp, i32, i64, ctypes.c_int, vp]
    L.pbr_blend_maps_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i64, ctypes.c_int, ctypes.c_int, vp]
    L.pbr_blend_maps_backward.restype = ctypes.c_int
    L.pbr_blend_sigmoid_mask.argtypes = [vp, vp, vp, i64, ctypes.c_float, ctypes.c_float, vp]
    L.pbr_blend_gradient_mask.argtypes = [vp, i32, i32, ctypes.c_int, vp]
    L.pbr_blend_sigmoid_mask_backward.argtypes = [vp, vp, vp, vp, i64, ctypes.c_float, vp]
    L.pbr_blend_sigmoid_mask_backward.restype = ctypes.c_int
    for name in ("pbr_blend_maps", "pbr_blend_sigmoid_mask", "pbr_blend_gradient_mask"):
        getattr(L, name).restype = ctypes.c_int
    L.pbr_render_desc_size.restype = ctypes.c_size_t
    L.pbr_build_id.argtypes = []
    L.pbr_build_id.restype = ctypes.c_char_p
    L.pbr_tuning_init.argtypes = [ctypes.POINTER(Tuning)]
    L.pbr_tuning_init.restype = None
    if L.pbr_render_desc_size() != ctypes.sizeof(RenderDesc):
        raise NativeLibraryError("pbr_render_desc layout mismatch: library %d bytes, binding %d"
                                 % (L.pbr_render_desc_size(), ctypes.sizeof(RenderDesc)))
    if L.pbr_abi_version() != ABI_VERSION:
        raise NativeLibraryError("libpbr_hip.so ABI %d, binding expects %d" % (L.pbr_abi_version(), ABI_VERSION))
    for env, knob in tuple(("PBR_TUNE_" + name.upper(), index) for name, index in TUNE_NAMES.items()):      # profiling runs: knobs from the environment
        if os.environ.get(env, "") != "":
            L.pbr_set_tuning(knob, int(os.environ[env]))
    _lib = L
    return L


def source_hash():
    """The digest pbr_build_id() would return for the sources next to this package (the Makefile's recipe: sorted csrc/*.hip and *.hpp,
    include/pbr_hip.h, the Makefile), or None where the sources are not shipped."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")))
    files += [os.path.join(_HERE, "..", "include", "pbr_hip.h"), os.path.join(csrc, "Makefile")]
    if not files or not all(os.path.isfile(f) for f in files):
        return None
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build_stamp() -> dict:
    """What every evidence file says about the binary it measured: the commit (PBR_GIT_HEAD in the environment -- the GPU box has no
    .git -- else `git rev-parse HEAD`), the library's own SHA-256, the digest of the sources it was built from (compiled in) and
    whether that is still the digest of the sources on disk (`stale`: the library would not be what `make` builds now)."""
    import hashlib
    import subprocess
    head = os.environ.get("PBR_GIT_HEAD")
    if not head:
        try:
            root = os.path.dirname(_HERE)
            head = subprocess.check_output(["git", "-C", root, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
            if subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--untracked-files=no"], stderr=subprocess.DEVNULL).strip():
                head += "+uncommitted"
        except Exception:  # noqa: BLE001
            head = "unknown"
    with open(LIB_PATH, "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()
    built_from, now = lib().pbr_build_id().decode(), source_hash()
    return {"git_head": head, "libpbr_hip_sha256": sha, "built_from_sources": built_from, "sources_on_disk": now,
            "stale": now is not None and now != built_from}


def error_string(code: int) -> str:
    return lib().pbr_error_string(code).decode()


def check(code: int):
    """Maps C-ABI status codes onto the exceptions the reference raises for the same
    condition (cooktorrance.py:62-65, :115-118; base.py:219)."""
    if code == OK:
        return
    msg = error_string(code)
    if code in (ERR_WORKFLOW, ERR_LIGHT_TYPE, ERR_CHANNELS, ERR_SHAPE, ERR_NULL_MAP):
        raise ValueError(msg)
    if code == ERR_DTYPE:
        raise TypeError(msg)
    if code == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError("HIP error %d: %s" % (code, msg))


def require_device():
    """The product has no CPU path: anything that computes needs a ROCm device."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("pypbr_amd needs a ROCm/HIP device (MI355X); there is no CPU fallback")
