"""Tensor-level entry points of the MI355X Cook-Torrance path.

`cook_torrance` evaluates what the reference's CookTorranceBRDF.forward computes
(/root/reference/pypbr/models/cooktorrance.py:92-182) on raw planar maps with ONE
fused HIP kernel launch; the map-level classes in pypbr_amd.materials / .models
call into it.  torch is used for device memory and streams only: every number is
produced by libpbr_hip.so.

Build extensions over the reference:
  * a leading batch dimension: maps may be [B,C,H,W]; B > 1 equals a loop of calls;
  * several lights: `light` [L,3] with `light_intensity` [L,3]; each light's linear
    contribution is clamped to [0,1] (what one reference call does), contributions are
    summed, clamped, and encoded once;
  * fp16 map storage (fp32 arithmetic), optional fp16 output;
  * row bands: (y_offset, height_total) evaluate rows of a taller map (multi-GPU
    sharding of a single material);
  * `convert_to_diffuse_specular`: to_diffuse_specular_material (metallic.py:71-120)
    fused in front of the specular-workflow evaluation.
"""
import collections
import ctypes
import os
import threading
import weakref
from typing import Optional, Sequence, Tuple, Union

import torch

from . import _native as N

TensorLike = Union[torch.Tensor, Sequence[float]]

_LIGHT_TYPES = {"directional": N.LIGHT_DIRECTIONAL, "point": N.LIGHT_POINT}
_DTYPES = {torch.float32: N.F32, torch.float16: N.F16}


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# ------------------------------------------------------------------ caches keyed on tensor identity + version counter
# Three caches save repeated work on tensors that did not change between calls: the device copy of a CPU-resident material
# (models.CookTorranceBRDF), the host copy of device-resident light / view tensors (below) and the "already signed?" verdict
# of a normal map (materials).  They recognise "did not change" by object identity and `tensor._version` -- which is NOT
# bumped by `t.data.add_()`, by edits of a numpy array that shares the tensor's memory (`torch.from_numpy(a).float()` shares
# it for float32 arrays, and materials ingest arrays exactly so, as upstream does), or by kernels that write through raw
# pointers (`out=`).  The reference re-reads its maps and parameters on every call, so all three are OFF by default and
# opt-in: `set_caching(device_maps=True, parameters=True, decode_verdicts=True)`, or PBR_CACHE=maps,params,decode in the
# environment, for loops that are known not to edit their tensors behind autograd's back.  Inference tensors
# (torch.inference_mode) have no version counter at all and are never cached.
CACHING = {"device_maps": False, "parameters": False, "decode_verdicts": False}
for _tok, _key in (("maps", "device_maps"), ("params", "parameters"), ("decode", "decode_verdicts")):
    if _tok in os.environ.get("PBR_CACHE", "").split(","):
        CACHING[_key] = True


def set_caching(device_maps: Optional[bool] = None, parameters: Optional[bool] = None, decode_verdicts: Optional[bool] = None) -> dict:
    """Switches the identity + version keyed caches (see above) on or off; returns the previous settings."""
    old = dict(CACHING)
    for key, v in (("device_maps", device_maps), ("parameters", parameters), ("decode_verdicts", decode_verdicts)):
        if v is not None:
            CACHING[key] = bool(v)
    return old


def version_of(t: torch.Tensor):
    """`t._version`, or None for tensors that do not track one (created under torch.inference_mode): those are not cached."""
    try:
        return None if t.is_inference() else t._version
    except RuntimeError:
        return None


_HOST_COPIES = {}          # id(tensor) -> (weakref to it, its version, host copy) for device-resident parameter tensors
_HOST_COPIES_MAX = 64


def _host_vec3(v: TensorLike, rows: Optional[int] = None):
    """Host values of a light / view parameter for the kernel-argument segment.  Tensors that live on a ROCm device do NOT come here any more
    (DEVICE_PARAMETERS: the kernels read them from device memory, _device_parameter_tensors); with that switched off a device tensor costs one
    small blocking D2H copy per call, as the reference's `.to(device)` / scalar reads do, and with
    `set_caching(parameters=True)` the host values are remembered per tensor object and version counter, so a caller
    that keeps its light / view tensors on the GPU and passes the same unchanged tensors call after call (the
    reference's override_device usage) synchronises on the first call only.  Tensors that require grad are read
    detached; their gradient comes from the backward kernel (see _CookTorranceFn)."""
    if not isinstance(v, torch.Tensor):
        t = torch.as_tensor(v, dtype=torch.float32)
    elif v.is_cuda:
        ver = version_of(v) if CACHING["parameters"] else None
        hit = _HOST_COPIES.get(id(v)) if ver is not None else None
        if hit is not None and hit[0]() is v and hit[1] == ver:
            t = hit[2]
        else:
            t = v.detach().to("cpu", torch.float32)
            if ver is not None:
                for k in [k for k, e in _HOST_COPIES.items() if e[0]() is None]:
                    del _HOST_COPIES[k]
                if len(_HOST_COPIES) >= _HOST_COPIES_MAX:
                    _HOST_COPIES.clear()
                _HOST_COPIES[id(v)] = (weakref.ref(v), ver, t)
    else:
        t = v.detach().to(torch.float32)
    if rows is None:
        if t.numel() != 3:
            raise ValueError("expected a vector of 3 components, got shape %s" % (tuple(t.shape),))
        return [float(x) for x in t.reshape(3).tolist()]
    t = t.reshape(-1, 3)
    return [[float(x) for x in r] for r in t.tolist()]


DEVICE_PARAMETERS = True     # view / light / intensity tensors that live on the device are read there (pbr_render_desc.device_params), not copied to the host


_WARNED_PARAMETER_COPY = []


def _warn_parameter_copy(t, device):
    """A device-resident parameter that is not float32 / contiguous / on the maps' device is read through a COPY made now: later
    edits of the tensor (an optimiser step, a graph replay) are not seen by a plan built from it.  Said once."""
    if not _WARNED_PARAMETER_COPY:
        _WARNED_PARAMETER_COPY.append(True)
        import warnings
        warnings.warn("pypbr_amd: a view / light / intensity tensor (%s, %s on %s) is not a contiguous float32 tensor on the maps' device (%s); "
                      "the kernels read a copy made at planning time, so a RenderPlan or captured graph will not see later updates of it"
                      % (tuple(t.shape), t.dtype, t.device, device), stacklevel=3)


def _device_parameter_tensors(view_dir, light, light_intensity, device):
    """When one of view_dir / light / light_intensity is a tensor on a ROCm device: (view [3] | None, lights [L,3] | None, intensities [1|L,3] | None)
    -- the device-resident ones as contiguous float32 tensors on `device`, None for those the caller holds on the host (they travel in the
    descriptor as always).  The kernels then read the device ones from device memory: no blocking read-back, no copy in either direction, and a
    captured graph sees the tensors' CURRENT values at every replay (a light being fitted by an optimiser).  Else None."""
    if not DEVICE_PARAMETERS or not any(isinstance(t, torch.Tensor) and t.is_cuda for t in (view_dir, light, light_intensity)):
        return None
    on_device = lambda t: isinstance(t, torch.Tensor) and t.is_cuda

    def conv(t, rows):
        if not on_device(t):
            return None
        c = t.detach().to(device, torch.float32).reshape(rows).contiguous()
        if c.data_ptr() != t.data_ptr():
            _warn_parameter_copy(t, device)
        return c
    v, lt, it = conv(view_dir, (-1,)), conv(light, (-1, 3)), conv(light_intensity, (-1, 3))
    if v is not None and v.numel() != 3:
        raise ValueError("expected a vector of 3 components, got shape %s" % (tuple(v.shape),))
    return v, lt, it


def _as_batched(t: Optional[torch.Tensor], channels: Tuple[int, ...], name: str):
    if t is None:
        return None
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.dim() != 4 or t.shape[1] not in channels:
        raise ValueError("%s must be [%s,H,W] or [B,%s,H,W], got %s" % (name, channels, channels, tuple(t.shape)))
    if t.stride(-1) != 1 or t.stride(-2) != t.shape[-1]:   # rows must be contiguous; planes/batches may be strided
        t = t.contiguous()
    return t


def _pbr_map(t: Optional[torch.Tensor]) -> N.PbrMap:
    if t is None:
        return N.PbrMap(None, 0, 0)
    return N.PbrMap(t.data_ptr(), t.stride(0) if t.shape[0] > 1 else 0, t.stride(1))


def tile_counts(tile) -> Tuple[int, int]:
    """MaterialBase.tile(num_tiles) repeats both axes num_tiles times (base.py:524-537); (ny, nx) is accepted too."""
    ny, nx = (tile, tile) if isinstance(tile, int) else (int(tile[0]), int(tile[1]))
    if ny < 1 or nx < 1:
        raise ValueError("tile counts must be >= 1, got %s" % (tile,))
    return ny, nx


def build_descriptor(albedo, normal, roughness, metallic, specular, out, *, view_dir, light, light_intensity,
                     light_type, light_size, albedo_is_srgb, specular_is_srgb, return_srgb,
                     convert_to_diffuse_specular, y_offset, height_total, schedule=0, tile=(1, 1)) -> N.RenderDesc:
    """Fills the C-ABI descriptor (include/pbr_hip.h: pbr_render_desc) from [B,C,H,W] tensors.
    Pure host logic: no device access, so it is testable without a GPU.  `tile=(ny, nx)` != (1, 1): the maps
    repeat over a (ny*H, nx*W) output (wrap-around addressing); `out` [B,3,rows,nx*W] then holds the rows
    [y_offset, y_offset + rows) of it."""
    lt = str(light_type).lower()
    if lt not in _LIGHT_TYPES:   # cooktorrance.py:62-65
        raise ValueError(f"Unsupported light_type: {lt}. Must be 'directional' or 'point'.")
    if metallic is not None:
        workflow = N.WORKFLOW_CONVERTED if convert_to_diffuse_specular else N.WORKFLOW_METALLIC
    elif specular is not None:
        if convert_to_diffuse_specular:
            raise ValueError("convert_to_diffuse_specular needs a metallic map")
        workflow = N.WORKFLOW_SPECULAR
    else:                         # cooktorrance.py:115-118
        raise ValueError("Material must have either 'metallic' or 'specular' property.")

    B, _, H, W = albedo.shape
    for name, t in (("normal", normal), ("roughness", roughness), ("metallic", metallic), ("specular", specular)):
        if t is None:
            continue
        if t.shape[-2:] != (H, W) or t.shape[0] not in (1, B):
            raise ValueError("%s %s does not match albedo %s" % (name, tuple(t.shape), tuple(albedo.shape)))
        if t.dtype != albedo.dtype or t.device != albedo.device:
            raise TypeError("all maps must share dtype and device (%s is %s on %s)" % (name, t.dtype, t.device))
    if albedo.dtype not in _DTYPES or out.dtype not in _DTYPES:
        raise TypeError("maps must be float32 or float16, got %s" % albedo.dtype)

    dev_params = _device_parameter_tensors(view_dir, light, light_intensity, albedo.device)
    # parameters that live on the device stay there (pbr_prepare_device_params, plan_cook_torrance); the others travel in the descriptor
    dv, dl, di = dev_params if dev_params is not None else (None, None, None)
    lights = [[0.0, 0.0, 0.0]] * dl.shape[0] if dl is not None else _host_vec3(light, rows=-1)
    intens = [[0.0, 0.0, 0.0]] * di.shape[0] if di is not None else _host_vec3(light_intensity, rows=-1)
    if dv is not None:
        view_dir = [0.0, 0.0, 1.0]
    if len(intens) == 1 and len(lights) > 1:
        intens = intens * len(lights)
    if len(lights) != len(intens):
        raise ValueError("light [%d,3] and light_intensity [%d,3] disagree" % (len(lights), len(intens)))
    if not 1 <= len(lights) <= N.MAX_LIGHTS:
        raise ValueError("between 1 and %d lights are supported, got %d" % (N.MAX_LIGHTS, len(lights)))

    d = N.RenderDesc()
    d.abi_version = N.ABI_VERSION
    ny, nx = tile_counts(tile)
    if (ny, nx) == (1, 1):
        d.batch, d.height, d.width = B, H, W
        d.height_total = H if height_total is None else int(height_total)
    else:
        if height_total not in (None, ny * H):
            raise ValueError("with tile=%s the full map has %d rows, not height_total=%s" % ((ny, nx), ny * H, height_total))
        d.batch, d.height, d.width = B, out.shape[-2], nx * W
        d.height_total, d.map_height, d.map_width = ny * H, H, W
        if tuple(out.shape[-2:]) != (d.height, d.width) or int(y_offset) + d.height > d.height_total:
            raise ValueError("out %s is not a band of the tiled %dx%d map" % (tuple(out.shape), ny * H, nx * W))
    d.y_offset = int(y_offset)
    d.map_dtype, d.out_dtype = _DTYPES[albedo.dtype], _DTYPES[out.dtype]
    d.workflow, d.light_type, d.n_lights = workflow, _LIGHT_TYPES[lt], len(lights)
    d.albedo_is_srgb = int(bool(albedo_is_srgb))
    d.specular_is_srgb = int(bool(specular_is_srgb))
    d.return_srgb = int(bool(return_srgb))
    d.albedo, d.normal, d.roughness = _pbr_map(albedo), _pbr_map(normal), _pbr_map(roughness)
    d.metallic, d.specular = _pbr_map(metallic), _pbr_map(specular)
    d.out = out.data_ptr()
    if out.dim() == 4 and not out.is_contiguous():      # rows contiguous (checked by the caller); planes / materials strided
        d.out_batch_stride = out.stride(0) if out.shape[0] > 1 else 0
        d.out_channel_stride = out.stride(1)
    v = _host_vec3(view_dir)
    for c in range(3):
        d.view_dir[c] = v[c]
    d.light_size = float(light_size) if light_size else 0.0   # falsy -> 1.0 inside (cooktorrance.py:130)
    d.schedule = int(schedule)
    for i, (l, it) in enumerate(zip(lights, intens)):
        for c in range(3):
            d.lights[i][c] = l[c]
            d.intensities[i][c] = it[c]
    d._device_parameters = dev_params          # (view [3], lights [L,3], intensities [1|L,3]) on the maps' device, or None
    return d


def refill_parameters(d, view_dir, light, light_intensity):
    """Host values of view / light / intensity into an existing descriptor (same number of lights): what build_descriptor writes."""
    v = _host_vec3(view_dir)
    lights, intens = _host_vec3(light, rows=-1), _host_vec3(light_intensity, rows=-1)
    if len(intens) == 1 and len(lights) > 1:
        intens = intens * len(lights)
    if len(lights) != len(intens) or len(lights) != d.n_lights:
        raise ValueError("light [%d,3] and light_intensity [%d,3] disagree with the descriptor's %d lights" % (len(lights), len(intens), d.n_lights))
    for c in range(3):
        d.view_dir[c] = v[c]
    for i, (l, it) in enumerate(zip(lights, intens)):
        for c in range(3):
            d.lights[i][c] = l[c]
            d.intensities[i][c] = it[c]


class RenderPlan:
    """A filled descriptor plus the tensors it points into: `launch()` is one C-ABI call
    (ctypes + hipLaunchKernel, a few microseconds), for callers that evaluate the same
    maps repeatedly (bench.py, a rendering-loss loop) and must not pay the Python-side
    descriptor construction per step."""

    def __init__(self, desc, out, keep_alive, squeeze):
        self.desc, self.out, self._keep, self._squeeze = desc, out, keep_alive, squeeze
        self.device = out.device
        self._fn = N.lib().pbr_cook_torrance
        self._ref = ctypes.byref(desc)
        self._param_tensors = (None, None, None)            # device-resident view / lights / intensities (prepare_device_parameters)
        self._param_block, self._param_stream = None, None

    @property
    def result(self) -> torch.Tensor:
        return self.out[0] if self._squeeze else self.out

    @property
    def kernel_name(self) -> str:
        return N.lib().pbr_kernel_name(self._ref).decode()

    @property
    def bytes_per_pixel(self) -> int:
        return N.lib().pbr_bytes_per_pixel(self._ref)

    def set_tuning(self, **knobs):
        """Per-call schedule knobs of THIS plan (pbr_render_desc.tuning; names of _native.TUNE_NAMES, e.g. lds_bytes=0, block_log2=8):
        they travel with the descriptor, touch no process-wide state and change speed only, never results.  No arguments: back to the rules."""
        if knobs:
            self._tuning = N.Tuning.of(**knobs)
            self.desc.tuning = ctypes.pointer(self._tuning)
        else:
            self._tuning = None
            self.desc.tuning = None
        return self

    def autotune(self, stream: Optional[int] = None) -> int:
        """pbr_cook_torrance_autotune on this plan's buffers: blocks, rewrites `out`, stores and returns the
        fastest schedule.  Not inside a stream capture."""
        best = ctypes.c_int32(0)
        with torch.cuda.device(self.device):
            N.check(N.lib().pbr_cook_torrance_autotune(self._ref, _stream_ptr(self.device) if stream is None else stream,
                                                       ctypes.byref(best)))
        self.desc.schedule = best.value
        return best.value

    def prepare_device_parameters(self, tensors=None, stream: Optional[int] = None):
        """pbr_prepare_device_params: folds view / light / intensity tensors that live on the device into the block the kernels read
        (enqueued on the stream; no host access to their values).  Called by plan_cook_torrance; call it again -- or capture it in a graph
        in front of `launch()` -- after the tensors changed (`tensors` = (view [3] | None, lights [L,3] | None, intensities [1|L,3] | None) to switch to others; None = the
        descriptor's host values)."""
        if tensors is not None:
            self._param_tensors = tuple(tensors)
        v, lt, it = self._param_tensors
        lib = N.lib()
        if self._param_block is None:
            self._param_block = torch.empty((lib.pbr_device_params_bytes() + 3) // 4, dtype=torch.float32, device=self.device)
        ptr = lambda t: None if t is None else t.data_ptr()
        if (lt is not None and lt.shape[0] != self.desc.n_lights) or (it is not None and it.shape[0] not in (1, self.desc.n_lights)):
            raise ValueError("light / light_intensity rows disagree with the descriptor's %d lights" % self.desc.n_lights)
        st = _stream_ptr(self.device) if stream is None else stream
        with torch.cuda.device(self.device):
            N.check(lib.pbr_prepare_device_params(self._ref, ptr(v), ptr(lt), ptr(it), 1 if it is None else it.shape[0], self._param_block.data_ptr(), st))
        self.desc.device_params = self._param_block.data_ptr()
        self._param_stream = st                             # launches on another stream fold the parameters again there (launch)

    def attach_blend(self, blend_desc, workspace, keep_alive):
        """Turns the plan into blend + evaluate (pbr_cook_torrance_blend): material 2 and the mask."""
        self._blend, self._blend_ref, self._workspace = blend_desc, ctypes.byref(blend_desc), workspace
        self._keep = self._keep + tuple(keep_alive)

    def blend_normal_sign(self, stream: Optional[int] = None) -> torch.Tensor:
        """pbr_blend_normal_sign over the rows this plan holds: an int32 [B] tensor, 1 where the blended normal of
        material b has a negative component there.  For row bands: combine the bands' flags (max) and hand them to the
        plans as `blend_flags`."""
        if getattr(self, "_blend", None) is None:
            raise ValueError("not a blend plan")
        flags = torch.zeros(self.desc.batch, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            N.check(N.lib().pbr_blend_normal_sign(self._ref, self._blend_ref, flags.data_ptr(),
                                                  _stream_ptr(self.device) if stream is None else stream))
        return flags

    def use_blend_flags(self, flags: torch.Tensor):
        """Whole-map flags worked out by the caller (see `blend_normal_sign`)."""
        self._blend.sign_mode = N.BLEND_SIGN_GIVEN
        self._workspace = flags.to(self.device, torch.int32).contiguous()

    def launch(self, stream: Optional[int] = None) -> torch.Tensor:
        """Enqueue on `stream` (raw hipStream_t) or torch's current stream of the maps' device."""
        if self.out.numel() == 0 and (self._keep[3] is not None or self._keep[4] is not None):
            return self.result         # zero-sized maps: the reference's whole-map ops return an empty (3, H, W) image; nothing to enqueue
        st = _stream_ptr(self.device) if stream is None else stream
        if self._param_stream is not None and st != self._param_stream:
            # the parameter block was written by a kernel on ANOTHER stream: nothing orders this launch behind it.  Folding the
            # parameters again on the launch stream (one wave) does, and picks up the tensors' current values.
            self.prepare_device_parameters(stream=st)
        if getattr(self, "_blend", None) is not None:
            rc = N.lib().pbr_cook_torrance_blend(self._ref, self._blend_ref, self._workspace.data_ptr(), st)
        else:
            rc = self._fn(self._ref, st)
        if rc != N.OK:
            N.check(rc)
        return self.result


def plan_cook_torrance(albedo: torch.Tensor, normal: Optional[torch.Tensor], roughness: torch.Tensor,
                       metallic: Optional[torch.Tensor] = None, specular: Optional[torch.Tensor] = None, *,
                       view_dir: TensorLike, light: TensorLike, light_intensity: TensorLike,
                       light_type: str = "point", light_size: Optional[float] = None,
                       albedo_is_srgb: bool = True, specular_is_srgb: bool = True, return_srgb: bool = True,
                       convert_to_diffuse_specular: bool = False, y_offset: int = 0,
                       height_total: Optional[int] = None, out_dtype: Optional[torch.dtype] = None,
                       out: Optional[torch.Tensor] = None, schedule: int = N.SCHEDULE_AUTO,
                       autotune: bool = False, tile=1, rows: Optional[int] = None,
                       blend: Optional[Sequence[Optional[torch.Tensor]]] = None,
                       blend_flags: Optional[torch.Tensor] = None, tuning: Optional[dict] = None) -> RenderPlan:
    """Validates the maps, allocates the output and fills the C-ABI descriptor; see `cook_torrance`.
    `schedule`: workgroup order (N.SCHEDULE_AUTO | N.SCHEDULE_LINEAR | N.schedule_xcd(c)), results do not depend
    on it; `autotune=True` measures the candidates on these very buffers once (blocking, a few launches) and
    keeps the fastest -- for plans that are launched many times.
    `tile=n | (ny, nx)`: evaluate `material.tile(n)` (base.py:524-537) without materialising the repeated maps --
    the kernel wraps its texel addresses, so each texel leaves HBM once instead of ny*nx times; the result is the
    (ny*H, nx*W) image (or its rows [y_offset, y_offset + rows)).
    `blend=(albedo2, normal2, roughness2, metallic2, specular2, mask)`: blend_with_mask (blending/functional.py:64-145:
    mask * map1 + (1 - mask) * map2, normals normalised / blended / normalised, the blended normal re-decoded as on
    assignment) fused in front of the evaluation -- both materials are read once, the blended maps are never written.
    fp32 maps, both materials complete; `mask` [1,H,W] or [B,1,H,W].  Whether a blended normal map counts as already
    signed is decided over the WHOLE map (base.py:212); the launch works that out itself unless the maps are a row band
    of a taller untiled map -- then pass `blend_flags` (int32 [B] on the device: `RenderPlan.blend_normal_sign()` of
    every band, combined with max; distributed.cook_torrance_sharded does this).
    `tuning={"lds_bytes": 0, ...}`: per-call schedule knobs of this plan (RenderPlan.set_tuning); speed only, never results."""
    if not isinstance(albedo, torch.Tensor) or not albedo.is_cuda:
        raise RuntimeError("pypbr_amd.functional.cook_torrance needs maps on a ROCm device "
                           "(use material.to('cuda')); there is no CPU path")
    squeeze = albedo.dim() == 3
    a = _as_batched(albedo, (3,), "albedo")
    n = _as_batched(normal, (3,), "normal")
    r = _as_batched(roughness, (1,), "roughness")
    m = _as_batched(metallic, (1,), "metallic")
    s = _as_batched(specular, (3,), "specular")
    B, _, H, W = a.shape
    ny, nx = tile_counts(tile)
    if (ny, nx) != (1, 1):
        H, W = (ny * H - int(y_offset)) if rows is None else int(rows), nx * W
    elif rows is not None:
        raise ValueError("`rows` selects a band of a tiled map; without `tile` pass the band's own maps")
    if out is None:
        out = torch.empty((B, 3, H, W), dtype=out_dtype or torch.float32, device=a.device)
    elif (tuple(out.shape[-3:]) != (3, H, W) or out.device != a.device or out.stride(-1) != 1 or out.stride(-2) != W
          or (out.dim() == 4 and out.shape[0] != B) or out.dim() not in (3, 4)):
        raise ValueError("out must be a [B,3,H,W] tensor with contiguous rows on the maps' device")
    if out.dim() == 3:
        out = out.unsqueeze(0)
    desc = build_descriptor(a, n, r, m, s, out, view_dir=view_dir, light=light, light_intensity=light_intensity,
                            light_type=light_type, light_size=light_size, albedo_is_srgb=albedo_is_srgb,
                            specular_is_srgb=specular_is_srgb, return_srgb=return_srgb,
                            convert_to_diffuse_specular=convert_to_diffuse_specular, y_offset=y_offset,
                            height_total=height_total, schedule=schedule, tile=(ny, nx))
    plan = RenderPlan(desc, out, (a, n, r, m, s), squeeze and out.dim() == 4)
    if tuning:
        plan.set_tuning(**tuning)
    if desc._device_parameters is not None:
        plan.prepare_device_parameters(desc._device_parameters)
    if blend is not None:
        if len(blend) != 6:
            raise ValueError("blend = (albedo2, normal2, roughness2, metallic2, specular2, mask)")
        if a.dtype != torch.float32 or out.dtype != torch.float32:
            raise TypeError("the fused blend supports float32 maps and output")
        if n is None:
            raise ValueError("the fused blend needs a normal map in both materials")
        a2, n2, r2 = (_as_batched(blend[0], (3,), "albedo2"), _as_batched(blend[1], (3,), "normal2"),
                      _as_batched(blend[2], (1,), "roughness2"))
        m2, s2 = _as_batched(blend[3], (1,), "metallic2"), _as_batched(blend[4], (3,), "specular2")
        k = _as_batched(blend[5] if blend[5].dim() != 2 else blend[5].unsqueeze(0), (1,), "mask")
        second = m2 if m is not None else s2
        if a2 is None or n2 is None or r2 is None or second is None or k is None:
            raise ValueError("the fused blend needs albedo, normal, roughness and %s of material 2 and a mask"
                             % ("metallic" if m is not None else "specular"))
        for name, t, like in (("albedo2", a2, a), ("normal2", n2, n), ("roughness2", r2, r), ("metallic2|specular2", second, m if m is not None else s),
                              ("mask", k, r)):
            if t.shape[-2:] != like.shape[-2:] or t.shape[0] not in (1, B) or t.dtype != torch.float32 or t.device != a.device:
                raise ValueError("%s %s (%s on %s) does not match material 1" % (name, tuple(t.shape), t.dtype, t.device))
        bd = N.BlendDesc()
        bd.albedo, bd.normal, bd.roughness = _pbr_map(a2), _pbr_map(n2), _pbr_map(r2)
        bd.metallic = _pbr_map(m2 if m is not None else None)
        bd.specular = _pbr_map(s2 if m is None else None)
        bd.mask = _pbr_map(k)
        if blend_flags is not None:
            if blend_flags.dtype != torch.int32 or blend_flags.numel() != B or blend_flags.device != a.device:
                raise ValueError("blend_flags must be an int32 tensor of %d flags on the maps' device" % B)
            bd.sign_mode = N.BLEND_SIGN_GIVEN
            flags = blend_flags.contiguous()
        else:
            flags = torch.empty(B, dtype=torch.int32, device=a.device)
        plan.attach_blend(bd, flags, (a2, n2, r2, second, k))
    elif autotune:
        plan.autotune()
    return plan


def cook_torrance(albedo: torch.Tensor, normal: Optional[torch.Tensor], roughness: torch.Tensor,
                  metallic: Optional[torch.Tensor] = None, specular: Optional[torch.Tensor] = None,
                  **kwargs) -> torch.Tensor:
    """Fused Cook-Torrance evaluation on device-resident planar maps.

    albedo/normal/specular: [3,H,W] or [B,3,H,W]; roughness/metallic: [1,H,W] or
    [B,1,H,W]; all on one ROCm device, float32 or float16.  Keyword arguments:
    view_dir, light, light_intensity (required), light_type="point", light_size=None,
    albedo_is_srgb=True, specular_is_srgb=True, return_srgb=True,
    convert_to_diffuse_specular=False, y_offset=0, height_total=None, out_dtype=None, out=None,
    schedule=0, autotune=False, tile=1, rows=None, blend=None, blend_flags=None (see `plan_cook_torrance`).
    Returns [3,H,W] or [B,3,H,W] (float32 unless `out_dtype`), on the same device,
    enqueued on the current stream without synchronising.  Differentiable with respect to the maps and to
    view_dir / light / light_intensity when those are tensors that require grad (backward kernels); the plain
    evaluation goes through `torch.ops.pbr_hip.cook_torrance` when that extension is built, through the ctypes
    binding of the same C ABI otherwise (and for out= / schedule= / autotune= / blend=).
    """
    if kwargs.get("blend") is not None and torch.is_grad_enabled() and any(
            isinstance(t, torch.Tensor) and t.requires_grad
            for t in (albedo, normal, roughness, metallic, specular) + tuple(kwargs["blend"]) + tuple(kwargs.get(k) for k in _PARAM_KEYS)):
        if _fused_blend_backward_can_take(albedo, kwargs):
            kw = {k: v for k, v in kwargs.items() if k != "blend"}
            try:
                return _FusedBlendFn.apply(kw, albedo, normal, roughness, metallic, specular, *kwargs["blend"])
            except _StepNotServed:
                pass
        return _blend_then_render_with_grad(albedo, normal, roughness, metallic, specular, **kwargs)
    if USE_TORCH_OPS and _torch_op_can_take(albedo, kwargs):
        return _cook_torrance_via_torch_op(albedo, normal, roughness, metallic, specular, **kwargs)
    maps = (albedo, normal, roughness, metallic, specular)
    params = tuple(kwargs.get(k) for k in _PARAM_KEYS)
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in maps + params):
        kw = {k: v for k, v in kwargs.items() if k not in _PARAM_KEYS}
        return _CookTorranceFn.apply(albedo, normal, roughness, metallic, specular, *params, kw)
    plan = plan_cook_torrance(albedo, normal, roughness, metallic, specular, **kwargs)
    with torch.cuda.device(plan.device):
        return plan.launch()


def _fused_blend_backward_can_take(albedo, kw) -> bool:
    """pbr_cook_torrance_blend_backward covers gradients of the maps of both materials and of the mask: fp32, untiled, a fresh
    result.  Gradients of view / light parameters through a blend take the unfused differentiable pieces."""
    if kw.get("out") is not None or kw.get("out_dtype") not in (None, torch.float32):
        return False
    if any(isinstance(kw.get(k), torch.Tensor) and kw[k].requires_grad for k in _PARAM_KEYS):
        return False
    tensors = [albedo] + [t for t in kw["blend"] if t is not None]
    if not all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 for t in tensors):
        return False
    if kw.get("tile", 1) not in (1, (1, 1)):
        # tiled maps (round 6, ABI 8): every map with its own planes here (a shared map would own a sum over the batch of MAP-sized planes);
        # the library answers the rest when the plan exists (pbr_blend_backward_serves: one light, map rows of whole 4-texel groups, a whole
        # output or a band that holds a period of the map's rows) -- _FusedBlendFn.forward raises _StepNotServed and the unfused pieces run
        B = albedo.shape[0] if albedo.dim() == 4 else 1
        if B > 1 and any(t.dim() < 4 or t.shape[0] == 1 for t in tensors):
            return False
    return True


class _FusedBlendFn(torch.autograd.Function):
    """blend_with_mask -> re-assignment of the blended normal -> CookTorranceBRDF.forward as ONE kernel forward
    (pbr_cook_torrance_blend) and ONE kernel backward (pbr_cook_torrance_blend_backward): the gradients of both materials' maps
    and of the mask, what the reference's autograd derives through examples/example_blend.py:14-32 inside a rendering loss."""

    @staticmethod
    def forward(ctx, kwargs, *tensors):
        maps, blend = tensors[:5], tensors[5:11]
        det = lambda t: None if t is None else t.detach()
        plan = plan_cook_torrance(*[det(t) for t in maps], blend=tuple(det(t) for t in blend), **kwargs)
        if plan.desc.map_height and not N.lib().pbr_blend_backward_serves(ctypes.byref(plan.desc)):
            raise _StepNotServed()                                # tiled, and not a launch of the fused tiled backward: the caller's unfused pieces
        ctx.plan = plan
        ctx.shapes = [None if t is None else tuple(t.shape) for t in tensors]
        ctx.save_for_backward(*[t for t in tensors if t is not None])   # for autograd's in-place-modification check
        with torch.cuda.device(plan.device):
            result = plan.launch()
        plan.out = None
        return result

    @staticmethod
    def backward(ctx, grad_out):
        ctx.saved_tensors
        plan = ctx.plan
        d = plan.desc
        B, H, W = d.batch, d.height, d.width
        g = grad_out.reshape(B, 3, H, W).to(torch.float32).contiguous()
        if d.map_height and (d.map_height != d.height_total or d.map_width != d.width):
            H, W = d.map_height, d.map_width                      # tiled maps: MAP-sized gradients, every texel's sum over its repeats
        dev = g.device
        channels = (3, 3, 1, 1, 3)
        need = ctx.needs_input_grad[1:]                       # [0] is the kwargs dict

        def bufs(offset):
            out = []
            for i in range(5):
                want = need[offset + i] and ctx.shapes[offset + i] is not None
                out.append(torch.empty((B, channels[i], H, W), dtype=torch.float32, device=dev) if want else None)
            return out
        b1, b2 = bufs(0), bufs(5)
        gmask = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev) if need[10] else None
        G1, G2 = N.MapGrads(*[None if b is None else b.data_ptr() for b in b1]), N.MapGrads(*[None if b is None else b.data_ptr() for b in b2])
        bd = N.BlendDesc.from_buffer_copy(plan._blend)        # the flags the forward launch left in the workspace are the whole map's
        bd.sign_mode = N.BLEND_SIGN_GIVEN
        with torch.cuda.device(dev):
            N.check(N.lib().pbr_cook_torrance_blend_backward(ctypes.byref(d), ctypes.byref(bd), plan._workspace.data_ptr(), g.data_ptr(),
                                                             ctypes.byref(G1), ctypes.byref(G2), None if gmask is None else gmask.data_ptr(),
                                                             _stream_ptr(dev)))

        def shaped(buf, shape):
            if buf is None:
                return None
            lead = shape[0] if len(shape) == buf.dim() else 1
            if B > 1 and lead == 1:                           # shared by the whole batch: owns the sum over the materials
                folded = torch.empty((1,) + tuple(buf.shape[1:]), dtype=torch.float32, device=dev)
                with torch.cuda.device(dev):
                    N.check(N.lib().pbr_fold_gradient(buf.data_ptr(), folded.data_ptr(), B, buf.shape[1], H, W, 1, 1, 1, _stream_ptr(dev)))
                buf = folded
            return buf.reshape(shape)
        grads = [shaped(b, s) for b, s in zip(b1 + b2, ctx.shapes[:10])]
        return (None, *grads, shaped(gmask, ctx.shapes[10]))


def _blend_then_render_with_grad(albedo, normal, roughness, metallic, specular, *, blend, **kwargs):
    """Gradients that the fused backward kernel does not cover (view / light parameters through a blend, fp16; tiled maps where the fused
    tiled backward does not serve them):
    the same computation runs unfused through the differentiable pieces -- blend_maps (pbr_blend_maps + pbr_blend_maps_backward) for every map, the
    re-decode of the blended normal (decode_normal and its backward: base.py:191-242 runs again on assignment), then the
    evaluation with its backward kernel -- so a rendering loss on a blended material (example_blend.py:14-32 inside a
    training loop) reaches both materials, the mask and the light / view parameters."""
    from .blending import blend_maps
    if kwargs.get("out") is not None:
        raise NotImplementedError("gradients through the fused blend need out=None (the result must be a fresh tensor)")
    # tile=n (round 6): the blend and the re-decode are MAP-sized operations -- the reference blends, assigns, then repeats (base.py:524-537)
    # -- and the evaluation of the blended maps under tile=n is differentiable with map-sized (folded) gradients, so nothing here changes
    kwargs.pop("blend_flags", None)        # whole maps decide "already signed?" from their own values, as the reference does
    if kwargs.get("height_total") not in (None, albedo.shape[-2]):   # (the fused backward kernel takes row bands with given flags)
        # a ROW BAND cannot take that decision from its own rows (base.py:212 looks at the whole map), and the unfused
        # re-decode below has no way to be told the whole map's verdict
        raise NotImplementedError("gradients through a fused blend need the whole map, not a row band: blend the maps first "
                                  "(pypbr_amd.blending) and evaluate the blended material's bands")
    a2, n2, r2, m2, s2, mask = blend
    squeeze = albedo.dim() == 3

    def batched(t):
        return None if t is None else (t if t.dim() == 4 else t.unsqueeze(0))
    first = [batched(t) for t in (albedo, normal, roughness, metallic, specular)]
    second = [batched(t) for t in (a2, n2, r2, m2, s2)]
    k = batched(mask if mask.dim() != 2 else mask.unsqueeze(0))
    B = first[0].shape[0]

    def pick(t, b):
        return t[b if t.shape[0] > 1 else 0]
    outs = [[], [], [], [], []]
    for b in range(B):
        for i, (x, y) in enumerate(zip(first, second)):
            if x is None or y is None:
                outs[i].append(None)
                continue
            v = blend_maps(pick(x, b), pick(y, b), pick(k, b), is_normal=(i == 1))
            outs[i].append(decode_normal(v) if i == 1 else v)
    maps = [None if col[0] is None else torch.stack(col) for col in outs]
    if maps[3] is not None:
        maps[4] = None
    out = cook_torrance(*maps, **kwargs)
    return out[0] if squeeze else out


USE_TORCH_OPS = True       # tests flip this to compare the two bindings of the same C ABI


def _torch_op_can_take(albedo, kw) -> bool:
    """`torch.ops.pbr_hip.cook_torrance` (pypbr_amd/torch_ops.py) covers the plain evaluation; explicit output buffers,
    schedules, per-call tuning, autotuning and the fused blend stay on the ctypes plan."""
    if not isinstance(albedo, torch.Tensor) or not albedo.is_cuda or albedo.numel() == 0:      # zero-sized maps: RenderPlan.launch returns the empty image
        return False
    if kw.get("out") is not None or kw.get("blend") is not None or kw.get("autotune") or kw.get("tuning") or kw.get("schedule", N.SCHEDULE_AUTO) != N.SCHEDULE_AUTO:
        return False
    if kw.get("out_dtype") not in (None, torch.float32, torch.float16):
        return False
    from . import torch_ops
    return torch_ops.available()


def _param_tensor(v, rows):
    """view / light / intensity for the operator: a tensor that requires grad goes in as it is (the operator's autograd
    formula returns its gradient); anything else as a small CPU tensor, device tensors through the cached host copy."""
    if isinstance(v, torch.Tensor):
        if (v.requires_grad and torch.is_grad_enabled()) or not v.is_cuda or DEVICE_PARAMETERS:
            return v                                         # device tensors: the operator reads them on the device (ABI 5)
        return torch.tensor(_host_vec3(v, rows=rows), dtype=torch.float32)
    t = torch.tensor(v, dtype=torch.float32)                 # Python numbers: no round trip (traceable by torch.compile)
    return t.reshape(3) if rows is None else t.reshape(-1, 3)


def _cook_torrance_via_torch_op(albedo, normal, roughness, metallic=None, specular=None, *, view_dir, light, light_intensity,
                                light_type="point", light_size=None, albedo_is_srgb=True, specular_is_srgb=True, return_srgb=True,
                                convert_to_diffuse_specular=False, y_offset=0, height_total=None, out_dtype=None, tile=1, rows=None,
                                out=None, schedule=N.SCHEDULE_AUTO, autotune=False, blend=None, blend_flags=None, tuning=None):
    # (out / schedule / autotune / blend are at their defaults here -- _torch_op_can_take -- and named so that an unknown
    # keyword raises TypeError exactly as on the plan path; blend_flags without a blend means nothing on either path)
    lt = str(light_type).lower()
    if lt not in _LIGHT_TYPES:   # cooktorrance.py:62-65
        raise ValueError(f"Unsupported light_type: {lt}. Must be 'directional' or 'point'.")
    squeeze = albedo.dim() == 3
    a = _as_batched(albedo, (3,), "albedo")
    n = _as_batched(normal, (3,), "normal")
    r = _as_batched(roughness, (1,), "roughness")
    m = _as_batched(metallic, (1,), "metallic")
    s = _as_batched(specular, (3,), "specular")
    if m is None and s is None:                         # cooktorrance.py:115-118
        raise ValueError("Material must have either 'metallic' or 'specular' property.")
    if m is not None:
        s = None
    if out_dtype == torch.float16 and torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (a, n, r, m, s)):
        raise NotImplementedError("gradients need a float32 result (fp16 maps are fine: their gradients come back in fp16)")
    ny, nx = tile_counts(tile)
    out = torch.ops.pbr_hip.cook_torrance(
        a, n, r, m, s, _param_tensor(view_dir, None), _param_tensor(light, -1), _param_tensor(light_intensity, -1),
        float(light_size) if light_size else 0.0, _LIGHT_TYPES[lt], bool(albedo_is_srgb), bool(specular_is_srgb),
        bool(convert_to_diffuse_specular), bool(return_srgb), int(y_offset), int(height_total or 0), ny, nx, int(rows or 0),
        out_dtype == torch.float16)
    return out[0] if squeeze else out


_PARAM_KEYS = ("view_dir", "light", "light_intensity")


def _param_grad(g: torch.Tensor, like, n_lights: int):
    """Shapes a [L,3] (or [3]) gradient like the parameter the caller passed: same shape, dtype and device; an intensity
    given once for several lights owns the sum."""
    if not isinstance(like, torch.Tensor) or not like.requires_grad:
        return None
    if g.dim() == 2 and like.numel() == 3 and n_lights > 1:
        g = g.sum(dim=0)
    return g.reshape(like.shape).to(device=like.device, dtype=like.dtype)


class _CookTorranceFn(torch.autograd.Function):
    """Autograd bridge: forward = the fused kernel, backward = pbr_cook_torrance_backward[_params] (one more streaming
    kernel that recomputes the forward terms).  Gradients flow to the maps and -- like the reference's autograd graph,
    cooktorrance.py:95-96, :126-140 -- to view_dir / light / light_intensity when those are tensors that require grad."""

    @staticmethod
    def forward(ctx, albedo, normal, roughness, metallic, specular, view_dir, light, light_intensity, kwargs):
        if kwargs.get("out") is not None:
            raise NotImplementedError("gradients need out=None (the result must be a fresh tensor)")
        if kwargs.get("blend") is not None:
            raise NotImplementedError("this autograd bridge takes no blend (functional.cook_torrance routes a blend under a gradient to _FusedBlendFn)")
        maps = (albedo, normal, roughness, metallic, specular)
        plan = plan_cook_torrance(*[None if t is None else t.detach() for t in maps], view_dir=view_dir, light=light,
                                  light_intensity=light_intensity, **kwargs)
        if plan.desc.out_dtype != N.F32:
            raise NotImplementedError("gradients need a float32 result (fp16 maps are fine: their gradients come back in fp16)")
        ctx.plan = plan
        ctx.in_shapes = [None if t is None else tuple(t.shape) for t in maps]
        ctx.params = (view_dir, light, light_intensity)
        ctx.save_for_backward(*[t for t in maps if t is not None])   # for autograd's in-place-modification check
        with torch.cuda.device(plan.device):
            result = plan.launch()
        plan.out = None          # backward recomputes; keeping the output in ctx would be a reference cycle
        return result

    @staticmethod
    def backward(ctx, grad_out):
        ctx.saved_tensors        # raises if a map was modified in place since forward (the kernel re-reads the maps)
        plan = ctx.plan
        d = plan.desc
        B, H, W = d.batch, d.height, d.width
        g = grad_out.reshape(B, 3, H, W).to(torch.float32).contiguous()
        channels = (3, 3, 1, 1, 3)
        present = (True, bool(d.normal.data), True, bool(d.metallic.data), bool(d.specular.data))
        gdtype = torch.float32 if d.map_dtype == N.F32 else torch.float16      # gradients in the maps' storage type
        want_params = any(ctx.needs_input_grad[5:8])
        wanted = [bool(ctx.needs_input_grad[i] and present[i] and ctx.in_shapes[i] is not None) for i in range(5)]
        tiled = bool(d.map_height) and (d.map_height, d.map_width) != (d.height_total, W)
        shared = [wanted[i] and B > 1 and (len(ctx.in_shapes[i]) == 3 or ctx.in_shapes[i][0] == 1) for i in range(5)]
        lib = N.lib()
        if tiled and not want_params and not any(shared):
            # MaterialBase.tile (base.py:524-537) under autograd: a texel owns the SUM over its repeats.  pbr_cook_torrance_backward_folded
            # hands out MAP-sized gradients -- one kernel that walks the maps and accumulates over the repeats in registers where it serves
            # the launch (one light, map rows of whole 4-texel groups), else backward + fold through its workspace
            h, w = d.map_height, d.map_width
            bufs = [torch.empty((B, channels[i], h, w), dtype=gdtype, device=g.device) if wanted[i] else None for i in range(5)]
            ws_bytes = lib.pbr_backward_folded_workspace_bytes(ctypes.byref(d))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=g.device) if ws_bytes else None
            with torch.cuda.device(g.device):
                N.check(lib.pbr_cook_torrance_backward_folded(ctypes.byref(d), g.data_ptr(), *[None if b is None else b.data_ptr() for b in bufs],
                                                              None if ws is None else ws.data_ptr(), _stream_ptr(g.device)))
            grads = [None if b is None else b.reshape(shape) for b, shape in zip(bufs, ctx.in_shapes)]
            return (*grads, None, None, None, None)
        bufs = [torch.empty((B, channels[i], H, W), dtype=gdtype, device=g.device) if wanted[i] else None for i in range(5)]
        ptrs = [None if b is None else b.data_ptr() for b in bufs]
        with torch.cuda.device(g.device):
            if want_params:
                L = d.n_lights
                gp = torch.empty(3 + 6 * L, dtype=torch.float32, device=g.device)
                ws = torch.empty(max(1, lib.pbr_param_grad_workspace_bytes(ctypes.byref(d)) // 4), dtype=torch.float32, device=g.device)
                N.check(lib.pbr_cook_torrance_backward_params(ctypes.byref(d), g.data_ptr(), *ptrs, gp.data_ptr(), ws.data_ptr(),
                                                              _stream_ptr(g.device)))
            else:
                N.check(lib.pbr_cook_torrance_backward(ctypes.byref(d), g.data_ptr(), *ptrs, _stream_ptr(g.device)))
        grads = []
        for i, (b, shape) in enumerate(zip(bufs, ctx.in_shapes)):
            if b is None:
                grads.append(None)
                continue
            # a map repeated by a fused tile(), or shared by the whole batch, owns the SUM of the per-output-pixel gradients
            if tiled and H != d.height_total:
                raise NotImplementedError("gradients of a tiled evaluation with light / view gradients or batch-shared maps need the whole output, not a row band")
            if tiled or shared[i]:
                h, w = (d.map_height, d.map_width) if tiled else (H, W)
                folded = torch.empty((1 if shared[i] else B, b.shape[1], h, w), dtype=gdtype, device=b.device)
                with torch.cuda.device(b.device):      # fp16 gradients: summed in fp32, rounded once
                    N.check(N.lib().pbr_fold_gradient_typed(b.data_ptr(), folded.data_ptr(), B, b.shape[1], h, w, H // h, W // w,
                                                            int(shared[i]), _DTYPES[gdtype], _stream_ptr(b.device)))
                b = folded
            grads.append(b.reshape(shape))
        pgrads = [None, None, None]
        if want_params:
            L = d.n_lights
            view, light, inten = ctx.params
            pgrads = [_param_grad(gp[0:3], view, 1) if ctx.needs_input_grad[5] else None,
                      _param_grad(gp[3:3 + 3 * L].reshape(L, 3), light, 1) if ctx.needs_input_grad[6] else None,
                      _param_grad(gp[3 + 3 * L:].reshape(L, 3), inten, L) if ctx.needs_input_grad[7] else None]
        return (*grads, *pgrads, None)


# ------------------------------------------------------------------ the rendering-loss step as one kernel
class _StepNotServed(Exception):
    """pbr_cook_torrance_mse_step does not serve this descriptor as one pass (raised before anything is launched)."""


class _MseStepFn(torch.autograd.Function):
    """loss = mean((cook_torrance(maps) - target)^2) with its gradients from ONE kernel (pbr_cook_torrance_mse_step): forward
    evaluates, compares and differentiates in a single pass over the maps (32 + 12 bytes read, 32 written per pixel) and keeps
    the four gradients; backward hands them over, scaled by the upstream gradient on the device (no host synchronisation; a
    scalar of exactly 1 -- `loss.backward()` -- costs one early-out launch per map)."""

    @staticmethod
    def _launch(plan, target, maps, wanted):
        """One pbr_cook_torrance_mse_step: -> (loss, gradient buffers in the maps' own shapes | None)."""
        d = plan.desc
        B, H, W = d.batch, d.height, d.width
        dev = plan.device                                   # the maps' device: a target handed over on the CPU, or on another GPU, is brought here
        tgt = target.detach().to(dev, torch.float32).reshape(B, 3, H, W).contiguous()       # H x W: the OUTPUT (tiled maps: all repeats)
        gdtype = torch.float32 if d.map_dtype == N.F32 else torch.float16
        # in the map's OWN shape ([C,H,W] or [B,C,H,W]: the same memory layout), so that backward returns the buffer itself, not a view of
        # it -- autograd takes ownership of such a gradient instead of cloning it (a 4096^2 fp16 albedo: 73 us per step)
        bufs = [torch.empty(tuple(maps[i].shape), dtype=gdtype, device=dev) if wanted[i] else None for i in range(5)]
        loss = torch.empty((), dtype=torch.float32, device=dev)
        lib = N.lib()
        ws_bytes = lib.pbr_mse_step_workspace_bytes(ctypes.byref(d))
        if ws_bytes == 0:                                   # e.g. tiled maps with several lights or ragged map rows: not one pass (pbr_hip.h)
            raise _StepNotServed()
        ws = torch.empty(max(1, ws_bytes // 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            N.check(lib.pbr_cook_torrance_mse_step(ctypes.byref(d), tgt.data_ptr(), *[None if b is None else b.data_ptr() for b in bufs],
                                                   loss.data_ptr(), ws.data_ptr(), _stream_ptr(dev)))
        return loss, bufs

    # A training loop calls the step with the SAME leaf tensors every iteration (the optimiser updates them in place): the filled
    # descriptor of the last step is kept -- pointers, never values of maps -- keyed on the very tensors (weakly held: a dropped material
    # frees its maps), their addresses and shapes and the scalar arguments; light / view VALUES are re-read every call.  Small maps make the
    # step host-bound (150 us at 256^2 against 23 us captured into a graph); the descriptor is a third of that.
    _PLANS = collections.OrderedDict()
    _PLANS_MAX = 8
    _LOCK = threading.Lock()                                # guards _PLANS itself; a kept descriptor is refilled and launched under ITS entry's lock

    @staticmethod
    def _kept_plan(maps, kwargs):
        params = tuple(kwargs.get(k) for k in _PARAM_KEYS)
        vals = []
        for v in params:
            if isinstance(v, torch.Tensor):
                if v.is_cuda or v.requires_grad:
                    return None, None, None                 # parameters on the device / with gradients: the general path every time
                vals.append(v.tolist())
            elif isinstance(v, (list, tuple)):
                vals.append([list(r) if isinstance(r, (list, tuple)) else r for r in v])
            else:
                return None, None, None
        rest = tuple(sorted((k, v if not isinstance(v, list) else tuple(v)) for k, v in kwargs.items() if k not in _PARAM_KEYS))
        try:
            hash(rest)
        except TypeError:
            return None, None, None
        for t in maps:
            if t is not None and (t.stride(-1) != 1 or t.stride(-2) != t.shape[-1]):
                return None, None, None                     # strided rows are evaluated through a copy (_as_batched): nothing to keep
        key = (tuple(None if t is None else (id(t), t.data_ptr(), tuple(t.shape), t.dtype) for t in maps), rest)
        hit = _MseStepFn._PLANS.get(key)
        if hit is not None:
            if all((r is None and t is None) or (r is not None and r() is t) for r, t in zip(hit[1], maps)):
                _MseStepFn._PLANS.move_to_end(key)
                return hit, key, vals                       # [plan, weak maps, parameter values, the entry's lock]: refilled by the caller under that lock
            del _MseStepFn._PLANS[key]
        return None, key, vals

    @staticmethod
    def forward(ctx, albedo, normal, roughness, metallic, specular, target, kwargs):
        maps = (albedo, normal, roughness, metallic, specular)
        # The process-wide lock covers the cache lookup and refill only: plan construction, the target's upload, the workspace
        # and the launch run outside it, under the ENTRY's own lock -- threads that drive different materials / GPUs do not serialise.
        with _MseStepFn._LOCK:
            hit, key, vals = _MseStepFn._kept_plan(maps, kwargs)
        plan = entry_lock = None
        if hit is not None:
            if hit[3].acquire(False):
                entry_lock, plan = hit[3], hit[0]
            else:                                           # another thread is launching through this very plan: build a private one, keep nothing
                key = None
        kept = plan is not None
        try:
            if kept and hit[2] != vals:                     # light / view VALUES are re-read every call
                try:
                    refill_parameters(plan.desc, *vals)
                    hit[2] = vals
                except ValueError:                          # another number of lights: another kernel, another plan
                    with _MseStepFn._LOCK:
                        if _MseStepFn._PLANS.get(key) is hit:
                            del _MseStepFn._PLANS[key]
                    plan, kept = None, False
            if plan is None:
                plan = plan_cook_torrance(*[None if t is None else t.detach() for t in maps], **kwargs)
                plan.out = None                                 # the colour is never written
            d = plan.desc
            present = (True, bool(d.normal.data), True, bool(d.metallic.data), bool(d.specular.data))
            wanted = [bool(ctx.needs_input_grad[i] and present[i] and maps[i] is not None) for i in range(5)]
            try:
                loss, bufs = _MseStepFn._launch(plan, target, maps, wanted)
            except _StepNotServed:
                if kept:                                        # never again through the cache: the fallback must not meet it on every call
                    with _MseStepFn._LOCK:
                        _MseStepFn._PLANS.pop(key, None)
                raise
            if (not kept and key is not None and plan._param_block is None
                    and all(p is None or p.data_ptr() == t.data_ptr() for p, t in zip(plan._keep, maps))):
                plan._keep = ()                                 # the cache holds the maps weakly (their owner keeps them alive while it wants the plan)
                with _MseStepFn._LOCK:                          # cached only AFTER a launch that was served
                    _MseStepFn._PLANS[key] = [plan, tuple(None if t is None else weakref.ref(t) for t in maps), vals, threading.Lock()]
                    while len(_MseStepFn._PLANS) > _MseStepFn._PLANS_MAX:
                        _MseStepFn._PLANS.popitem(last=False)
        finally:
            if entry_lock is not None:
                entry_lock.release()
        # the plan (descriptor + strong references to every map) is NOT kept: a second backward rebuilds it from the saved tensors
        ctx.kwargs, ctx.wanted, ctx.grads = kwargs, wanted, bufs
        ctx.present = [t is not None for t in maps]
        ctx.save_for_backward(*[t for t in maps if t is not None], target)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        saved = ctx.saved_tensors                           # in-place edits of the maps since forward are detected, as for any op
        grads = ctx.grads
        ctx.grads = None                                    # handed over below: autograd may keep the very buffers (e.g. as .grad)
        if grads is None:
            # differentiated again (retain_graph=True on the earlier backward): the first call gave its buffers away, so the step is
            # evaluated once more for fresh ones -- the common single backward never pays for a copy
            it = iter(saved[:-1])
            maps = [next(it) if p else None for p in ctx.present]
            plan = plan_cook_torrance(*[None if t is None else t.detach() for t in maps], **ctx.kwargs)
            plan.out = None
            _, grads = _MseStepFn._launch(plan, saved[-1], maps, ctx.wanted)
        live = [b for b in grads if b is not None]
        if live:                                            # all gradients scaled in ONE launch (they share a dtype and a device)
            k = grad_loss.detach().to(live[0].device, torch.float32).reshape(1).contiguous()
            ptrs = (ctypes.c_void_p * len(live))(*[b.data_ptr() for b in live])
            counts = (ctypes.c_size_t * len(live))(*[b.numel() for b in live])
            with torch.cuda.device(live[0].device):
                N.check(N.lib().pbr_scale_list_by_device_scalar(ptrs, counts, len(live), _DTYPES[live[0].dtype], k.data_ptr(), _stream_ptr(live[0].device)))
        return (*grads, None, None)


def rendering_loss_mse(albedo: torch.Tensor, normal: Optional[torch.Tensor], roughness: torch.Tensor,
                       metallic: Optional[torch.Tensor] = None, specular: Optional[torch.Tensor] = None, *,
                       target: torch.Tensor, **kwargs) -> torch.Tensor:
    """`torch.nn.MSELoss()(cook_torrance(albedo, normal, roughness, metallic | specular, **kwargs), target)` -- the rendering loss of
    docs/source/tutorials/06_advanced.rst:73-107 for the predicted material -- as a 0-dim tensor on the maps' device.  When a map
    requires grad and the evaluation qualifies (whole untiled maps of the batch, fp32 result, light / view parameters without
    gradients, every map with its own planes) loss and gradients come out of ONE kernel, pbr_cook_torrance_mse_step; otherwise
    it is the differentiable evaluation followed by torch's MSE (same value to fp32 rounding, same gradients)."""
    maps = (albedo, normal, roughness, metallic, specular)
    params = tuple(kwargs.get(k) for k in _PARAM_KEYS)
    grad_maps = torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in maps)
    grad_other = torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in params + (target,))
    B = albedo.shape[0] if albedo.dim() == 4 else 1
    shared = any(t is not None and B > 1 and (t.dim() == 3 or t.shape[0] == 1) for t in maps)
    plain = all(kwargs.get(k) in (None, d) for k, d in (("out", None), ("blend", None), ("out_dtype", torch.float32))) and \
        kwargs.get("rows") is None and not kwargs.get("autotune")
    if grad_maps and not grad_other and plain and not shared and albedo.is_cuda:
        # tiled maps (tile=n: MaterialBase.tile fused) take the one pass too -- the repeat-inner kernel leaves map-sized gradients -- where
        # the library serves them (one light, map rows of whole 4-texel groups); otherwise the three steps below
        kw = {k: v for k, v in kwargs.items() if k not in ("out", "blend", "out_dtype", "autotune", "rows")}
        try:
            return _MseStepFn.apply(albedo, normal, roughness, metallic, specular, target, kw)
        except _StepNotServed:
            pass
    out = cook_torrance(albedo, normal, roughness, metallic, specular, **kwargs)
    return torch.nn.functional.mse_loss(out.float(), target.to(out.device, torch.float32).reshape(out.shape))


# ------------------------------------------------------------------ stand-alone conversions
_PINNED_OUT = []            # weak references to page-locked results still held by callers
PINNED_RESULT_CAP = int(os.environ.get("PBR_PINNED_RESULT_CAP", str(1 << 30)))


def to_host(t: torch.Tensor, device=torch.device("cpu")) -> torch.Tensor:
    """Device -> CPU for results handed back to CPU-resident materials (the reference's default).  `t.cpu()` allocates a
    fresh pageable tensor every time: 26 ms for the 192 MiB result of a 4096^2 material, page faults included.  The
    pinned caching allocator hands back recycled page-locked blocks instead: 3.5 ms, the rate of the link
    (tools/pcie_path_probe.py).  Page-locked memory is a bounded resource and torch never returns such blocks to the OS,
    so at most PINNED_RESULT_CAP bytes (PBR_PINNED_RESULT_CAP, default 1 GiB) of results that callers still hold are
    page-locked; beyond that (a loop that stores its results) the copy is an ordinary pageable one.  Synchronises the
    current stream, as `.cpu()` does.  Plain copy when a gradient is attached (autograd has to see the transfer)."""
    if not t.is_cuda or t.requires_grad:
        return t.to(device)
    nbytes = t.numel() * t.element_size()
    _PINNED_OUT[:] = [r for r in _PINNED_OUT if r() is not None]
    held = sum(r().numel() * r().element_size() for r in _PINNED_OUT if r() is not None)
    if held + nbytes > PINNED_RESULT_CAP:
        return t.to(device)
    try:
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    except RuntimeError:                  # page-locking refused (RLIMIT_MEMLOCK, exhausted pool): the pageable way still works
        return t.to(device)
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    _PINNED_OUT.append(weakref.ref(host))
    return host


def _needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def _device_tensor(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("%s needs a tensor on a ROCm device; there is no CPU path" % what)
    if t.dtype not in _DTYPES:
        raise TypeError("%s supports float32/float16, got %s" % (what, t.dtype))
    return t.contiguous()


def _grad_like(g: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """The upstream gradient in the maps' storage type, contiguous (the backward kernels read it as the maps are stored)."""
    return g.to(like.dtype).contiguous()


def _colour_raw(t: torch.Tensor, to_linear: bool) -> torch.Tensor:
    out = torch.empty_like(t)
    fn = N.lib().pbr_srgb_to_linear if to_linear else N.lib().pbr_linear_to_srgb
    with torch.cuda.device(t.device):
        N.check(fn(t.data_ptr(), out.data_ptr(), t.numel(), _DTYPES[t.dtype], _stream_ptr(t.device)))
    return out


class _ColourFn(torch.autograd.Function):
    """srgb_to_linear / linear_to_srgb with their backward kernels (pbr_*_backward): the reference's colour transfers are plain
    torch ops (functions.py:31-66), so a rendering loss differentiates through material.to_linear() / linear_albedo."""

    @staticmethod
    def forward(ctx, texture, to_linear):
        t = _device_tensor(texture.detach(), "srgb_to_linear" if to_linear else "linear_to_srgb")
        ctx.save_for_backward(t)
        ctx.to_linear = to_linear
        return _colour_raw(t, to_linear)

    @staticmethod
    def backward(ctx, grad_out):
        (t,) = ctx.saved_tensors
        g = _grad_like(grad_out, t)
        gin = torch.empty_like(t)
        fn = N.lib().pbr_srgb_to_linear_backward if ctx.to_linear else N.lib().pbr_linear_to_srgb_backward
        with torch.cuda.device(t.device):
            N.check(fn(t.data_ptr(), g.data_ptr(), gin.data_ptr(), t.numel(), _DTYPES[t.dtype], _stream_ptr(t.device)))
        return gin, None


def srgb_to_linear(texture: torch.Tensor) -> torch.Tensor:
    """utils.srgb_to_linear (pypbr/utils/functions.py:31-47) on the device; differentiable (its own backward kernel)."""
    if _needs_grad(texture):
        return _ColourFn.apply(texture, True)
    return _colour_raw(_device_tensor(texture, "srgb_to_linear"), True)


def linear_to_srgb(texture: torch.Tensor) -> torch.Tensor:
    """utils.linear_to_srgb (pypbr/utils/functions.py:50-66) on the device; differentiable (its own backward kernel)."""
    if _needs_grad(texture):
        return _ColourFn.apply(texture, False)
    return _colour_raw(_device_tensor(texture, "linear_to_srgb"), False)


def _m2ds_raw(a, m, albedo_is_srgb):
    diffuse, spec = torch.empty_like(a), torch.empty_like(a)
    P = a.shape[-1] * a.shape[-2]
    with torch.cuda.device(a.device):
        N.check(N.lib().pbr_metallic_to_specular(a.data_ptr(), m.data_ptr(), diffuse.data_ptr(), spec.data_ptr(),
                                                 a.numel() // (3 * P), P, int(albedo_is_srgb), _DTYPES[a.dtype],
                                                 _stream_ptr(a.device)))
    return diffuse, spec


class _MetallicToSpecularFn(torch.autograd.Function):
    """to_diffuse_specular_material's arithmetic (metallic.py:98-108) with its backward kernel."""

    @staticmethod
    def forward(ctx, albedo, metallic, albedo_is_srgb):
        a, m = albedo.detach(), metallic.detach()
        ctx.save_for_backward(a, m)
        ctx.srgb = bool(albedo_is_srgb)
        return _m2ds_raw(a, m, albedo_is_srgb)

    @staticmethod
    def backward(ctx, g_diffuse, g_specular):
        a, m = ctx.saved_tensors
        gd = None if g_diffuse is None else _grad_like(g_diffuse, a)
        gs = None if g_specular is None else _grad_like(g_specular, a)
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gm = torch.empty_like(m) if ctx.needs_input_grad[1] else None
        P = a.shape[-1] * a.shape[-2]
        ptr = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(a.device):
            N.check(N.lib().pbr_metallic_to_specular_backward(a.data_ptr(), m.data_ptr(), ptr(gd), ptr(gs), ptr(ga), ptr(gm),
                                                              a.numel() // (3 * P), P, int(ctx.srgb), _DTYPES[a.dtype],
                                                              _stream_ptr(a.device)))
        return ga, gm, None


def metallic_to_diffuse_specular(albedo: torch.Tensor, metallic: torch.Tensor, albedo_is_srgb: bool = False):
    """Arithmetic of to_diffuse_specular_material (metallic.py:98-108).  albedo [..,3,H,W],
    metallic [..,1,H,W] -> (diffuse, specular) both [..,3,H,W] in linear space.  Differentiable w.r.t. both maps."""
    a = _device_tensor(albedo, "metallic_to_diffuse_specular")
    m = _device_tensor(metallic, "metallic_to_diffuse_specular")
    if a.shape[-3] != 3 or m.shape[-3] != 1 or a.shape[-2:] != m.shape[-2:] or a.shape[:-3] != m.shape[:-3]:
        raise ValueError("albedo [..,3,H,W] / metallic [..,1,H,W] expected, got %s / %s" % (tuple(a.shape), tuple(m.shape)))
    if m.dtype != a.dtype:
        m = m.to(a.dtype)
    if _needs_grad(a, m):
        return _MetallicToSpecularFn.apply(a, m, bool(albedo_is_srgb))
    return _m2ds_raw(a, m, albedo_is_srgb)


def _ds2bm_raw(d, s, albedo_is_srgb):
    base, met = torch.empty_like(d), torch.empty_like(d)
    with torch.cuda.device(d.device):
        N.check(N.lib().pbr_specular_to_metallic(d.data_ptr(), s.data_ptr(), base.data_ptr(), met.data_ptr(),
                                                 d.numel(), int(albedo_is_srgb), _DTYPES[d.dtype], _stream_ptr(d.device)))
    return base, met


class _SpecularToMetallicFn(torch.autograd.Function):
    """to_basecolor_metallic_material's arithmetic (diffuse.py:128-147) with its backward kernel (torch's sub-gradients through
    clamp / where; the thresholded selects are re-taken with the forward's own arithmetic)."""

    @staticmethod
    def forward(ctx, diffuse, specular, albedo_is_srgb):
        d, s = diffuse.detach(), specular.detach()
        ctx.save_for_backward(d, s)
        ctx.srgb = bool(albedo_is_srgb)
        return _ds2bm_raw(d, s, albedo_is_srgb)

    @staticmethod
    def backward(ctx, g_basecolor, g_metallic):
        d, s = ctx.saved_tensors
        gb = None if g_basecolor is None else _grad_like(g_basecolor, d)
        gm = None if g_metallic is None else _grad_like(g_metallic, d)
        gd = torch.empty_like(d) if ctx.needs_input_grad[0] else None
        gs = torch.empty_like(s) if ctx.needs_input_grad[1] else None
        ptr = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(d.device):
            N.check(N.lib().pbr_specular_to_metallic_backward(d.data_ptr(), s.data_ptr(), ptr(gb), ptr(gm), ptr(gd), ptr(gs), d.numel(),
                                                              int(ctx.srgb), _DTYPES[d.dtype], _stream_ptr(d.device)))
        return gd, gs, None


def diffuse_specular_to_basecolor_metallic(diffuse: torch.Tensor, specular: torch.Tensor, albedo_is_srgb: bool = False):
    """Arithmetic of to_basecolor_metallic_material (diffuse.py:128-147): RAW specular in,
    (basecolor, 3-channel metallic) out.  Differentiable w.r.t. both maps."""
    d = _device_tensor(diffuse, "diffuse_specular_to_basecolor_metallic")
    s = _device_tensor(specular, "diffuse_specular_to_basecolor_metallic")
    if d.shape != s.shape:
        raise ValueError("diffuse %s and specular %s must have the same shape" % (tuple(d.shape), tuple(s.shape)))
    if s.dtype != d.dtype:
        s = s.to(d.dtype)
    if _needs_grad(d, s):
        return _SpecularToMetallicFn.apply(d, s, bool(albedo_is_srgb))
    return _ds2bm_raw(d, s, albedo_is_srgb)


def pack_maps(*maps: Optional[torch.Tensor], device=None, reserve_output: bool = False, material_major: bool = False):
    """Copies the maps of a material (or of a batch) into a single device allocation and returns views of it (same
    shapes, dtypes and values; `None` stays `None`).  Pure data movement (torch copies), no arithmetic.  Why: a launch
    streams all planes of a material at once, and planes that live in one allocation sit close together in the
    address space; `reserve_output=True` also appends room for the fp32 result and returns it last, so that `out=`
    can be placed next to its inputs.  `material_major=True` (batched [B,C,H,W] maps): material b's planes and its
    result next to each other, materials one pitch apart -- strided views, which the C ABI takes as they are.  It
    is an option, not the default: measured 3-4 % ahead for 16 x 4096^2, level for 64 x 2048^2, 2-5 % behind for
    64 x 1024^2 (tools/batch_layout_probe.py).  See DESIGN.md, "Data layout in HBM", for the measurements."""
    present = [t for t in maps if t is not None]
    if not present:
        return tuple(maps) + ((None,) if reserve_output else ())
    dev = torch.device(device) if device is not None else present[0].device

    def padded(nbytes):                                  # every map starts 256-byte aligned
        return -(-nbytes // 256) * 256
    batch = max([t.shape[0] for t in present if t.dim() == 4] or [1])
    if batch > 1 and material_major:
        return _pack_material_major(maps, batch, dev, reserve_output, padded)
    out_shape = None
    if reserve_output:
        out_shape = tuple(present[0].shape[:-3]) + (3,) + tuple(present[0].shape[-2:])

    def pitch_of(shape, esz):
        """Bytes from one plane of a map to the next.  Planes whose size is a multiple of 8 MiB (2048^2, 4096^2 fp32 ...)
        all start on the same HBM channel group when packed back to back; PLANE_SKEW_BYTES (when set) more per plane
        spread them (tools/skew_probe.py: 16 x 2048^2 with every tensor skewed, linear order: 6.16 -> 6.44 TB/s; inside
        one arena the effect is ~1 %, see below).  Other sizes stay dense."""
        plane = shape[-2] * shape[-1] * esz
        return plane + (PLANE_SKEW_BYTES if PLANE_SKEW_BYTES and plane % (8 << 20) == 0 else 0)

    def extent(shape, esz):
        n_planes = 1
        for d in shape[:-2]:
            n_planes *= d
        return padded(n_planes * pitch_of(shape, esz))
    sizes = [0 if t is None else extent(t.shape, t.element_size()) for t in maps]
    out_bytes = extent(out_shape, 4) if reserve_output else 0
    arena = _aligned_arena(sum(sizes) + out_bytes, dev)

    def view(off, shape, dtype, esz):
        pitch = pitch_of(shape, esz) // esz
        strides = [1, shape[-1]]
        step = pitch
        for d in reversed(shape[:-2]):
            strides.append(step)
            step *= d
        strides = tuple(reversed(strides))            # (..., planes, rows, 1): dense rows, `pitch` elements between planes
        typed = arena.view(dtype)
        return typed.as_strided(tuple(shape), strides, typed.storage_offset() + off // esz)
    views, off = [], 0
    for t, nbytes in zip(maps, sizes):
        if t is None:
            views.append(None)
            continue
        v = view(off, t.shape, t.dtype, t.element_size())
        v.copy_(t)
        views.append(v)
        off += nbytes
    if reserve_output:
        views.append(view(off, out_shape, torch.float32, 4))
    return tuple(views)


# The page-locked staging area of upload_packed, one per thread, reused: allocating one is a hipHostMalloc (4 ms for 10 MB) and torch's
# caching host allocator handed a recycled block back only some of the time -- the upload of examples/example_brdf.py's material took 0.8
# or 4.5 ms by that alone (tools/example_bench.py, per-repeat times).  Grown geometrically; requests beyond the cap get a block of their own.
UPLOAD_STAGE_CAP = int(os.environ.get("PBR_UPLOAD_STAGE_CAP", str(256 << 20)))
_UPLOAD_STAGE = threading.local()


def _upload_stage(nbytes: int):
    """-> (`nbytes` of page-locked uint8 -- pageable where page-locking is refused: still one transfer --, the slot to leave the copy's
    event in or None).  The previous copy out of the slot is waited for before its memory is handed out again."""
    def fresh(n):
        try:
            return torch.empty(n, dtype=torch.uint8, pin_memory=True)
        except RuntimeError:
            return torch.empty(n, dtype=torch.uint8)
    if nbytes > UPLOAD_STAGE_CAP:
        return fresh(nbytes), None
    slot = getattr(_UPLOAD_STAGE, "slot", None)
    if slot is None or slot[0].numel() < nbytes:
        grown = max(nbytes, 2 * slot[0].numel() if slot is not None else 0)
        slot = _UPLOAD_STAGE.slot = [fresh(min(grown, UPLOAD_STAGE_CAP)), None]
    if slot[1] is not None:
        slot[1].synchronize()
        slot[1] = None
    return slot[0][:nbytes], slot


def release_upload_stage():
    """Drops the CALLING thread's page-locked staging block (up to UPLOAD_STAGE_CAP bytes stay pinned per uploading thread otherwise:
    loader pools and server threads that are done uploading call this, or set PBR_UPLOAD_STAGE_CAP lower).  The copy still in flight
    out of it is waited for first."""
    slot = getattr(_UPLOAD_STAGE, "slot", None)
    if slot is not None:
        if slot[1] is not None:
            slot[1].synchronize()
        _UPLOAD_STAGE.slot = None


# Host tensor -> its place in the staging area.  Up to this many bytes per upload the copy is a plain memcpy on the calling thread, NOT
# Tensor.copy_: ATen spreads a host copy over its whole OpenMP pool (128 threads on a GPU box's 256-core host), whose workers then spin
# on every core -- inside the box's CPU quota (16 cores) that stalled this very thread for 70-170 ms at a time (CFS throttling: every other
# upload of examples/example_brdf.py's 10 MB of samples), and each munmap that followed paid TLB shootdowns to all of them (2-5 ms to free
# the samples).  Measured with tools/upload_phase_probe.py: 0.4-0.6 ms, every time, for the same bytes by memcpy.  Above the limit (a
# 4096^2 float material is 537 MB) the pool's bandwidth is worth more than that risk.
STAGE_MEMCPY_LIMIT = int(os.environ.get("PBR_STAGE_MEMCPY_LIMIT", str(128 << 20)))


def _page_locked_range(samples):
    """Dense sample arrays that all live in ONE page-locked storage, each dword-aligned, covering a range not much larger than their
    bytes -> (address of the range's first byte, its length, the range as a uint8 tensor); None otherwise."""
    first = samples[0]
    storage = first.untyped_storage()
    if not all(t.untyped_storage().data_ptr() == storage.data_ptr() for t in samples) or not first.is_pinned():
        return None
    lo = min(t.data_ptr() for t in samples)
    hi = max(t.data_ptr() + t.numel() * t.element_size() for t in samples)
    used = sum(t.numel() * t.element_size() for t in samples)
    if any((t.data_ptr() - lo) % 4 for t in samples) or lo % 4 or hi - lo > 2 * used + 4096:
        return None
    whole = torch.empty(0, dtype=torch.uint8).set_(storage)
    start = lo - storage.data_ptr()
    return lo, hi - lo, whole[start:start + (hi - lo)]


def _stage_copy(dst_bytes: torch.Tensor, src: torch.Tensor, upload_bytes: int):
    # a host memcpy: only a HOST source may take it (a device pointer here would be read by the CPU, unordered against the kernels
    # still queued on that device); anything else goes through copy_, which knows about devices and streams
    if upload_bytes <= STAGE_MEMCPY_LIMIT and src.device.type == "cpu" and src.is_contiguous():
        ctypes.memmove(dst_bytes.data_ptr(), src.data_ptr(), dst_bytes.numel())
    else:
        dst_bytes.view(src.dtype).view(src.shape).copy_(src)


ENCODED_DTYPES = (torch.uint8, torch.uint16)       # an image's own samples (materials._image_to_tensor(..., defer=True)); float32 / 255 or / 65535 once decoded


def is_encoded(t) -> bool:
    return t is not None and t.dtype in ENCODED_DTYPES


def _dense_samples(t: torch.Tensor):
    """(C,H,W) view of image samples -> (the dense array behind it, (stride_c, stride_h, stride_w) in samples).  PIL's (H,W,C) array
    seen as (C,H,W) travels as it is; anything that is not one dense block is copied to (C,H,W) order first."""
    C, H, W = t.shape
    hwc = t.permute(1, 2, 0)
    if hwc.is_contiguous():
        return hwc, (1, W * C, C)
    t = t.contiguous()
    return t, (H * W, W, 1)


def unpack_image(samples: torch.Tensor, bits: int, strides, shape, out: torch.Tensor, decode_normal: bool = False) -> torch.Tensor:
    """MaterialBase._to_tensor for PIL images (base.py:143-164) on the device: `bits`-wide samples (8 | 16) starting at `samples`'
    first byte, resident on `out`'s device, addressed [c*strides[0] + y*strides[1] + x*strides[2]] -> float32 (C,H,W) `out` (dense);
    with `decode_normal` base.py:191-242 follows in the same pass and `out` is (3,H,W) (pbr_unpack_image)."""
    C, H, W = shape
    if out.dtype != torch.float32 or not out.is_contiguous() or tuple(out.shape) != ((3 if decode_normal else C), H, W) or samples.device != out.device:
        raise ValueError("unpack_image: `out` must be a contiguous float32 (C,H,W) tensor on the samples' device")
    with torch.cuda.device(out.device):
        N.check(N.lib().pbr_unpack_image(samples.data_ptr(), bits, C, H, W, strides[0], strides[1], strides[2],
                                         out.data_ptr(), 1 if decode_normal else 0, _stream_ptr(out.device)))
    return out


def upload_packed(tensors: Sequence[torch.Tensor], device, tail_planes: int = 0, encoded_normal: Optional[int] = None):
    """CPU tensors -> tensors on `device` with ONE host-to-device copy: the maps are laid out in a page-locked host arena exactly as
    they will sit in the device allocation, which then arrives as a single DMA transfer (five separate `t.to(device)` of pageable
    tensors are five transfers, each bounced through the runtime's own staging buffers).  Maps that share dtype and (H, W) -- a
    material's maps as a rule -- are packed as DENSE planes, so that the whole material is one [P,H,W] block: whole-material
    operations (MaterialBase.resize) then take one launch over all planes; `tail_planes` more planes of that shape are left free
    behind them (a float normal map's decoded form lands there).  Otherwise every map starts 256-byte aligned.

    Maps that are still an image's samples (uint8 / uint16, `is_encoded`) travel AS SAMPLES -- a quarter / half of the bytes -- in a
    staging area in front of the block and are turned into float32 on arrival (pbr_unpack_image, one launch per map; the map at index
    `encoded_normal` is a normal map and is decoded on the way, base.py:191-242); their float planes sit behind the planes of the
    maps that arrived as floats, so that the copy stays ONE contiguous transfer and the block stays dense.

    Returns (views in the order given, the [P + tail_planes, H, W] block or None)."""
    dev = torch.device(device)
    ts = [t.detach() for t in tensors]
    if not ts:
        return [], None
    enc = [is_encoded(t) for t in ts]
    out_dtype = [torch.float32 if e else t.dtype for t, e in zip(ts, enc)]
    out_shape = [((3,) + tuple(t.shape[1:]) if i == encoded_normal else tuple(t.shape)) for i, t in enumerate(ts)]
    same = all(d == out_dtype[0] and t.dim() == 3 and t.shape[-2:] == ts[0].shape[-2:] for t, d in zip(ts, out_dtype))

    def slot_bytes(i):
        n = out_dtype[i].itemsize
        for e in out_shape[i]:
            n *= e
        return n if same else -(-n // 256) * 256

    # staging area (samples), then the maps that arrive as floats, then the unpacked maps, then the tail
    dense, stage_off, off = {}, {}, 0
    for i, t in enumerate(ts):
        if enc[i]:
            dense[i] = _dense_samples(t)
            stage_off[i] = off
            off += -(-dense[i][0].numel() * t.element_size() // 256) * 256
    # Samples the loader decoded straight into ONE page-locked block (io.load_material_from_folder) are already where a DMA transfer can
    # read them, laid out for it: the block's used range goes up as it is -- no staging copy, and nothing to free but the block itself.
    direct = _page_locked_range([dense[i][0] for i in range(len(ts))]) if all(enc) and dev.type == "cuda" else None
    if direct is not None:
        base, off = direct[0], -(-direct[1] // 256) * 256
        stage_off = {i: dense[i][0].data_ptr() - base for i in range(len(ts))}
    staged = off
    offs = {}
    for i in [i for i in range(len(ts)) if not enc[i]] + [i for i in range(len(ts)) if enc[i]]:
        offs[i] = off
        off += slot_bytes(i)
    sent = staged + sum(slot_bytes(i) for i in range(len(ts)) if not enc[i])           # bytes of the one transfer
    plane = ts[0].shape[-2] * ts[0].shape[-1] * out_dtype[0].itemsize
    total = off + (tail_planes * plane if same else 0)
    if direct is not None:
        host, stage = direct[2], None
    else:
        host, stage = _upload_stage(sent)
        for i, t in enumerate(ts):
            src, o = (dense[i][0], stage_off[i]) if enc[i] else (t, offs[i])
            _stage_copy(host[o:o + src.numel() * src.element_size()], src, sent)
    arena = _aligned_arena(total, dev)
    arena[:host.numel()].copy_(host, non_blocking=True)
    if stage is not None and dev.type == "cuda":
        stage[1] = torch.cuda.Event()
        stage[1].record(torch.cuda.current_stream(dev))
    views = []
    for i, t in enumerate(ts):
        n = out_dtype[i].itemsize
        for e in out_shape[i]:
            n *= e
        view = arena[offs[i]:offs[i] + n].view(out_dtype[i]).view(out_shape[i])
        if enc[i]:
            unpack_image(arena[stage_off[i]:], 8 * t.element_size(), dense[i][1], tuple(t.shape), view, decode_normal=(i == encoded_normal))
        views.append(view)
    block = None
    if same:
        block = arena[staged:].view(out_dtype[0]).view(-1, ts[0].shape[-2], ts[0].shape[-1])
    return views, block


# 0 = dense planes (the default).  Measured with 4352 (17 x 256 B, tools/skew_ab.sh): batches of 2048^2 maps +1 %
# (16 maps: 467.6 -> 462.5 us), 64 x 2048^2 +0.6 %, 4 x 4096^2 level, the bench workload (one 4096^2 material) 1.5 % SLOWER
# (114.0 -> 116.1 us) -- so it stays an experiment knob.
PLANE_SKEW_BYTES = int(os.environ.get("PBR_PLANE_SKEW_BYTES", "0"))


def _aligned_arena(nbytes, dev):
    """`nbytes` of uint8 whose first byte is 256-byte aligned IN MEMORY, whatever the allocator hands out (the device
    allocator already aligns to 512; the host allocator only to 64)."""
    raw = torch.empty(nbytes + 255, dtype=torch.uint8, device=dev)
    skip = -raw.data_ptr() % 256
    return raw[skip:skip + nbytes]


def _pack_material_major(maps, batch, dev, reserve_output, padded):
    """Batched maps [B,C,H,W]: material b's planes (and its result) next to each other, materials one pitch apart.
    The views keep their [B,C,H,W] shapes; only the batch stride differs from a free-standing tensor, which the C ABI
    takes per map.  Maps shared by the whole batch ([1,C,H,W]) are stored once, behind the materials."""
    per_material = [t for t in maps if t is not None and t.dim() == 4 and t.shape[0] == batch]
    if any(t is not None and not (t.dim() == 4 and t.shape[0] in (1, batch)) for t in maps):
        raise ValueError("batched maps must all be [B,C,H,W] or [1,C,H,W]")
    h, w = per_material[0].shape[-2:]
    pitch = sum(padded(t[0].numel() * t.element_size()) for t in per_material)
    out_bytes = padded(3 * h * w * 4) if reserve_output else 0
    pitch += out_bytes
    shared_bytes = sum(padded(t.numel() * t.element_size()) for t in maps if t is not None and t.shape[0] == 1)
    arena = _aligned_arena(batch * pitch + shared_bytes, dev)
    views, off, shared_off = [], 0, batch * pitch
    for t in maps:
        if t is None:
            views.append(None)
            continue
        es, plane = t.element_size(), t.shape[-2] * t.shape[-1]
        typed = arena.view(t.dtype)
        if t.shape[0] == 1:
            v = typed.as_strided(tuple(t.shape), (t[0].numel(), plane, t.shape[-1], 1), typed.storage_offset() + shared_off // es)
            shared_off += padded(t.numel() * es)
        else:
            v = typed.as_strided(tuple(t.shape), (pitch // es, plane, t.shape[-1], 1), typed.storage_offset() + off // es)
            off += padded(t[0].numel() * es)
        v.copy_(t)
        views.append(v)
    if reserve_output:
        typed = arena.view(torch.float32)
        views.append(typed.as_strided((batch, 3, h, w), (pitch // 4, h * w, w, 1), typed.storage_offset() + off // 4))
    return tuple(views)


def _resize_raw(t: torch.Tensor, ho: int, wo: int, antialias: bool) -> torch.Tensor:
    h, w = t.shape[-2:]
    planes = t.numel() // (h * w)
    out = torch.empty(t.shape[:-2] + (ho, wo), dtype=t.dtype, device=t.device)
    lib = N.lib()
    ws = torch.empty(lib.pbr_resize_workspace_bytes(planes, h, wo) // 4, dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        N.check(lib.pbr_resize_bilinear(t.data_ptr(), out.data_ptr(), planes, h, w, ho, wo, int(bool(antialias)),
                                        ws.data_ptr(), _stream_ptr(t.device)))
    return out


class _ResizeFn(torch.autograd.Function):
    """MaterialBase.resize for one map with its backward kernel (the transposed tap matrices: pbr_resize_bilinear_backward)."""

    @staticmethod
    def forward(ctx, texture, ho, wo, antialias):
        t = texture.detach().contiguous()
        ctx.geom = (tuple(t.shape), ho, wo, bool(antialias))
        return _resize_raw(t, ho, wo, antialias)

    @staticmethod
    def backward(ctx, grad_out):
        shape, ho, wo, antialias = ctx.geom
        h, w = shape[-2:]
        g = grad_out.to(torch.float32).contiguous()
        planes = g.numel() // (ho * wo)
        gin = torch.empty(shape, dtype=torch.float32, device=g.device)
        lib = N.lib()
        ws = torch.empty(max(1, lib.pbr_resize_backward_workspace_bytes(planes, h, w, ho, wo) // 4), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            N.check(lib.pbr_resize_bilinear_backward(g.data_ptr(), gin.data_ptr(), planes, h, w, ho, wo, int(antialias), ws.data_ptr(),
                                                     _stream_ptr(g.device)))
        return gin, None, None, None


def resize(texture: torch.Tensor, size, antialias: bool = True) -> torch.Tensor:
    """MaterialBase.resize for one map (base.py:490-504 -> torchvision resize of a float tensor):
    bilinear, align_corners=False, optional antialiasing.  `size` = (h, w), or an int that fixes the
    SMALLER edge and keeps the aspect ratio (torchvision semantics).  [..., H, W] float32 on device.
    Differentiable (its own backward kernel), as F.interpolate is upstream."""
    if not texture.is_cuda:
        raise RuntimeError("resize needs a tensor on a ROCm device; there is no CPU path")
    if texture.dtype != torch.float32:
        raise TypeError("resize supports float32 maps, got %s" % texture.dtype)
    h, w = texture.shape[-2:]
    if isinstance(size, (list, tuple)) and len(size) == 1:
        size = size[0]
    if isinstance(size, int):
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = size, int(size * long / short)
        size = (new_long, new_short) if w <= h else (new_short, new_long)
    ho, wo = int(size[0]), int(size[1])
    if _needs_grad(texture):
        return _ResizeFn.apply(texture, ho, wo, bool(antialias))
    return _resize_raw(texture.contiguous(), ho, wo, antialias)


def _decode_normal_raw(t: torch.Tensor, out: Optional[torch.Tensor] = None):
    C, H, W = t.shape
    if out is None:
        out = torch.empty((3, H, W), dtype=t.dtype, device=t.device)
    flag = torch.empty(1, dtype=torch.int32, device=t.device)
    with torch.cuda.device(t.device):
        N.check(N.lib().pbr_decode_normal(t.data_ptr(), out.data_ptr(), C, H * W, _DTYPES[t.dtype],
                                          flag.data_ptr(), _stream_ptr(t.device)))
    return out, flag


class _DecodeNormalFn(torch.autograd.Function):
    """A predicted normal map assigned to a material in a rendering loss (06_advanced.rst:73-107) must keep its
    gradient: forward = pbr_decode_normal, backward = pbr_decode_normal_backward (float32)."""

    @staticmethod
    def forward(ctx, normal_map):
        t = normal_map.detach().contiguous()
        out, flag = _decode_normal_raw(t)
        ctx.save_for_backward(normal_map)
        ctx.flag = flag
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (normal_map,) = ctx.saved_tensors
        t, g = normal_map.detach().contiguous(), grad_out.to(torch.float32).contiguous()
        gin = torch.empty_like(t)
        with torch.cuda.device(t.device):
            N.check(N.lib().pbr_decode_normal_backward(t.data_ptr(), g.data_ptr(), gin.data_ptr(), t.shape[0],
                                                       t.shape[1] * t.shape[2], ctx.flag.data_ptr(), _stream_ptr(t.device)))
        return gin


def decode_normal(normal_map: torch.Tensor) -> torch.Tensor:
    """MaterialBase._process_normal_map (base.py:191-242) on the device: (2|3,H,W) -> (3,H,W).  Differentiable for
    float32 maps (its own backward kernel)."""
    if normal_map.dim() != 3 or normal_map.shape[0] not in (2, 3):
        raise ValueError("Normal map must have 2 or 3 channels.")
    if normal_map.requires_grad and torch.is_grad_enabled():
        if not normal_map.is_cuda or normal_map.dtype != torch.float32:
            raise NotImplementedError("gradients through decode_normal need a float32 map on a ROCm device")
        return _DecodeNormalFn.apply(normal_map)
    return _decode_normal_raw(_device_tensor(normal_map, "decode_normal"))[0]
