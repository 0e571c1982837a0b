"""Material containers with the surface the Cook-Torrance path reads.

Mirrors the part of pypbr.materials the hot path touches (SURVEY.md 8a rows H13-H16,
8b "Material surface"): the name -> tensor dict `_maps`, attribute access to maps,
`device`, `albedo_is_srgb` / `specular_is_srgb`, `linear_albedo` / `linear_specular`,
`to()`, `to_linear()` / `to_srgb()`, the two workflow conversions, `tile()`.
Reference: /root/reference/pypbr/materials/{base,metallic,diffuse}.py.

Every computation (normal decode, colour transfer, workflow conversion) runs in
libpbr_hip.so on a ROCm device; with no device present those calls raise -- there is no
ATen/CPU arithmetic in this package.

Where the maps live.  A material's `device` is where its maps are HANDED OUT (the reference's
default: the CPU, e.g. examples/example_brdf.py).  Internally a map stays where it was last
produced: a CPU material's first whole-material operation (`resize`, `to_linear`, a workflow
conversion, a blend) uploads all its maps in ONE host-to-device copy (functional.upload_packed),
runs on the device and leaves the results THERE; later operations and `CookTorranceBRDF` read them
in place.  Only a caller that looks at a map (`material.albedo`, `_maps`, `as_dict()`, `clone()`,
`linear_albedo` ...) brings it home, once.  Round 3 staged every map through the device and back
per operation: load -> resize -> tile -> render was 5 x (H2D + launch + D2H) for the resize, the
same again for the normal decode, and one more upload of everything for the render
("What's missing" #3); now it is one upload and the download of the image.

Differences from the reference, all supersets:
  * maps may be any floating torch.Tensor on any device (the reference only files
    CPU float32 `torch.FloatTensor`s into `_maps`) and may carry a
    leading batch dimension [B,C,H,W] (F2).  A ROCm-resident tensor assigned to a material
    that still sits on its default device (cpu) pulls the MATERIAL onto the tensor's
    device instead of being copied to the host -- the rendering-loss loop of
    docs/source/tutorials/06_advanced.rst builds a material from predicted device tensors
    every step, and a silent device -> host -> device round trip of every map cost 9.8 ms
    per step at 2048^2 against 0.2 ms of kernels (tools/flow_trace.py);
  * `to_diffuse_specular_material(specular_is_srgb=...)` exposes the flag the upstream
    conversion forgets to set (F6); the default keeps upstream behaviour.
"""
import copy
import enum
import weakref
from typing import Optional, Tuple

import numpy as np
import torch

from . import _native
from . import functional as F_


class NormalConvention(enum.Enum):
    """pypbr/utils/enums.py:12-14 (carried on materials; the BRDF never reads it)."""
    OPENGL = "opengl"
    DIRECTX = "directx"


def _is_image(value) -> bool:
    try:
        from PIL import Image
        return isinstance(value, Image.Image)
    except ImportError:  # pragma: no cover
        return False


def _is_map_value(value) -> bool:
    if value is None or isinstance(value, (np.ndarray, ImageMap)) or _is_image(value):
        return True
    return isinstance(value, torch.Tensor) and value.is_floating_point()


def _image_to_tensor(image, defer: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PIL image -> (C,H,W) float32 in [0,1] (base.py:143-164: 16-bit modes / 65535,
    mode F as is, RGBA -> RGB, everything else uint8 / 255).
    `defer=True`: 8- and 16-bit images come back as their own SAMPLES -- a (C,H,W) uint8 / uint16 view of PIL's (H,W,C) array, no
    transposing, no arithmetic -- and the division waits for `_samples_to_float` (host) or pbr_unpack_image (device).  `out`: a 1-D
    uint8 tensor of exactly the samples' size (the loader passes a slice of its page-locked block): the samples are put there."""
    if image.mode in ("I", "I;16", "I;16B", "I;16L", "I;16N"):
        arr = np.array(image, dtype=np.uint16)
        if defer:
            return _samples_tensor(arr, out)
        return (torch.from_numpy(arr.astype(np.float32)) / 65535.0).unsqueeze(0)
    if image.mode == "F":
        return torch.from_numpy(np.array(image, dtype=np.float32)).unsqueeze(0)
    if image.mode == "RGBA":
        image = image.convert("RGB")
    arr = np.array(image)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    if defer and arr.dtype == np.uint8:
        return _samples_tensor(arr, out)
    t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
    return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t.to(torch.float32)


def _samples_tensor(arr: np.ndarray, out: Optional[torch.Tensor]) -> torch.Tensor:
    """(H,W) or (H,W,C) samples -> the (C,H,W) VIEW of them as a tensor; copied into `out` first when it has their size."""
    if out is not None and out.dtype == torch.uint8 and out.dim() == 1 and out.numel() == arr.nbytes and out.is_contiguous():
        np.copyto(out.numpy().view(arr.dtype).reshape(arr.shape), arr)
        t = out.view(torch.uint16 if arr.dtype == np.uint16 else torch.uint8).view(arr.shape)
    else:
        t = torch.from_numpy(arr)
    return t.unsqueeze(0) if t.dim() == 2 else t.permute(2, 0, 1)


def _samples_to_float(t: torch.Tensor) -> torch.Tensor:
    """The arithmetic `_image_to_tensor(image, defer=True)` left out, on the host, exactly as base.py:143-164 would have done it."""
    if t.dtype == torch.uint8:
        return t.contiguous().to(torch.float32).div(255)
    return t.to(torch.float32) / 65535.0


class ImageMap:
    """A PIL image the loader's worker threads already turned into a tensor (io.py): float (C,H,W), or -- conversion deferred --
    the image's samples (`_image_to_tensor(image, defer=True)`).  Filed into a material like the image itself."""
    __slots__ = ("tensor",)

    def __init__(self, tensor: torch.Tensor):
        self.tensor = tensor


# Whether maps that come from image files stay as SAMPLES (and a normal map undecoded) until the first operation that needs floats:
# None = when a ROCm device is there to do it on arrival; True / False force it (tests of the host side run with True).
DEFER_IMAGE_DECODE: Optional[bool] = None


def _defer_images() -> bool:
    return torch.cuda.is_available() if DEFER_IMAGE_DECODE is None else bool(DEFER_IMAGE_DECODE)


def _through_device(t: torch.Tensor, fn):
    """Runs a libpbr_hip op on `t`; CPU-resident maps are staged through the device."""
    if t.is_cuda:
        return fn(t)
    _native.require_device()
    res = fn(t.to("cuda"))
    if isinstance(res, tuple):
        return tuple(F_.to_host(r, t.device) for r in res)
    return F_.to_host(res, t.device)


def _normalised(device) -> torch.device:
    """torch.device("cuda") names the current device: spelled out, so that it compares equal to the tensors that live there."""
    device = device if isinstance(device, torch.device) else torch.device(device)
    if device.type == "cuda" and device.index is None and torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return device


def _compute_device(home) -> torch.device:
    """Where a material handed out on `home` is computed: `home` itself when it is a ROCm device, else the current one."""
    home = torch.device(home)
    if home.type == "cuda":
        return home
    _native.require_device()
    return torch.device("cuda", torch.cuda.current_device())


# "Already signed?" (base.py:212: `normal_map.min() < 0`) is a property of the tensor's VALUES, and for a signed map the
# reference returns the very tensor it was given.  The rendering-loss loop wraps the same decoded normal map in a new
# material every step; with functional.set_caching(decode_verdicts=True) the first decode of a tensor leaves its verdict
# in a 4-byte device flag, the second assignment of the same unchanged tensor (same object, same version counter) reads
# that flag once, and from then on a signed map is handed back as it is -- no kernel, no copy, the reference's object
# identity -- while an encoded map is decoded afresh.  Off by default: the version counter does not see every edit
# (functional.CACHING), and the reference looks at the values every time.
_DECODE_VERDICTS = {}          # id(tensor) -> [weakref, version, device flag, host verdict or None]
_DECODE_VERDICTS_MAX = 64


def _decode_remembering(t: torch.Tensor) -> torch.Tensor:
    ver = F_.version_of(t) if F_.CACHING["decode_verdicts"] else None
    if ver is None:                 # the default (functional.set_caching): decide afresh, on the device, every assignment
        return F_.decode_normal(t)
    hit = _DECODE_VERDICTS.get(id(t))
    if hit is not None and hit[0]() is t and hit[1] == ver:
        if hit[3] is None:
            hit[3] = bool(hit[2].item())       # the decode that wrote it ran an assignment ago: no wait in practice
        if hit[3]:
            return t
        return F_.decode_normal(t)
    out, flag = F_._decode_normal_raw(t)
    for k in [k for k, e in _DECODE_VERDICTS.items() if e[0]() is None]:
        del _DECODE_VERDICTS[k]
    if len(_DECODE_VERDICTS) >= _DECODE_VERDICTS_MAX:
        _DECODE_VERDICTS.clear()
    _DECODE_VERDICTS[id(t)] = [weakref.ref(t), ver, flag, None]
    return out


class MaterialBase:
    """Dynamic bag of texture maps (base.py:34-120)."""

    def __init__(self, albedo=None, albedo_is_srgb: bool = True, normal=None, roughness=None,
                 normal_convention: NormalConvention = NormalConvention.OPENGL,
                 device: torch.device = torch.device("cpu"), **kwargs):
        self.device = _normalised(device)
        self.normal_convention = normal_convention
        self._maps = {}
        self.albedo_is_srgb = albedo_is_srgb
        for name, value in (("albedo", albedo), ("normal", normal), ("roughness", roughness)):
            if value is not None:
                setattr(self, name, value)
        for name, value in kwargs.items():
            setattr(self, name, value)

    # -- attribute protocol: map-like values are filed in _maps, the rest are plain attributes
    def __setattr__(self, name, value):
        if name in ("albedo_is_srgb", "_maps", "device", "normal_convention", "specular_is_srgb"):
            object.__setattr__(self, name, value)
        elif _is_map_value(value):
            self.materialize_blend()                       # an assignment to a lazily blended material lands on blended maps
            if self.__dict__.get("_lazy_tile", (1, 1)) != (1, 1):
                self.materialize_tile()                    # ... and one to a material with a recorded tile(n) on REPEATED maps: the
                                                           # new map is not repeated again on read (base.py:524-537 tiles what is there)
            self._raw[name] = self._ingest(name, value)
        else:
            object.__setattr__(self, name, value)

    def __getattr__(self, name):
        d = self.__dict__
        pending = d.get("_lazy_blend")
        if name in d.get("_store", {}) or (pending is not None and name in pending[0]):   # a map only material 2 has
            return self._maps[name]       # anything but the BRDF sees blended, repeated maps at home (CookTorranceBRDF reads _store itself)
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    # `_maps` is the reference's name -> tensor dict: the PUBLIC view.  Reading it first resolves a pending lazy blend
    # (blending.blend_with_mask(..., lazy=True)) and a pending tile(n, lazy=True), decodes a normal map whose decode was
    # deferred, and brings every map that still sits on the compute device home to `self.device` -- whoever looks at the maps
    # sees what the reference's dict would hold.  The package's own code goes through `_raw` (the same dict, as it stands) and `_resident()`.
    @property
    def _maps(self):
        d = self.__dict__
        if d.get("_lazy_blend") is not None:
            self.materialize_blend()
        if d.get("_lazy_tile", (1, 1)) != (1, 1):
            self.materialize_tile()
        self._bring_home()
        return d.setdefault("_store", {})

    @_maps.setter
    def _maps(self, value):
        self.__dict__["_store"] = value

    @property
    def _raw(self) -> dict:
        """The name -> tensor dict as it stands: maps wherever they were last produced, a deferred normal still encoded."""
        return self.__dict__.setdefault("_store", {})

    def _is_away(self) -> bool:
        """A map of this material sits on another device than the one it is handed out on."""
        return any(t is not None and t.device != self.device for t in self._raw.values())

    def _has_pending(self) -> bool:
        """Some map is not yet what the reference's dict would hold: still an image's samples, or a normal map still undecoded."""
        return bool(self.__dict__.get("_raw_normal")) or any(F_.is_encoded(t) for t in self._raw.values())

    def _samples_on_host(self):
        """Maps that are still an image's samples become float32 where they are (the host): base.py:143-164's arithmetic."""
        store = self._raw
        for name, t in store.items():
            if F_.is_encoded(t):
                store[name] = _samples_to_float(t)

    def _bring_home(self):
        store, home = self._raw, self.device
        self._samples_on_host()
        if self.__dict__.get("_raw_normal"):
            self.__dict__["_raw_normal"] = False
            store["normal"] = self._process_normal_map(store["normal"])
        for name, t in store.items():
            if t is not None and t.device != home:
                store[name] = F_.to_host(t, home) if home.type == "cpu" else t.to(home)

    def _resident(self, keep: bool = True) -> dict:
        """name -> float tensor on the compute device for every map that is present.  Maps that are still on the host travel in ONE
        copy (functional.upload_packed): maps that are still an image's samples go as samples and become float32 on arrival, the
        normal map decoded in the same pass; a float normal map whose decode was deferred goes first, its decoded form written
        behind the others -- either way the material ends up as one dense block of planes.  `keep=True`: the device tensors become
        the material's maps (the operation that asked is about to replace them anyway); `keep=False` (CookTorranceBRDF): host maps
        stay the material's maps -- the reference re-reads them every call -- except maps nobody can have seen yet (samples, a
        deferred normal), whose float form on the device is the map from now on."""
        self.materialize_blend()
        d, store = self.__dict__, self._raw
        compute = _compute_device(self.device)
        names = [k for k, t in store.items() if t is not None]
        away = [k for k in names if store[k].device != compute]
        out = {k: store[k] for k in names}
        pending = bool(d.get("_raw_normal"))
        unseen = [k for k in away if F_.is_encoded(store[k])]
        if away:
            # only HOST maps ride the packed upload (its staging is a host memcpy); a map that requires grad takes the differentiable
            # copy, a map on ANOTHER GPU (material.to("cuda:1") from cuda:0) the device-to-device copy, ordered on its stream
            alone = [k for k in away if store[k].requires_grad or store[k].device.type != "cpu"]
            for k in alone:
                out[k] = store[k].to(compute)
            plain = [k for k in away if k not in alone]
            in_flight = pending and "normal" in plain
            as_samples = in_flight and F_.is_encoded(store["normal"])
            if in_flight and not as_samples:                     # raw float normal first: everything behind it is one dense block
                plain = ["normal"] + [k for k in plain if k != "normal"]
            if plain:
                views, block = F_.upload_packed([store[k] for k in plain], compute, tail_planes=3 if in_flight and not as_samples else 0,
                                                encoded_normal=plain.index("normal") if as_samples else None)
                out.update(zip(plain, views))
                if as_samples:
                    pending = False
                elif in_flight and block is not None:
                    out["normal"] = F_._decode_normal_raw(out["normal"], out=block[-3:])[0]
                    pending = False
        if pending:
            out["normal"] = F_.decode_normal(out["normal"])
        if d.get("_raw_normal"):
            d["_raw_normal"] = False
            store["normal"] = out["normal"]
        for k in (away if keep else unseen):
            store[k] = out[k]
        return out

    def materialize_blend(self):
        """Carries out a pending lazy blend: the maps become real blended tensors (blend.hip kernels)."""
        pending = self.__dict__.get("_lazy_blend")
        if pending is not None:
            from .blending import _blend_dicts
            self.__dict__["_lazy_blend"] = None
            other, mask = pending
            store = self.__dict__["_store"]
            for name, result in _blend_dicts(dict(store), other, mask).items():
                store[name] = self._settle(name, result)      # normals pass through _process_normal_map, as upstream
        return self

    def _settle(self, name, t):
        """A map PRODUCED by this package (a blend, a conversion) becomes the material's: it stays on the device it was computed on,
        a normal map passes through _process_normal_map as on any assignment upstream."""
        if t is None:
            return None
        if name == "normal":
            return self._process_normal_map(t)
        return t

    def _ingest(self, name, value):
        if value is None:
            return None
        fresh = False                                # nobody else holds the tensor: its decode may wait for the first device operation
        if isinstance(value, torch.Tensor):
            if value.is_cuda and self.device.type == "cpu":
                self.device = value.device           # a device tensor pulls the material onto its device (module docstring)
            t = value.to(self.device)
        elif isinstance(value, np.ndarray):
            t = torch.from_numpy(value).float().to(self.device)
        elif _is_image(value) or isinstance(value, ImageMap):
            # base.py:143-164 converts and base.py:191-242 decodes at assignment.  A map that comes out of an image file lives on the
            # host and nobody can look at it but through this material: it is stored as loaded -- the image's own samples, a normal map
            # undecoded -- and becomes float32 on the device together with the first operation that uploads the material (a quarter of
            # the bytes on the way, no conversion on the host), or on the host when somebody reads it first.
            fresh = self.device.type == "cpu" and _defer_images()
            t = value.tensor if isinstance(value, ImageMap) else _image_to_tensor(value, defer=fresh)
            if F_.is_encoded(t) and not fresh:
                t = _samples_to_float(t)
            fresh = fresh and t.dim() == 3
            if not F_.is_encoded(t):
                t = t.to(self.device)
        else:  # pragma: no cover  (guarded by _is_map_value)
            raise TypeError(f"Unsupported image type: {type(value)}. Supported types are PIL.Image.Image, "
                            "np.ndarray, and torch.FloatTensor.")
        if name == "normal":
            self.__dict__["_raw_normal"] = False
            if fresh:
                if t.shape[0] not in (2, 3):
                    raise ValueError("Normal map must have 2 or 3 channels.")
                self.__dict__["_raw_normal"] = True
                return t
            return self._process_normal_map(t)
        return t

    @staticmethod
    def _process_normal_map(normal_map: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """base.py:191-242 on the device; a batch is decoded map by map (the "already
        signed?" test of :212 is per map)."""
        if normal_map is None:
            return None
        if normal_map.shape[-3] not in (2, 3):
            raise ValueError("Normal map must have 2 or 3 channels.")
        if normal_map.dim() == 4:
            return torch.stack([_through_device(n, F_.decode_normal) for n in normal_map], dim=0)
        if (normal_map.is_cuda and not normal_map.requires_grad and normal_map.dim() == 3 and normal_map.shape[0] == 3
                and normal_map.is_contiguous() and normal_map.dtype in (torch.float32, torch.float16)):
            return _decode_remembering(normal_map)
        return _through_device(normal_map, F_.decode_normal)

    # -- device management (base.py:245-259)
    def to(self, device):
        """base.py:245-259.  Moving to a ROCm device takes ONE host-to-device copy of all maps into one allocation
        (functional.upload_packed): a launch streams every plane of the material at once, and planes that share an allocation stay
        close together in the address space (1-6 % faster and steadier than maps scattered over the heap)."""
        self.device = _normalised(device)
        self.materialize_blend()
        if self.device.type == "cuda":
            self._resident(keep=True)
        else:
            self._bring_home()
        return self

    def _bring_home_normal(self):
        self.__dict__["_raw_normal"] = False
        self._raw["normal"] = self._process_normal_map(self._raw["normal"])

    # -- properties
    @property
    def linear_albedo(self):
        """base.py:262-277: a fresh linear copy when the stored map is sRGB."""
        albedo = self._maps.get("albedo")
        if albedo is None:
            return None
        return _through_device(albedo, F_.srgb_to_linear) if self.albedo_is_srgb else albedo

    @property
    def size(self) -> Optional[Tuple[int, int]]:
        """(height, width) of the first map that is present (base.py:293-307)."""
        ny, nx = self.lazy_tile
        for t in self.__dict__.get("_store", {}).values():       # a pending blend does not change the size
            if t is not None:
                return (t.shape[-2] * ny, t.shape[-1] * nx)
        return None

    def as_dict(self):
        self.materialize_tile()
        return dict(self._maps)

    def cache_on_device(self, enable: bool = True):
        """Opt-in: CookTorranceBRDF keeps the device copy of this CPU-resident material between calls (one upload for a
        loop over an unchanged material instead of one per call: 9.6 ms for a 4096^2 material).  The copy is recognised
        as current by tensor identity and version counter, which do not see `.data` edits or edits of a numpy array that
        shares a map's memory -- hence opt-in (functional.set_caching(device_maps=True) turns it on for every material).
        All cached copies together are bounded (models.DEVICE_CACHE_CAP, least recently used evicted first)."""
        object.__setattr__(self, "_cache_on_device", bool(enable))
        if not enable:
            self.drop_device_cache()
        return self

    def drop_device_cache(self):
        """Frees the device copy CookTorranceBRDF keeps of a CPU-resident material between calls."""
        self.__dict__.pop("_device_cache", None)
        return self

    def _convert_in_place(self, name, flag, fn):
        """One map through a colour transfer, on the device, the result left there (module docstring)."""
        if self._raw.get(name) is not None:
            self._raw[name] = fn(self._resident(keep=True)[name])

    # -- colour space, in place, returning self (base.py:754-778)
    def to_linear(self):
        self.materialize_blend()
        if self._raw.get("albedo") is not None and self.albedo_is_srgb:
            self._convert_in_place("albedo", "albedo_is_srgb", F_.srgb_to_linear)
            self.albedo_is_srgb = False
        return self

    def to_srgb(self):
        self.materialize_blend()
        if self._raw.get("albedo") is not None and not self.albedo_is_srgb:
            self._convert_in_place("albedo", "albedo_is_srgb", F_.linear_to_srgb)
            self.albedo_is_srgb = True
        return self

    # -- the two calls around the BRDF in examples/example_brdf.py:11
    def resize(self, size, antialias: bool = True):
        """Resize every map (base.py:490-504): bilinear, antialiased by default; in place, returns self.  All float32 (C,H,W) maps
        of one size -- a material's maps as a rule -- go through ONE pbr_resize_bilinear launch over all their planes; the results
        stay on the device (module docstring)."""
        self.materialize_tile()
        maps = self._resident(keep=True)
        store = self._raw
        groups = {}
        for name, t in maps.items():
            if t.dim() == 3 and t.dtype == torch.float32 and not (t.requires_grad and torch.is_grad_enabled()):
                groups.setdefault(tuple(t.shape[-2:]), []).append(name)
            else:
                store[name] = F_.resize(t, size, antialias=antialias)
        for names in groups.values():
            names.sort(key=lambda k: maps[k].data_ptr())
            ts = [maps[k] for k in names]
            out = F_.resize(_as_block(ts), size, antialias=antialias)
            p = 0
            for k, t in zip(names, ts):
                store[k] = out[p:p + t.shape[0]]
                p += t.shape[0]
        return self

    # -- pure indexing (base.py:524-537); no arithmetic involved
    def tile(self, num_tiles: int, lazy: bool = False):
        """Repeat every map num_tiles x num_tiles (base.py:524-537).  `lazy=True` (build extension) only records
        the repeat: the maps stay as they are, `CookTorranceBRDF` hands the count to the kernel, which evaluates every texel at
        all its repeats -- each texel then leaves HBM once instead of num_tiles^2 times and no copy is made; whoever else looks at
        the maps sees the repeated ones.  A material whose maps are waiting on the compute device (module docstring), or live there,
        records the repeat likewise."""
        if num_tiles <= 0:                 # upstream: map.repeat(1, 0, 0) -> empty maps; a negative count is torch's RuntimeError
            self.materialize_tile()
            self._samples_on_host()
            store = self._raw
            for name, t in store.items():
                if t is not None:
                    store[name] = t.repeat(*((1,) * (t.dim() - 2) + (num_tiles, num_tiles)))
            return self
        on_device = [t.is_cuda for t in self._raw.values() if t is not None]
        if lazy or self._is_away() or self._has_pending() or (on_device and all(on_device)):
            # maps nobody has seen yet, or maps that live on the compute device: the repeat is recorded and the kernels evaluate (and
            # differentiate) every texel at all its repeats -- the example's `material.resize(512).tile(2)` inside a rendering loss
            # (06_advanced.rst:73-107) then moves a quarter of the map bytes and no repeated copy exists; whoever LOOKS at the maps
            # (`_maps`, attribute access, `as_dict`, `clone`, another map operation) gets the repeated ones, as upstream
            ny, nx = self.lazy_tile
            object.__setattr__(self, "_lazy_tile", (ny * num_tiles, nx * num_tiles))
            return self
        self.materialize_tile()
        store = self._maps
        for name, t in store.items():
            if t is not None:
                reps = (1,) * (t.dim() - 2) + (num_tiles, num_tiles)
                store[name] = t.repeat(*reps)
        return self

    @property
    def lazy_tile(self) -> Tuple[int, int]:
        """Pending (ny, nx) repeat recorded by tile(n, lazy=True); (1, 1) if none."""
        return self.__dict__.get("_lazy_tile", (1, 1))

    def materialize_tile(self):
        """Turns a pending lazy repeat into real maps (anything but the BRDF needs them), where the maps are."""
        ny, nx = self.lazy_tile
        if (ny, nx) != (1, 1):
            object.__setattr__(self, "_lazy_tile", (1, 1))
            self.materialize_blend()
            self._samples_on_host()
            if self.__dict__.get("_raw_normal"):
                self._bring_home_normal()
            store = self._raw
            for name, t in store.items():
                if t is not None:
                    store[name] = t.repeat(*((1,) * (t.dim() - 2) + (ny, nx)))
        return self

    def clone(self):
        """Deep copy: tensors cloned, flags copied (base.py:880-912)."""
        self.materialize_blend()          # a pending lazy blend is carried out first: the copy must not blend again
        self._samples_on_host()
        if self.__dict__.get("_raw_normal"):
            self._bring_home_normal()
        new = copy.copy(self)
        new.__dict__.pop("_device_cache", None)
        new.__dict__.pop("_plan_cache", None)
        new.__dict__.pop("_plan_seen", None)
        new.__dict__["_lazy_blend"] = None
        object.__setattr__(new, "_maps", {k: (None if v is None else v.clone()) for k, v in self._raw.items()})
        return new

    @classmethod
    def _assemble(cls, device, maps: dict, **flags):
        """A material of this class around maps this package produced (a conversion): they stay where they were computed."""
        new = cls(device=device, **flags)
        for name, t in maps.items():
            if t is not None:
                new._raw[name] = new._settle(name, t)
        return new

    def __repr__(self):
        body = ", ".join(f"{k}={None if v is None else tuple(v.shape)}" for k, v in self._raw.items())
        return f"{type(self).__name__}({body})"


def _as_block(ts):
    """[P,H,W] over the planes of (C_i,H,W) tensors: a view when they sit back to back in one allocation (functional.upload_packed,
    a previous resize), else a copy."""
    first = ts[0]
    plane = first.shape[-2] * first.shape[-1]
    adjacent = all(t.is_contiguous() for t in ts) and all(
        a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and a.data_ptr() + a.numel() * a.element_size() == b.data_ptr()
        for a, b in zip(ts, ts[1:]))
    if adjacent:
        return first.as_strided((sum(t.shape[0] for t in ts), first.shape[-2], first.shape[-1]), (plane, first.shape[-1], 1))
    return torch.cat(ts, dim=0)


class BasecolorMetallicMaterial(MaterialBase):
    """Metallic workflow (metallic.py:22-69)."""

    def __init__(self, albedo=None, albedo_is_srgb: bool = True, normal=None, roughness=None, metallic=None,
                 **kwargs):
        super().__init__(albedo=albedo, albedo_is_srgb=albedo_is_srgb, normal=normal, roughness=roughness, **kwargs)
        if metallic is not None:
            self.metallic = metallic

    @property
    def basecolor(self):
        return self.albedo

    @basecolor.setter
    def basecolor(self, value):
        self.albedo = value

    def to_diffuse_specular_material(self, albedo_is_srgb: bool = False, specular_is_srgb: bool = True):
        """metallic.py:71-120.  The new material shares normal/roughness, holds LINEAR
        diffuse and specular maps and -- as upstream -- is flagged specular_is_srgb=True
        unless told otherwise."""
        self.materialize_tile()
        self.materialize_blend()
        if self._raw.get("albedo") is None or self._raw.get("metallic") is None:
            raise ValueError("Both albedo and metallic maps are required for conversion.")
        maps = self._resident(keep=True)
        albedo, metallic = maps["albedo"], maps["metallic"]
        if metallic.shape[-2:] != albedo.shape[-2:]:       # metallic.py:93-96: TF.resize(metallic, albedo.shape[1:], antialias=True)
            metallic = F_.resize(metallic, tuple(albedo.shape[-2:]), antialias=True)
        diffuse, specular = F_.metallic_to_diffuse_specular(albedo, metallic, albedo_is_srgb=self.albedo_is_srgb)
        return DiffuseSpecularMaterial._assemble(
            self.device, dict(albedo=diffuse, normal=maps.get("normal"), roughness=maps.get("roughness"), specular=specular),
            albedo_is_srgb=albedo_is_srgb, specular_is_srgb=specular_is_srgb)


class DiffuseSpecularMaterial(MaterialBase):
    """Specular workflow (diffuse.py:23-91)."""

    def __init__(self, albedo=None, albedo_is_srgb: bool = True, normal=None, roughness=None, specular=None,
                 specular_is_srgb: bool = True, **kwargs):
        super().__init__(albedo=albedo, albedo_is_srgb=albedo_is_srgb, normal=normal, roughness=roughness, **kwargs)
        self.specular_is_srgb = specular_is_srgb
        if specular is not None:
            self.specular = specular

    @property
    def diffuse(self):
        return self.albedo

    @diffuse.setter
    def diffuse(self, value):
        self.albedo = value

    @property
    def linear_specular(self):
        """diffuse.py:76-91."""
        specular = self._maps.get("specular")
        if specular is None:
            return None
        return _through_device(specular, F_.srgb_to_linear) if self.specular_is_srgb else specular

    def to_basecolor_metallic_material(self, albedo_is_srgb: bool = False):
        """diffuse.py:93-158: RAW specular (not linear_specular), 3-channel metallic."""
        self.materialize_tile()
        self.materialize_blend()
        if self._raw.get("albedo") is None or self._raw.get("specular") is None:
            raise ValueError("Both albedo (diffuse) and specular maps are required for conversion.")
        maps = self._resident(keep=True)
        albedo, specular = maps["albedo"], maps["specular"]
        if specular.shape[-2:] != albedo.shape[-2:]:       # diffuse.py:117-118: TF.resize(self.specular, diffuse.shape[1:], antialias=True)
            specular = F_.resize(specular, tuple(albedo.shape[-2:]), antialias=True)
        base, metallic = F_.diffuse_specular_to_basecolor_metallic(albedo, specular, albedo_is_srgb=self.albedo_is_srgb)
        return BasecolorMetallicMaterial._assemble(
            self.device, dict(albedo=base, normal=maps.get("normal"), roughness=maps.get("roughness"), metallic=metallic),
            albedo_is_srgb=albedo_is_srgb)

    def to_linear(self):
        """diffuse.py:160-175."""
        super().to_linear()
        if self._raw.get("specular") is not None and self.specular_is_srgb:
            self._convert_in_place("specular", "specular_is_srgb", F_.srgb_to_linear)
            self.specular_is_srgb = False
        return self

    def to_srgb(self):
        """diffuse.py:177-190."""
        super().to_srgb()
        if self._raw.get("specular") is not None and not self.specular_is_srgb:
            self._convert_in_place("specular", "specular_is_srgb", F_.linear_to_srgb)
            self.specular_is_srgb = True
        return self
