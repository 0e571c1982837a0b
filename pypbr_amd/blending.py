"""Material blending in front of the BRDF: mirrors the surface of
pypbr.blending that examples/example_blend.py uses -- the functional API
(/root/reference/pypbr/blending/functional.py:25-286) and the callable classes
(/root/reference/pypbr/blending/blending.py:28-214).  Every blend returns `(material, mask)`
exactly like the reference.  All arithmetic (mask generation, per-map lerp, normal blend)
runs in libpbr_hip.so; CPU-resident materials are staged through the device."""
import contextlib
from abc import ABC, abstractmethod

import torch

from . import _native
from . import functional as F_
from .materials import MaterialBase, _compute_device


# ---------------------------------------------------------------- device ops (fp32 planar maps)
def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _check_f32(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s needs tensors on a ROCm device; there is no CPU path" % what)
    if t.dtype != torch.float32:
        raise TypeError("%s supports float32 maps, got %s" % (what, t.dtype))
    return t.contiguous()


def _blend_maps_raw(a, b, m, is_normal):
    out = torch.empty_like(a)
    with torch.cuda.device(a.device):
        _native.check(_native.lib().pbr_blend_maps(a.data_ptr(), b.data_ptr(), m.data_ptr(), out.data_ptr(), a.shape[0],
                                                   a.shape[1] * a.shape[2], int(bool(is_normal)), _stream(a)))
    return out


class _BlendMapsFn(torch.autograd.Function):
    """blend_maps with a gradient (pbr_blend_maps_backward): the reference's blend is plain torch arithmetic
    (functional.py:103-110, :119-145), so a rendering loss on a blended material reaches both materials and the mask."""

    @staticmethod
    def forward(ctx, a, b, m, is_normal):
        ctx.save_for_backward(a, b, m)
        ctx.is_normal = bool(is_normal)
        return _blend_maps_raw(a, b, m, is_normal)

    @staticmethod
    def backward(ctx, grad_out):
        a, b, m = ctx.saved_tensors
        g = grad_out.to(torch.float32).contiguous()
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        gm = torch.empty_like(m) if ctx.needs_input_grad[2] else None
        with torch.cuda.device(a.device):
            _native.check(_native.lib().pbr_blend_maps_backward(
                a.data_ptr(), b.data_ptr(), m.data_ptr(), g.data_ptr(), None if ga is None else ga.data_ptr(),
                None if gb is None else gb.data_ptr(), None if gm is None else gm.data_ptr(), a.shape[0], a.shape[1] * a.shape[2],
                int(ctx.is_normal), 0, _stream(a)))
        return ga, gb, gm, None


def blend_maps(map1: torch.Tensor, map2: torch.Tensor, mask: torch.Tensor, is_normal: bool = False) -> torch.Tensor:
    """mask * map1 + (1 - mask) * map2 for one (C,H,W) map and a (1,H,W) mask; `is_normal`:
    normalise both, blend, re-normalise (functional.py:119-145).  Differentiable (its own backward kernel)."""
    wants_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (map1, map2, mask))
    a, b, m = _check_f32(map1, "blend_maps"), _check_f32(map2, "blend_maps"), _check_f32(mask, "blend_maps")
    if a.dim() == 3 and b.dim() == 3 and a.shape[0] != b.shape[0] and 1 in (a.shape[0], b.shape[0]) and a.shape[1:] == b.shape[1:]:
        # `mask * map1 + (1 - mask) * map2` broadcasts a 1-channel map against a 3-channel one upstream (e.g. the
        # 3-channel metallic map to_basecolor_metallic_material returns, diffuse.py:147, against a 1-channel one)
        c = max(a.shape[0], b.shape[0])
        a, b = a.expand(c, -1, -1).contiguous(), b.expand(c, -1, -1).contiguous()
    if a.shape != b.shape or a.dim() != 3 or m.numel() != a.shape[1] * a.shape[2]:
        raise ValueError("maps %s / %s and mask %s do not match" % (tuple(a.shape), tuple(b.shape), tuple(m.shape)))
    if wants_grad:
        return _BlendMapsFn.apply(a, b, m.reshape(1, a.shape[1], a.shape[2]), bool(is_normal))
    return _blend_maps_raw(a, b, m, is_normal)


def _sigmoid_mask_raw(a, b, blend_width, shift):
    out = torch.empty_like(a)
    with torch.cuda.device(a.device):
        _native.check(_native.lib().pbr_blend_sigmoid_mask(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(),
                                                           float(shift), float(blend_width), _stream(a)))
    return out


class _SigmoidMaskFn(torch.autograd.Function):
    """The height / property blend mask with its backward kernel: upstream it is torch.sigmoid of plain arithmetic on the two
    property maps (functional.py:184-193), so a loss on the blended material reaches the height maps through the mask."""

    @staticmethod
    def forward(ctx, a, b, blend_width, shift):
        out = _sigmoid_mask_raw(a.detach(), b.detach(), blend_width, shift)
        ctx.save_for_backward(out)
        ctx.blend_width = float(blend_width)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (mask,) = ctx.saved_tensors
        g = grad_out.to(torch.float32).contiguous()
        g1 = torch.empty_like(mask) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(mask) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(mask.device):
            _native.check(_native.lib().pbr_blend_sigmoid_mask_backward(mask.data_ptr(), g.data_ptr(), None if g1 is None else g1.data_ptr(),
                                                                        None if g2 is None else g2.data_ptr(), mask.numel(), ctx.blend_width,
                                                                        _stream(mask)))
        return g1, g2, None, None


def sigmoid_mask(prop1: torch.Tensor, prop2: torch.Tensor, blend_width: float, shift: float = 0.0) -> torch.Tensor:
    """sigmoid((prop1 + shift - prop2) / (blend_width + 1e-6)) (functional.py:184-193, :227-236).  Differentiable w.r.t. both maps."""
    a, b = _check_f32(prop1, "sigmoid_mask"), _check_f32(prop2, "sigmoid_mask")
    if a.shape != b.shape:
        raise ValueError("property maps %s / %s differ in shape" % (tuple(a.shape), tuple(b.shape)))
    if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
        return _SigmoidMaskFn.apply(a, b, float(blend_width), float(shift))
    return _sigmoid_mask_raw(a, b, blend_width, shift)


def gradient_mask(height: int, width: int, direction: str, device) -> torch.Tensor:
    """linspace(0, 1) along x ('horizontal') or y ('vertical'), shape (1,H,W) (functional.py:262-281)."""
    if direction not in ("horizontal", "vertical"):
        raise ValueError("Direction must be 'horizontal' or 'vertical'.")
    out = torch.empty((1, height, width), dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        _native.check(_native.lib().pbr_blend_gradient_mask(out.data_ptr(), height, width, int(direction == "vertical"),
                                                            _stream(out)))
    return out


# ---------------------------------------------------------------- functional API (functional.py)
_LAZY = False


@contextlib.contextmanager
def lazy_blending(enabled: bool = True):
    """Inside this context every blend of this module (functions and the Blend* classes) is lazy, see
    `blend_with_mask(..., lazy=True)`:   with lazy_blending(): blended, mask = HeightBlend(0.1, -0.5)(m1, m2)"""
    global _LAZY
    previous, _LAZY = _LAZY, bool(enabled)
    try:
        yield
    finally:
        _LAZY = previous


def _blend_dicts(maps1: dict, maps2: dict, mask: torch.Tensor) -> dict:
    """The per-map loop of blend_with_mask (functional.py:96-112) on two name -> tensor dicts."""
    out = {}
    for name in list(maps1.keys()) + [k for k in maps2.keys() if k not in maps1]:
        m1, m2 = maps1.get(name), maps2.get(name)
        if m1 is None or m2 is None:
            out[name] = m2 if m1 is None else m1
        else:
            a = m1 if m1.is_cuda else m1.to(_compute_device(m1.device))
            out[name] = blend_maps(a, m2.to(a.device), mask.to(a.device), is_normal=(name == "normal"))
    return out


def _fusable(maps1: dict, maps2: dict, mask: torch.Tensor) -> bool:
    """Can CookTorranceBRDF hand both materials to pbr_cook_torrance_blend?  Both complete, same workflow, same
    (C,H,W) float32 shapes."""
    second = "metallic" if maps1.get("metallic") is not None else "specular"
    for name in ("albedo", "normal", "roughness", second):
        a, b = maps1.get(name), maps2.get(name)
        if a is None or b is None or a.dim() != 3 or a.shape != b.shape or a.dtype != torch.float32 or b.dtype != torch.float32:
            return False
    if second == "metallic" and maps2.get("metallic") is None:
        return False
    return mask.dtype == torch.float32 and tuple(mask.shape[-2:]) == tuple(maps1["albedo"].shape[-2:])


def blend_with_mask(material1: MaterialBase, material2: MaterialBase, mask: torch.Tensor, lazy: bool = False):
    """functional.py:64-116.  Returns (blended material of material1's class, mask as (1,H,W)).
    `lazy=True` (build extension): nothing is blended yet -- the result remembers both materials and the mask,
    `CookTorranceBRDF` evaluates it with the fused blend + render kernel (both materials read once, no blended
    copy written), and the first look at its maps (`material.albedo`, `_maps`, resize, save ...) blends them for real."""
    if mask.dim() == 2:
        mask = mask.unsqueeze(0)
    elif mask.dim() != 3 or mask.size(0) != 1:
        raise ValueError("Mask must have shape [1, H, W] or [H, W].")
    material1.materialize_tile(); material2.materialize_tile()      # blending reads the maps themselves
    blended = material1.__class__()
    blended.device = material1.device
    recorded = False
    if lazy or _LAZY:                    # only RECORDS the blend: the maps stay where they are, no device is needed yet
        for m in (material1, material2):
            m.materialize_blend()
            if m._has_pending():
                m._resident(keep=True)   # the fused kernel reads float maps and decoded normals
        if _fusable(material1._raw, material2._raw, mask):
            blended.__dict__["_store"] = dict(material1._raw)
            blended.__dict__["_lazy_blend"] = (dict(material2._raw), mask)
            recorded = True
    if not recorded:
        # both materials on the compute device (each in ONE upload when it still sits on the host); the blended maps stay there until
        # somebody looks at them (materials.py, module docstring)
        maps1, maps2 = material1._resident(keep=True), material2._resident(keep=True)
        for name, result in _blend_dicts(maps1, maps2, mask).items():
            blended._raw[name] = blended._settle(name, result)          # normals pass through _process_normal_map again, as upstream
    blended.albedo_is_srgb = material1.albedo_is_srgb
    return blended, mask


def _resized_like(prop2: torch.Tensor, prop1: torch.Tensor) -> torch.Tensor:
    if prop1.shape == prop2.shape:
        return prop2
    return F_.resize(prop2, tuple(prop1.shape[1:]), antialias=True)


def _handed_out(mask: torch.Tensor, material: MaterialBase) -> torch.Tensor:
    """The mask a blend returns lives where the material's maps are handed out (upstream: the materials' own device)."""
    home = torch.device(material.device)
    if mask.device == home:
        return mask
    return F_.to_host(mask, home) if home.type == "cpu" else mask.to(home)


def blend_on_height(material1: MaterialBase, material2: MaterialBase, blend_width: float = 0.1, shift: float = 0.0):
    """functional.py:148-196."""
    material1.materialize_tile(); material2.materialize_tile()
    if material1._raw.get("height") is None or material2._raw.get("height") is None:
        raise ValueError("Both materials must have height maps for height-based blending.")
    h1, h2 = material1._resident(keep=True)["height"], material2._resident(keep=True)["height"]
    mask = sigmoid_mask(h1, _resized_like(h2.to(h1.device), h1), blend_width, shift)
    blended, _ = blend_with_mask(material1, material2, mask)
    return blended, _handed_out(mask, material1)


def blend_on_properties(material1: MaterialBase, material2: MaterialBase, property_name: str = "metallic",
                        blend_width: float = 0.1):
    """functional.py:199-239."""
    material1.materialize_tile(); material2.materialize_tile()
    if material1._raw.get(property_name) is None or material2._raw.get(property_name) is None:
        raise ValueError(f"Both materials must have '{property_name}' maps for property-based blending.")
    p1, p2 = material1._resident(keep=True)[property_name], material2._resident(keep=True)[property_name]
    mask = sigmoid_mask(p1, _resized_like(p2.to(p1.device), p1), blend_width, 0.0)
    blended, _ = blend_with_mask(material1, material2, mask)
    return blended, _handed_out(mask, material1)


def blend_with_gradient(material1: MaterialBase, material2: MaterialBase, direction: str = "horizontal"):
    """functional.py:242-286."""
    size = material1.size
    if size is None:
        raise ValueError("Materials must have at least one map to determine size.")
    if direction not in ("horizontal", "vertical"):
        raise ValueError("Direction must be 'horizontal' or 'vertical'.")
    mask = gradient_mask(size[0], size[1], direction, _compute_device(material1.device))
    blended, _ = blend_with_mask(material1, material2, mask)
    return blended, _handed_out(mask, material1)


def blend_materials(material1: MaterialBase, material2: MaterialBase, method: str = "mask", **kwargs):
    """functional.py:25-61."""
    if method == "mask":
        mask = kwargs.get("mask")
        if mask is None:
            raise ValueError("Mask must be provided for 'mask' blending method.")
        return blend_with_mask(material1, material2, mask)
    if method == "height":
        return blend_on_height(material1, material2, kwargs.get("blend_width", 0.1))
    if method == "properties":
        return blend_on_properties(material1, material2, kwargs.get("property_name", "metallic"), kwargs.get("blend_width", 0.1))
    if method == "gradient":
        return blend_with_gradient(material1, material2, kwargs.get("direction", "horizontal"))
    raise ValueError(f"Unknown blending method: {method}")


import sys as _sys
functional = _sys.modules[__name__]      # `pypbr.blending.functional` is this module's function set


# ---------------------------------------------------------------- class API (blending.py)
class BlendMethod(ABC):
    @abstractmethod
    def __call__(self, material1: MaterialBase, material2: MaterialBase):
        ...


class MaskBlend(BlendMethod):
    def __init__(self, mask: torch.Tensor):
        self.mask = mask

    def __call__(self, material1, material2):
        return blend_with_mask(material1, material2, self.mask)


class HeightBlend(BlendMethod):
    def __init__(self, blend_width: float = 0.1, shift: float = 0.0):
        self.blend_width, self.shift = blend_width, shift

    def __call__(self, material1, material2):
        return blend_on_height(material1, material2, self.blend_width, self.shift)


class PropertyBlend(BlendMethod):
    def __init__(self, property_name: str = "metallic", blend_width: float = 0.1):
        self.property_name, self.blend_width = property_name, blend_width

    def __call__(self, material1, material2):
        return blend_on_properties(material1, material2, self.property_name, self.blend_width)


class GradientBlend(BlendMethod):
    def __init__(self, direction: str = "horizontal"):
        self.direction = direction

    def __call__(self, material1, material2):
        return blend_with_gradient(material1, material2, self.direction)


class BlendFactory:
    _methods = {"mask": MaskBlend, "height": HeightBlend, "properties": PropertyBlend, "gradient": GradientBlend}

    @staticmethod
    def get_blend_method(method_name: str, **kwargs) -> BlendMethod:
        cls = BlendFactory._methods.get(method_name.lower())
        if cls is None:
            raise ValueError(f"Unknown blending method: {method_name}")
        return cls(**kwargs)
