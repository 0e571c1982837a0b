"""Colour transfer functions with the reference's names (pypbr/utils/functions.py:31-66),
evaluated by libpbr_hip.so.  CPU tensors are staged through the device; with no ROCm
device present the call raises (no CPU arithmetic in this package)."""
import torch

from . import functional as _F
from .materials import NormalConvention, _through_device  # noqa: F401


def srgb_to_linear(texture: torch.Tensor) -> torch.Tensor:
    """sRGB -> linear, shape preserved: clamp to [0,1], x/12.92 below 0.04045,
    ((x+0.055)/1.055)**2.4 above, clamp."""
    return _through_device(texture, _F.srgb_to_linear)


def linear_to_srgb(texture: torch.Tensor) -> torch.Tensor:
    """linear -> sRGB, shape preserved: clamp to [0,1], 12.92x below 0.0031308,
    1.055 x**(1/2.4) - 0.055 above, clamp."""
    return _through_device(texture, _F.linear_to_srgb)
