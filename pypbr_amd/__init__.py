"""pypbr_amd -- MI355X-native Cook-Torrance evaluation of PBR material maps.

A from-scratch gfx950 implementation of ONE hot path of giuvecchio/PyPBR:
`pypbr.models.CookTorranceBRDF` and the map conversions it pulls in, behind the
reference's own Python surface (pypbr_amd.models / .materials / .utils mirror
pypbr.models / .materials / .utils for that path) and a C ABI (include/pbr_hip.h).
"""
from . import blending, functional, io, materials, models, utils  # noqa: F401
from .materials import BasecolorMetallicMaterial, DiffuseSpecularMaterial, MaterialBase  # noqa: F401
from .models import BRDFModel, CookTorranceBRDF  # noqa: F401

__version__ = "0.1.0"
__all__ = ["blending", "functional", "io", "materials", "models", "utils", "MaterialBase", "BasecolorMetallicMaterial",
           "DiffuseSpecularMaterial", "BRDFModel", "CookTorranceBRDF"]
