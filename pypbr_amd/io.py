"""Folder -> material loader, the call in front of the BRDF in examples/example_brdf.py:8
(SURVEY.md 8f, row N1).  Mirrors pypbr.io.load_material_from_folder / select_material_class
(/root/reference/pypbr/io.py:27-186): same file-name conventions, same PIL mode handling,
same workflow selection and warnings.  Host-side only (PIL decode); no arithmetic."""
import os
import warnings
from typing import Dict, List, Optional, Type

from .materials import BasecolorMetallicMaterial, DiffuseSpecularMaterial, MaterialBase

# map type -> accepted file stems, in lookup order (io.py:44-54)
DEFAULT_MAP_NAMES: Dict[str, List[str]] = {
    "basecolor": ["albedo", "basecolor"],
    "diffuse": ["diffuse"],
    "normal": ["normal", "normalmap"],
    "height": ["height", "displacement", "bump"],
    "roughness": ["roughness"],
    "metallic": ["metallic", "metalness"],
    "specular": ["specular"],
}
EXTENSIONS = ("png", "jpg", "jpeg", "tiff", "bmp", "exr")            # io.py:56
_RGB_MAPS = ("basecolor", "diffuse", "normal", "specular")
_DEEP_MODES = ("I", "I;16", "I;16B", "I;16L", "I;16N", "F")


def _find(folder: str, stems: List[str]) -> Optional[str]:
    for stem in stems:
        for ext in EXTENSIONS:
            path = os.path.join(folder, f"{stem}.{ext}")
            if os.path.isfile(path):
                return path
    return None


def _open(path: str, map_type: str):
    from PIL import Image
    image = Image.open(path)
    if map_type in _RGB_MAPS:
        return image.convert("RGB")
    if map_type == "height" and image.mode in _DEEP_MODES:           # keep 16-bit / float heights (io.py:71-80)
        return image
    return image if image.mode == "L" else image.convert("L")


def _decoded(image):
    """PIL opens lazily; `load()` runs the decoder now (in the calling thread)."""
    image.load()
    return image


def select_material_class(loaded_maps: Dict[str, object], preferred_workflow: Optional[str] = None) -> Type[MaterialBase]:
    """io.py:132-186.  Pops the map of the workflow that is not chosen when both are present."""
    has_metallic, has_specular = "metallic" in loaded_maps, "specular" in loaded_maps
    if has_metallic and has_specular:
        if preferred_workflow == "specular":
            warnings.warn("Both metallic and specular maps are present. Using specular workflow as preferred.")
            loaded_maps.pop("metallic", None)
            return DiffuseSpecularMaterial
        if preferred_workflow == "metallic":
            warnings.warn("Both metallic and specular maps are present. Using metallic workflow as preferred.")
        else:
            warnings.warn("Both metallic and specular maps are present. Specify preferred_workflow to choose. "
                          "Defaulting to metallic workflow.")
        loaded_maps.pop("specular", None)
        return BasecolorMetallicMaterial
    if has_metallic:
        return BasecolorMetallicMaterial
    if has_specular:
        return DiffuseSpecularMaterial
    if "basecolor" in loaded_maps:
        return BasecolorMetallicMaterial
    if "diffuse" in loaded_maps:
        return DiffuseSpecularMaterial
    warnings.warn("Neither metallic nor specular map found, and no albedo map found. "
                  "Defaulting to BasecolorMetallicMaterial.")
    return BasecolorMetallicMaterial


def load_material_from_folder(folder_path: str, map_names: Optional[Dict[str, List[str]]] = None,
                              preferred_workflow: Optional[str] = None, is_srgb: bool = True) -> MaterialBase:
    """io.py:27-129: scan `folder_path` for <stem>.<ext> files, pick the workflow, build the material
    (maps are decoded on the CPU like the reference; `.to("cuda")` moves them)."""
    names = DEFAULT_MAP_NAMES if map_names is None else map_names
    found = [(map_type, path) for map_type, path in ((t, _find(folder_path, stems)) for t, stems in names.items()) if path is not None]
    # The files are decoded side by side: PIL's decoders release the interpreter lock, and PNG inflate is what this function spends its
    # time on (five 1024^2 maps: 123 ms one after the other, the whole rest of examples/example_brdf.py 7 ms -- tools/example_bench.py).
    # Same images, same dict order as the sequential loop of io.py:58-85.
    if len(found) > 1:
        from concurrent.futures import ThreadPoolExecutor
        workers = min(len(found), len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4)
        with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
            images = list(pool.map(lambda tp: _decoded(_open(tp[1], tp[0])), found))
    else:
        images = [_decoded(_open(path, map_type)) for map_type, path in found]
    loaded = {map_type: image for (map_type, _), image in zip(found, images)}
    cls = select_material_class(loaded, preferred_workflow)
    if issubclass(cls, BasecolorMetallicMaterial):
        albedo = loaded.get("basecolor")
        if albedo is None:
            warnings.warn("Basecolor map not found for metallic workflow. Looking for 'albedo' or 'basecolor' maps.")
    else:
        albedo = loaded.get("diffuse")
        if albedo is None:
            warnings.warn("Diffuse map not found for specular workflow. Looking for 'diffuse' map.")
    kwargs = {k: v for k, v in loaded.items() if k not in ("basecolor", "diffuse")}
    kwargs["albedo"] = albedo
    return cls(**kwargs, albedo_is_srgb=is_srgb, specular_is_srgb=is_srgb)
