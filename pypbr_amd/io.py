"""Folder -> material loader, the call in front of the BRDF in examples/example_brdf.py:8
(SURVEY.md 8f, row N1).  Mirrors pypbr.io.load_material_from_folder / select_material_class
(/root/reference/pypbr/io.py:27-186): same file-name conventions, same PIL mode handling,
same workflow selection and warnings.  Host-side only (PIL decode); no arithmetic."""
import os
import warnings
from typing import Dict, List, Optional, Type

from .materials import BasecolorMetallicMaterial, DiffuseSpecularMaterial, ImageMap, MaterialBase, _defer_images, _image_to_tensor

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


_SIXTEEN_BIT = ("I", "I;16", "I;16B", "I;16L", "I;16N")


def _converted(image, map_type: str):
    """The mode a map is read in (io.py:58-85): colour maps as RGB, 16-bit / float heights as they are, everything else as L."""
    if map_type in _RGB_MAPS:
        return image.convert("RGB")
    if map_type == "height" and image.mode in _DEEP_MODES:           # keep 16-bit / float heights (io.py:71-80)
        return image
    return image if image.mode == "L" else image.convert("L")


def _open(path: str, map_type: str):
    from PIL import Image
    return _converted(Image.open(path), map_type)


def _sample_bytes(image, map_type: str) -> int:
    """Size of the samples `_decoded(..., defer=True)` will keep of a just-opened image (its header is read, nothing decoded yet);
    0 when the map does not stay as samples (a float height)."""
    w, h = image.size
    if map_type in _RGB_MAPS:
        return 3 * w * h
    if map_type == "height" and image.mode in _DEEP_MODES:
        return 2 * w * h if image.mode in _SIXTEEN_BIT else 0
    return w * h


def _decoded(image, map_type: str, defer: bool, out=None):
    """Runs the decoder now, in the calling thread -- PIL opens lazily -- and the image becomes a tensor there too (base.py:143-164;
    with `defer` only its samples are taken over -- materials.py, `_ingest` -- and nothing is computed on the host at all; `out`: where
    the samples go, a slice of the loader's page-locked block)."""
    image = _converted(image, map_type)
    image.load()
    return ImageMap(_image_to_tensor(image, defer=defer, out=out))


def _sample_block(sizes):
    """One page-locked allocation for the samples of all maps of a material, each map 256-byte aligned: the decoders write into it, and
    functional.upload_packed sends it to the device as it is -- no staging copy in between, one block to free.  (Five arrays allocated
    by five worker threads and freed by the caller's cost 2-3 ms of munmap on a GPU box; tools/upload_phase_probe.py.)  -> per map a
    1-D uint8 slice, or None for maps that do not stay as samples / when page-locking is not to be had."""
    import torch
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += -(-n // 256) * 256
    if total == 0 or not torch.cuda.is_available():
        return [None] * len(sizes)
    try:
        block = torch.empty(total, dtype=torch.uint8, pin_memory=True)
    except RuntimeError:
        return [None] * len(sizes)
    return [block[o:o + n] if n else None for o, n in zip(offs, sizes)]


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
    # The image -> tensor step runs in the same worker (numpy and torch release the lock as well).  Same dict order as the
    # sequential loop of io.py:58-85.
    defer = _defer_images()                # the material is built on the CPU (the ctor's default), like upstream
    from PIL import Image
    opened = {map_type: Image.open(path) for map_type, path in found}           # headers only: nothing is decoded yet
    try:
        # The workflow is decided by WHICH maps are there (io.py:132-186), so it is decided before the decoders run: the maps the chosen
        # workflow does not take (the other workflow's albedo and specular / metallic map) are opened -- a file that is no image raises as
        # upstream -- but not decoded, and the samples of the others sit back to back in one block.
        loaded = dict(opened)
        cls = select_material_class(loaded, preferred_workflow)       # pops the map of the workflow not chosen
        if issubclass(cls, BasecolorMetallicMaterial):
            albedo_key = "basecolor"
            if "basecolor" not in loaded:
                warnings.warn("Basecolor map not found for metallic workflow. Looking for 'albedo' or 'basecolor' maps.")
        else:
            albedo_key = "diffuse"
            if "diffuse" not in loaded:
                warnings.warn("Diffuse map not found for specular workflow. Looking for 'diffuse' map.")
        wanted = [(t, im) for t, im in loaded.items() if t not in ("basecolor", "diffuse") or t == albedo_key]
        slots = _sample_block([_sample_bytes(im, t) for t, im in wanted]) if defer else [None] * len(wanted)
        jobs = [(im, t, defer, slot) for (t, im), slot in zip(wanted, slots)]
        if len(jobs) > 1:
            from concurrent.futures import ThreadPoolExecutor
            workers = min(len(jobs), len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4)
            with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
                images = list(pool.map(lambda job: _decoded(*job), jobs))
        else:
            images = [_decoded(*job) for job in jobs]
    finally:
        for im in opened.values():         # every file handle, decoded or not, whatever a decoder raised (close() is idempotent)
            im.close()
    loaded = {t: image for (t, _), image in zip(wanted, images)}
    albedo = loaded.get(albedo_key)
    kwargs = {k: v for k, v in loaded.items() if k not in ("basecolor", "diffuse")}
    kwargs["albedo"] = albedo
    return cls(**kwargs, albedo_is_srgb=is_srgb, specular_is_srgb=is_srgb)
