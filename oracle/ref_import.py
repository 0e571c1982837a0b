"""Import the upstream reference (giuvecchio/PyPBR) from /root/reference.

TEST INFRASTRUCTURE, DEVELOPMENT CONTAINER ONLY.  /root/reference does not exist
on the GPU box; nothing under tests/ -m gpu, bench.py or smoke() may call this.
It is used by oracle/gen_golden.py (fixture generation) and by the `refpin`
tests that skip themselves when /root/reference is absent.

The reference cannot be imported as-is here: torchvision is not
installed and pypbr/_version.py is generated at build time.  Both are
import-time-only for the Cook-Torrance path, so two in-memory stand-ins are
registered in sys.modules (nothing is written into the reference tree):

* torchvision.transforms.functional with `to_tensor` (uint8 HWC -> float CHW/255,
  what torchvision does for 8-bit PIL images) and `resize` (antialiased bilinear
  F.interpolate, what torchvision does for float tensors) so that the PNG
  fixtures load and examples/example_brdf.py's resize/tile path runs;
* pypbr._version with version = "0".
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pypbr"))


class _VersionFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname == "pypbr._version":
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        module.version = "0"
        module.__version__ = "0"


def _install_torchvision_stub():
    if "torchvision" in sys.modules:
        return
    import numpy as np
    import torch
    import torch.nn.functional as F

    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    fn = types.ModuleType("torchvision.transforms.functional")

    def to_tensor(pic):
        arr = np.asarray(pic)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
        if t.dtype == torch.uint8:
            return t.to(torch.float32).div(255)
        return t.to(torch.float32)

    def resize(img, size, interpolation=None, max_size=None, antialias=True):
        # torchvision semantics for float tensors: an int (or 1-tuple) size fixes the SMALLER edge and
        # keeps the aspect ratio; bilinear, align_corners=False, antialias as given
        if isinstance(size, (list, tuple)) and len(size) == 1:
            size = size[0]
        if isinstance(size, int):
            h, w = img.shape[-2:]
            short, long = (w, h) if w <= h else (h, w)
            new_short, new_long = size, int(size * long / short)
            size = (new_long, new_short) if w <= h else (new_short, new_long)
        return F.interpolate(img[None], size=tuple(size), mode="bilinear",
                             align_corners=False, antialias=bool(antialias))[0]

    fn.to_tensor = to_tensor
    fn.resize = resize
    tr.functional = fn
    tv.transforms = tr
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tr
    sys.modules["torchvision.transforms.functional"] = fn


def import_reference():
    """Returns the reference's `pypbr` package (imported from /root/reference)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at " + REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    _install_torchvision_stub()
    if not any(isinstance(f, _VersionFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _VersionFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import pypbr  # noqa: the reference package
    if not pypbr.__file__.startswith(REFERENCE_ROOT):
        raise RuntimeError("`pypbr` resolved to %s, not the reference" % pypbr.__file__)
    return pypbr
