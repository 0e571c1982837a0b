import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm device (MI355X); run with -m gpu on the GPU box")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests must never pass silently without a device."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no ROCm device in this container (run with gpurun)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN_DIR, "MANIFEST.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
        return cache[name]
    return load


RANDOM_SETS = ("rand64", "rand37x53", "rand1x1", "rand1x17", "rand5x1", "real48")


def parse_case(key, manifest):
    """out_<workflow>_<light>_<srgb|lin>[_<extra>] -> kwargs shared by oracle and product calls."""
    parts = key[4:].split("_")
    kind, lk, cs = parts[:3]
    extra = parts[3] if len(parts) > 3 else ""
    ltype, lvec, lsize = manifest["lights"][lk]
    view = manifest["view1"] if extra == "view1" else manifest["view0"]
    inten = manifest["intensity1"] if extra == "view1" else manifest["intensity0"]
    return dict(kind=kind, light_key=lk, extra=extra, light_type=ltype, light=lvec, light_size=lsize,
                view=view, intensity=inten, return_srgb=(cs == "srgb"),
                linear_maps=(extra == "linmaps"), no_normal=(extra == "nonormal"))


def render_keys(z):
    return sorted(k for k in z if k.startswith("out_") and not k.startswith(("out_conv_", "out_back_")))
