#!/usr/bin/env python3
"""load -> render of a 4096^2 material from PNG files (the tiles maps of tests/golden scaled up by PIL and saved to a scratch folder): what the
loader's sample path (materials.DEFER_IMAGE_DECODE: samples kept, decoded into one page-locked block, unpacked on the device) is worth
where it matters -- 4096^2 maps are 160 MB of samples and 604 MB as float32.  Medians over `--repeat` runs after the first."""
import argparse
import os
import statistics
import sys
import tempfile
import time
import warnings

import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pypbr_amd.materials as M  # noqa: E402
from pypbr_amd.io import load_material_from_folder  # noqa: E402
from pypbr_amd.models import CookTorranceBRDF  # noqa: E402

warnings.simplefilter("ignore")
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--repeat", type=int, default=5)
args = ap.parse_args()
src = os.path.join(ROOT, "tests", "golden", "tiles")
V, L, I = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0])
brdf = CookTorranceBRDF(light_type="point")
with tempfile.TemporaryDirectory() as folder:
    for name in ("basecolor", "normal", "roughness", "metallic", "height"):
        im = Image.open(os.path.join(src, name + ".png"))
        im.resize((args.size, args.size), Image.BILINEAR).save(os.path.join(folder, name + ".png"), compress_level=1)
    print("files:", {f: os.path.getsize(os.path.join(folder, f)) >> 20 for f in sorted(os.listdir(folder))}, "MiB", flush=True)
    images = {}
    for defer in (True, False, True, False):
        M.DEFER_IMAGE_DECODE = defer
        rows = {"load": [], "render + download": [], "total": []}
        for rep in range(args.repeat + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m = load_material_from_folder(folder, preferred_workflow="metallic")
            t1 = time.perf_counter()
            c = brdf(m, V, L, I, 1.0)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if rep:
                rows["load"].append((t1 - t0) * 1e3); rows["render + download"].append((t2 - t1) * 1e3); rows["total"].append((t2 - t0) * 1e3)
        images[defer] = c
        print("samples kept until the device needs them" if defer else "float maps made at assignment (base.py:143-242)  ",
              {k: round(statistics.median(v), 1) for k, v in rows.items()}, "ms", flush=True)
    print("same image:", torch.equal(images[True], images[False]))
