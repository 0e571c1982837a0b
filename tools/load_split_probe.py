import os, sys, time, statistics, warnings
sys.path.insert(0, os.getcwd())
import torch
import pypbr_amd.materials as M
from pypbr_amd.io import load_material_from_folder
from pypbr_amd.models import CookTorranceBRDF
warnings.simplefilter("ignore")
V, L, I = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0])
brdf = CookTorranceBRDF(light_type="point")
for defer in (False, True, False, True):
    M.DEFER_IMAGE_DECODE = defer
    rows = {"load": [], "resize": [], "render": [], "total": []}
    for rep in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = load_material_from_folder("tests/golden/tiles", preferred_workflow="metallic")
        torch.cuda.synchronize(); t1 = time.perf_counter()
        m.resize((512, 512)); m.tile(2)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        c = brdf(m, V, L, I, 1.0)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if rep:
            rows["load"].append((t1 - t0) * 1e3); rows["resize"].append((t2 - t1) * 1e3); rows["render"].append((t3 - t2) * 1e3); rows["total"].append((t3 - t0) * 1e3)
    print("defer=%s" % defer, {k: round(statistics.median(v), 3) for k, v in rows.items()}, flush=True)
