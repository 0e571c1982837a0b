#!/usr/bin/env python3
"""The reference's two example scripts, statement by statement, on the CPU-resident materials the loader returns:

    /root/reference/examples/example_brdf.py:8-26     load -> resize((512, 512)) -> tile(2) -> CookTorranceBRDF("point") -> image
    /root/reference/examples/example_blend.py:14-32   two loads -> HeightBlend(0.1, -0.5) -> resize -> tile(2) -> render -> image

Wall time per statement (device drained after each, so a statement is charged what it enqueued), the transfers it caused
(functional.upload_packed / to_host calls, bytes), and beside it the same statements through the CPU oracle (oracle/: the ATen
restatement of the reference; torchvision's resize of a float tensor is F.interpolate(..., antialias=True)) on this host's usable cores.
`python bench.py --example brdf|blend` runs this; `--size N` renders at another resize target; `--repeat K` repeats (first run = cold).
One JSON line per example on stdout."""
import argparse
import json
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
VIEW, LIGHT, INTEN = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0])


def usable_cores():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


class Stages:
    def __init__(self, sync):
        self.sync, self.rows, self.t0 = sync, [], None

    def __call__(self, name, fn):
        self.sync()
        t0 = time.perf_counter()
        out = fn()
        self.sync()
        self.rows.append((name, (time.perf_counter() - t0) * 1e3))
        return out


def run_hip(example, size):
    from pypbr_amd import blending as B, functional as F
    from pypbr_amd.io import load_material_from_folder
    from pypbr_amd.models import CookTorranceBRDF
    moved = {"h2d": 0, "h2d_bytes": 0, "d2h": 0, "d2h_bytes": 0}
    up, down = F.upload_packed, F.to_host

    def counted_up(ts, *a, **k):
        moved["h2d"] += 1
        moved["h2d_bytes"] += sum(t.numel() * t.element_size() for t in ts)
        return up(ts, *a, **k)

    def counted_down(t, *a, **k):
        if t.is_cuda:
            moved["d2h"] += 1
            moved["d2h_bytes"] += t.numel() * t.element_size()
        return down(t, *a, **k)
    F.upload_packed, F.to_host = counted_up, counted_down
    try:
        st = Stages(torch.cuda.synchronize)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            material = st("load (PNG inflate on the host, samples into one page-locked block)", lambda: load_material_from_folder(os.path.join(GOLDEN, "tiles"), preferred_workflow="metallic"))
            if example == "blend":
                material2 = st("load 2 (PNG inflate on the host)", lambda: load_material_from_folder(os.path.join(GOLDEN, "rocks"), preferred_workflow="metallic"))
                material, mask = st("HeightBlend (2 uploads, mask, blend; mask handed out)", lambda: B.HeightBlend(blend_width=0.1, shift=-0.5)(material, material2))
        st("resize (samples uploaded, unpacked, normal decoded; one resize launch)" if example == "brdf" else "resize (one launch)", lambda: material.resize((size, size)))
        st("tile(2) (recorded)", lambda: material.tile(2))
        brdf = CookTorranceBRDF(light_type="point")
        color = st("render + download of the image", lambda: brdf(material, VIEW, LIGHT, INTEN, 1.0))
        assert color.device.type == "cpu"
        st("render again, image left on the device", lambda: CookTorranceBRDF(light_type="point", override_device=torch.device("cuda"))(material, VIEW, LIGHT, INTEN, 1.0))
    finally:
        F.upload_packed, F.to_host = up, down
    return color, st.rows, moved


def run_oracle(example, size, threads):
    """The same statements through oracle/ (test infrastructure; here it is the thing TIMED as the CPU baseline, never the product)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import blend_oracle as BO
    import torch_oracle as O
    import torch.nn.functional as TF
    from pypbr_amd.io import DEFAULT_MAP_NAMES, _find, _open            # host-side file lookup + PIL mode handling (no arithmetic)
    from pypbr_amd.materials import _image_to_tensor
    torch.set_num_threads(threads)
    st = Stages(lambda: None)

    def load(folder):
        maps = {}
        for kind, stems in DEFAULT_MAP_NAMES.items():
            path = _find(os.path.join(GOLDEN, folder), stems)
            if path is not None and kind not in ("diffuse", "specular"):
                maps["albedo" if kind == "basecolor" else kind] = _image_to_tensor(_open(path, kind))
        maps["normal"] = O.decode_normal(maps["normal"])
        return maps
    maps = st("load + normal decode", lambda: load("tiles"))
    if example == "blend":
        maps2 = st("load 2 + normal decode", lambda: load("rocks"))

        def blend():
            mask = BO.sigmoid_mask(maps["height"], maps2["height"], 0.1, -0.5)
            return BO.blend_materials(maps, maps2, mask)          # the blended normal is decoded again inside, as on assignment upstream
        maps = st("HeightBlend", blend)
    maps = st("resize", lambda: {k: TF.interpolate(v[None], (size, size), mode="bilinear", align_corners=False, antialias=True)[0] for k, v in maps.items()})
    maps = st("tile(2)", lambda: {k: v.repeat(1, 2, 2) for k, v in maps.items()})
    color = st("render", lambda: O.cook_torrance(maps["albedo"], maps["normal"], maps["roughness"], maps["metallic"], None, view=VIEW, light=LIGHT,
                                                 intensity=INTEN, light_type="point", light_size=1.0))
    return color, st.rows


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--example", choices=["brdf", "blend", "both"], default="both")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--repeat", type=int, default=6)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args(argv)
    assert torch.cuda.is_available(), "needs a ROCm device"
    cores = usable_cores()
    for example in (["brdf", "blend"] if args.example == "both" else [args.example]):
        import statistics
        first_total, later = None, []
        for _ in range(args.repeat):                       # a script runs once: results of earlier repeats are dropped like a finished script's
            color = None
            color, rows, moved = run_hip(example, args.size)
            if first_total is None:
                first_total = sum(ms for name, ms in rows[:-1])
            else:
                later.append(rows)
        later = later or [rows]
        # per statement the MEDIAN over the repeats after the first (the first run pays imports, HIP start-up and code-object loading and is
        # reported by itself; single repeats jitter -- a busy host's PNG decode, a page-locking call -- by tens of milliseconds)
        rows = [(name, statistics.median(r[i][1] for r in later)) for i, (name, _) in enumerate(later[0])]
        rec = {"example": example, "statements": "examples/example_%s.py" % example, "resize": args.size, "image": list(color.shape), "repeats": args.repeat,
               "hip_ms": {name: round(ms, 3) for name, ms in rows}, "hip_total_ms": round(sum(ms for name, ms in rows[:-1]), 3),
               "hip_ms_each_repeat": {name: [round(r[i][1], 3) for r in later] for i, (name, _) in enumerate(later[0])},
               "hip_total_ms_first_run": round(first_total, 3), "transfers": moved}
        if not args.no_cpu:
            ref, crow = run_oracle(example, args.size, cores)
            rec["cpu_oracle_ms"] = {name: round(ms, 3) for name, ms in crow}
            rec["cpu_oracle_total_ms"] = round(sum(ms for _, ms in crow), 3)
            rec["cpu_threads"] = cores
            rec["max_abs_diff_vs_oracle"] = float((color - ref).abs().max())
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
