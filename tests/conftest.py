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

# ---- the suite's parity table: one row per input set, filled by test_gpu_parity.parity_report, closed by
# tests/test_gpu_zz_parity_table.py.  Thresholds: the largest roughness at which a value of the set is more than 1e-5 from the
# reference's fp32 output, MEASURED on the GPU and committed (tests/golden/parity_thresholds.json); criterion (i) is asserted above it,
# and the measured value of a later run must not exceed it.
PARITY_SETS = {}
PARITY_DEFAULT_THRESHOLD = 0.185          # sets without a recorded threshold (new sets): round 4's constant


def _recorded_thresholds():
    try:
        with open(os.path.join(GOLDEN_DIR, "parity_thresholds.json")) as f:
            return json.load(f).get("sets", {})
    except (OSError, ValueError):
        return {}


PARITY_RECORDED = _recorded_thresholds()


# The GATE is the recorded measurement plus a real margin: a toolchain bump that moves one ill-conditioned value across 1e-5
# at a slightly higher roughness is not a parity regression.  The measured value is still printed and written (parity_table.json), and
# profiles/README.md says how to re-record it.
PARITY_GATE_MARGIN = 0.01


def parity_threshold(set_name):
    if set_name in PARITY_RECORDED:
        return float(PARITY_RECORDED[set_name]) + PARITY_GATE_MARGIN
    return PARITY_DEFAULT_THRESHOLD


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm device (MI355X); run with -m gpu on the GPU box")
    # Multi-process GPU tests start their ranks from a fork server that is launched HERE, before anything in this
    # process has touched the GPU: a process that has initialised HIP must not exec another program, and the ranks
    # must not inherit an initialised HIP runtime through fork() either.
    try:
        import multiprocessing
        from multiprocessing import forkserver
        multiprocessing.get_context("forkserver")
        forkserver.ensure_running()
    except Exception:  # pragma: no cover  (the CPU suite does not need it)
        pass


# ---- order by PURPOSE, not by file name (VERDICT r5 #1c): `pytest -x -m gpu` on a foreign box must reach every golden / oracle parity
# test before any test that starts subprocesses, runs bench.py or depends on the host in any way.  Files not listed sort in the middle.
GPU_ORDER_FIRST = (
    "test_gpu_00_baseline_configs",     # all five BASELINE.json configs against the oracles
    "test_gpu_parity",                  # forward hot path against the golden vectors (outputs of the real reference)
    "test_gpu_edge_golden", "test_gpu_edge_values", "test_gpu_reference_style", "test_gpu_full_shapes",
    "test_gpu_torch_ops", "test_gpu_map_op_gradients", "test_gpu_blending", "test_gpu_blend_specular", "test_gpu_example_blend",
    "test_gpu_image_decode", "test_gpu_backward", "test_gpu_blend_backward", "test_gpu_loss_step",
    "test_gpu_host_paths", "test_gpu_device_params", "test_gpu_round4", "test_gpu_round5", "test_gpu_round6",
)
GPU_ORDER_LAST = (
    "test_gpu_zz_parity_table",         # closes the parity table: behind every parity_report caller, ahead of the harness tests
    "test_gpu_distributed",             # starts ranks (forkserver)
    "test_gpu_bench_ranks",             # runs bench.py in subprocesses, CPU legs included
)


def _order_key(item):
    stem = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if stem in GPU_ORDER_FIRST:
        return (0, GPU_ORDER_FIRST.index(stem))
    if stem in GPU_ORDER_LAST:
        return (2, GPU_ORDER_LAST.index(stem))
    return (1, 0)


def pytest_collection_modifyitems(config, items):
    """Purpose order (stable within a file), and: `-m gpu` tests must never pass silently without a device."""
    items.sort(key=_order_key)
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


def oracle_render(z, case, prefix="in_", dtype=None):
    """The ATen-level oracle (oracle/torch_oracle.py, pinned bit-equal to the reference) on one golden case;
    dtype=torch.float64 gives the float64 evaluation of the reference's formulas on the same inputs."""
    import torch
    import torch_oracle as O

    def T(x):
        t = torch.from_numpy(x)
        return t if dtype is None else t.to(dtype)
    kind = case["kind"]
    a, r = T(z[prefix + "albedo"]), T(z[prefix + "roughness"])
    n = None if case["no_normal"] else T(z[prefix + "normal"])
    fdt = torch.float32 if dtype is None else dtype
    kw = dict(view=torch.tensor(case["view"], dtype=fdt), light=torch.tensor(case["light"], dtype=fdt),
              intensity=torch.tensor(case["intensity"], dtype=fdt), light_type=case["light_type"],
              light_size=case["light_size"], return_srgb=case["return_srgb"])
    if kind == "converted":
        return O.cook_torrance_converted(a, n, r, T(z[prefix + "metallic"]), quirk_specular_srgb=(case["extra"] == "quirk"), **kw)
    lin = case["linear_maps"]
    return O.cook_torrance(a, n, r, T(z[prefix + "metallic"]) if kind == "metallic" else None,
                           T(z[prefix + "specular"]) if kind == "specular" else None,
                           albedo_is_srgb=not lin, specular_is_srgb=not lin, **kw)
