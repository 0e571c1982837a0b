#!/usr/bin/env python3
"""bench.py -- Mpixels/s of fused Cook-Torrance evaluation on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1|2|3|4|5] [--size S] [--batch B] [--no-cpu-baseline]
                  [--layout arena|separate] [--settle L] [--cpu-budget SECONDS]

`--config` selects the workload by SURVEY.md 8d's numbering (= BASELINE.json configs[config - 1]); a "step" is one pass of
the hot path over this rank's batch = ONE launch of the fused kernel through the C ABI, inputs already resident in HBM:
  1  BASELINE.json configs[0] on the GPU: ONE 256x256 material through CookTorranceBRDF.__call__ -- device-resident (eager, and captured
     into a HIP graph) and CPU-resident (upload, evaluate, image back) -- with the host / launch / kernel split (`latency_us`) and the
     CPU oracle's time for the same call beside it (run_config1);
  2 (default, the configuration the metric is quoted on)  one 4096x4096 BasecolorMetallicMaterial PER GPU (albedo sRGB,
     decoded normal, roughness, metallic; fp32 planar), point light, sRGB out.  WEAK scaling: every rank owns its own
     material;
  3  B=64 2048x2048 materials, directional light, sRGB -> linear and the metallic -> diffuse/specular conversion fused
     in front of the specular-workflow evaluation (upstream's F6 flag setting timed, the other one beside it);
  4  B=512 1024x1024 materials, point light;
  5  B=32 4096x4096 materials, 16 point lights on a ring accumulated in-kernel, fp16 maps, fp32 accumulate and result.
Configs 3-5 are STRONG-scaled: the job is the whole batch, rank r generates and owns the materials of
pypbr_amd.distributed.partition(B, H, N, r) -- nobody ever holds the whole batch -- and evaluates them through
pypbr_amd.distributed.cook_torrance_sharded(owned=...), which broadcasts the light / view block from rank 0 over RCCL
(backend "nccl") before the plan is built.  The data path has no collective; results stay sharded.

`python bench.py --gpus N` with N > 1 starts the N ranks itself: the parent process (which never touches a GPU)
runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
bench.py ...` as a child and exits with its code.  Started under torch.distributed.run already (RANK in the
environment), the script is a rank.

Timing protocol.  Two timed regions, both K steps bracketed by barrier + synchronize, MAX over ranks:
  cold    W warm-up launches from whatever state the GPU was in, then K timed steps -- the literal
          "W warm-up, K steps" protocol -> `ms_per_step_cold`, `value_cold`;
  steady  `--settle` further untimed launches (default: 300 for config 2, else what fills ~40 ms) so that power
          management has settled the clocks (tools/transient_probe.py), then K timed steps -> `ms_per_step`, `value`:
          the sustained rate.
  slope   2K more steps, timed the same way: (wall(2K) - wall(K)) / K is the per-step time without the fixed cost of entering and
          leaving a region (~80 us: 4 us per step of a 20-step region) -> `ms_per_step_slope`, beside the literal `ms_per_step`.
`warmup` echoes the argument; `launches_before_value` says how many launches really preceded the region `value` is taken from.
Rank 0 prints ONE JSON line (contract in the task statement).  Extra objects:
  roofline     -- achieved algorithmic HBM GB/s of the kernel (bytes per pixel x this rank's pixels per launch / average
                  launch duration from HIP events on the launch stream over the STEADY region -- the K steps `value` is quoted on;
                  rounds 2-4 averaged the cold region in, which alone moved `frac` by 4 %: profiles/history/r05_driver_line_ab.md) vs 8 TB/s;
                  for N > 1 the slowest rank's.  Config 2 also runs the BARE access pattern of the kernel (tools/boxcal.hip: the same
                  planes, the same bytes per lane, no arithmetic) on the same buffers right behind the steady region:
                  `box_pattern_us`, `kernel_over_box_pattern` -- a slow line with a ratio near 1 is a slow box, not a slow binary
  roofline_valu-- config 5 only: the launch is VALU-bound; vector instructions issued per second against the chip's issue rate
  per_rank     -- kernel_us of every rank (steady region), its shard, and the latency of the light-block broadcast
  parity       -- bands of the timed output of this run against the float64 C oracle and the ATen restatement
  cpu_baseline -- the ATen-level restatement of the reference's CPU path (oracle/torch_oracle.py, kind "port": bit-equal to the
                  reference in the dev container) timed on this host's cores, rank 0, N=1 only: 1 thread and all host
                  cores at 256^2 / 1024^2 / 2048^2, CPU model stated.

`python bench.py --example brdf|blend|both [--size 512]`: the reference's example scripts statement by statement on CPU-resident
materials (tools/example_bench.py): time per statement, transfers, the CPU oracle's time for the same statements.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_BUFFER_SETS = 3              # config 2 rotates map sets: 704 MiB per step never re-hits the 256 MiB Infinity Cache anyway
VIEW = [0.0, 0.0, 1.0]
RING16 = [[math.cos(2 * math.pi * i / 16), math.sin(2 * math.pi * i / 16), 1.0] for i in range(16)]

# SURVEY.md 8d.  light/intensity are lists of lights; `flags` go to the evaluation as they are.
CONFIGS = {
    2: dict(batch=1, size=4096, scaling="weak", dtype=torch.float32, light_type="point", light=[[0.1, 0.1, 1.0]],
            intensity=[[1.0, 1.0, 1.0]], light_size=1.0, flags={}, steps=500, warmup=50,
            name="Batch=1 {S}x{S} BasecolorMetallicMaterial per GPU, point light, fused HIP kernel, fp32 maps, sRGB in/out "
                 "(BASELINE.json configs[1])"),
    3: dict(batch=64, size=2048, scaling="strong", dtype=torch.float32, light_type="directional", light=[[0.3, -0.2, 1.0]],
            intensity=[[1.0, 1.0, 1.0]], light_size=None, flags=dict(convert_to_diffuse_specular=True, specular_is_srgb=True),
            steps=40, warmup=5,
            name="Batch={B} {S}x{S} materials, directional light, sRGB->linear + metallic->diffuse-specular conversion fused "
                 "(specular_is_srgb=True: upstream's flag, SURVEY.md F6), fp32 (BASELINE.json configs[2])"),
    4: dict(batch=512, size=1024, scaling="strong", dtype=torch.float32, light_type="point", light=[[0.1, 0.1, 1.0]],
            intensity=[[1.0, 1.0, 1.0]], light_size=1.0, flags={}, steps=40, warmup=5,
            name="Batch={B} {S}x{S} materials sharded over the ranks, point light, fp32 (BASELINE.json configs[3])"),
    5: dict(batch=32, size=4096, scaling="strong", dtype=torch.float16, light_type="point", light=RING16,
            intensity=[[1.0 / 16] * 3] * 16, light_size=1.0, flags={}, steps=20, warmup=3,
            name="Batch={B} {S}x{S} materials, 16 point lights accumulated in-kernel, fp16 maps, fp32 accumulate and result "
                 "(BASELINE.json configs[4])"),
}
# config 5 is VALU-bound.  Vector instructions per (pixel, light) of the batch-inner kernel are COUNTED, not assumed: rocprofv3
# SQ_INSTS_VALU x 64 / (pixels x lights) of the newest committed per-kernel evidence set (recorded_valu below) -- the algorithmic
# figure its VALU roofline uses.  Peak = one vector instruction per SIMD per 4 cycles for a wave of 64 -- the right order for THIS kernel, whose
# stream is packed fp32 (measured issue rates, tools/valu_occupancy.hip / profiles/r06_valu_issue_rates.txt: packed 4.3-4.8 cycles per wave
# instruction, transcendentals 8.2; plain fp32 issues every 2.5 cycles once two waves feed a SIMD, which a packed kernel does not use):
VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 4 * 64          # lane-instructions per ns: 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles x 64 lanes


def recorded_valu():
    """(vector instructions per (pixel, light) of the 16-light batch-inner kernel, the committed file they come from): the newest
    profiles/rNN_kernels.json that holds the `fwd_16_lights` case (4 x 4096^2 fp16 maps, 16 point lights) with its SQ pass."""
    import glob
    found = glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_kernels.json")) + glob.glob(os.path.join(ROOT, "profiles", "history", "r[0-9][0-9]_kernels.json"))
    for path in sorted(found, key=os.path.basename, reverse=True):
        try:
            with open(path) as f:
                recs = json.load(f)
        except (OSError, ValueError):
            continue
        for r in recs if isinstance(recs, list) else []:
            if "kernel" not in r:                   # the collection's stamp (tools/stamp_profiles.py)
                continue
            n = (r.get("sq") or {}).get("valu_wave_instructions")
            if n and str(r.get("case", "")).startswith("fwd_16_lights") and "batch_kernel" in r.get("kernel", ""):
                return n * 64 / (4 * 4096 * 4096 * 16), os.path.relpath(path, ROOT), (r.get("sq") or {}).get("shader_clock_GHz")
    return None, None, None


def synth_material(size, device, seed, dtype=torch.float32, rows=None):
    """SURVEY.md 8d config 2 recipe: U[0,1) albedo/metallic, roughness remapped to [0.05,1],
    normal = normalize([U(-.5,.5), U(-.5,.5), 1]) stored decoded.  `rows=(y0, y1)`: that band of the same material."""
    g = torch.Generator(device=device).manual_seed(seed)
    H = W = size
    albedo = torch.rand(3, H, W, device=device, generator=g)
    nxy = torch.rand(2, H, W, device=device, generator=g) - 0.5
    normal = torch.cat([nxy, torch.ones(1, H, W, device=device)], 0)
    normal = normal / normal.norm(dim=0, keepdim=True)
    rough = torch.rand(1, H, W, device=device, generator=g) * 0.95 + 0.05
    metal = torch.rand(1, H, W, device=device, generator=g)
    maps = [t.to(dtype) for t in (albedo, normal, rough, metal)]
    if rows is not None:
        maps = [t[:, rows[0]:rows[1]].contiguous() for t in maps]
    return maps


def synth_shard(cfg, shard, device, seed0):
    """This rank's materials [batch_start, batch_stop) x rows [row_start, row_stop): material b is drawn from seed0 + b,
    whoever owns it, so the job is the same however many ranks split it."""
    nb, h, S = shard.batch_stop - shard.batch_start, shard.row_stop - shard.row_start, cfg["size"]
    out = [torch.empty((nb, c, h, S), dtype=cfg["dtype"], device=device) for c in (3, 3, 1, 1)]
    band = None if h == S else (shard.row_start, shard.row_stop)
    for i, b in enumerate(range(shard.batch_start, shard.batch_stop)):
        for dst, src in zip(out, synth_material(S, device, seed0 + b, cfg["dtype"], band)):
            dst[i].copy_(src)
    return out


def _oracle_path():
    p = os.path.join(ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)


def _oracle_eval(cfg, maps, y_offset=0, H_total=None, dtype=None):
    """The ATen restatement (oracle/torch_oracle.py) of this configuration on CPU maps [C,h,W] (fp32, or float64 twins)."""
    import torch_oracle as O
    dt = dtype or torch.float32
    a, n, r, m = [t.to(dt) for t in maps]
    kw = dict(view=torch.tensor(VIEW, dtype=dt), light_type=cfg["light_type"], light_size=cfg["light_size"],
              y_offset=y_offset, H_total=H_total)
    lights, inten = torch.tensor(cfg["light"], dtype=dt), torch.tensor(cfg["intensity"], dtype=dt)
    if cfg["flags"].get("convert_to_diffuse_specular"):
        return O.cook_torrance_converted(a, n, r, m, quirk_specular_srgb=cfg["flags"].get("specular_is_srgb", True),
                                         light=lights[0], intensity=inten[0], **kw)
    if len(cfg["light"]) > 1:
        return O.cook_torrance_multi(a, n, r, m, None, lights=lights, intensities=inten, **kw)
    return O.cook_torrance(a, n, r, m, None, light=lights[0], intensity=inten[0], **kw)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """The cores this process may actually run on: its affinity mask, capped by the cgroup's CPU quota (a GPU box grants a share of
    the host; os.cpu_count() reports the whole machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(cfg, budget_s=30.0):
    """SURVEY.md 8d / BASELINE.md 4: the ATen restatement of the reference's CPU path with torch.set_num_threads(n) for
    n = 1 and n = the cores this process can really use (affinity mask and cgroup quota: `usable_cores`; `host_cores` =
    os.cpu_count() is reported beside it, and no leg runs more threads than are usable), at 256^2 (the size BASELINE configs[0]
    names), 1024^2 and 2048^2 (several lights: 256^2, 512^2, 1024^2), this configuration's light and flags, CPU model stated.
    Bounded: a leg that the previous (smaller) leg of its thread count predicts to overrun what is left of `budget_s`
    is skipped, and the line says so.  `value` / `cores`: the fastest thread count at the largest size measured."""
    _oracle_path()
    host, usable = os.cpu_count() or 1, usable_cores()
    saved = torch.get_num_threads()
    table, skipped = [], []
    legs = ((256, 5), (1024, 2), (2048, 1)) if len(cfg["light"]) == 1 else ((256, 2), (512, 1), (1024, 1))
    # the size the metric is quoted on (SURVEY.md 8d "256^2, 1024^2, 4096^2"): one pass at ONE thread (8-9 s) when the budget allows
    metric_leg = (4096, 1) if cfg["size"] >= 4096 and len(cfg["light"]) == 1 else None
    t_cold = time.perf_counter()
    if budget_s > 0:
        _oracle_eval(cfg, synth_material(128, "cpu", 98))           # cold first call of the process (1 s here, several on a loaded box): outside the budget
    t_start = time.perf_counter()                                   # the budget clock starts AFTER it (VERDICT r5 #1a)
    cold_s = t_start - t_cold
    for threads in sorted({1, usable}):
        torch.set_num_threads(threads)
        per_pixel = None
        for size, passes in legs + ((metric_leg,) if metric_leg and threads == 1 else ()):
            left = budget_s - (time.perf_counter() - t_start)
            first = not table and budget_s > 0                      # the first leg of the first thread count always runs: no host is slow enough to empty the table
            if not first and (left <= 0 or (per_pixel is not None and per_pixel * size * size * passes > left)):
                skipped.append(f"{size}^2 x {threads} threads")
                continue
            maps = synth_material(size, "cpu", 99)
            t0 = time.perf_counter()
            for _ in range(passes):
                _oracle_eval(cfg, maps)
            dt = (time.perf_counter() - t0) / passes
            per_pixel = dt / (size * size)
            table.append({"size": size, "threads": threads, "ms": round(dt * 1e3, 1), "Mpixels_per_s": round(size * size / dt / 1e6, 3)})
    torch.set_num_threads(saved)
    rec = {"value": None, "unit": "Mpixels/s", "cores": None, "usable_cores": usable, "host_cores": host, "kind": "port", "cpu_model": cpu_model(),
           "table": table, "cold_first_call_s": round(cold_s, 2)}
    legs_txt = " / ".join(f"{sz}^2 ({p} passes)" for sz, p in legs) + (f" and, at 1 thread, {metric_leg[0]}^2 (1 pass: the metric's own size)" if metric_leg else "")
    if not table:                                                   # e.g. --cpu-budget 0: a skipped baseline, not a lost bench line
        rec["sample"] = f"nothing measured within the {budget_s:.0f} s budget; skipped: {skipped}"
        return rec
    biggest = max(e["size"] for e in table)
    best = max((e for e in table if e["size"] == biggest), key=lambda e: e["Mpixels_per_s"])
    rec.update(value=best["Mpixels_per_s"], cores=best["threads"], value_at=f"{biggest}^2")
    rec["sample"] = (f"oracle/torch_oracle.py (the reference's ATen ops, bit-equal to it in the dev container), this configuration's light and flags, "
                     f"one material of {legs_txt} at 1 and {usable} threads (the cores this process may use; the host has {host}), "
                     f"{time.perf_counter() - t_start:.1f} s in all; value = the fastest thread count at {biggest}^2" +
                     (f"; skipped for time: {skipped}" if skipped else ""))
    return rec


def parity_of_timed_output(cfg, out, maps, shard, band_rows):
    """The checker's leg (like cpu_baseline: the only place bench.py touches oracle/): row bands of the output the timed
    launches wrote -- first rows of the first material, middle, LAST rows of the last material this rank owns -- against
    the float64 C oracle (oracle/ct_oracle.c) and the fp32 ATen restatement of the reference (oracle/torch_oracle.py)."""
    _oracle_path()
    import numpy as np
    import c_oracle
    S, h = cfg["size"], shard.row_stop - shard.row_start
    nb = shard.batch_stop - shard.batch_start
    picks = sorted({(0, 0), (nb // 2, max(0, (h - band_rows) // 2)), (nb - 1, max(0, h - band_rows))})
    worst64 = worst32 = ref_gap = rough_needed = 0.0
    over = ref_over = values = 0
    fp16_out = out.dtype == torch.float16
    for i, y0 in picks:
        sl = slice(y0, min(h, y0 + band_rows))
        crop = [t[i, :, sl, :].float().cpu() for t in maps]                       # fp16 maps: the oracle gets their exact up-casts
        got = out[i, :, sl, :].float().cpu().numpy()
        yg = shard.row_start + y0
        converted = bool(cfg["flags"].get("convert_to_diffuse_specular"))
        ref64 = c_oracle.render(*[t.numpy() for t in crop], None, view=VIEW, lights=cfg["light"], intensities=cfg["intensity"],
                                light_type=cfg["light_type"], light_size=cfg["light_size"], y_offset=yg, H_total=S,
                                workflow="converted" if converted else "metallic",
                                specular_is_srgb=cfg["flags"].get("specular_is_srgb", True), dtype=np.float64)
        ref32 = _oracle_eval(cfg, crop, y_offset=yg, H_total=S).numpy()
        worst64 = max(worst64, float(np.abs(got - ref64).max()))
        d32 = np.abs(got - ref32)
        worst32 = max(worst32, float(d32.max()))
        over += int((d32 > 1e-5).sum())
        if (d32 > 1e-5).any():                                                       # the roughness above which every value of the sample is within 1e-5
            rough_needed = max(rough_needed, float(np.broadcast_to(crop[2].numpy(), d32.shape)[d32 > 1e-5].max()))
        r = np.abs(ref32 - ref64)                                                     # the reference's own fp32 rounding against its float64 evaluation
        ref_gap = max(ref_gap, float(r.max()))
        ref_over += int((r > 1e-5).sum())
        values += got.size
    return {"max_abs_err_vs_fp64_oracle": worst64, "max_abs_err_vs_reference_fp32": worst32,
            "values_over_1e-5_vs_reference_fp32": over, "values": values,
            "reference_fp32_max_abs_err_vs_its_fp64": ref_gap, "reference_fp32_values_over_1e-5_vs_its_fp64": ref_over,
            # count <= 2e-5 N -- a statement about the build only where the reference's own fp32 run meets it
            "count_bound_2e-5_N": round(2e-5 * values, 1), "count_within_bound": over <= 2e-5 * values,
            "reference_count_within_bound": ref_over <= 2e-5 * values,
            "roughness_above_which_every_value_is_within_tolerance": round(rough_needed, 4),
            "tolerance": 4.9e-4 if fp16_out else 1e-5,
            "sample": f"{len(picks)} bands of {band_rows} rows x {S} columns (local material, first row) = "
                      f"{[(shard.batch_start + i, shard.row_start + y) for i, y in picks]} of the output written by rank 0's timed launches; "
                      f"fp64: oracle/ct_oracle.c, fp32: oracle/torch_oracle.py (reference's ATen ops)"}


def recorded_traffic(kernel_name):
    """HBM bytes per launch from the PMC passes (rocprofv3 --pmc, collected separately and committed
    under profiles/): FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 + WRITE_SIZE.
    Returns (bytes or None, which committed run they come from)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        ent = rec.get(kernel_name)
        if ent is None:
            return None, None
        return ent.get("hbm_bytes_per_launch"), ent.get("run", "profiles/pmc_traffic.json: " + ent.get("collected", ""))
    except (OSError, ValueError):
        return None, None


class BoxPattern:
    """tools/boxcal.hip on the buffers of one plan of config 2 (fp32 metallic maps, one material): `.launch(stream)` enqueues the bare
    8-in / 3-out access pattern over the plan's own planes.  Bench-side helper; None when the helper library is not built."""

    def __init__(self, lib, maps, out):
        import ctypes
        self.lib, self.ctypes = lib, ctypes
        planes = []
        for t in maps:                                   # [1,C,H,W] fp32, rows dense
            for c in range(t.shape[1]):
                planes.append(t.data_ptr() + c * t.stride(1) * 4)
        outs = [out.data_ptr() + c * out.stride(1) * 4 for c in range(3)]
        assert len(planes) == 8 and all(p % 16 == 0 for p in planes + outs)
        self.inp = (ctypes.c_void_p * 8)(*planes)
        self.out = (ctypes.c_void_p * 3)(*outs)
        self.pixels = maps[0].shape[-2] * maps[0].shape[-1]
        self.keep = (maps, out)

    def launch(self, stream):
        rc = self.lib.boxcal_forward_pattern(self.inp, self.out, self.ctypes.c_uint64(self.pixels), self.ctypes.c_void_p(stream))
        if rc != 0:
            raise RuntimeError("boxcal_forward_pattern: %d" % rc)


def load_boxcal():
    import ctypes
    path = os.path.join(ROOT, "tools", "libboxcal.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    lib.boxcal_forward_pattern.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint64, ctypes.c_void_p]
    lib.boxcal_forward_pattern.restype = ctypes.c_int
    return lib


SHARE_GPU = os.environ.get("PBR_BENCH_SHARE_GPU") == "1"     # test hook: N ranks on fewer GPUs (rank r on device r mod count), the
                                                              # small collectives over gloo -- exercises the N > 1 code path on a 1-GPU box


def launch_ranks(n):
    """`python bench.py --gpus N`, N > 1, typed as it stands: this process has not touched the GPU (and never will);
    the N ranks run under torch.distributed.run as a CHILD process, and its exit code becomes ours."""
    if torch.cuda.device_count() < n and not SHARE_GPU:          # (on ROCm wheels without amdsmi this count CAN initialise the runtime: harmless, the ranks are a CHILD)
        raise SystemExit(f"bench.py --gpus {n}: only {torch.cuda.device_count()} ROCm device(s) visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def timed_region(plans, steps, stream, barrier, offset=0):
    """K launches bracketed by barrier + synchronize on both sides; returns (wall seconds, kernel ms per launch from
    HIP events recorded on the launch stream)."""
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                                   # same stream the kernels are launched on
    for i in range(steps):
        plans[(offset + i) % len(plans)].launch(stream)
    ev1.record()
    while not ev1.query():                         # spin on the end event: a blocking wait adds tens of microseconds of wake-up
        pass                                       # latency, which is 1-2 % of a 20-step region
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0             # this rank's K steps; MAX over ranks by the caller
    barrier()
    return elapsed, ev0.elapsed_time(ev1) / steps


def run_config1(args, real_stdout):
    """BASELINE.json configs[0] on the GPU: ONE 256x256 BasecolorMetallicMaterial, point light, through the reference-shaped callable
    `CookTorranceBRDF.__call__` (examples/example_brdf.py:14-23) -- where the cost is the host layer and the launch, not the kernel.
    Legs (median of `--steps` repeats of 200 calls each): device-resident material, eager and captured into a HIP graph; CPU-resident
    material (the reference's default: upload, evaluate, image back), eager; the bare plan launch; the kernel by HIP events; and the CPU
    oracle's time for the same call on this host (34.4 ms on the survey's 8 cores)."""
    import statistics
    from pypbr_amd import functional as F
    from pypbr_amd.materials import BasecolorMetallicMaterial
    from pypbr_amd.models import CookTorranceBRDF
    S = args.size or 256
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    view, light, inten = torch.tensor(VIEW), torch.tensor([0.1, 0.1, 1.0]), torch.tensor([1.0, 1.0, 1.0])
    host_maps = synth_material(S, "cpu", 0)
    brdf = CookTorranceBRDF(light_type="point")
    on_dev = BasecolorMetallicMaterial(albedo=host_maps[0], roughness=host_maps[2], metallic=host_maps[3])
    on_dev._raw["normal"] = host_maps[1]
    on_dev.to(dev)
    on_cpu = BasecolorMetallicMaterial(albedo=host_maps[0], roughness=host_maps[2], metallic=host_maps[3])
    on_cpu._raw["normal"] = host_maps[1]
    calls, reps = 200, max(3, args.steps or 7)

    def per_call(fn, sync_each=False):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        issue, drained = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            for _ in range(calls):
                fn()
                if sync_each:
                    torch.cuda.synchronize()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            issue.append((t1 - t0) / calls * 1e6)
            drained.append((t2 - t0) / calls * 1e6)
        return round(statistics.median(issue), 2), round(statistics.median(drained), 2)

    eager_issue, eager = per_call(lambda: brdf(on_dev, view, light, inten, 1.0))
    _, eager_sync = per_call(lambda: brdf(on_dev, view, light, inten, 1.0), sync_each=True)          # a caller that looks at every image
    _, cpu_resident = per_call(lambda: brdf(on_cpu, view, light, inten, 1.0))                        # returns a host tensor: synchronous by nature
    store = on_dev._raw
    plan = F.plan_cook_torrance(store["albedo"], store["normal"], store["roughness"], store["metallic"], view_dir=VIEW, light=[0.1, 0.1, 1.0],
                                light_intensity=[1.0, 1.0, 1.0], light_type="point", light_size=1.0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    plan_issue, plan_drained = per_call(lambda: plan.launch(stream))
    _, kernel_ms = timed_region([plan], 2000, stream, lambda: None)
    # the same call captured into a HIP graph (a render loop over a fixed material: replay costs one graph launch)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):
            brdf(on_dev, view, light, inten, 1.0)
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured_out = brdf(on_dev, view, light, inten, 1.0)
    graph_issue, graph_drained = per_call(graph.replay)
    eager_out = brdf(on_dev, view, light, inten, 1.0)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(captured_out, eager_out)
    line = {
        "metric": "Mpixels/s Cook-Torrance eval, BASELINE.json configs[0] (one 256x256 material through CookTorranceBRDF.__call__)",
        "value": round(S * S / eager, 3), "unit": "Mpixels/s", "n_gpus": 1, "steps": reps, "warmup": 20,
        "ms_per_step": round(eager * 1e-3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"Single {S}x{S} BasecolorMetallicMaterial, point light, CookTorranceBRDF.__call__ on a device-resident material, eager "
                               "(BASELINE.json configs[0]; examples/example_brdf.py path)", "config": 1, "kernel": plan.kernel_name,
                   "pixels_per_step": S * S, "bytes_per_pixel": plan.bytes_per_pixel, "lights": 1, "map_dtype": "f32",
                   "timing": f"a step = one call; medians over {reps} repeats of {calls} back-to-back calls, GPU drained at the end of each repeat"},
        "latency_us": {
            "eager_device_resident": eager, "eager_device_resident_host_issue": eager_issue, "eager_device_resident_sync_every_call": eager_sync,
            "graph_replay_device_resident": graph_drained, "graph_replay_host_issue": graph_issue,
            "eager_cpu_resident_upload_evaluate_download": cpu_resident,
            "plan_launch": plan_drained, "plan_launch_host_issue": plan_issue, "kernel_hip_events": round(kernel_ms * 1e3, 2),
            "eager_over_graph": round(eager / graph_drained, 2),
            "note": "host layer = eager - plan_launch; launch = plan_launch - kernel; the kernel itself is launch-latency-sized at 256^2"},
        "roofline": {"bound": "hbm", "achieved": round(plan.bytes_per_pixel * S * S / (kernel_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(plan.bytes_per_pixel * S * S / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernel_us": round(kernel_ms * 1e3, 2), "note": "2.9 MB per launch: 0.36 us at peak -- this configuration measures latency, not bandwidth"},
    }
    from pypbr_amd import _native
    line["build"] = _native.build_stamp()
    if not args.no_cpu_baseline:
        _oracle_path()
        import torch_oracle as O
        a, n, r, m = host_maps
        ref = O.cook_torrance(a, n, r, m, None, view=view, light=light, intensity=inten, light_type="point", light_size=1.0)
        err = (eager_out.cpu() - ref).abs()
        ref64 = O.cook_torrance(a.double(), n.double(), r.double(), m.double(), None, view=view.double(), light=light.double(), intensity=inten.double(),
                                light_type="point", light_size=1.0)
        over, ref_over = err > 1e-5, (ref.double() - ref64).abs() > 1e-5
        line["parity"] = {"max_abs_err_vs_reference_fp32": float(err.max()), "max_abs_err_vs_fp64_oracle": float((eager_out.cpu().double() - ref64).abs().max()),
                          "values_over_1e-5_vs_reference_fp32": int(over.sum()), "reference_fp32_values_over_1e-5_vs_its_fp64": int(ref_over.sum()),
                          "values": ref.numel(), "count_bound_2e-5_N": round(2e-5 * ref.numel(), 1), "tolerance": 1e-5,
                          "roughness_above_which_every_value_is_within_tolerance": round(float(r.expand_as(err)[over].max()), 4) if bool(over.any()) else 0.0,
                          "sample": "the whole image against oracle/torch_oracle.py (the reference's ATen ops; float64: the same ops in double); "
                                    "synthetic roughness in [0.05, 1]: below ~0.15 the reference's own fp32 output is not reproducible to 1e-5 (DESIGN.md 4)"}
        table = []
        saved = torch.get_num_threads()
        for threads in sorted({1, usable_cores()}):
            torch.set_num_threads(threads)
            O.cook_torrance(a, n, r, m, None, view=view, light=light, intensity=inten, light_type="point", light_size=1.0)
            t0 = time.perf_counter()
            for _ in range(10):
                O.cook_torrance(a, n, r, m, None, view=view, light=light, intensity=inten, light_type="point", light_size=1.0)
            dt = (time.perf_counter() - t0) / 10
            table.append({"size": S, "threads": threads, "ms": round(dt * 1e3, 2), "Mpixels_per_s": round(S * S / dt / 1e6, 3)})
        torch.set_num_threads(saved)
        best = max(table, key=lambda e: e["Mpixels_per_s"])
        line["cpu_baseline"] = {"value": best["Mpixels_per_s"], "unit": "Mpixels/s", "cores": best["threads"], "usable_cores": usable_cores(),
                                "host_cores": os.cpu_count() or 1, "kind": "port", "cpu_model": cpu_model(), "table": table,
                                "sample": f"oracle/torch_oracle.py, the same {S}^2 call, 10 passes at 1 and {usable_cores()} threads (BASELINE.md 2: the reference itself 34.4 ms on the survey's 8 cores)"}
    os.write(real_stdout, (json.dumps(line) + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=[1] + sorted(CONFIGS), help="workload, SURVEY.md 8d numbering (BASELINE.json configs[config-1])")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--size", type=int, default=None, help="override the configuration's map edge (tests)")
    ap.add_argument("--batch", type=int, default=None, help="override the configuration's batch (tests; configs 3-5)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the checker's legs (cpu_baseline and parity)")
    ap.add_argument("--settle", type=int, default=None, help="untimed clock-settle launches between the cold and the steady timed region")
    ap.add_argument("--layout", choices=("separate", "arena"), default="arena",
                    help="config 2.  arena (default): the material's maps and result in one allocation, as Material.to(device) lays "
                         "them out (F.pack_maps); separate: five tensors as torch's allocator places them")
    ap.add_argument("--cpu-budget", type=float, default=30.0, help="seconds the cpu_baseline leg may take (legs predicted to overrun are skipped and named)")
    ap.add_argument("--spawn", action="store_true", help="start the ranks through torch.distributed.run even for --gpus 1")
    ap.add_argument("--no-rccl", action="store_true", help="--gpus 1 started plainly: do not form the one-rank RCCL group (no collectives at all)")
    ap.add_argument("--example", choices=("brdf", "blend", "both"), default=None,
                    help="instead of the kernel benchmark: examples/example_brdf.py / example_blend.py statement by statement (tools/example_bench.py)")
    args = ap.parse_args()
    if args.example:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import example_bench
        return example_bench.main(["--example", args.example] + (["--size", str(args.size)] if args.size else []) +
                                  (["--no-cpu"] if args.no_cpu_baseline else []))

    if "RANK" not in os.environ and (args.gpus > 1 or args.spawn):
        launch_ranks(args.gpus)                    # does not return

    # The contract is ONE JSON line on stdout.  RCCL prints a five-line version banner to the C-level stdout when a communicator is
    # created (and other libraries may print what they like): from here on file descriptor 1 IS stderr, and the line goes to the
    # descriptor that was stdout when the process started.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    # the host driver of this pool only supports dmabuf IPC: without this RCCL's buffer exchange between ranks fails with
    # `hipIpcGetMemHandle: invalid argument` (the image exports it; a launcher that builds its own environment may not)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.config == 1:
        if args.gpus != 1:
            raise SystemExit("--config 1 is the single-material latency case: one GPU")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm device; pypbr_amd has no CPU path")
        return run_config1(args, real_stdout)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ and (world > 1 or args.spawn or os.environ.get("PBR_BENCH_FORCE_DIST") == "1")
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    cfg = dict(CONFIGS[args.config])
    if args.size:
        cfg["size"] = args.size
    if args.batch:
        if args.config == 2:
            raise SystemExit("--batch applies to configs 3-5 (config 2 is one material per GPU)")
        cfg["batch"] = args.batch
    steps = args.steps if args.steps is not None else cfg["steps"]
    warmup = args.warmup if args.warmup is not None else cfg["warmup"]
    S = cfg["size"]

    from pypbr_amd import functional as F
    from pypbr_amd.distributed import Shard, broadcast_light_block, cook_torrance_sharded, partition

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device; pypbr_amd has no CPU path")
    device = torch.device("cuda", local_rank % torch.cuda.device_count() if SHARE_GPU else local_rank)
    torch.cuda.set_device(device)
    coll = device                                  # where the small collectives' buffers live
    backend = None
    if distributed:
        import torch.distributed as dist
        if SHARE_GPU:                              # RCCL refuses two ranks on one device
            dist.init_process_group("gloo")
            coll = torch.device("cpu")
        else:
            dist.init_process_group("nccl", device_id=device)
        backend = dist.get_backend()
    late_rccl = world == 1 and not distributed and not args.no_rccl       # a single rank started plainly: see form_group_of_one below

    def barrier(always=False):
        # a group of ONE has nobody to wait for: the timed regions of an N = 1 run contain no collective at all
        if distributed and (world > 1 or always):
            dist.barrier() if SHARE_GPU else dist.barrier(device_ids=[local_rank])

    # light/view parameters: owned by rank 0, broadcast over RCCL/xGMI (404 B, once per change)
    params = dict(view_dir=VIEW, light=cfg["light"], light_intensity=cfg["intensity"], light_size=cfg["light_size"])

    def broadcast_latency():
        lat = []
        for _ in range(20):                        # what one parameter change costs: pack, broadcast, unpack (D2H of 404 B)
            barrier(always=True)
            t0 = time.perf_counter()
            broadcast_light_block(params if rank == 0 else None, device=device, src=0)
            lat.append((time.perf_counter() - t0) * 1e6)
        return sorted(lat)[len(lat) // 2]
    bcast_us = broadcast_latency() if distributed else None

    plans, map_sets = [], []
    share_plan = None
    if args.config == 2:
        # weak scaling: every rank its own material; the block still comes from rank 0
        if distributed:
            params = broadcast_light_block(params if rank == 0 else None, device=device, src=0)
        shard = Shard(0, 1, 0, S)
        for i in range(N_BUFFER_SETS):
            a, n, r, m = synth_material(S, device, 1234 + rank * 16 + i)
            out = None
            if args.layout == "arena":      # the maps of a material and its result in ONE allocation (F.pack_maps)
                a, n, r, m, out = F.pack_maps(a, n, r, m, reserve_output=True)
                out = out.unsqueeze(0)
            map_sets.append([t.unsqueeze(0) for t in (a, n, r, m)])
            plans.append(F.plan_cook_torrance(a, n, r, m, view_dir=params["view_dir"], light=params["light"],
                                              light_intensity=params["light_intensity"], light_type=cfg["light_type"],
                                              light_size=params["light_size"], out=out))
        global_pixels = world * S * S
    else:
        # strong scaling: the job is the whole batch; this rank generates and owns partition(B, S, world, rank)
        shard = partition(cfg["batch"], S, world, rank)
        a, n, r, m = synth_shard(cfg, shard, device, 4000 * args.config)
        maps = {"albedo": a, "normal": n, "roughness": r, "metallic": m}
        if distributed:
            _, plan = cook_torrance_sharded(maps, params if rank == 0 else None, light_type=cfg["light_type"], owned=shard,
                                            global_shape=(cfg["batch"], S), plan=True, **cfg["flags"])
        else:
            plan = F.plan_cook_torrance(a, n, r, m, view_dir=VIEW, light=cfg["light"], light_intensity=cfg["intensity"],
                                        light_type=cfg["light_type"], light_size=cfg["light_size"], y_offset=shard.row_start,
                                        height_total=S, **cfg["flags"])
        if plan is None:
            raise SystemExit(f"rank {rank}: empty shard {tuple(shard)} -- more ranks than rows")
        plans.append(plan)
        map_sets.append([a, n, r, m])
        global_pixels = cfg["batch"] * S * S
        if world == 1 and cfg["batch"] % 8 == 0 and cfg["batch"] >= 16:
            # beside the N = 1 point of the strong-scaling curve: the share one GPU of an 8-GPU node gets (its first B/8 materials)
            k = cfg["batch"] // 8
            share_plan = F.plan_cook_torrance(a[:k], n[:k], r[:k], m[:k], view_dir=VIEW, light=cfg["light"], light_intensity=cfg["intensity"],
                                              light_type=cfg["light_type"], light_size=cfg["light_size"], out=plan.out[:k], **cfg["flags"])
    kernel = plans[0].kernel_name
    bpp = plans[0].bytes_per_pixel
    local_pixels = (shard.batch_stop - shard.batch_start) * (shard.row_stop - shard.row_start) * S
    stream = torch.cuda.current_stream(device).cuda_stream

    # ---- cold: W warm-up launches, K timed steps (the literal protocol)
    for i in range(warmup):
        plans[i % len(plans)].launch(stream)
    cold_s, cold_kernel_ms = timed_region(plans, steps, stream, barrier, offset=warmup)
    # ---- steady: clock settle (from an idle GPU the first ~20 launches run at boost clocks, the next ~150 up to 25 %
    # slower while power management reins them in, steady from launch ~300 of the 0.115 ms kernel on: tools/transient_probe.py)
    settle = args.settle if args.settle is not None else (300 if args.config == 2 else max(2, int(math.ceil(40.0 / max(cold_kernel_ms, 1e-3)))))
    for i in range(settle):
        plans[i % len(plans)].launch(stream)
    elapsed, kernel_ms = timed_region(plans, steps, stream, barrier)
    # ---- slope: 2K more steps; (wall(2K) - wall(K)) / K is a step without the region's fixed entry / exit cost
    elapsed2k, kernel2k_ms = timed_region(plans, 2 * steps, stream, barrier)

    extras = {}
    box_ms = None
    if args.config == 2 and cfg["dtype"] == torch.float32:
        boxlib = load_boxcal()
        if boxlib is not None:                     # the bare access pattern on the same buffers, same rotation, same protocol
            boxes = [BoxPattern(boxlib, ms, pl.out) for ms, pl in zip(map_sets, plans)]
            for i in range(10):
                boxes[i % len(boxes)].launch(stream)
            _, box_ms = timed_region(boxes, steps, stream, barrier)
            for pl in plans:                       # the pattern wrote sums into the result planes: evaluate again (the parity leg reads plans[0].out)
                pl.launch(stream)
    if share_plan is not None:
        _, share_ms = timed_region([share_plan], max(5, steps), stream, barrier)
        k = cfg["batch"] // 8
        extras["per_gpu_share_of_8"] = {"batch": k, "kernel_us": round(share_ms * 1e3, 2), "Mpixels_per_s": round(k * S * S / share_ms / 1e3, 1),
                                        "hbm_GBps_algorithmic": round(bpp * k * S * S / share_ms / 1e6, 1),
                                        "note": "the first B/8 materials of this very batch, same buffers: what one GPU of an 8-GPU node evaluates per step"}
        plans[0].launch(stream)                    # the parity leg reads the full result: write it again
    if args.config == 3:
        # the other setting of upstream's F6 flag (specular decoded once), same maps, same protocol
        alt = F.plan_cook_torrance(*map_sets[0], view_dir=params["view_dir"], light=params["light"], light_intensity=params["light_intensity"],
                                   light_type=cfg["light_type"], light_size=params["light_size"], y_offset=shard.row_start, height_total=S,
                                   out=plans[0].out, **dict(cfg["flags"], specular_is_srgb=False))
        _, alt_ms = timed_region([alt], max(5, steps), stream, barrier)
        extras["specular_is_srgb_false"] = {"kernel_us": round(alt_ms * 1e3, 2), "Mpixels_per_s": round(local_pixels / alt_ms / 1e3, 1),
                                            "note": "this rank's shard with the flag a user sets by hand (decoded once); same kernel, same bytes"}
        plans[0].launch(stream)

    if late_rccl:
        # `python bench.py` as the driver types it for N = 1: an RCCL group of ONE formed in this process AFTER the timed regions -- with
        # RCCL initialised before them the same launches measured 1.0-1.3 % slower (five alternating pairs, round 4) -- so that the
        # line has still been through init_process_group("nccl"), the light-block broadcast, a barrier and the all-reduce / all-gather
        # of the N > 1 path below, and says so (`per_rank.backend`, `ranks_seen`).  Never fatal: without RCCL the line stands as measured.
        import datetime
        import torch.distributed as dist
        try:
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device,
                                    timeout=datetime.timedelta(seconds=60))
            distributed, backend = True, dist.get_backend()
            bcast_us = broadcast_latency()
        except Exception as e:                     # noqa: BLE001 -- whatever RCCL or the rendezvous raises
            backend = "none (init_process_group('nccl') failed: %s)" % str(e).splitlines()[0][:120]
    ranks_seen = dist.get_world_size() if distributed else 1

    per_rank_us, per_rank_px = [kernel_ms * 1e3], [local_pixels]
    per_rank_all = [0.5 * (kernel_ms + cold_kernel_ms) * 1e3]
    shards = [tuple(shard)]
    if distributed:
        t = torch.tensor([elapsed, cold_s, elapsed2k], device=coll, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, cold_s, elapsed2k = float(t[0].item()), float(t[1].item()), float(t[2].item())
        mine = torch.tensor([kernel_ms * 1e3, 0.5 * (kernel_ms + cold_kernel_ms) * 1e3, float(local_pixels)] + [float(v) for v in shard],
                            device=coll, dtype=torch.float64)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank_us = [float(g[0].item()) for g in gathered]
        per_rank_all = [float(g[1].item()) for g in gathered]
        per_rank_px = [int(g[2].item()) for g in gathered]
        shards = [tuple(int(v.item()) for v in g[3:7]) for g in gathered]

    assert bool(torch.isfinite(plans[0].result).all())

    if rank == 0:
        value = global_pixels * steps / elapsed / 1e6
        slow = max(range(world), key=lambda i: per_rank_us[i])        # the rank whose launches of the STEADY region -- the K steps `value`
        kernel_all_ms = per_rank_all[slow] * 1e-3                     # is quoted on -- took longest; (kernel_all: cold + steady, rounds 2-4's basis)
        achieved = bpp * per_rank_px[slow] / (per_rank_us[slow] * 1e-6) / 1e9
        traffic, traffic_run = recorded_traffic(kernel) if args.config == 2 and S == 4096 else (None, None)
        line = {
            "metric": "Mpixels/s Cook-Torrance eval, 4K maps" if args.config == 2 else
                      f"Mpixels/s Cook-Torrance eval, BASELINE.json configs[{args.config - 1}]",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 5), "higher_is_better": True, "scaling": cfg["scaling"],
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "ms_per_step_cold": round(cold_s / steps * 1e3, 5),
            "ms_per_step_slope": round((elapsed2k - elapsed) / steps * 1e3, 5),
            "value_slope": round(global_pixels * steps / max(elapsed2k - elapsed, 1e-9) / 1e6, 1),
            "launches_before_value": warmup + steps + settle,
            "value_cold": round(global_pixels * steps / cold_s / 1e6, 1),
            "config": {"workload": cfg["name"].format(S=S, B=cfg["batch"]), "config": args.config,
                       "kernel": kernel, "global_batch": world if args.config == 2 else cfg["batch"], "map_size": [S, S],
                       "pixels_per_step": global_pixels, "bytes_per_pixel": bpp, "lights": len(cfg["light"]),
                       "map_dtype": "f16" if cfg["dtype"] == torch.float16 else "f32",
                       "parallelism": (f"material-sharded x{world}: " + ("one material per rank (weak)" if args.config == 2 else
                                       "pypbr_amd.distributed.partition over the batch, every rank generates and owns its slice (strong)"))
                                      + (" (TEST HOOK: ranks share GPUs, collectives over gloo)" if SHARE_GPU else ""),
                       "timing": f"value/ms_per_step: wall time of {steps} steps after {warmup} warm-up + {steps} cold-timed + "
                                 f"{settle} clock-settle launches = {warmup + steps + settle} launches (sustained rate; `warmup` echoes the argument); "
                                 f"value_cold/ms_per_step_cold: the {steps} steps right after the {warmup} warm-up launches; "
                                 f"value_slope/ms_per_step_slope: (wall of {2 * steps} further steps - wall of the {steps}) / {steps}, i.e. "
                                 f"without the ~{max(0.0, (2 * elapsed - elapsed2k)) * 1e6:.0f} us a region costs to enter and leave",
                       "clock_settle_launches": settle},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_run,
                         "kernel_us": round(per_rank_us[slow], 2), "kernel_us_steady": round(per_rank_us[slow], 2),
                         "kernel_us_cold_and_steady": round(kernel_all_ms * 1e3, 2),
                         "basis": "HIP events on the launch stream around the steady region's launches (the region `value` is quoted on)",
                         "pixels_per_launch": per_rank_px[slow], "rank": slow},
            "per_rank": {"kernel_us": [round(u, 2) for u in per_rank_us], "pixels_per_launch": per_rank_px,
                         "shard_batch_rows": shards, "backend": backend, "ranks_seen": ranks_seen,
                         "light_block_broadcast_us": None if bcast_us is None else round(bcast_us, 1)},
        }
        if args.config == 2:
            line["config"]["layout"] = ("arena: the 8 map planes of a material and its 3 result planes in one allocation "
                                        "(pypbr_amd.functional.pack_maps, what Material.to(device) does)" if args.layout == "arena" else
                                        "separate: albedo, normal, roughness, metallic and the result as five tensors wherever torch's allocator put them")
            line["roofline"]["kernel_us_cold"] = round(cold_kernel_ms * 1e3, 2)
            line["roofline"]["kernel_us_2k_region"] = round(kernel2k_ms * 1e3, 2)
            if box_ms is not None:
                # the bare 8-in / 3-out access pattern (tools/boxcal.hip) on the same buffers right behind the steady region: what THIS box
                # gives any kernel with this access pattern.  kernel_over_box_pattern ~ 1: the kernel is at its pattern's ceiling on this box
                line["roofline"]["box_pattern_us"] = round(box_ms * 1e3, 2)
                line["roofline"]["box_pattern_frac"] = round(bpp * per_rank_px[0] / (box_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                line["roofline"]["kernel_over_box_pattern"] = round(per_rank_us[0] / (box_ms * 1e3), 4)
        if len(cfg["light"]) > 1:
            L = len(cfg["light"])
            per_pl, valu_src, clock = recorded_valu()
            evals = per_rank_px[slow] * L / (kernel_all_ms * 1e-3) / 1e9
            ginstr = None if per_pl is None else per_pl * evals
            line["roofline_valu"] = {"bound": "valu", "achieved": None if ginstr is None else round(ginstr, 1), "peak": round(VALU_PEAK_GINSTR, 1),
                                     "unit": "G lane-instr/s", "frac": None if ginstr is None else round(ginstr / VALU_PEAK_GINSTR, 4),
                                     "light_evals_per_s_G": round(evals, 1),
                                     # the chip does not hold 2.4 GHz under this launch (power-capped): against the clock the committed SQ
                                     # pass measured DURING this kernel (GRBM_GUI_ACTIVE / time), the schedule is at this fraction of issue
                                     "shader_clock_GHz_recorded": clock,
                                     "frac_at_recorded_clock": None if (ginstr is None or not clock) else round(ginstr / (VALU_PEAK_GINSTR * clock / 2.4), 4),
                                     "basis": (f"{per_pl:.2f} vector instructions per (pixel, light) (rocprofv3 SQ_INSTS_VALU x 64 / (pixels x lights), {valu_src}) "
                                               if per_pl is not None else "no committed SQ pass for the 16-light kernel found under profiles/: ") +
                                              "against 256 CUs x 4 SIMDs x 16 lanes per clock at 2.4 GHz; the launch is VALU-bound, its HBM "
                                              "fraction is not the measure (SURVEY.md 8d)"}
        line.update(extras)
        from pypbr_amd import _native
        line["build"] = _native.build_stamp()          # the commit, the library's sha256, the sources it was built from (stale = not these)
        if not args.no_cpu_baseline:
            line["parity"] = parity_of_timed_output(cfg, plans[0].out, map_sets[0], shard, 8 if S >= 2048 and len(cfg["light"]) == 1 else 4)
            if world == 1:
                line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_budget)
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if distributed:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
