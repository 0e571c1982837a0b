#!/usr/bin/env python3
"""bench.py -- Mpixels/s of fused Cook-Torrance evaluation on MI355X.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): one
4096x4096 BasecolorMetallicMaterial per GPU (albedo sRGB, decoded normal, roughness,
metallic; fp32 planar), point light, sRGB output.  A "step" is one pass of the hot
path over that batch = ONE launch of the fused kernel through the C ABI
(pbr_cook_torrance), inputs already resident in HBM.  With --gpus N every rank owns
its own material (independent materials shard with no data-path collective, weak
scaling); the light/view parameter block is broadcast once from rank 0 over RCCL.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size 4096] [--no-cpu-baseline] [--layout arena|separate] [--settle 300]

`python bench.py --gpus N` with N > 1 starts the N ranks itself: the parent process (which never touches a GPU)
runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
bench.py ...` as a child and exits with its code.  Started under torch.distributed.run already (RANK in the
environment), the script is a rank.

Timing protocol.  Two timed regions, both K steps bracketed by barrier + synchronize, MAX over ranks:
  cold    W warm-up launches from whatever state the GPU was in, then K timed steps -- the literal
          "W warm-up, K steps" protocol -> `ms_per_step_cold`, `value_cold`;
  steady  `--settle` (default 300) further untimed launches so that power management has settled the clocks
          (tools/transient_probe.py), then K timed steps -> `ms_per_step`, `value`: the sustained rate.
Rank 0 prints ONE JSON line (contract in the task statement).  Extra objects:
  roofline     -- achieved algorithmic HBM GB/s of the kernel (44 B/pixel x pixels per launch / average launch
                  duration from HIP events on the launch stream over ALL 2K timed launches, cold + steady: the figure a
                  rocprofv3 --kernel-trace average of the same command reproduces) vs 8 TB/s
  per_rank     -- kernel_us of every rank (steady region) and the latency of the light-block broadcast
  parity       -- bands of the timed output of this run against the float64 C oracle and the ATen restatement
  cpu_baseline -- the ATen-level restatement of the reference's CPU path (oracle/torch_oracle.py,
                  kind "port": bit-equal to the reference in the dev container) timed on this
                  host's cores on a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_BUFFER_SETS = 3              # rotate map sets: 704 MiB per step never re-hits the 256 MiB Infinity Cache anyway
VIEW, LIGHT, INTENSITY, LIGHT_SIZE = [0.0, 0.0, 1.0], [0.1, 0.1, 1.0], [1.0, 1.0, 1.0], 1.0


def synth_material(size, device, seed):
    """SURVEY.md 8d config 2 recipe: U[0,1) albedo/metallic, roughness remapped to [0.05,1],
    normal = normalize([U(-.5,.5), U(-.5,.5), 1]) stored decoded."""
    g = torch.Generator(device=device).manual_seed(seed)
    H = W = size
    albedo = torch.rand(3, H, W, device=device, generator=g)
    nxy = torch.rand(2, H, W, device=device, generator=g) - 0.5
    normal = torch.cat([nxy, torch.ones(1, H, W, device=device)], 0)
    normal = normal / normal.norm(dim=0, keepdim=True)
    rough = torch.rand(1, H, W, device=device, generator=g) * 0.95 + 0.05
    metal = torch.rand(1, H, W, device=device, generator=g)
    return albedo, normal, rough, metal


def _oracle_path():
    p = os.path.join(ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)


def cpu_baseline(sample_size, passes):
    """Times the ATen-level oracle (the reference's op sequence) on the host cores."""
    _oracle_path()
    import torch_oracle
    a, n, r, m = [t.cpu() for t in synth_material(sample_size, "cpu", 99)]
    kw = dict(view=torch.tensor(VIEW), light=torch.tensor(LIGHT), intensity=torch.tensor(INTENSITY),
              light_type="point", light_size=LIGHT_SIZE)
    small = [t[:, :512, :512].contiguous() for t in (a, n, r, m)]
    torch_oracle.cook_torrance(*small, None, **kw)                                  # warm-up (cold first call ~1 s)
    # ATen's intra-op pool defaults to every host core; the reference's op mix scales badly past a few
    # threads, so give it its best thread count: probe on a 512x512 crop
    best, best_t = torch.get_num_threads(), float("inf")
    for nt in sorted({1, 4, 8, 16, 32, torch.get_num_threads()}):
        if nt > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        torch_oracle.cook_torrance(*small, None, **kw)
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = nt, dt
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    for _ in range(passes):
        torch_oracle.cook_torrance(a, n, r, m, None, **kw)
    dt = time.perf_counter() - t0
    return {"value": round(sample_size * sample_size * passes / dt / 1e6, 3), "unit": "Mpixels/s",
            "cores": torch.get_num_threads(), "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"{passes} passes of one {sample_size}x{sample_size} material, same recipe and light as the GPU "
                      f"workload, oracle/torch_oracle.py (ATen ops of the reference) at its fastest ATen thread count, {dt:.1f} s"}


def parity_of_timed_output(plan, maps, size, band_rows=8):
    """The checker's leg (like cpu_baseline: the only place bench.py touches oracle/): three row bands of the
    output the timed launches wrote -- first rows, middle, LAST rows -- against the float64 C oracle
    (oracle/ct_oracle.c) and the fp32 ATen restatement of the reference (oracle/torch_oracle.py)."""
    _oracle_path()
    import numpy as np
    import c_oracle
    import torch_oracle
    out = plan.result
    a, n, r, m = maps
    worst64 = worst32 = 0.0
    over = values = 0
    bands = sorted({0, (size - band_rows) // 2, size - band_rows})
    for y0 in bands:
        sl = slice(y0, y0 + band_rows)
        ca, cn, cr, cm = [t[:, sl, :].cpu() for t in (a, n, r, m)]
        got = out[:, sl, :].cpu().numpy()
        ref64 = c_oracle.render(ca.numpy(), cn.numpy(), cr.numpy(), cm.numpy(), view=VIEW, lights=LIGHT, intensities=INTENSITY,
                                light_type="point", light_size=LIGHT_SIZE, y_offset=y0, H_total=size, dtype=np.float64)
        ref32 = torch_oracle.cook_torrance(ca, cn, cr, cm, None, view=torch.tensor(VIEW), light=torch.tensor(LIGHT),
                                           intensity=torch.tensor(INTENSITY), light_type="point", light_size=LIGHT_SIZE,
                                           y_offset=y0, H_total=size).numpy()
        worst64 = max(worst64, float(np.abs(got - ref64).max()))
        d32 = np.abs(got - ref32)
        worst32 = max(worst32, float(d32.max()))
        over += int((d32 > 1e-5).sum())
        values += got.size
    return {"max_abs_err_vs_fp64_oracle": worst64, "max_abs_err_vs_reference_fp32": worst32,
            "values_over_1e-5_vs_reference_fp32": over, "values": values, "tolerance": 1e-5,
            "sample": f"rows {[f'{y}..{y + band_rows - 1}' for y in bands]} (all {size} columns) of the output written by the "
                      f"timed launches of map set 0; fp64: oracle/ct_oracle.c, fp32: oracle/torch_oracle.py (reference's ATen ops)"}


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


SHARE_GPU = os.environ.get("PBR_BENCH_SHARE_GPU") == "1"     # test hook: N ranks on fewer GPUs (rank r on device r mod count), the
                                                              # small collectives over gloo -- exercises the N > 1 code path on a 1-GPU box


def launch_ranks(n):
    """`python bench.py --gpus N`, N > 1, typed as it stands: this process has not touched the GPU (and never will);
    the N ranks run under torch.distributed.run as a CHILD process, and its exit code becomes ours."""
    if torch.cuda.device_count() < n and not SHARE_GPU:          # counting devices does not initialise HIP
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
        plans[(offset + i) % N_BUFFER_SETS].launch(stream)
    ev1.record()
    while not ev1.query():                         # spin on the end event: a blocking wait adds tens of microseconds of wake-up
        pass                                       # latency, which is 1-2 % of a 20-step region
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0             # this rank's K steps; MAX over ranks by the caller
    barrier()
    return elapsed, ev0.elapsed_time(ev1) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the checker's legs (cpu_baseline and parity)")
    ap.add_argument("--settle", type=int, default=300, help="untimed clock-settle launches between the cold and the steady timed region")
    ap.add_argument("--layout", choices=("separate", "arena"), default="arena",
                    help="arena (default): the material's maps and result in one allocation, as Material.to(device) lays "
                         "them out (F.pack_maps); separate: five tensors as torch's allocator places them")
    ap.add_argument("--cpu-sample", type=int, default=2048)
    ap.add_argument("--cpu-passes", type=int, default=4)
    ap.add_argument("--spawn", action="store_true", help="start the ranks through torch.distributed.run even for --gpus 1")
    args = ap.parse_args()

    if "RANK" not in os.environ and (args.gpus > 1 or args.spawn):
        launch_ranks(args.gpus)                    # does not return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ and (world > 1 or args.spawn or os.environ.get("PBR_BENCH_FORCE_DIST") == "1")
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from pypbr_amd import functional as F
    from pypbr_amd.distributed import broadcast_light_block

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device; pypbr_amd has no CPU path")
    device = torch.device("cuda", local_rank % torch.cuda.device_count() if SHARE_GPU else local_rank)
    torch.cuda.set_device(device)
    coll = device                                  # where the small collectives' buffers live
    if distributed:
        import torch.distributed as dist
        if SHARE_GPU:                              # RCCL refuses two ranks on one device
            dist.init_process_group("gloo")
            coll = torch.device("cpu")
        else:
            dist.init_process_group("nccl", device_id=device)

    def barrier():
        if distributed:
            dist.barrier() if SHARE_GPU else dist.barrier(device_ids=[local_rank])

    # light/view parameters: owned by rank 0, broadcast over RCCL/xGMI (404 B, once per change)
    params = dict(view_dir=VIEW, light=[LIGHT], light_intensity=[INTENSITY], light_size=LIGHT_SIZE)
    bcast_us = None
    if distributed:
        params = broadcast_light_block(params if rank == 0 else None, device=device, src=0)
        lat = []
        for _ in range(20):                        # what one parameter change costs: pack, broadcast, unpack (D2H of 404 B)
            barrier()
            t0 = time.perf_counter()
            broadcast_light_block(params if rank == 0 else None, device=device, src=0)
            lat.append((time.perf_counter() - t0) * 1e6)
        bcast_us = sorted(lat)[len(lat) // 2]

    plans, map_sets = [], []
    for i in range(N_BUFFER_SETS):
        a, n, r, m = synth_material(args.size, device, 1234 + rank * 16 + i)
        out = None
        if args.layout == "arena":      # the maps of a material and its result in ONE allocation (F.pack_maps)
            a, n, r, m, out = F.pack_maps(a, n, r, m, reserve_output=True)
            out = out.unsqueeze(0)
        map_sets.append((a, n, r, m))
        plans.append(F.plan_cook_torrance(a, n, r, m, view_dir=params["view_dir"], light=params["light"],
                                          light_intensity=params["light_intensity"], light_type="point",
                                          light_size=params["light_size"], out=out))
    kernel = plans[0].kernel_name
    bpp = plans[0].bytes_per_pixel
    pixels = args.size * args.size
    stream = torch.cuda.current_stream(device).cuda_stream

    # ---- cold: W warm-up launches, K timed steps (the literal protocol)
    for i in range(args.warmup):
        plans[i % N_BUFFER_SETS].launch(stream)
    cold_s, cold_kernel_ms = timed_region(plans, args.steps, stream, barrier, offset=args.warmup)
    # ---- steady: clock settle (from an idle GPU the first ~20 launches run at boost clocks, the next ~150 up to 25 %
    # slower while power management reins them in, steady from launch ~300 on: tools/transient_probe.py), K timed steps
    for i in range(args.settle):
        plans[i % N_BUFFER_SETS].launch(stream)
    elapsed, kernel_ms = timed_region(plans, args.steps, stream, barrier)

    per_rank_us = [kernel_ms * 1e3]
    if distributed:
        t = torch.tensor([elapsed, cold_s], device=coll, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, cold_s = float(t[0].item()), float(t[1].item())
        gathered = [torch.zeros(1, device=coll, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor([kernel_ms * 1e3], device=coll, dtype=torch.float64))
        per_rank_us = [float(g.item()) for g in gathered]

    assert bool(torch.isfinite(plans[0].result).all())

    if rank == 0:
        value = world * pixels * args.steps / elapsed / 1e6
        kernel_all_ms = 0.5 * (kernel_ms + cold_kernel_ms)      # every timed launch of this run (K cold + K steady): what a
        achieved = bpp * pixels / (kernel_all_ms * 1e-3) / 1e9   # rocprofv3 --kernel-trace average of the same command shows
        traffic, traffic_run = recorded_traffic(kernel)
        line = {
            "metric": "Mpixels/s Cook-Torrance eval, 4K maps",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "ms_per_step_cold": round(cold_s / args.steps * 1e3, 5),
            "value_cold": round(world * pixels * args.steps / cold_s / 1e6, 1),
            "config": {"workload": f"Batch=1 {args.size}x{args.size} BasecolorMetallicMaterial per GPU, point light, "
                                   f"fused HIP kernel, fp32 maps, sRGB in/out (BASELINE.json configs[1])",
                       "kernel": kernel, "pixels_per_launch": pixels, "bytes_per_pixel": bpp,
                       "parallelism": f"material-sharded x{world}" + (" (TEST HOOK: ranks share GPUs, collectives over gloo)" if SHARE_GPU else ""),
                       "timing": f"value/ms_per_step: {args.steps} steps after {args.warmup} warm-up + {args.steps} cold-timed + "
                                 f"{args.settle} clock-settle launches (sustained rate); value_cold/ms_per_step_cold: the "
                                 f"{args.steps} steps right after the {args.warmup} warm-up launches",
                       "clock_settle_launches": args.settle,
                       "layout": "arena: the 8 map planes of a material and its 3 result planes in one allocation "
                                 "(pypbr_amd.functional.pack_maps, what Material.to(device) does)"
                                 if args.layout == "arena" else "separate: albedo, normal, roughness, metallic and the result as "
                                                                "five tensors wherever torch's allocator put them"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_run,
                         "kernel_us": round(kernel_all_ms * 1e3, 2), "kernel_us_steady": round(kernel_ms * 1e3, 2),
                         "kernel_us_cold": round(cold_kernel_ms * 1e3, 2)},
            "per_rank": {"kernel_us": [round(u, 2) for u in per_rank_us],
                         "light_block_broadcast_us": None if bcast_us is None else round(bcast_us, 1)},
        }
        if not args.no_cpu_baseline:
            line["parity"] = parity_of_timed_output(plans[0], map_sets[0], args.size)
            if world == 1:
                line["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.cpu_passes)
        print(json.dumps(line), flush=True)
    if distributed:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
