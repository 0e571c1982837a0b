#!/usr/bin/env python3
"""A/B of whole source REVISIONS in one process on one box: each revision is a copy of the
package built at a side path (`tools/ab_revisions.sh` extracts `git archive REV pypbr_amd include` to
tools/bin/REV/ and runs its own Makefile), imported here under an alias so that every revision fills ITS OWN
descriptor layout with ITS OWN host code and launches ITS OWN libpbr_hip.so.  All revisions see the same input
tensors; rounds are interleaved (rev A, rev B, rev C, rev A, ...) so that clock / box drift hits all alike.

    python tools/ab_revisions.py --revs r2=tools/bin/8600504,pre5=tools/bin/1dbff36,head=. \
        --cases headline,backward,config4 --rounds 7 --launches 500 --out profiles/r04_ab_revisions.json
"""
import argparse
import ctypes
import hashlib
import importlib.util
import json
import os
import statistics
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_material  # noqa: E402


def load_revision(alias, path):
    pkg_dir = os.path.join(os.path.abspath(path), "pypbr_amd")
    name = "pypbr_rev_" + alias
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    F = importlib.import_module(name + ".functional")
    N = importlib.import_module(name + "._native")
    if hasattr(F, "USE_TORCH_OPS"):
        F.USE_TORCH_OPS = False                       # the registered operators exist once per process: every revision through its ctypes plan
    lib = N.lib()
    sha = hashlib.sha256(open(N.LIB_PATH, "rb").read()).hexdigest()[:16]
    return dict(alias=alias, F=F, N=N, lib=lib, lib_path=os.path.relpath(N.LIB_PATH, ROOT), lib_sha256_16=sha, abi=lib.pbr_abi_version())


def build_case(case, rev, sets, dev):
    """-> (launch(i), kernel name, bytes per launch) for this revision on the shared input tensors."""
    F, N, lib = rev["F"], rev["N"], rev["lib"]
    stream = torch.cuda.current_stream(dev).cuda_stream
    if case in ("headline", "backward"):
        kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
        plans = []
        for s in sets["one4096"]:
            *packed, out = F.pack_maps(*s, reserve_output=True)
            plans.append(F.plan_cook_torrance(*packed, out=out, **kw))
        if case == "headline":
            fn = lib.pbr_cook_torrance
            return (lambda i: fn(ctypes.byref(plans[i % len(plans)].desc), stream)), plans[0].kernel_name, 44 * 4096 * 4096
        g = torch.rand(1, 3, 4096, 4096, device=dev)
        grads = [torch.empty(1, c, 4096, 4096, device=dev) for c in (3, 3, 1, 1)]
        ptrs = [t.data_ptr() for t in grads] + [None]
        fn = lib.pbr_cook_torrance_backward
        return (lambda i: fn(ctypes.byref(plans[i % len(plans)].desc), g.data_ptr(), *ptrs, stream)), "backward:" + plans[0].kernel_name, 76 * 4096 * 4096
    if case in ("backward_f16", "loss_f16"):              # the streamed kernels of the fp16-map family
        kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
        maps = [t.half() for t in sets["one4096"][0]]
        plan = F.plan_cook_torrance(*maps, **kw)
        g = torch.rand(1, 3, 4096, 4096, device=dev)
        grads = [torch.empty(1, c, 4096, 4096, device=dev, dtype=torch.float16) for c in (3, 3, 1, 1)]
        ptrs = [t.data_ptr() for t in grads] + [None]
        keep = (maps, g, grads)
        if case == "backward_f16":
            fn = lib.pbr_cook_torrance_backward
            return (lambda i, keep=keep: fn(ctypes.byref(plan.desc), g.data_ptr(), *ptrs, stream)), "backward fp16 maps", 44 * 4096 * 4096
        loss = torch.empty((), device=dev)
        ws = torch.empty(max(1, lib.pbr_mse_step_workspace_bytes(ctypes.byref(plan.desc)) // 4), device=dev)
        fn = lib.pbr_cook_torrance_mse_step
        return (lambda i, keep=keep: fn(ctypes.byref(plan.desc), g.data_ptr(), *ptrs, loss.data_ptr(), ws.data_ptr(), stream)), "loss step fp16 maps", 44 * 4096 * 4096
    if case == "config4":
        kw = dict(view_dir=[0, 0, 1], light=[0.1, 0.1, 1.0], light_intensity=[1, 1, 1], light_type="point", light_size=1.0)
        plan = F.plan_cook_torrance(*sets["b64_1024"], **kw)
        fn = lib.pbr_cook_torrance
        return (lambda i: fn(ctypes.byref(plan.desc), stream)), plan.kernel_name, 44 * 64 * 1024 * 1024
    raise SystemExit("unknown case " + case)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--revs", required=True, help="alias=path[,alias=path...]; path holds pypbr_amd/ with its built libpbr_hip.so")
    ap.add_argument("--cases", default="headline,backward,config4")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--launches", type=int, default=500)
    ap.add_argument("--settle", type=int, default=400)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    revs = [load_revision(*kv.split("=")) for kv in args.revs.split(",")]
    sets = {"one4096": [synth_material(4096, dev, 1234 + i) for i in range(3)]}
    one = synth_material(1024, dev, 99)
    sets["b64_1024"] = [torch.stack([t] * 64).contiguous() for t in one]
    for b in range(1, 64):                                  # distinct materials (cheap: perturb, keep ranges)
        sets["b64_1024"][0][b].mul_(1.0 - 0.005 * b)
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        head = os.environ.get("PBR_GIT_HEAD", "unknown")     # the GPU box has no .git: the caller passes it
    record = dict(tool="tools/ab_revisions.py", git_head=head, device=torch.cuda.get_device_name(0), rounds=args.rounds, launches=args.launches,
                  revisions=[{k: r[k] for k in ("alias", "lib_path", "lib_sha256_16", "abi")} for r in revs], cases=[])
    for case in args.cases.split(","):
        launchers = [build_case(case, r, sets, dev) for r in revs]
        # same values out of every revision (bit for bit where the arithmetic did not change; reported, not asserted)
        for (launch, _, _) in launchers:
            for i in range(args.settle // len(launchers)):
                assert launch(i) == 0
        torch.cuda.synchronize()
        times = [[] for _ in revs]
        for rnd in range(args.rounds):
            order = list(range(len(revs)))
            order = order[rnd % len(revs):] + order[:rnd % len(revs)]      # rotate who goes first
            for k in order:
                launch = launchers[k][0]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(args.launches):
                    launch(i)
                e1.record()
                e1.synchronize()
                times[k].append(e0.elapsed_time(e1) * 1e3 / args.launches)
        entry = dict(case=case, results=[])
        base = statistics.median(times[-1])
        for r, t, (_, kname, nbytes) in zip(revs, times, launchers):
            med = statistics.median(t)
            entry["results"].append(dict(alias=r["alias"], kernel=kname, us_rounds=[round(x, 2) for x in t], us_median=round(med, 2),
                                         us_min=round(min(t), 2), tb_per_s=round(nbytes / med / 1e6, 3), vs_last=round(med / base, 4)))
            print("%-9s %-6s median %8.2f us  min %8.2f  (%5.3f TB/s)  x%.4f of %s   %s" % (case, r["alias"], med, min(t), nbytes / med / 1e6, med / base,
                                                                                              revs[-1]["alias"], " ".join("%.1f" % x for x in t)), flush=True)
        record["cases"].append(entry)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(record, f, indent=1)
        print("wrote", args.out)


if __name__ == "__main__":
    main()
