"""bench.py's multi-rank path end to end: `python bench.py --gpus 2` typed as it stands starts its own ranks (the parent
never touches the GPU), every rank times its own material, the MAX over ranks is taken, rank 0 prints ONE JSON line
with per-rank kernel times and the light-block broadcast latency.  On a box with fewer than 2 GPUs the ranks share
cuda:0 and the small collectives go over gloo (PBR_BENCH_SHARE_GPU=1: RCCL refuses two ranks on one device); the
timed data path has no collective either way."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_starts_its_own_ranks_and_reports_the_whole_job():
    env = dict(os.environ)
    shared = torch.cuda.device_count() < 2
    if shared:
        env["PBR_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--settle", "10",
           "--size", "1024", "--no-cpu-baseline"]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout                                   # rank 0 only, one line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 10 and line["warmup"] == 3 and line["scaling"] == "weak"
    assert len(line["per_rank"]["kernel_us"]) == 2 and all(u > 0 for u in line["per_rank"]["kernel_us"])
    assert line["per_rank"]["light_block_broadcast_us"] > 0
    # whole-job throughput: both ranks' pixels over the slowest rank's time
    assert abs(line["value"] - 2 * 1024 * 1024 * 10 / (line["ms_per_step"] * 10 * 1e-3) / 1e6) <= 0.01 * line["value"]
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1
    assert ("TEST HOOK" in line["config"]["parallelism"]) == shared
