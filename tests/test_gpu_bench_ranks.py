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


def _bench(*flags, gpus=1, timeout=900):
    env = dict(os.environ)
    shared = torch.cuda.device_count() < gpus
    if shared:
        env["PBR_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus)] + [str(f) for f in flags]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout
    assert run.stdout.strip() == lines[0], run.stdout          # ONE line and nothing else on stdout (RCCL's version banner goes to stderr)
    return json.loads(lines[0]), shared


def test_config4_strong_scaled_over_two_ranks_at_reduced_size():
    """`python bench.py --gpus 2 --config 4` (BASELINE.json configs[3]: B=512 1024^2 over the GPUs of a node) at a reduced
    size: every rank generates and owns partition(B, H, 2, rank)'s materials, the light block comes from rank 0's broadcast
    inside distributed.cook_torrance_sharded, the line reports the WHOLE job (strong scaling) and every rank's share."""
    line, shared = _bench("--config", 4, "--batch", 10, "--size", 256, "--steps", 6, "--warmup", 2, "--settle", 4, gpus=2)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["config"] == 4
    assert line["config"]["global_batch"] == 10 and line["config"]["pixels_per_step"] == 10 * 256 * 256
    assert line["per_rank"]["shard_batch_rows"] == [[0, 5, 0, 256], [5, 10, 0, 256]]
    assert line["per_rank"]["pixels_per_launch"] == [5 * 256 * 256] * 2 and all(u > 0 for u in line["per_rank"]["kernel_us"])
    assert line["per_rank"]["light_block_broadcast_us"] > 0
    assert abs(line["value"] - 10 * 256 * 256 * 6 / (line["ms_per_step"] * 6 * 1e-3) / 1e6) <= 0.01 * line["value"]
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1 and line["config"]["bytes_per_pixel"] == 44
    par = line["parity"]                                         # bands of rank 0's timed output against both oracles
    assert par["max_abs_err_vs_fp64_oracle"] <= 2e-6 and par["values"] > 0
    assert ("TEST HOOK" in line["config"]["parallelism"]) == shared


def test_config4_row_bands_when_the_batch_is_smaller_than_the_world():
    """One material over two ranks: the shard is a row band, generated as the band of the same seeded material."""
    line, _ = _bench("--config", 4, "--batch", 1, "--size", 256, "--steps", 4, "--warmup", 1, "--settle", 2, gpus=2)
    assert line["per_rank"]["shard_batch_rows"] == [[0, 1, 0, 128], [0, 1, 128, 256]]
    assert line["parity"]["max_abs_err_vs_fp64_oracle"] <= 2e-6


@pytest.mark.parametrize("config,flags,kernel,bpp", [
    (3, ("--batch", 4, "--size", 512), "ct_directional_converted_f32_f32_v4", 44),
    (5, ("--batch", 4, "--size", 512), "ctb_point_metallic_f16_f32_v2_b4", 28),
    (4, ("--batch", 16, "--size", 256), "ct_point_metallic_f32_f32_v4", 44),
])
def test_named_configs_on_one_gpu_at_reduced_size(config, flags, kernel, bpp):
    line, _ = _bench("--config", config, *flags, "--steps", 4, "--warmup", 1, "--settle", 2, "--no-cpu-baseline")
    assert line["n_gpus"] == 1 and line["config"]["config"] == config and line["config"]["kernel"] == kernel
    assert line["config"]["bytes_per_pixel"] == bpp and line["scaling"] == "strong"
    if config == 3:
        assert line["specular_is_srgb_false"]["kernel_us"] > 0
    if config == 5:
        assert line["roofline_valu"]["bound"] == "valu" and line["config"]["lights"] == 16 and line["config"]["map_dtype"] == "f16"
    if config == 4:
        assert line["per_gpu_share_of_8"]["batch"] == 2 and line["per_gpu_share_of_8"]["kernel_us"] > 0


def _cpu_baseline_is_well_formed(cb):
    """STRUCTURE only -- no assertion here depends on how fast the host is (VERDICT r5 #1b): a record either holds measured legs with
    a value taken from them, or says which legs it skipped."""
    assert cb["kind"] == "port" and cb["cpu_model"] and cb["unit"] == "Mpixels/s"
    assert 1 <= cb["usable_cores"] <= cb["host_cores"]
    assert {e["threads"] for e in cb["table"]} <= {1, cb["usable_cores"]}
    if cb["table"]:
        assert all(e["ms"] >= 0 and e["size"] > 0 for e in cb["table"])
        assert cb["value"] in {e["Mpixels_per_s"] for e in cb["table"]} and cb["cores"] in (1, cb["usable_cores"])
    else:
        assert cb["value"] is None and "skipped" in cb["sample"]


def test_named_config_parity_leg_at_reduced_size():
    """The checker's legs of configs 3 and 5 (converted / 16 lights on fp16 maps): bands of the timed output against both oracles."""
    for config, flags in ((3, ("--batch", 2, "--size", 256)), (5, ("--batch", 2, "--size", 256))):
        line, _ = _bench("--config", config, *flags, "--steps", 3, "--warmup", 1, "--settle", 1, "--cpu-budget", 2)
        assert line["parity"]["max_abs_err_vs_fp64_oracle"] <= 2e-6, (config, line["parity"])
        _cpu_baseline_is_well_formed(line["cpu_baseline"])


def test_rccl_runs_under_the_suite_with_one_rank():
    """the "nccl" (= RCCL) branch of the N > 1 path executed by every GPU test run -- no PBR_BENCH_SHARE_GPU,
    so the rank started by torch.distributed.run goes through init_process_group("nccl"), the 404-byte light-block broadcast,
    the barriers, the all-reduce (MAX of the times) and the all-gather of the per-rank records, all on RCCL; the line says which
    backend served it and how many ranks it saw (so that a SCALE record answers "did RCCL see N ranks?" by itself)."""
    env = {k: v for k, v in os.environ.items() if k != "PBR_BENCH_SHARE_GPU"}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--config", "4", "--batch", "8", "--size", "256",
           "--steps", "4", "--warmup", "1", "--settle", "2", "--no-cpu-baseline"]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout
    line = json.loads(lines[0])
    assert line["per_rank"]["backend"] == "nccl" and line["per_rank"]["ranks_seen"] == 1 and line["n_gpus"] == 1
    assert line["per_rank"]["light_block_broadcast_us"] > 0 and line["per_rank"]["shard_batch_rows"] == [[0, 8, 0, 256]]
    assert "TEST HOOK" not in line["config"]["parallelism"]


def test_plain_single_gpu_line_forms_an_rccl_group_of_one():
    """`python bench.py` as the driver types it (no torch.distributed.run): the rank forms a one-rank RCCL group in-process, so the
    headline line too has been through the collectives and carries backend / ranks_seen; --no-rccl is the line without any."""
    line, _ = _bench("--size", 512, "--steps", 5, "--warmup", 2, "--settle", 5, "--no-cpu-baseline")
    assert line["per_rank"]["backend"] == "nccl" and line["per_rank"]["ranks_seen"] == 1 and line["per_rank"]["light_block_broadcast_us"] > 0
    plain, _ = _bench("--size", 512, "--steps", 5, "--warmup", 2, "--settle", 5, "--no-cpu-baseline", "--no-rccl")
    assert plain["per_rank"]["backend"] is None and plain["per_rank"]["ranks_seen"] == 1 and plain["per_rank"]["light_block_broadcast_us"] is None


def test_cpu_baseline_runs_on_the_cores_the_process_has():
    """no leg with more threads than the process may use; an exhausted budget gives a skipped record, not a lost line."""
    line, _ = _bench("--size", 256, "--steps", 3, "--warmup", 1, "--settle", 1, "--cpu-budget", 6)
    _cpu_baseline_is_well_formed(line["cpu_baseline"])
    assert line["cpu_baseline"]["table"], "a positive budget always measures the first leg, however slow the host"
    none, _ = _bench("--size", 256, "--steps", 3, "--warmup", 1, "--settle", 1, "--cpu-budget", 0)
    assert none["cpu_baseline"]["value"] is None and none["cpu_baseline"]["table"] == [] and "skipped" in none["cpu_baseline"]["sample"]
