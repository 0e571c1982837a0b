#!/bin/bash
# round-2 first GPU pass: tests, bench (plain + self-spawned ranks), probes
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r2a
mkdir -p "$OUT"
cd "$R"
rocm-smi --showid 2>/dev/null | head -5 > "$OUT/smi.txt"
python3 -c "import torch; print('devices', torch.cuda.device_count())" > "$OUT/devices.txt" 2>&1
timeout 1500 python3 -m pytest tests -m gpu -x -q -s > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -5 "$OUT/pytest.log"
timeout 300 python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_driver.json" 2> "$OUT/bench_driver.err"
cat "$OUT/bench_driver.json"
timeout 300 python3 bench.py --spawn --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/bench_spawn.json" 2> "$OUT/bench_spawn.err"
echo "spawn rc=$?"; cat "$OUT/bench_spawn.json"; tail -3 "$OUT/bench_spawn.err"
timeout 60 python3 -c "
import torch, subprocess
torch.zeros(1, device='cuda'); torch.cuda.synchronize()
r = subprocess.run(['echo', 'child-after-hip-init-ok'], capture_output=True, text=True)
print('rc', r.returncode, r.stdout, r.stderr)
" > "$OUT/subprocess_probe.txt" 2>&1
cat "$OUT/subprocess_probe.txt"
