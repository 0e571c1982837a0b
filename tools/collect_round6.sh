#!/bin/bash
# Round-6 evidence (run ON the MI355X box).  From the dev container:
#     HEAD=$(git rev-parse HEAD); gpurun --timeout 1150 -- "PBR_GIT_HEAD=$HEAD bash tools/collect_round6.sh r6x headline"
#     ... "PBR_GIT_HEAD=$HEAD bash tools/collect_round6.sh r6y configs"      ... r6z kernels      ... r6w examples
# Every collection starts by writing <out>/stamp.json (the commit, sha256 of libpbr_hip.so, the digest of the sources the library
# was built from, the digest of the sources on the box) and REFUSES to measure a library that is not what these sources build
# (round 3's committed rocprof summary predated the last kernel change).  tools/stamp_profiles.py copies a
# collection into profiles/ with the stamp written into every file.
set -u
TAG=${1:-r6x}
WHAT=${2:-headline}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
python3 - "$OUT/stamp.json" <<'PY' || { echo "REFUSED: libpbr_hip.so was not built from the sources on this box (rebuild: python -c 'import __graft_entry__ as g; g.build()')"; exit 3; }
import json, sys, time
import torch
from pypbr_amd import _native as N
stamp = N.build_stamp()
stamp["collected_utc"] = time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())
stamp["device"] = torch.cuda.get_device_name(0) if torch.cuda.is_available() else None
json.dump(stamp, open(sys.argv[1], "w"), indent=1)
print(json.dumps(stamp))
sys.exit(1 if stamp["stale"] else 0)
PY
export TMPDIR=/tmp
case "$WHAT" in
headline)
    bash "$R/tools/collect_evidence.sh" "$TAG" > "$OUT/collect_evidence.log" 2>&1
    tail -3 "$OUT/collect_evidence.log"
    ;;
configs)
    cd /tmp
    timeout 300 python3 "$R/bench.py" --config 1 > "$OUT/bench_c1.json" 2> "$OUT/bench_c1.err"
    echo "config 1 done: $(head -c 300 "$OUT/bench_c1.json")"
    for c in 3 4 5; do
        timeout 300 python3 "$R/bench.py" --config $c > "$OUT/bench_c$c.json" 2> "$OUT/bench_c$c.err"
        timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c$c" -o run -- \
            python3 "$R/bench.py" --config $c --no-cpu-baseline > "$OUT/trace_c$c.log" 2>&1
        echo "config $c done: $(head -c 300 "$OUT/bench_c$c.json")"
    done
    ;;
kernels)
    bash "$R/tools/collect_kernels.sh" "$TAG" > "$OUT/collect_kernels.log" 2>&1
    tail -3 "$OUT/collect_kernels.log"
    # the bare access patterns of the resize family beside its kernels, on the same box (tools/membench.hip)
    [ -x "$R/tools/bin/membench" ] || { mkdir -p "$R/tools/bin"; /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 "$R/tools/membench.hip" -o "$R/tools/bin/membench" 2> /dev/null; }
    timeout 120 "$R/tools/bin/membench" 4096 resize > "$OUT/membench_resize.txt" 2>&1
    cat "$OUT/membench_resize.txt"
    ;;
examples)
    cd /tmp
    timeout 300 python3 "$R/bench.py" --example both > "$OUT/examples.jsonl" 2> "$OUT/examples.err"
    cat "$OUT/examples.jsonl"
    for e in brdf blend; do
        timeout 300 rocprofv3 --memory-copy-trace --kernel-trace --stats --output-format csv -d "$OUT/copytrace_$e" -o run -- \
            python3 "$R/tools/example_bench.py" --example $e --repeat 1 --no-cpu > "$OUT/copytrace_$e.log" 2>&1
        cat "$OUT/copytrace_$e"/*memory_copy_stats.csv
    done
    ;;
ab)
    # HEAD against earlier revisions in ONE process (tools/ab_revisions.sh built them into tools/bin/<rev> in the dev container)
    timeout 600 python3 "$R/tools/ab_revisions.py" --revs "${PBR_AB_REVS:-r2=tools/bin/8600504,pre5=tools/bin/1dbff36,head=.}" --rounds 7 --launches 500 \
        --out "$OUT/ab_revisions.json" > "$OUT/ab_revisions.log" 2>&1
    cat "$OUT/ab_revisions.log"
    ;;
*)
    echo "unknown collection $WHAT"; exit 2 ;;
esac
find "$OUT" -name "*.csv" | head -20
