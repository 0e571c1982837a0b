# SQ / LDS counters of the resize-gradient kernels (run ON the GPU box): bash tools/prof_resize_bwd.sh [tag]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-rzb}
mkdir -p $R/gpurun_out/$T
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/$T/pmc_SQ -o run -- python3 $R/tools/run_kernels.py 3 resize_bwd > $R/gpurun_out/$T/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/$T/pmc_MIX -o run -- python3 $R/tools/run_kernels.py 3 resize_bwd > $R/gpurun_out/$T/mix.log 2>&1
python3 - $R/gpurun_out/$T <<'PY'
import csv, glob, sys, collections
for sub in ("pmc_SQ", "pmc_MIX"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for p in glob.glob(sys.argv[1] + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"][:70]
            if "resize" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (k, r["Dispatch_Id"]) not in seen:
                seen.add((k, r["Dispatch_Id"])); n[k] += 1
    for k, c in acc.items():
        print(sub, k, n[k], {a: round(b / n[k]) for a, b in c.items()})
PY
