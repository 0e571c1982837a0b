# LDS / SQ counters of the resize-gradient kernel (run ON the GPU box): bash tools/prof_resize_bwd.sh [tag]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-rzb}
mkdir -p $R/gpurun_out/$T
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$T/pmc_LDS -o run -- python3 $R/tools/run_kernels.py 3 resize_bwd > $R/gpurun_out/$T/lds.log 2>&1
python3 - $R/gpurun_out/$T/pmc_LDS <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); n[k] += 1
for k, c in acc.items():
    print(k, n[k], {a: round(b / n[k]) for a, b in c.items()})
PY
