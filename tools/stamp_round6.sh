#!/bin/bash
# Collections of tools/collect_round6.sh -> profiles/r06_* (dev container, after the gpurun calls merged gpurun_out/):
#     bash tools/stamp_round6.sh r6Z          # expects gpurun_out/r6Z (headline), r6Z_c (configs), r6Z_k (kernels), r6Z_e (examples), r6Z_ab (ab)
# Derived files are made here from the raw rocprofv3 output (pmc_to_json, kernels_summary, configs_trace_summary); every file gets the
# collection's stamp (commit, library SHA-256, digest of its sources) written into it (tools/stamp_profiles.py).
set -eu
T=${1:?tag}
cd "$(dirname "$0")/.."
G=gpurun_out
python tools/pmc_to_json.py $G/$T r06_pmc_traffic.json > $G/$T/pmc_traffic.json
python tools/stamp_profiles.py $G/$T bench.json=r06_bench_n1.json trace/run_kernel_stats.csv=r06_kernel_stats.csv pmc_traffic.json=r06_pmc_traffic.json configs.jsonl=r06_configs_1gpu.jsonl
cp profiles/r06_pmc_traffic.json profiles/pmc_traffic.json
python tools/configs_trace_summary.py $G/${T}_c > $G/${T}_c/configs_kernel_trace.json
cat $G/${T}_c/bench_c1.json $G/${T}_c/bench_c3.json $G/${T}_c/bench_c4.json $G/${T}_c/bench_c5.json | grep '^{' > $G/${T}_c/bench_configs.jsonl
python tools/stamp_profiles.py $G/${T}_c bench_configs.jsonl=r06_bench_configs_1gpu.jsonl configs_kernel_trace.json=r06_configs_kernel_trace.json \
    trace_c3/run_kernel_stats.csv=r06_config3_kernel_stats.csv trace_c4/run_kernel_stats.csv=r06_config4_kernel_stats.csv trace_c5/run_kernel_stats.csv=r06_config5_kernel_stats.csv
python tools/kernels_summary.py $G/${T}_k > $G/${T}_k/kernels.json
python tools/stamp_profiles.py $G/${T}_k kernels.json=r06_kernels.json cases.jsonl=r06_kernels_cases.jsonl trace/run_kernel_stats.csv=r06_kernels_kernel_stats.csv membench_resize.txt=r06_membench_resize.txt
python tools/stamp_profiles.py $G/${T}_e examples.jsonl=r06_examples.jsonl copytrace_brdf/run_memory_copy_stats.csv=r06_example_brdf_memory_copy_stats.csv \
    copytrace_blend/run_memory_copy_stats.csv=r06_example_blend_memory_copy_stats.csv copytrace_brdf/run_kernel_stats.csv=r06_example_brdf_kernel_stats.csv \
    copytrace_blend/run_kernel_stats.csv=r06_example_blend_kernel_stats.csv
