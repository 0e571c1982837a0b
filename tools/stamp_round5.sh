#!/bin/bash
# Collections of tools/collect_round5.sh -> profiles/r05_* (dev container, after the gpurun calls merged gpurun_out/):
#     bash tools/stamp_round5.sh r5Z          # expects gpurun_out/r5Z (headline), r5Z_c (configs), r5Z_k (kernels), r5Z_e (examples), r5Z_ab (ab)
# Derived files are made here from the raw rocprofv3 output (pmc_to_json, kernels_summary, configs_trace_summary); every file gets the
# collection's stamp (commit, library SHA-256, digest of its sources) written into it (tools/stamp_profiles.py).
set -eu
T=${1:?tag}
cd "$(dirname "$0")/.."
G=gpurun_out
python tools/pmc_to_json.py $G/$T r05_pmc_traffic.json > $G/$T/pmc_traffic.json
python tools/stamp_profiles.py $G/$T bench.json=r05_bench_n1.json trace/run_kernel_stats.csv=r05_kernel_stats.csv pmc_traffic.json=r05_pmc_traffic.json configs.jsonl=r05_configs_1gpu.jsonl
cp profiles/r05_pmc_traffic.json profiles/pmc_traffic.json
python tools/configs_trace_summary.py $G/${T}_c > $G/${T}_c/configs_kernel_trace.json
cat $G/${T}_c/bench_c1.json $G/${T}_c/bench_c3.json $G/${T}_c/bench_c4.json $G/${T}_c/bench_c5.json | grep '^{' > $G/${T}_c/bench_configs.jsonl
python tools/stamp_profiles.py $G/${T}_c bench_configs.jsonl=r05_bench_configs_1gpu.jsonl configs_kernel_trace.json=r05_configs_kernel_trace.json \
    trace_c3/run_kernel_stats.csv=r05_config3_kernel_stats.csv trace_c4/run_kernel_stats.csv=r05_config4_kernel_stats.csv trace_c5/run_kernel_stats.csv=r05_config5_kernel_stats.csv
python tools/kernels_summary.py $G/${T}_k > $G/${T}_k/kernels.json
python tools/stamp_profiles.py $G/${T}_k kernels.json=r05_kernels.json cases.jsonl=r05_kernels_cases.jsonl trace/run_kernel_stats.csv=r05_kernels_kernel_stats.csv membench_resize.txt=r05_membench_resize.txt
python tools/stamp_profiles.py $G/${T}_e examples.jsonl=r05_examples.jsonl copytrace_brdf/run_memory_copy_stats.csv=r05_example_brdf_memory_copy_stats.csv \
    copytrace_blend/run_memory_copy_stats.csv=r05_example_blend_memory_copy_stats.csv copytrace_brdf/run_kernel_stats.csv=r05_example_brdf_kernel_stats.csv \
    copytrace_blend/run_kernel_stats.csv=r05_example_blend_kernel_stats.csv
