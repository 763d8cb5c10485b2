#!/bin/bash
# Runs HERE after tools/r05_collect_on_box.sh ran on the GPU box: gpurun_out/{r05,prof_r05,prof_pipeline_r05,chain_work_new.json}
# -> profiles/r05_* and the two stamped files profiles/{traffic,chain_work}_latest.json.
set -e
cd "$(dirname "$0")/.."
G=gpurun_out/r05
NOTE=${1:-"Round 5, final build."}
python tools/make_traffic_json.py gpurun_out/prof_r05
cp gpurun_out/chain_work_new.json profiles/chain_work_latest.json
python tools/summarize_profile.py gpurun_out/prof_r05 profiles/r05_kernel.md "$NOTE \`python bench.py\` under rocprofv3 (tools/profile_bench.sh r05)."
cp gpurun_out/prof_r05/stats/bench_kernel_stats.csv profiles/r05_kernel_rocprofv3_kernel_stats.csv
python tools/summarize_pipeline_profile.py gpurun_out/prof_pipeline_r05 profiles/r05_full_pipeline_kernels.md $G/pipeline_scheme0.json $G/pipeline_100k.json $G/pipeline_pmc.txt "$NOTE"
cp gpurun_out/prof_pipeline_r05/pipe_kernel_stats.csv profiles/r05_full_pipeline_rocprofv3_kernel_stats.csv
for f in bench bench_kitti bench_kitti_chain pipeline_100k pipeline_kitti pipeline_parity_100k streaming odometry_options_parity odometry_options_parity_30pct_mismatches \
         bench_single_process_1x bench_single_process_2x; do cp $G/$f.json profiles/r05_$f.json; done
for s in 0 1 2; do cp $G/pipeline_scheme$s.json profiles/r05_pipeline_stages_scheme$s.json; done
cp $G/bench_kitti_chain_1.json profiles/r05_bench_kitti_chain_one_at_a_time.json
grep -v '^\[Gloo\]' $G/bench_kitti_chain_2ranks.json > profiles/r05_bench_kitti_chain_2ranks.json
for f in residual_families full_batch_parity solve_latency; do cp $G/$f.jsonl profiles/r05_$f.jsonl; done
cp $G/pipeline_pmc.txt profiles/r05_pipeline_pmc.txt
cp $G/trace_2ranks/t_kernel_stats.csv profiles/r05_two_ranks_one_gpu_rocprofv3_kernel_stats.csv
git status --short profiles | head -40
