#!/bin/bash
# Runs HERE after tools/r04_profiles.sh ran on the GPU box: gpurun_out/{r04,prof_r04,prof_pipeline_r04} -> profiles/r04_*.
set -e
cd "$(dirname "$0")/.."
G=gpurun_out/r04
python tools/make_traffic_json.py gpurun_out/prof_r04
python tools/summarize_profile.py gpurun_out/prof_r04 profiles/r04_kernel.md "Round 4, final build (cost-first refinement): \`python bench.py\` under rocprofv3 (tools/profile_bench.sh r04)."
cp gpurun_out/prof_r04/stats/bench_kernel_stats.csv profiles/r04_kernel_rocprofv3_kernel_stats.csv
python tools/summarize_pipeline_profile.py gpurun_out/prof_pipeline_r04 profiles/r04_full_pipeline_kernels.md $G/pipeline.json $G/pipeline_100k.json $G/pipeline_pmc.txt "Round 4, final build."
cp gpurun_out/prof_pipeline_r04/pipe_kernel_stats.csv profiles/r04_full_pipeline_rocprofv3_kernel_stats.csv
for f in bench bench_kitti bench_kitti_chain pipeline pipeline_100k pipeline_kitti pipeline_parity_100k streaming; do cp $G/$f.json profiles/r04_$f.json; done
cp $G/bench_kitti_chain_1.json profiles/r04_bench_kitti_chain_one_at_a_time.json
grep -v '^\[Gloo\]' $G/bench_kitti_chain_2ranks.json > profiles/r04_bench_kitti_chain_2ranks.json
cp $G/residual_families.jsonl $G/ransac_forms.jsonl $G/full_batch_parity.jsonl profiles/ 2>/dev/null && \
  for f in residual_families ransac_forms full_batch_parity; do mv profiles/$f.jsonl profiles/r04_$f.jsonl; done
cp $G/pipeline_pmc.txt profiles/r04_pipeline_pmc.txt
[ -s $G/odometry_options_parity.json ] && cp $G/odometry_options_parity.json profiles/r04_odometry_options_parity.json
[ -s $G/solve_latency.jsonl ] && cp $G/solve_latency.jsonl profiles/r04_solve_latency.jsonl
git status --short profiles | head -40
