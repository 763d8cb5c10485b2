#!/bin/bash
# Runs HERE after tools/r06_collect_on_box.sh ran on the GPU box: gpurun_out/{r06,prof_r06,prof_pipeline_r06,prof_pipeline_r06_s2,chain_work_new.json}
# -> profiles/r06_* and the two stamped files profiles/{traffic,chain_work}_latest.json.
set -e
cd "$(dirname "$0")/.."
G=gpurun_out/r06
NOTE=${1:-"Round 6, final build."}
python tools/make_traffic_json.py gpurun_out/prof_r06
cp gpurun_out/chain_work_new.json profiles/chain_work_latest.json
python tools/summarize_profile.py gpurun_out/prof_r06 profiles/r06_kernel.md "$NOTE \`python bench.py\` under rocprofv3 (tools/profile_bench.sh r06)."
cp gpurun_out/prof_r06/stats/bench_kernel_stats.csv profiles/r06_kernel_rocprofv3_kernel_stats.csv
P100=$G/pipeline_100k.json; [ -f $P100 ] || P100=$G/pipeline_scheme2.json
python tools/summarize_pipeline_profile.py gpurun_out/prof_pipeline_r06_s2 profiles/r06_full_pipeline_kernels_scheme2.md $G/pipeline_scheme2.json $P100 $G/pipeline_pmc_scheme2.txt "$NOTE Eigensolver scheme 2 (the C++ facade's default)."
cp gpurun_out/prof_pipeline_r06_s2/pipe_kernel_stats.csv profiles/r06_full_pipeline_scheme2_rocprofv3_kernel_stats.csv
python tools/summarize_pipeline_profile.py gpurun_out/prof_pipeline_r06 profiles/r06_full_pipeline_kernels.md $G/pipeline_scheme0.json $P100 $G/pipeline_pmc.txt "$NOTE Eigensolver scheme 0 (the C ABI's default)."
cp gpurun_out/prof_pipeline_r06/pipe_kernel_stats.csv profiles/r06_full_pipeline_rocprofv3_kernel_stats.csv
cp $G/pipeline_pmc.txt profiles/r06_pipeline_pmc.txt
cp $G/pipeline_pmc_scheme2.txt profiles/r06_pipeline_pmc_scheme2.txt
cp $G/bench.json profiles/r06_bench.json
cp $G/config4_group_form_ab.jsonl profiles/r06_config4_group_form_ab.jsonl
cp $G/numeric_jacobian_config4.jsonl profiles/r06_numeric_jacobian_config4.jsonl
[ -f $G/config4_pmc.txt ] && cp $G/config4_pmc.txt profiles/r06_config4_pmc.txt
for s in 0 1 2; do cp $G/pipeline_scheme$s.json profiles/r06_pipeline_stages_scheme$s.json; done
for f in bench_kitti bench_kitti_chain bench_kitti_chain_scheme0 pipeline_100k pipeline_kitti pipeline_parity_100k streaming odometry_options_parity \
         bench_single_process_1x bench_single_process_2x; do [ -f $G/$f.json ] && cp $G/$f.json profiles/r06_$f.json; done
[ -f $G/bench_kitti_chain_1.json ] && cp $G/bench_kitti_chain_1.json profiles/r06_bench_kitti_chain_one_at_a_time.json
[ -f $G/bench_kitti_chain_2ranks.json ] && grep -v '^\[Gloo\]' $G/bench_kitti_chain_2ranks.json > profiles/r06_bench_kitti_chain_2ranks.json
for f in residual_families full_batch_parity solve_latency; do [ -f $G/$f.jsonl ] && cp $G/$f.jsonl profiles/r06_$f.jsonl; done
git status --short profiles | head -40
