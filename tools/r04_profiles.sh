#!/bin/bash
# Runs ON THE GPU BOX: everything profiles/r04_* is made from.  Outputs under gpurun_out/r04/.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04
mkdir -p $OUT
cd $REPO
# 1. the headline: bench line, kernel trace + PMC passes (HBM traffic, SQ counters)
python bench.py > $OUT/bench.json 2> $OUT/bench.err
bash tools/profile_bench.sh r04 > $OUT/profile_bench.log 2>&1
# 2. the chain: stage timings, kernel trace, SQ counters of the front kernels
python tools/bench_pipeline.py 20000 > $OUT/pipeline.json 2> /dev/null
python tools/bench_pipeline.py 100000 > $OUT/pipeline_100k.json 2> /dev/null
python tools/bench_pipeline_kitti.py > $OUT/pipeline_kitti.json 2> /dev/null
bash tools/profile_pipeline.sh r04 > $OUT/profile_pipeline.log 2>&1
bash tools/pmc_pipeline.sh r04 > $OUT/pipeline_pmc.txt 2>&1
# 3. other bench lines
python bench.py --workload kitti_all --steps 50 --warmup 10 > $OUT/bench_kitti.json 2> /dev/null
python bench.py --workload kitti_all --chain --steps 30 --warmup 5 > $OUT/bench_kitti_chain.json 2> /dev/null
python bench.py --workload kitti_all --chain --steps 30 --warmup 5 --in-flight 1 > $OUT/bench_kitti_chain_1.json 2> /dev/null
python bench.py --gpus 2 --share-gpu --workload kitti_all --chain --steps 10 --warmup 3 > $OUT/bench_kitti_chain_2ranks.json 2> /dev/null
# 4. residual families with roofline blocks
python tools/bench_modes.py 100000 > $OUT/residual_families.jsonl 2> /dev/null
# 5. RANSAC forms on one box (A/B record)
for f in 2 3; do PNEC_RANSAC_FORM=$f python tools/ab_ransac_forms.py uniform 20000 >> $OUT/ransac_forms.jsonl 2> /dev/null; done
# 6. parity at scale
python tools/verify_full_batch.py 100000 target > $OUT/full_batch_parity.jsonl 2> /dev/null
python tools/verify_pipeline.py 100000 > $OUT/pipeline_parity_100k.json 2> $OUT/pipeline_parity_100k.err
python tools/bench_streaming.py > $OUT/streaming.json 2> /dev/null
# 7. one PNEC::Solve per frame through the facade (default options / the odometry's / the timed overload)
for n in 100 512 700 2000; do for m in "" vo timed; do ./pnec_amd/pnec_host_demo $n solve_latency 300 $m 2>/dev/null | tail -1 >> $OUT/solve_latency.jsonl; done; done
python tools/verify_odometry_options.py 2000 > $OUT/odometry_options_parity.json 2> /dev/null
ls -la $OUT
