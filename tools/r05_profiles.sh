#!/bin/bash
# Runs ON THE GPU BOX: everything profiles/r05_* is made from.  Outputs under gpurun_out/r05/.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
cd $REPO
# 1. the headline: bench line, kernel trace + PMC passes (HBM traffic, SQ counters)
python bench.py > $OUT/bench.json 2> $OUT/bench.err
bash tools/profile_bench.sh r05 > $OUT/profile_bench.log 2>&1
# 2. the chain: stage timings per eigensolver scheme, kernel trace, SQ counters of the front kernels
for s in 0 1 2; do python tools/bench_pipeline.py 20000 512 $s > $OUT/pipeline_scheme$s.json 2> /dev/null; done
python tools/bench_pipeline.py 100000 > $OUT/pipeline_100k.json 2> /dev/null
python tools/bench_pipeline_kitti.py > $OUT/pipeline_kitti.json 2> /dev/null
bash tools/profile_pipeline.sh r05 > $OUT/profile_pipeline.log 2>&1
bash tools/pmc_pipeline.sh r05 > $OUT/pipeline_pmc.txt 2>&1
# 3. other bench lines
python bench.py --workload kitti_all --steps 50 --warmup 10 > $OUT/bench_kitti.json 2> /dev/null
python bench.py --workload kitti_all --chain --steps 30 --warmup 5 > $OUT/bench_kitti_chain.json 2> /dev/null
python bench.py --workload kitti_all --chain --steps 30 --warmup 5 --in-flight 1 > $OUT/bench_kitti_chain_1.json 2> /dev/null
python bench.py --gpus 2 --share-gpu --workload kitti_all --chain --steps 10 --warmup 3 > $OUT/bench_kitti_chain_2ranks.json 2> /dev/null
python bench.py --gpus 2 --share-gpu --single-process --pairs 50000 --steps 10 --warmup 3 > $OUT/bench_single_process_2x.json 2> /dev/null
python bench.py --gpus 1 --single-process --pairs 100000 --steps 10 --warmup 3 > $OUT/bench_single_process_1x.json 2> /dev/null
# 4. residual families with roofline blocks
python tools/bench_modes.py 100000 > $OUT/residual_families.jsonl 2> /dev/null
# 5. parity at scale
python tools/verify_full_batch.py 100000 target > $OUT/full_batch_parity.jsonl 2> /dev/null
python tools/verify_pipeline.py 100000 > $OUT/pipeline_parity_100k.json 2> $OUT/pipeline_parity_100k.err
python tools/verify_eigensolver_schemes.py 2000 > $OUT/odometry_options_parity.json 2> /dev/null
python tools/verify_eigensolver_schemes.py 2000 0.3 > $OUT/odometry_options_parity_30pct_mismatches.json 2> /dev/null
python tools/bench_streaming.py > $OUT/streaming.json 2> /dev/null
# 6. one PNEC::Solve per frame through the facade (default options / the odometry's / the timed overload)
rm -f $OUT/solve_latency.jsonl
for n in 100 512 700 2000; do for m in default vo timed; do for s in 2 0; do ./pnec_amd/pnec_host_demo $n solve_latency 300 $m $s 2>/dev/null | tail -1 >> $OUT/solve_latency.jsonl; done; done; done
# 7. the two-ranks-on-one-GPU run under a kernel trace (what the 7x per step is made of)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_2ranks -o t -- python $REPO/bench.py --gpus 2 --share-gpu --workload kitti_all --chain --steps 6 --warmup 2 > $OUT/trace_2ranks.log 2>&1
find $OUT -type f -size +4M -delete
ls -la $OUT
