#!/bin/bash
# Runs ON THE GPU BOX: everything profiles/r06_* is made from.  Outputs under gpurun_out/r06/.
# usage: tools/r06_profiles.sh [quick]   (quick: the headline + the chain under scheme 2 only)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06
QUICK=${1:-}
mkdir -p $OUT
cd $REPO
# 1. the headline: bench line, kernel trace + PMC passes (HBM traffic, SQ counters)
python bench.py > $OUT/bench.json 2> $OUT/bench.err
bash tools/profile_bench.sh r06 > $OUT/profile_bench.log 2>&1
# 2. the chain: stage timings per eigensolver scheme; kernel trace + SQ counters of the front kernels under the facade's
#    default scheme (2) AND under the C ABI's (0)
for s in 2 0 1; do python tools/bench_pipeline.py 20000 512 $s > $OUT/pipeline_scheme$s.json 2> /dev/null; done
PNEC_ES_SCHEME=2 bash tools/profile_pipeline.sh r06_s2 > $OUT/profile_pipeline_s2.log 2>&1
PNEC_ES_SCHEME=2 bash tools/pmc_pipeline.sh r06_s2 > $OUT/pipeline_pmc_scheme2.txt 2>&1
bash tools/profile_pipeline.sh r06 > $OUT/profile_pipeline.log 2>&1
bash tools/pmc_pipeline.sh r06 > $OUT/pipeline_pmc.txt 2>&1
# 3. config 4 (multi-hypothesis): one-solve-per-block against the group form, same box
for g in 0 1; do PNEC_SOLVE_GROUPS=$g python tools/ab_config4.py 2>/dev/null | grep '^{' | sed "s/^{/{\"PNEC_SOLVE_GROUPS\": $g, /"; done > $OUT/config4_group_form_ab.jsonl
python tools/verify_numeric_jacobian.py $OUT/numeric_jacobian_config4.jsonl > /dev/null 2>&1
bash tools/pmc_config4.sh r06 > $OUT/config4_pmc.txt 2>&1
if [ -z "$QUICK" ]; then
python tools/bench_pipeline.py 100000 512 2 > $OUT/pipeline_100k.json 2> /dev/null
PNEC_ES_SCHEME=2 python tools/bench_pipeline_kitti.py > $OUT/pipeline_kitti.json 2> /dev/null
# 4. other bench lines
python bench.py --workload kitti_all --steps 50 --warmup 10 > $OUT/bench_kitti.json 2> /dev/null
python bench.py --workload kitti_all --chain --steps 30 --warmup 5 > $OUT/bench_kitti_chain.json 2> /dev/null
python bench.py --workload kitti_all --chain --es-scheme 0 --steps 30 --warmup 5 > $OUT/bench_kitti_chain_scheme0.json 2> /dev/null
python bench.py --workload kitti_all --chain --steps 30 --warmup 5 --in-flight 1 > $OUT/bench_kitti_chain_1.json 2> /dev/null
python bench.py --gpus 2 --share-gpu --workload kitti_all --chain --steps 10 --warmup 3 > $OUT/bench_kitti_chain_2ranks.json 2> /dev/null
python bench.py --gpus 2 --share-gpu --single-process --pairs 50000 --steps 10 --warmup 3 > $OUT/bench_single_process_2x.json 2> /dev/null
python bench.py --gpus 1 --single-process --pairs 100000 --steps 10 --warmup 3 > $OUT/bench_single_process_1x.json 2> /dev/null
# 5. residual families with roofline blocks
python tools/bench_modes.py 100000 > $OUT/residual_families.jsonl 2> /dev/null
# 6. parity at scale
python tools/verify_full_batch.py 100000 target > $OUT/full_batch_parity.jsonl 2> /dev/null
python tools/verify_pipeline.py 100000 > $OUT/pipeline_parity_100k.json 2> $OUT/pipeline_parity_100k.err
python tools/verify_eigensolver_schemes.py 2000 > $OUT/odometry_options_parity.json 2> /dev/null
python tools/bench_streaming.py > $OUT/streaming.json 2> /dev/null
# 7. one PNEC::Solve per frame through the facade (default options / the odometry's / the timed overload)
rm -f $OUT/solve_latency.jsonl
for n in 100 512 700 2000; do for m in default vo timed; do for s in 2 0; do ./pnec_amd/pnec_host_demo $n solve_latency 300 $m $s 2>/dev/null | tail -1 >> $OUT/solve_latency.jsonl; done; done; done
fi
find $OUT -type f -size +4M -delete
ls -la $OUT
