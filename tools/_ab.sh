for s in 0 1; do echo "split=$s"; PNEC_SELECT_SPLIT=$s timeout -s KILL 100 python tools/bench_pipeline.py 20000 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(round(d['gpu_ms']['inlier_extraction'],3), round(d['gpu_ms_one_call_pipeline'],3))"; done
for n in 100 512 2000; do timeout -s KILL 60 ./pnec_amd/pnec_host_demo $n solve_latency 300 | cut -c150-230; done
