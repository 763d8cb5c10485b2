# Runs ON THE GPU BOX: work counts of the -DPNEC_WORK_COUNT build (tools/build_front_variant.sh count "-DPNEC_WORK_COUNT" first), then everything profiles/r06_* is made from (tools/r06_profiles.sh).
set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}
PNEC_HIP_LIB=pnec_amd/csrc/build/var_count/libpnec_hip.so python tools/count_chain_work.py > gpurun_out/chain_work_new.json 2> gpurun_out/chain_work_new.err
cp gpurun_out/chain_work_new.json profiles/chain_work_latest.json   # (bench.py's chain rooflines read it; stamped with these sources)
bash tools/r06_profiles.sh $1 > gpurun_out/r06_run.log 2>&1
tail -5 gpurun_out/r06_run.log
