for r in 1 2; do for L in default basefull; do
  if [ $L = default ]; then unset PNEC_HIP_LIB; else export PNEC_HIP_LIB=$PWD/pnec_amd/csrc/build/var_$L/libpnec_hip.so; fi
  python tools/digest_solve.py 40000 10 2>&1 | tail -1 | cut -c1-230
  python bench.py --no-cpu-baseline --workload kitti_all --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   kitti_all', '%.4g'%d['value'], d['ms_per_step'])"
done; done
unset PNEC_HIP_LIB
