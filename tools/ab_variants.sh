# usage (on the GPU box): bash tools/ab_variants.sh "<bench args>" v1 v2 ...   (variant 'default' = the in-tree library)
ARGS=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ $v = default ]; then unset PNEC_HIP_LIB; else export PNEC_HIP_LIB=$PWD/pnec_amd/csrc/build/var_$v/libpnec_hip.so; fi
  python bench.py --no-cpu-baseline $ARGS 2>&1 | tail -1 > /tmp/b.json
  python -c "
import json; d=json.load(open('/tmp/b.json')); print('$v', '%.4g'%d['value'], '%.4f'%d['roofline']['kernel_ms'], '%.4f'%d['ms_per_step'])"
done
done
