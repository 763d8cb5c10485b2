# A/B of PNEC_RANSAC_TAIL_SINGLES on one box (digest + chain / RANSAC-stage times per setting)
mkdir -p gpurun_out/s2
for rep in 1 2; do
for s in ${SINGLES:-0 1024 2048}; do
  PNEC_RANSAC_TAIL_SINGLES=$s timeout 300 python tools/ab_ransac_forms.py uniform ${PAIRS:-20000} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('singles', $s, 'chain %.3f ransac %.3f' % (d['chain_ms'], d['ransac_stage_ms']), d['digest'][:12], d['mean_ransac_iterations'], d['mean_inliers'])"
done; done | tee gpurun_out/s2/ab_singles.txt
