# same-box A/B of library builds: tools/ab_quick.sh [reps] lib...   ('default' = the in-tree build)
R=${1:-3}; shift
bash tools/ab_libs.sh "python tools/ab_ransac_forms.py uniform 20000 | python -c \"import sys,json; d=json.loads(sys.stdin.read()); print('chain %.3f ransac %.3f' % (d['chain_ms'], d['ransac_stage_ms']), d['digest'][:12])\"" $R "$@"
