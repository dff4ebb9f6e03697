#!/bin/bash
# tuning: the pending phase search in n parts between the next capture's relaxation passes (QAMPY_POST_PARTS: 1 = one launch, 0 = automatic)
cd $GRAFT_REPO_ROOT
R=gpurun_out/parts; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_functional.py -q -m gpu -x > $R/gpu_tests.txt 2>&1; tail -3 $R/gpu_tests.txt
for n in $*; do
  QAMPY_POST_PARTS=$n timeout 300 python bench.py --bank 0 --no-cpu-baseline --no-extra-shapes --exact-steps 1 > $R/bench_$n.json 2> $R/bench_$n.err
  python - $n <<'PY'
import json, sys
n=sys.argv[1]
d=json.load(open('gpurun_out/parts/bench_%s.json' % n))
st=d['tier_b']['stages']
print('parts', n, 'value', d['value'], 'ms', d['ms_per_step'], d['stages_ms'], [s.get('pass_ms_by_pass') for s in st], 'three', (d.get('three_in_flight') or {}).get('value'), 'frac', d['roofline'].get('frac'), d['roofline'].get('kernel'))
PY
done
