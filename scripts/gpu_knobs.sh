#!/bin/bash
# tuning: CUs kept free of stream 2 (QAMPY_HIP_RESERVED_CUS) x parts of the pending phase search (QAMPY_POST_PARTS); args: "cus:parts" ...
cd $GRAFT_REPO_ROOT
R=gpurun_out/knobs; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
for cfg in $*; do
  cus=${cfg%%:*}; n=${cfg##*:}
  QAMPY_HIP_RESERVED_CUS=$cus QAMPY_POST_PARTS=$n timeout 300 python bench.py --bank 0 --no-cpu-baseline --no-extra-shapes --exact-steps 1 > $R/bench_${cus}_$n.json 2> $R/bench_${cus}_$n.err
  python - $cus $n <<'PY'
import json, sys
cus, n = sys.argv[1:3]
d=json.load(open('gpurun_out/knobs/bench_%s_%s.json' % (cus, n)))
st=d['tier_b']['stages']
print('reserved_cus', cus, 'parts', n, 'value', d['value'], 'ms', d['ms_per_step'], d['stages_ms'], [s.get('pass_ms_by_pass') for s in st], 'frac', d['roofline'].get('frac'))
PY
done
