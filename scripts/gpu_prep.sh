#!/bin/bash
# measurement: per-pass hook behind (0) or in front of (1) the trainer launch x parts of the phase search
cd $GRAFT_REPO_ROOT
R=gpurun_out/prep; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
for cfg in $*; do
  g=${cfg%%:*}; n=${cfg##*:}
  QAMPY_HIP_HOOK_BEFORE=$g QAMPY_POST_PARTS=$n timeout 300 python bench.py --bank 0 --no-cpu-baseline --no-extra-shapes --exact-steps 1 > $R/bench_${g}_$n.json 2> $R/bench_${g}_$n.err
  python - $g $n <<'PY'
import json, sys
g, n = sys.argv[1:3]
d=json.load(open('gpurun_out/prep/bench_%s_%s.json' % (g, n)))
st=d['tier_b']['stages']
print('hook_before', g, 'parts', n, 'value', d['value'], 'ms', d['ms_per_step'], d['stages_ms'], [s.get('pass_ms_by_pass') for s in st])
PY
done
