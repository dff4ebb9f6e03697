#!/bin/bash
# iteration check: whole GPU suite, then the default bench line without the CPU legs
cd $GRAFT_REPO_ROOT
R=gpurun_out/iter; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x > $R/gpu_tests.txt 2>&1; tail -5 $R/gpu_tests.txt
timeout 900 python bench.py --bank 0 --no-cpu-baseline > $R/bench_c3.json 2> $R/bench_c3.err; tail -2 $R/bench_c3.err
python scripts/show_bench.py $R/bench_c3.json > $R/bench_c3.txt 2>&1; head -12 $R/bench_c3.txt | cut -c1-400
python - <<'PY'
import json
d=json.load(open('gpurun_out/iter/bench_c3.json'))
r=d['roofline']; print(r['launch_ms_by_pass'], r['launch_ms'], r['frac'])
for st in d['tier_b']['stages']: print(st['stage'], st.get('pass_ms_by_pass'))
PY
