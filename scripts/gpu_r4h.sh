#!/bin/bash
cd $GRAFT_REPO_ROOT
R=gpurun_out/r4h; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pit.py tests/test_gpu_fullsize.py tests/test_abi.py tests/test_pilot_receiver.py -q -m gpu > $R/gpu_tests.txt 2>&1; tail -3 $R/gpu_tests.txt
timeout 900 python scripts/pit_exp.py --workload c3 --reps 3 --no-exact --variants default,gear4,gear16,a:2112:2112,a:1024:2048,a:2112:6336,a:4224:4224,noacq 2>&1 | grep "^##" | cut -c1-700 > $R/acq.txt
timeout 900 python scripts/pit_exp.py --workload c2 --reps 3 --no-exact --variants default,gear4,gear16,a:1024:1024,a:1024:3072 2>&1 | grep "^##" | cut -c1-500 >> $R/acq.txt
timeout 900 python bench.py > $R/bench_c3.json 2> $R/bench_c3.err; tail -2 $R/bench_c3.err
python scripts/show_bench.py $R/bench_c3.json > $R/bench_c3.txt 2>&1
timeout 900 python bench.py --workload c5 --steps 3 > $R/bench_c5.json 2> $R/bench_c5.err; tail -2 $R/bench_c5.err
ls -la $R
