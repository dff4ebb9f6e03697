#!/bin/bash
# segment kernel: 8 against 16 lanes per chain at C3 (pass times of both stages)
cd $GRAFT_REPO_ROOT
R=gpurun_out/lanes; rm -rf $R; mkdir -p $R
for L in 8 16 default; do
  if [ $L = default ]; then unset QAMPY_HIP_SEG_LANES; else export QAMPY_HIP_SEG_LANES=$L; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extra-shapes --steps 10 --warmup 2 > $R/b_$L.json 2> $R/b_$L.err
done
python - <<'PY'
import json
for n in ("8","16","default"):
    try:
        d=json.loads(open(f"gpurun_out/lanes/b_{n}.json").read().strip().splitlines()[-1]); tb=d["tier_b"]
        print(n, d["value"], d["ms_per_step"], [(s["stage"], s["S"], s["seg_len"], s["P"], s["pass_ms"]) for s in tb["stages"]], tb["stages_ms"], tb.get("certified"))
    except Exception as e: print(n, "fail", e)
PY
