#!/bin/bash
# final-tree soak: GPU suite twice, the driver's bench command three times, smoke()
cd $GRAFT_REPO_ROOT
R=gpurun_out/r06s; mkdir -p $R
for i in 1 2; do python -m pytest tests -q -m gpu -x -p no:cacheprovider > $R/t$i.txt 2>&1; tail -1 $R/t$i.txt; done
for i in 1 2 3; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $R/b$i.json 2> $R/b$i.err
  python -c "
import json;t=open('$R/b$i.json').read();d=json.loads(t);print(d['value'],d['ms_per_step'],d['tier_b']['certified'],d['tier_b_tight']['certified'],len(t))"
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
