#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests/test_train_hip.py -x -q -m gpu 2>&1 | tail -2
for k in 1 2; do echo "== default (node,pair,dir,front) run $k"; python3 tools/repro_check.py 40 2>&1 | grep "runs that"; done
for FB in "node,pair,dir,front" "node,pair,dir" "node,pair,dir,front" "node,pair,dir"; do
  DIFFSPECTRA_FUSED_BWD=$FB python3 bench.py --mode train --steps 40 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/ab_t.json 2>/dev/null
  python3 -c "import json; r=json.load(open('gpurun_out/ab_t.json')); print('fused backward kernels $FB:', round(r['value']), 'molecules/s', round(r['ms_per_step'],2), 'ms/step')"
done
