#!/bin/bash
# after a chain-kernel change: the chain / config-5 / curve tests, then the training step with and without the fused chains
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python3 -m pytest tests/test_train_hip.py -x -q -m gpu -k "fused or config5 or curves" 2>&1 | tail -4
for i in 1 2; do
  python3 bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/s14_t.json 2> gpurun_out/s14_t.err || (tail -20 gpurun_out/s14_t.err; exit 1)
  python3 -c "import json; r=json.load(open('gpurun_out/s14_t.json')); print('train', round(r['value']), round(r['ms_per_step'],2), 'host issue', round(r['config']['host_issue_ms_per_step'],2))"
done
