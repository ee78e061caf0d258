#!/bin/bash
# fused chains with bf16 weights: kernel times, the chain / config-5 tests, the training step
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 120 python3 tools/chain_bench.py 2>&1 | grep -v amdgpu.ids | tail -6
timeout -k 10 120 python3 tools/chain_bench.py 2>&1 | grep -v amdgpu.ids | tail -6
python3 -m pytest tests/test_train_hip.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
  python3 bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/s14_t.json 2> gpurun_out/s14_t.err || (tail -20 gpurun_out/s14_t.err; exit 1)
  python3 -c "import json; r=json.load(open('gpurun_out/s14_t.json')); print('train', round(r['value']), round(r['ms_per_step'],2), 'host issue', round(r['config']['host_issue_ms_per_step'],2))"
done
