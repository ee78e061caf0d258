#!/bin/bash
# session 4: fused chains - the training step with and without them, then correctness tests, flash attention vs torch, full GPU suite
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
for MODE in 1 0 1 0; do
  DIFFSPECTRA_FUSED_CHAIN=$MODE python3 bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/s4_train_fused$MODE.json 2> gpurun_out/s4_train_fused$MODE.err || (tail -20 gpurun_out/s4_train_fused$MODE.err; exit 1)
  python3 -c "import json; r=json.load(open('gpurun_out/s4_train_fused$MODE.json')); print('fused=$MODE', round(r['value']), round(r['ms_per_step'],2), 'loss', r['config']['last_loss'])"
done
timeout -k 10 600 python3 -m pytest tests/test_train_hip.py -q -x -s -k "fused_pair_chain or flash_attention_vs_torch or pair_sum_zbuild or config5_as_benchmarked" > gpurun_out/s4_tests.log 2>&1 || (tail -60 gpurun_out/s4_tests.log; exit 1)
grep "^\[" gpurun_out/s4_tests.log | cut -c1-300; tail -2 gpurun_out/s4_tests.log
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_gputests_2.log 2>&1 || (tail -30 gpurun_out/r05_gputests_2.log; exit 1)
tail -3 gpurun_out/r05_gputests_2.log
