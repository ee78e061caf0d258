#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python3 tools/chain_bench.py
timeout -k 10 600 python3 -m pytest tests/test_train_hip.py -q -x -s -k "fused_pair_chain or config5_as_benchmarked" > gpurun_out/s8_tests.log 2>&1 || (tail -40 gpurun_out/s8_tests.log; exit 1)
grep "^\[fused" gpurun_out/s8_tests.log | cut -c1-300; tail -1 gpurun_out/s8_tests.log
for CFG in 1 0 1 0; do
  DIFFSPECTRA_FUSED_CHAIN=$CFG python3 bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/s8_t.json 2> gpurun_out/s8_t.err || (tail -20 gpurun_out/s8_t.err; exit 1)
  python3 -c "import json; r=json.load(open('gpurun_out/s8_t.json')); print('fused=$CFG', round(r['value']), round(r['ms_per_step'],2))"
done
