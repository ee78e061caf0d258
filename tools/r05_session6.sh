#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 600 python3 -m pytest tests/test_train_hip.py -q -x -s -k "fused_pair_chain or config5_as_benchmarked" > gpurun_out/s6_tests.log 2>&1 || (tail -40 gpurun_out/s6_tests.log; exit 1)
grep "^\[" gpurun_out/s6_tests.log | cut -c1-300; tail -2 gpurun_out/s6_tests.log
for MODE in 1 0 1 0; do
  DIFFSPECTRA_FUSED_CHAIN=$MODE python3 bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/s6_train_fused$MODE.json 2> gpurun_out/s6_train_fused$MODE.err || (tail -20 gpurun_out/s6_train_fused$MODE.err; exit 1)
  python3 -c "import json; r=json.load(open('gpurun_out/s6_train_fused$MODE.json')); print('fused=$MODE', round(r['value']), round(r['ms_per_step'],2), 'loss', r['config']['last_loss'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_train -- python3 $R/bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic > $R/gpurun_out/kt_train.log 2>&1
cp $(find $R/gpurun_out/kt_train -name '*kernel_stats.csv' | head -1) $R/gpurun_out/kt_train_kernel_stats.csv
rm -rf $R/gpurun_out/kt_train
cd $R
grep "chain_fwd\|front_fwd" gpurun_out/kt_train_kernel_stats.csv | cut -d, -f1-7 | sed 's/(anonymous namespace):://g' | cut -c1-200
