#!/bin/bash
# after a training-kernel change: the whole training test file, then the training step (default and with the fused chains off)
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python3 -m pytest tests/test_train_hip.py -x -q -m gpu 2>&1 | tail -3
for CFG in "1 2" "0 2" "1 1" "1 2"; do
  set -- $CFG
  DIFFSPECTRA_FUSED_CHAIN=$1 DIFFSPECTRA_DW_STREAMS=$2 python3 bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/s14_t.json 2> gpurun_out/s14_t.err || (tail -20 gpurun_out/s14_t.err; exit 1)
  python3 -c "import json; r=json.load(open('gpurun_out/s14_t.json')); print('train fused=$1 dw_streams=$2', round(r['value']), round(r['ms_per_step'],2), 'host issue', round(r['config']['host_issue_ms_per_step'],2))"
done
