#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
for CFG in "1 1" "0 1" "1 0" "0 0" "1 1" "0 1"; do
  set -- $CFG
  DIFFSPECTRA_FUSED_CHAIN=$1 DIFFSPECTRA_NODE_STREAM=$2 python3 bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/s7_t.json 2> gpurun_out/s7_t.err || (tail -20 gpurun_out/s7_t.err; exit 1)
  python3 -c "import json; r=json.load(open('gpurun_out/s7_t.json')); print('fused=$1 node_stream=$2', round(r['value']), round(r['ms_per_step'],2))"
done
for BATCH in 5000 10000 5000 10000; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --batch $BATCH --no-cpu-baseline --no-live-traffic --no-config5 > gpurun_out/s7_b$BATCH.json 2> gpurun_out/s7_b.err || (tail -20 gpurun_out/s7_b.err; exit 1)
  python3 -c "import json; r=json.load(open('gpurun_out/s7_b$BATCH.json')); print('batch $BATCH', round(r['value'],2), round(r['ms_per_step'],1), r['roofline']['avg_launch_ms'])"
done
