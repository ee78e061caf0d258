#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
for TB in 8 64 256; do
  python3 bench.py --mode train --train-batch $TB --steps 30 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/s13_t$TB.json 2> gpurun_out/s13_t.err || (tail -20 gpurun_out/s13_t.err; exit 1)
  python3 -c "import json; r=json.load(open('gpurun_out/s13_t$TB.json')); print('train batch $TB', round(r['value']), round(r['ms_per_step'],2), 'host issue', round(r['config']['host_issue_ms_per_step'],2))"
done
