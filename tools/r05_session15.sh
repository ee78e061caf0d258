#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
for X in 0 1 0 1; do
  DIFFSPECTRA_X_SKIPFF=$X python3 bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/s15_t.json 2> gpurun_out/s15_t.err || (tail -5 gpurun_out/s15_t.err; true)
  python3 -c "import json; r=json.load(open('gpurun_out/s15_t.json')); print('train skipff=$X', round(r['value']), round(r['ms_per_step'],2), 'host issue', round(r['config']['host_issue_ms_per_step'],2))" || true
done
