#!/bin/bash
# final evidence, part C: the training step - bench lines (bf16 with live traffic + CPU baseline, fp32), SQ counters, kernel stats + launches per step
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python3 bench.py --mode train --steps 20 --warmup 5 > gpurun_out/r05_train_bench_final_bf16.json 2> gpurun_out/r05_train_bf16.err
python3 bench.py --mode train --precision fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/r05_train_bench_final_fp32.json 2> gpurun_out/r05_train_fp32.err
python3 -c "
import json
for f in ('bf16','fp32'):
    r=json.load(open('gpurun_out/r05_train_bench_final_'+f+'.json')); print('train', f, r['value'], r['ms_per_step'], r['roofline'].get('traffic'))"
tools/pmc_sq_passes.sh sq_train_final python3 $R/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic
head -14 gpurun_out/sq_train_final_table.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_train -- python3 $R/bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline --no-live-traffic > $R/gpurun_out/kt_train.log 2>&1
cp $(find $R/gpurun_out/kt_train -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r05_train_kernel_stats_final.csv
rm -rf $R/gpurun_out/kt_train
cd $R
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r05_train_kernel_stats_final.csv')))
calls=sum(int(r['Calls']) for r in rows); tot=sum(float(r['TotalDurationNs']) for r in rows)
print('launches in the profiled run (8 timed + 2 warm-up + 1 instrumented step = 11 steps):', calls, '->', round(calls/11), 'per step; kernel time per step', round(tot/11/1e6,2), 'ms')
PY
