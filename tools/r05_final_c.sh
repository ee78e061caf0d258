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
# timeline of the step (idle / one kernel / several; per-phase kernels), the isolated kernel benches and the repetition check
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_train2 -- python3 $R/bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline --no-live-traffic > $R/gpurun_out/kt_train2.log 2>&1
cd $R
python3 tools/trace_timeline.py gpurun_out/kt_train2 --phases > gpurun_out/r05_train_timeline_final.txt
rm -rf gpurun_out/kt_train2
head -8 gpurun_out/r05_train_timeline_final.txt
python3 tools/sfa_bench.py > gpurun_out/r05_sfa_bench_final.txt 2>/dev/null
python3 tools/chain_bench.py 2>/dev/null | tail -7 > gpurun_out/r05_chain_bench_final.txt
( echo "default build, 3 x 40 repeated steps (every gradient compared bit for bit with the first run and with the run before):"
  for k in 1 2 3; do python3 tools/repro_check.py 40 2>/dev/null | tail -1; done ) | cut -c1-200 > gpurun_out/r05_train_repetition_check.txt
cat gpurun_out/r05_sfa_bench_final.txt gpurun_out/r05_chain_bench_final.txt gpurun_out/r05_train_repetition_check.txt | cut -c1-160
rm -f gpurun_out/r05_train_ab_final.txt
for CFG in "1 2 node,pair,dir" "0 2 node,pair,dir" "1 1 node,pair,dir" "1 3 node,pair,dir" "1 2 node" "1 2 none"; do
  set -- $CFG
  DIFFSPECTRA_FUSED_CHAIN=$1 DIFFSPECTRA_DW_STREAMS=$2 DIFFSPECTRA_FUSED_BWD=$3 python3 bench.py --mode train --steps 30 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/ab_t.json 2>/dev/null
  python3 -c "import json; r=json.load(open('gpurun_out/ab_t.json')); print('fused chains=$1 weight-gradient streams=$2 fused backward kernels=$3:', round(r['value']), 'molecules/s', round(r['ms_per_step'],2), 'ms/step, host issue', round(r['config']['host_issue_ms_per_step'],2), 'ms')" | tee -a gpurun_out/r05_train_ab_final.txt
done
