#!/bin/bash
# session 3: the new training tests, the training step's counters, then the whole GPU suite
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python3 -m pytest tests/test_train_hip.py -q -x -s -k "config5_as_benchmarked or curves_agree or hip_trainer_produced" > gpurun_out/s3_tests_b.log 2>&1 || (tail -60 gpurun_out/s3_tests_b.log; exit 1)
grep "^\[" gpurun_out/s3_tests_b.log | cut -c1-400; tail -3 gpurun_out/s3_tests_b.log
python3 bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s3_train_bf16.json 2> gpurun_out/s3_train_bf16.err
python3 -c "import json; r=json.load(open('gpurun_out/s3_train_bf16.json')); print('train bf16', r['value'], r['ms_per_step'])"
tools/pmc_sq_passes.sh sq_train python3 $R/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline
cat gpurun_out/sq_train_table.txt
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_gputests_1.log 2>&1 || (tail -30 gpurun_out/r05_gputests_1.log; exit 1)
tail -3 gpurun_out/r05_gputests_1.log
