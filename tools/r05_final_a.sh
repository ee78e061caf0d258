#!/bin/bash
# final evidence, part A: GPU test suite, the driver's command (plain and under rocprofv3), sampling counters
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > gpurun_out/r05_gpu_tests_final.log 2>&1 || (tail -30 gpurun_out/r05_gpu_tests_final.log; exit 1)
tail -2 gpurun_out/r05_gpu_tests_final.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_cmd_final.json 2> gpurun_out/r05_bench_driver_cmd_final.err ) 2> gpurun_out/r05_bench_driver_cmd_final.time
python3 -c "import json; r=json.load(open('gpurun_out/r05_bench_driver_cmd_final.json')); print('driver cmd', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic'], (r.get('config5') or {}).get('ms_per_step'))"
cat gpurun_out/r05_bench_driver_cmd_final.time
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_bench -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --no-config5 > $R/gpurun_out/r05_bench_driver_cmd_final_under_rocprof.json 2> $R/gpurun_out/kt_bench.err
cp $(find $R/gpurun_out/kt_bench -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r05_bench_driver_cmd_final_kernel_stats.csv
rm -rf $R/gpurun_out/kt_bench
cd $R
head -8 gpurun_out/r05_bench_driver_cmd_final_kernel_stats.csv | cut -d, -f1-5 | cut -c1-160
