#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_train -- python3 $R/bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic > $R/gpurun_out/kt_train.log 2>&1
cp $(find $R/gpurun_out/kt_train -name '*kernel_stats.csv' | head -1) $R/gpurun_out/kt_train_kernel_stats.csv
rm -rf $R/gpurun_out/kt_train
cd $R
grep "chain_fwd\|front_fwd" gpurun_out/kt_train_kernel_stats.csv | cut -d, -f1-7 | sed 's/(anonymous namespace):://g' | cut -c1-200
