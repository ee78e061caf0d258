#!/bin/bash
# kernel trace of the training bench -> timeline view (idle / one kernel / several) + kernel stats
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_train -- python3 $R/bench.py --mode train --steps 8 --warmup 2 --no-cpu-baseline --no-live-traffic > $R/gpurun_out/kt_train.log 2>&1
cd $R
python3 tools/trace_timeline.py gpurun_out/kt_train --phases | tee gpurun_out/s12_timeline.txt
cp $(find gpurun_out/kt_train -name '*kernel_stats.csv' | head -1) gpurun_out/s12_kernel_stats.csv
head -3 $(find gpurun_out/kt_train -name '*kernel_trace.csv' | head -1) > gpurun_out/s12_trace_head.txt
rm -rf gpurun_out/kt_train
