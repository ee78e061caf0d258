#!/bin/bash
# sampling kernels: sensitivity to the weight stream (variant halfw: the lo plane re-reads the hi plane's lines) - per-kernel times
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for V in "" _halfw; do
  DIFFSPECTRA_HIP_LIB=$R/diffspectra_amd/libdiffspectra_hip$V.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt16$V -- python3 $R/tools/time_forward.py --mols 5000 --iters 5 > $R/gpurun_out/kt16$V.log 2>&1
  cp $(find $R/gpurun_out/kt16$V -name '*kernel_stats.csv' | head -1) $R/gpurun_out/s16_stats$V.csv
  rm -rf $R/gpurun_out/kt16$V
  tail -1 $R/gpurun_out/kt16$V.log
done
cd $R
python3 tools/kernel_table.py gpurun_out/s16_stats.csv 5000 gpurun_out/s16_stats_halfw.csv 5000 | head -14
