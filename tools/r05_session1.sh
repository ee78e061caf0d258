#!/bin/bash
# first GPU session of round 5: per-kernel tables at 1250 / 5000 molecules, SQ counters at 4096 (round-4 kernels = variant "base"),
# same-session A/B of the k_attn_fused variants, GPU tests on the product library
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export DIFFSPECTRA_HIP_LIB=$R/diffspectra_amd/libdiffspectra_hip_r04.so
for M in 5000 1250; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$M -- python3 $R/tools/time_forward.py --mols $M --iters 20 > $R/gpurun_out/kt_$M.log 2>&1
  cp $(find $R/gpurun_out/kt_$M -name '*kernel_stats.csv' | head -1) $R/gpurun_out/kt_${M}_kernel_stats.csv
  rm -rf $R/gpurun_out/kt_$M
done
cd $R
python3 tools/kernel_table.py gpurun_out/kt_1250_kernel_stats.csv 1250 gpurun_out/kt_5000_kernel_stats.csv 5000 > gpurun_out/kt_1250_vs_5000.txt
cat gpurun_out/kt_1250_vs_5000.txt
tools/pmc_sq_passes.sh sq_sampling python3 $R/tools/time_forward.py --mols 4096 --iters 1
cat gpurun_out/sq_sampling_table.txt
cd /tmp
for V in base fma v2 pf prod r04; do
  if [ $V = prod ]; then unset DIFFSPECTRA_HIP_LIB; else export DIFFSPECTRA_HIP_LIB=$R/diffspectra_amd/libdiffspectra_hip_$V.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$V -- python3 $R/tools/time_forward.py --mols 5000 --iters 20 > $R/gpurun_out/ab_$V.log 2>&1
  cp $(find $R/gpurun_out/ab_$V -name '*kernel_stats.csv' | head -1) $R/gpurun_out/ab_${V}_kernel_stats.csv
  rm -rf $R/gpurun_out/ab_$V
  echo "== $V"; grep -h "k_attn_fused\|k_equi_pairs" $R/gpurun_out/ab_${V}_kernel_stats.csv | cut -d, -f1-4 | sed 's/(anonymous namespace):://g'
  grep -h "ms per forward\|forward" $R/gpurun_out/ab_$V.log | head -2
done
unset DIFFSPECTRA_HIP_LIB
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_gputests_0.log 2>&1 || (tail -30 gpurun_out/r05_gputests_0.log; exit 1)
tail -3 gpurun_out/r05_gputests_0.log
