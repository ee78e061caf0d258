#!/bin/bash
# SQ counters of the SpecFormer flash attention kernels alone (three --pmc passes of tools/sfa_bench.py)
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
bash tools/pmc_sq_passes.sh s10_sfa python3 $R/tools/sfa_bench.py --iters 3
cat gpurun_out/s10_sfa_table.txt
