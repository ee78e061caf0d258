#!/bin/bash
# SpecFormer flash attention: same-session A/B of variant builds (tools/variant_build.py --source ds_train_attn.hip)
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
for V in ${VARIANTS:-"" _pro ""}; do
  [ "$V" = "product" ] && V=""
  echo "== variant ${V:-product}" | tee -a gpurun_out/s9_sfa.log
  DIFFSPECTRA_HIP_LIB=$R/diffspectra_amd/libdiffspectra_hip$V.so timeout -k 10 120 python3 tools/sfa_bench.py | tee -a gpurun_out/s9_sfa.log
done
