#!/bin/bash
# is it k_tr_gemm_bf16<128,128,256>?  The pair / directed backward kernels with an ordinary LDS request (80 kB: anything may share the CU), with and
# without that tile (DST_GEMM_BN=64 keeps every non-wide product on 64-column tiles)
R=$GRAFT_REPO_ROOT
cd $R
for BN in 64 0; do
  for k in 1 2 3; do echo "== lds80 variant, DST_GEMM_BN=$BN run $k"; DST_GEMM_BN=$BN DIFFSPECTRA_HIP_LIB=$R/diffspectra_amd/libdiffspectra_hip_lds80.so python3 tools/repro_check.py 40 2>&1 | grep "runs that"; done
done
for CFG in "_lds80 64" " 0" "_lds80 64" " 0"; do
  set -- $CFG
  DST_GEMM_BN=$2 DIFFSPECTRA_HIP_LIB=$R/diffspectra_amd/libdiffspectra_hip$1.so python3 bench.py --mode train --steps 40 --warmup 5 --no-cpu-baseline --no-live-traffic > gpurun_out/ab_t.json 2>/dev/null
  python3 -c "import json; r=json.load(open('gpurun_out/ab_t.json')); print('variant ${1:-product} DST_GEMM_BN=$2:', round(r['value']), 'molecules/s', round(r['ms_per_step'],2), 'ms/step')"
done
