#!/bin/bash
# which co-resident kernel class matters for the pair / directed backward kernels: LDS request 128 kB (every <=32 kB kernel may share the CU),
# 142 kB (only the 18 kB ones), 150 kB (product: none of the products)
R=$GRAFT_REPO_ROOT
cd $R
for V in _lds128 _lds142; do
  for k in 1 2 3; do echo "== variant $V run $k"; DIFFSPECTRA_HIP_LIB=$R/diffspectra_amd/libdiffspectra_hip$V.so python3 tools/repro_check.py 40 2>&1 | grep "runs that"; done
done
