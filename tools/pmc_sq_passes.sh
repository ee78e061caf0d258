#!/bin/bash
# three rocprofv3 --pmc passes of one command -> gpurun_out/$1_{A,B,C}; usage: tools/pmc_sq_passes.sh TAG python3 prog args...
# (counters only with --kernel-trace; the program itself stands behind `--`, never a shell or env wrapper)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/${tag}_A -- "$@" > $R/gpurun_out/${tag}_A.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/${tag}_B -- "$@" > $R/gpurun_out/${tag}_B.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${tag}_C -- "$@" > $R/gpurun_out/${tag}_C.log 2>&1
cd $R
python3 tools/pmc_sq_table.py gpurun_out/${tag}_A gpurun_out/${tag}_B gpurun_out/${tag}_C > gpurun_out/${tag}_table.txt
# the raw per-dispatch CSVs are large: keep the table only
rm -rf gpurun_out/${tag}_A gpurun_out/${tag}_B gpurun_out/${tag}_C
