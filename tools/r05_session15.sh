#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
echo "== all fused, streams=1"; DIFFSPECTRA_DW_STREAMS=1 python3 tools/repro_check.py 40 2>&1 | grep "runs that"
echo "== all fused, streams=3"; DIFFSPECTRA_DW_STREAMS=3 python3 tools/repro_check.py 40 2>&1 | grep "runs that"
echo "== all fused, streams=2"; DIFFSPECTRA_DW_STREAMS=2 python3 tools/repro_check.py 40 2>&1 | grep "runs that"
