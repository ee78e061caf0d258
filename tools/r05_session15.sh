#!/bin/bash
# repeated-step check of the training step (bit-reproducibility), default and with every fused backward kernel
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests/test_train_hip.py -x -q -m gpu 2>&1 | tail -2
echo "== default"; python3 tools/repro_check.py 40 2>&1 | grep "runs that"
