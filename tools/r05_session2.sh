#!/bin/bash
# session 2: two-stream forward A/B at 5000 / 1250 molecules, the new GPU tests, then the whole GPU suite
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
for M in 5000 1250; do
  for MODE in "one DIFFSPECTRA_TWO_STREAM=0" "two DIFFSPECTRA_TWO_STREAM=1" "base DIFFSPECTRA_TWO_STREAM=0 DIFFSPECTRA_HIP_LIB=$R/diffspectra_amd/libdiffspectra_hip_base.so" "two_again DIFFSPECTRA_TWO_STREAM=1" "one_again DIFFSPECTRA_TWO_STREAM=0"; do
    set -- $MODE; tag=$1; shift
    env "$@" python3 tools/time_forward.py --mols $M --iters 30 > gpurun_out/s2_${M}_$tag.log 2>&1
    echo "$M $tag: $(grep 'ms/forward' gpurun_out/s2_${M}_$tag.log)"
  done
done
timeout -k 10 900 python3 -m pytest tests/test_hip_parity.py -q -x -k "two_stream or stages or g4_forward or full_size or largest or launch_order or graph_replay" > gpurun_out/s2_tests_a.log 2>&1 || (tail -40 gpurun_out/s2_tests_a.log; exit 1)
tail -3 gpurun_out/s2_tests_a.log
timeout -k 10 900 python3 -m pytest tests/test_train_hip.py -q -x -s -k "config5_as_benchmarked or curves_agree or hip_trainer_produced" > gpurun_out/s2_tests_b.log 2>&1 || (tail -60 gpurun_out/s2_tests_b.log; exit 1)
grep "^\[" gpurun_out/s2_tests_b.log | cut -c1-300; tail -3 gpurun_out/s2_tests_b.log
