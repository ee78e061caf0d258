#!/bin/bash
# final evidence, part B: sampling SQ counters + HBM traffic per kernel, batch curve, configs 1 and 4
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
tools/pmc_sq_passes.sh sq_sampling_final python3 $R/tools/time_forward.py --mols 4096 --iters 1
cat gpurun_out/sq_sampling_final_table.txt | head -12
cd /tmp && export TMPDIR=/tmp
for CNT in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_$CNT -- python3 $R/bench.py --mode resident --mols 4096 --steps 1 --warmup 0 --denoise-steps 8 --steps-per-pass 1 --no-cpu-baseline --no-live-traffic --no-config5 > $R/gpurun_out/pmc_$CNT.log 2>&1
done
cd $R
python3 tools/pmc_traffic_json.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE 4096 gpurun_out/r05_pmc_traffic.json
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
rm -f gpurun_out/r05_throughput_vs_batch.jsonl
python3 tools/batch_curve.py --out gpurun_out/r05_throughput_vs_batch.jsonl --mols 256,1250,2500,5000,10000
python3 bench.py --mode resident --spectra ir --mols 64 --denoise-steps 50 --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --no-config5 > gpurun_out/bench_r05_config1_ir_b64_s50.json 2>/dev/null
python3 bench.py --unconditional --mode resident --steps 4 --warmup 1 --no-cpu-baseline --no-live-traffic --no-config5 > gpurun_out/bench_r05_config4_unconditional.json 2>/dev/null
python3 -c "
import json
for f in ('bench_r05_config1_ir_b64_s50','bench_r05_config4_unconditional'):
    r=json.load(open('gpurun_out/'+f+'.json')); print(f, r['value'])"
