#!/bin/bash
# K2f (spmv_colfused.hip) tuning sweep on one box: lock-step lag, balanced tiles, block width -> gpurun_out/k2f_sweep.log
# Usage (GPU box): bash tools/k2f_sweep.sh
out=gpurun_out/k2f_sweep.log
mkdir -p gpurun_out
: > "$out"
run() {  # label, env..., -- args
    echo "### $1" >> "$out"; shift
    env "$@" 2>/dev/null | grep -v amdgpu.ids | grep -E "^==|colfused|colblock 2|auto" >> "$out"
}
for lag in 0 1 2 3 4; do
    run "powerlaw lag=$lag" SMH_COLFUSED_LAG=$lag timeout -k 10 200 python3 tools/quick_bench.py --cases powerlaw --lanes 8 --cb-shifts 17,18 --only-blocked
done
for lag in 0 1 2 3; do
  run "uniform lag=$lag" SMH_COLFUSED_LAG=$lag timeout -k 10 200 python3 tools/quick_bench.py --cases uniform --lanes 8 --cb-shifts 18,19 --only-blocked
done
cat "$out"
