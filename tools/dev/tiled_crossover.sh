#!/bin/bash
# K2f against K2t on uniform-column matrices of several sizes / densities -> gpurun_out/tiled_crossover.log
out=gpurun_out/tiled_crossover.log; : > $out
for spec in "1000000 16" "1000000 32" "2000000 16" "2000000 32" "4000000 8" "4000000 16" "4000000 32" "10000000 8" "10000000 16" "10000000 64"; do
  set -- $spec
  for c in uniform uniform64; do
    echo "### rows $1 k $2 $c" >> $out
    timeout -k 10 300 python3 tools/quick_bench.py --rows $1 --k $2 --cases $c --tiled $c --lanes 8 --cb-shifts 18 --only-blocked 2>&1 | grep -v "amdgpu.ids\|colblock 2" >> $out || exit 1
  done
done
cat $out
