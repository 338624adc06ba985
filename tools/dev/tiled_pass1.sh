#!/bin/bash
out=gpurun_out/tiled_pass1b.log; : > $out
for u in 2 4; do for p in 8 12 16 24; do
  echo "##### unroll $u parts $p" >> $out
  SMH_TILED_UNROLL=$u SMH_TILED_PARTS=$p timeout -k 10 300 python3 tools/quick_bench.py --cases uniform,powerlaw --lanes 8 --only-blocked 2>&1 | grep "tiled (" | sed 's/.*median/median/' >> $out
done; done
cat $out
