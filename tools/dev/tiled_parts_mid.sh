#!/bin/bash
out=gpurun_out/tiled_parts_mid.log; : > $out
for spec in "2000000 16" "4000000 8" "4000000 16" "1000000 16"; do set -- $spec
 for p in 2 4 8 16 32; do for u in 2 8; do
  echo "### rows $1 k $2 parts $p unroll $u" >> $out
  SMH_TILED_PARTS=$p SMH_TILED_UNROLL=$u timeout -k 10 200 python3 tools/quick_bench.py --rows $1 --k $2 --cases uniform,uniform64 --tiled uniform,uniform64 --lanes 8 --only-blocked 2>&1 | grep "tiled (" | sed 's/.*median/median/' | cut -c1-40 >> $out
 done; done
done
cat $out
