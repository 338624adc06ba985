#!/bin/bash
out=gpurun_out/tiled_tile.log; : > $out
for t in 36 42 48 52 56; do
  echo "##### tile target $t" >> $out
  SMH_TILED_TILE=$t timeout -k 10 300 python3 tools/quick_bench.py --cases uniform,powerlaw,uniform64 --lanes 8 --only-blocked 2>&1 | grep "tiled (" >> $out
done
cat $out
