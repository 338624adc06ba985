#!/bin/bash
out=gpurun_out/tiled_cap.log; : > $out
for cap in 1536 1280 1024 768; do
  echo "##### f64 cap $cap" >> $out
  SMH_TILED_CAP=$cap timeout -k 10 300 python3 tools/quick_bench.py --cases powerlaw --lanes 8 --only-blocked --tiled powerlaw 2>&1 | grep "tiled" >> $out
  for k in 8 16; do
    SMH_TILED_CAP=$cap timeout -k 10 300 python3 tools/quick_bench.py --k $k --cases uniform64 --lanes 8 --only-blocked --tiled uniform64 2>&1 | grep "tiled (" >> $out
  done
done
for cap in 3072 2048 1536; do
  echo "##### f32 cap $cap" >> $out
  SMH_TILED_CAP=$cap timeout -k 10 300 python3 tools/quick_bench.py --k 8 --cases uniform --lanes 8 --only-blocked --tiled uniform 2>&1 | grep "tiled (" >> $out
  SMH_TILED_CAP=$cap timeout -k 10 300 python3 tools/quick_bench.py --rows 3000000 --cases powerlaw32 --lanes 8 --only-blocked --tiled powerlaw32 2>&1 | grep "tiled (" >> $out
done
cat $out
