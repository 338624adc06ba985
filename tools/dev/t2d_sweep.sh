#!/bin/bash
out=gpurun_out/t2d.log; : > $out
for V in a b c; do
echo "##### variant $V (a: as shipped, b: conflict-free LDS reads instead of gathers, c: no stores) -- b, c: timing only, results wrong" >> $out
T2D_MEAN=48 T2D_C=16384 T2D_PARTS=4 timeout -k 10 200 tools/dev/t2d_probe_$V f32 2>&1 | grep "expand  " >> $out
T2D_MEAN=48 timeout -k 10 200 tools/dev/t2d_probe_$V f64 10000000 0 1 2>&1 | grep "expand  " >> $out
done
cat $out
