#!/bin/bash
# sweeps of the 2-D tiled probe -> gpurun_out/t2d.log
out=gpurun_out/t2d.log; : > $out
r() { echo "### $*" >> $out; env "$@" timeout -k 10 200 tools/dev/t2d_probe_$V ${ARGS:-f32} 2>&1 | grep -v "wave of row" >> $out || exit 1; }
for V in a b c; do
echo "##### variant $V" >> $out
r T2D_MEAN=48 T2D_C=16384 T2D_PARTS=4
ARGS="f64 10000000 0 1" r T2D_MEAN=48
done
cat $out
