#!/bin/bash
# A/B of two builds of the library on the headline: the product .so against sparsemat_amd/libsparsemat_hip_ab.so, interleaved, one box
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4; do
  for lib in product ab; do
    if [ $lib = ab ]; then export SPARSEMAT_HIP_LIB=$GRAFT_REPO_ROOT/sparsemat_amd/libsparsemat_hip_ab.so; else unset SPARSEMAT_HIP_LIB; fi
    timeout -k 10 200 python3 bench.py --no-traffic --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', 'kernel_ms mean %.4f median %.4f min %.4f cold %.4f step %.4f' % (r['kernel_ms'], r['kernel_ms_median'], r['kernel_ms_min'], r['cold']['kernel_ms_median'], d['ms_per_step']))" || exit 1
  done
done
