#!/bin/bash
# K1s XD against the classic XS body on the 512^3 Laplacian (C4): its tests first, then SpMV / CG times interleaved on one box
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python3 -m pytest tests/test_stream_gpu.py tests/test_par_overlap_gpu.py -x -q > gpurun_out/xd_tests.log 2>&1 || { tail -30 gpurun_out/xd_tests.log; exit 1; }
tail -3 gpurun_out/xd_tests.log
for rep in 1 2 3; do
  for xd in 0 1; do
    echo "== SMH_STREAM_XD=$xd (run $rep)"
    SMH_STREAM_XD=$xd timeout -k 10 300 python3 tools/cg_bench.py 2>&1 | grep -E "SpMV auto|ms_per_iteration" || exit 1
  done
done
