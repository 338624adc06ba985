#!/bin/bash
# first GPU contact of the rewritten K2t: its tests, then old (SMH_TILED_V1=1) against new on C2-uniform and C3
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_tiled_gpu.py -x -q 2>&1 | tail -25 || exit 1
for v in 1 0 1 0; do
  echo "== SMH_TILED_V1=$v"
  SMH_TILED_V1=$v timeout -k 10 300 python3 tools/quick_bench.py --cases uniform,powerlaw --only-blocked 2>&1 | grep -E "tiled|==" || exit 1
done
