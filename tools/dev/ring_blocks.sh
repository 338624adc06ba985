#!/bin/bash
# K1r: row ranges (workgroups) per CU, SMH_RING_BLOCKS_PER_CU, on the banded / window / f64 shapes (single launches, one box)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for b in 8 6; do
  echo "== SMH_RING_BLOCKS_PER_CU=$b"
  SMH_RING_BLOCKS_PER_CU=$b timeout -k 10 300 python3 tools/quick_bench.py --cases banded,window,banded64 --lanes 8 2>&1 | grep -E "^==|K1r|auto" | cut -c1-120 || exit 1
done
done
