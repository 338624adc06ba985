#!/bin/bash
# A/B of K2t's XCD-aware block -> tile mapping (SMH_TILED_XCD: bit 0 pass 1, bit 1 pass 2), one box, one call
cd "$GRAFT_REPO_ROOT"
for x in 0 1 2 3 0 3; do
  echo "== SMH_TILED_XCD=$x"
  SMH_TILED_XCD=$x timeout -k 10 300 python3 tools/quick_bench.py --cases uniform,powerlaw --only-blocked 2>&1 | grep -E "tiled|auto|==" || exit 1
done
