#!/bin/bash
# A/B of K2t's pipeline depths inside one call (SMH_TILED_BATCH: tiles per batch in pass 2; SMH_TILED_AHEAD: chunks in flight in pass 1)
cd "$GRAFT_REPO_ROOT"
for cfg in "4 3" "8 3" "2 3" "4 2" "4 4" "8 4" "4 3"; do
  set -- $cfg
  echo "== batch $1 ahead $2"
  SMH_TILED_BATCH=$1 SMH_TILED_AHEAD=$2 timeout -k 10 300 python3 tools/quick_bench.py --cases uniform,powerlaw --only-blocked 2>&1 | grep -E "tiled \(K2t\)" | cut -c1-150 || exit 1
done
