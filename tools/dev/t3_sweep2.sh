#!/bin/bash
# K2t with the lock-stepped pass 2: products per tile x most rows per row block (the LDS then holds sums for 160 KiB / (4 or 8 x rows) wavefronts)
cd "$GRAFT_REPO_ROOT"
for cfg in "174 3328" "130 2496" "200 4096" "150 2880" "256 4992" "110 2048" "174 3328"; do
  set -- $cfg
  echo "== uniform tile $1 cap $2"
  SMH_TILED_TILE=$1 SMH_TILED_CAP=$2 timeout -k 10 300 python3 tools/quick_bench.py --cases uniform --only-blocked 2>&1 | grep -E "tiled \(K2t\)" | cut -c1-150 || exit 1
done
for cfg in "60 1664" "45 1248" "75 2048" "90 2496" "52 1456" "60 1664"; do
  set -- $cfg
  echo "== powerlaw tile $1 cap $2"
  SMH_TILED_TILE=$1 SMH_TILED_CAP=$2 timeout -k 10 300 python3 tools/quick_bench.py --cases powerlaw --only-blocked 2>&1 | grep -E "tiled \(K2t\)" | cut -c1-150 || exit 1
done
