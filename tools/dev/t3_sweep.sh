#!/bin/bash
# K2t (round 3 form): per-kernel times (rocprofv3 kernel trace) and a sweep of its geometry knobs on C2-uniform and C3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_tiled_gpu.py -x -q 2>&1 | tail -5 || exit 1
rm -rf gpurun_out/t3_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t3_trace -- python3 tools/quick_bench.py --cases uniform,powerlaw --only-blocked > gpurun_out/t3_trace.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/t3_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_t3_expand" in r["Name"] or "k_t3_reduce" in r["Name"]:
            print("%-50s calls %s avg %.1f us min %.1f max %.1f" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
for cfg in "uniform 174 3328" "uniform 120 3328" "uniform 140 2496" "uniform 200 4096" "uniform 100 2048" "powerlaw 60 1664" "powerlaw 44 1248" "powerlaw 88 2496" "powerlaw 100 3328" "powerlaw 36 1024"; do
  set -- $cfg
  echo "== $1 tile $2 cap $3"
  SMH_TILED_TILE=$2 SMH_TILED_CAP=$3 timeout -k 10 300 python3 tools/quick_bench.py --cases $1 --only-blocked 2>&1 | grep -E "tiled" || exit 1
done
for p in 4 6 12 16; do
  echo "== parts $p"
  SMH_TILED_PARTS=$p timeout -k 10 300 python3 tools/quick_bench.py --cases uniform,powerlaw --only-blocked 2>&1 | grep -E "tiled" || exit 1
done
