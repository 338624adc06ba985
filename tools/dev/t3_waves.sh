#!/bin/bash
# K2t pass 2: wavefronts (= adjacent row blocks walking the slices in lock step) per workgroup, kernel trace, one box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in 1 2 4 6 12 3 4; do
  echo "== SMH_TILED_WAVES=$w"
  rm -rf gpurun_out/t3_trace
  SMH_TILED_WAVES=$w timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t3_trace -- python3 tools/quick_bench.py --cases uniform,powerlaw --only-blocked > gpurun_out/t3_trace.log 2>&1 || { tail -20 gpurun_out/t3_trace.log; exit 1; }
  grep -E "tiled \(K2t\)" gpurun_out/t3_trace.log | cut -c1-150
  python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/t3_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_t3_reduce" in r["Name"]:
            print("%-50s calls %s avg %.1f us min %.1f max %.1f" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  rm -rf gpurun_out/t3_trace
done
