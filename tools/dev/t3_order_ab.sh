#!/bin/bash
# K2t with the tiles' pairs in bank order against row order (SMH_TILED_ORDER=0), per-kernel times from a kernel trace, one box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for ord in 0 1; do
  echo "== SMH_TILED_ORDER=$ord (run $rep)"
  rm -rf gpurun_out/t3_trace
  SMH_TILED_ORDER=$ord timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t3_trace -- python3 tools/quick_bench.py --cases ${T3_CASES:-uniform,powerlaw} --only-blocked > gpurun_out/t3_trace.log 2>&1 || { tail -20 gpurun_out/t3_trace.log; exit 1; }
  grep -E "tiled \(K2t\)" gpurun_out/t3_trace.log | cut -c1-160
  python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/t3_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_t3_expand" in r["Name"] or "k_t3_reduce" in r["Name"] or "k_t3_bank" in r["Name"] or "k_t3_tile_entries" in r["Name"] or "k_t3_pair" in r["Name"]:
            print("%-50s calls %s avg %.1f us min %.1f max %.1f" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  rm -rf gpurun_out/t3_trace
done
done
