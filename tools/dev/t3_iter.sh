#!/bin/bash
# K2t development loop on the GPU box: its tests, per-kernel times (rocprofv3 kernel trace), end-to-end times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_tiled_gpu.py -x -q 2>&1 | tail -5 || exit 1
rm -rf gpurun_out/t3_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t3_trace -- python3 tools/quick_bench.py --cases ${T3_CASES:-uniform,powerlaw} --only-blocked > gpurun_out/t3_trace.log 2>&1 || { tail -20 gpurun_out/t3_trace.log; exit 1; }
grep -E "tiled" gpurun_out/t3_trace.log
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/t3_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_t3_expand" in r["Name"] or "k_t3_reduce" in r["Name"] or "k_t2_" in r["Name"]:
            print("%-50s calls %s avg %.1f us min %.1f max %.1f" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
find gpurun_out/t3_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/tiled_kernel_stats.csv \;
rm -rf gpurun_out/t3_trace
