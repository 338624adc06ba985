#!/bin/bash
# kernel timeline of the rank-like step (tools/dev/rank_like_step.py, overlap on): one step's kernels with start / end relative to the step's first kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/rl_trace
SMH_RANK_LIKE_TRACE=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rl_trace -- python3 tools/dev/rank_like_step.py > gpurun_out/rl_trace.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/rl_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 3 steps: find the interior launches (the long k_spmv_ring2 ones) and print what surrounds the last but one
long = [i for i, r in enumerate(rows) if "k_spmv_ring2" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 200000]
i0, i1 = long[-3], long[-2]
t0 = int(rows[i0]["Start_Timestamp"])
print("columns:", [c for c in rows[0].keys()][:14])
for r in rows[i0 - 8:i1 + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f us  %7.1f  q=%s  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].replace("void smh::", "")[:70]))
PY
rm -rf gpurun_out/rl_trace
