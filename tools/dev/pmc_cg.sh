#!/bin/bash
# HBM traffic of the CG iteration's kernels (C4) from rocprofv3 PMC counters: two passes (FETCH_SIZE, WRITE_SIZE), never mixed
# with tracing; the program itself follows `--`.  Output: gpurun_out/pmc_cg_summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_cg_$C
  timeout -k 5 300 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_cg_$C -- python3 tools/cg_bench.py --iters 20 > gpurun_out/pmc_cg_$C.log 2>&1
  echo "$C rc=$?"
done
python3 - <<'PY' > gpurun_out/pmc_cg_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/pmc_cg_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void smh::", "")
            if r["Counter_Name"] == c: agg[k][c].append(float(r["Counter_Value"]))
print("# per launch, KiB counted; bytes = 2.0 * FETCH_SIZE + 1.0 * WRITE_SIZE (factors calibrated by bench.py in the same round: 1.9999 / 1.0000)")
tot = 0.0
for k, d in sorted(agg.items()):
    if not any(s in k for s in ("k_spmv_stream", "k_cg_p", "k_cg_update", "k_ew", "k_dot")): continue
    f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])); w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
    b = (2.0 * f + w) * 1024
    print("%-60s launches %4d  FETCH %.0f KiB  WRITE %.0f KiB  -> %.3f GB" % (k[:60], len(d["FETCH_SIZE"]), f, w, b / 1e9))
PY
cat gpurun_out/pmc_cg_summary.txt
rm -rf gpurun_out/pmc_cg_FETCH_SIZE gpurun_out/pmc_cg_WRITE_SIZE
