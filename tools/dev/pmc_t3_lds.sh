#!/bin/bash
# LDS counters of K2t's reduce pass with the tiles in row order (SMH_TILED_ORDER=0) and in bank order
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for ord in 0 1; do
  rm -rf gpurun_out/pk_l
  SMH_TILED_ORDER=$ord timeout -k 5 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pk_l -- python3 tools/quick_bench.py --cases ${1:-uniform} --only-blocked > gpurun_out/pk_l.log 2>&1
  echo "== SMH_TILED_ORDER=$ord rc=$?"
  python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pk_l/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_t3_expand" not in k and "k_t3_reduce" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k[:70])
    for c, v in sorted(d.items()): print("   %-34s %.5g  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
  rm -rf gpurun_out/pk_l
done
