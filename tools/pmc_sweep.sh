#!/bin/bash
# Collect rocprofv3 PMC counters (one pass per counter group, never mixed with tracing) for a quick_bench case.
# usage (on the GPU box): tools/pmc_sweep.sh <outdir> <quick_bench args...>
# Counter groups respect the per-block slot limits of gfx950 (TCC 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2; SQ 8; TA 2
# base counters per pass; see MI355X_MICROARCH.md "rocprofv3 PMC slots").  Round 1 once put three TA_*_STALLED base
# counters and TCP_TCC_READ_REQ_LATENCY into ONE pass: rocprofiler refused the configuration at the first kernel launch
# ("Could not construct profile cfg failed with error code 38: Request exceeds the capabilities of the hardware to
# collect", gpurun_out/pmc_lap/g6.log of that run), aborted the child with signal 6 from inside the launch and the pass
# sat in its abort handler until the timeout -- which looked like a hang.  The stall counters are now split over passes 7
# and 8, two TA base counters each.
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
         "TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $C --output-format csv -d "$OUT/g$i" -- python3 tools/quick_bench.py "$@" > "$OUT/g$i.log" 2>&1
  echo "group $i ($C) rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "spmv" not in k: continue
        k = k.split("(")[0].replace("void smh::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, d in res.items():
    print(k)
    for c, v in sorted(d.items()): print("   %-40s %.4g" % (c, v))
PY
