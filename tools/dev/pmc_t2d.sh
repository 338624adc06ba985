#!/bin/bash
# counters of the 2-D tiled probe's two kernels (one pass per counter group; no tracing)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES" \
         "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM FETCH_SIZE WRITE_SIZE" \
         "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/pt_$i
  timeout -k 5 200 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pt_$i -- tools/dev/t2d_probe ${T2D_ARGS:-f32} > gpurun_out/pt_$i.log 2>&1
  echo "group $i rc=$?"
done
python3 - <<'PY' > gpurun_out/pmc_t2d_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pt_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k[:70])
    for c, v in sorted(d.items()): print("   %-34s %.4g  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
cat gpurun_out/pmc_t2d_summary.txt
rm -rf gpurun_out/pt_*
