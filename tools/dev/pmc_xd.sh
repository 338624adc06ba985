#!/bin/bash
# SQ / LDS / TA counters of the CSR-stream kernel on the 512^3 Laplacian (one pass per counter group; no tracing).
# usage: tools/dev/pmc_xd.sh   (SMH_STREAM_VDICT=0 in the environment: the kernel that reads the value array)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
         "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/px_$i
  timeout -k 5 300 rocprofv3 --pmc $C --output-format csv -d gpurun_out/px_$i -- python3 tools/quick_bench.py --cases lap512 --only-blocked --tiled none > gpurun_out/px_$i.log 2>&1
  echo "group $i rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/px_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_spmv_stream_xd" not in k: continue
        k = k[k.index("k_spmv_stream_xd"):].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k[:90])
    for c, v in sorted(d.items()): print("   %-34s %.5g  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
rm -rf gpurun_out/px_*
