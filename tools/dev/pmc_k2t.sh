#!/bin/bash
# SQ / LDS / TA / TCC counters of K2t's two passes on C3 (power law f64) -- and on C2-uniform f32 with CASE=uniform --, one rocprofv3
# --pmc pass per counter group (never mixed with tracing).  usage: [CASE=powerlaw|uniform] tools/dev/pmc_k2t.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CASE=${CASE:-powerlaw}
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_STALL_sum TCC_EA_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_avr"; do
  i=$((i+1))
  rm -rf gpurun_out/pk_$i
  timeout -k 5 300 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pk_$i -- python3 tools/pmc_kernels.py --child --case $CASE --variant tiled > gpurun_out/pk_$i.log 2>&1
  echo "group $i rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pk_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_t3_expand" not in k and "k_t3_reduce" not in k: continue
        k = k[k.index("k_t3_"):].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k[:90])
    for c, v in sorted(d.items()): print("   %-34s %.5g  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
for f in gpurun_out/pk_4.log gpurun_out/pk_5.log; do tail -n 3 "$f"; done
rm -rf gpurun_out/pk_*/
