#!/bin/bash
# SQ / LDS counters of K1s against K1s-p on the 512^3 Laplacian (one pass per counter group and kernel choice; no tracing)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for P in 0; do
  i=0
  for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1))
    rm -rf gpurun_out/pp_${P}_$i
    SMH_STREAM_PIPE=$P timeout -k 5 200 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pp_${P}_$i -- python3 tools/cg_bench.py --iters 3 > gpurun_out/pp_${P}_$i.log 2>&1
    echo "pipe=$P group $i rc=$?"
  done
done
python3 - <<'PY' > gpurun_out/pmc_pipe_summary.txt
import csv, glob, collections
for P in (0,):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/pp_%d_*/**/*counter_collection.csv" % P, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_spmv_stream" not in k: continue
            k = k.split("(")[0].replace("void smh::", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print("pipe=%d %s" % (P, k[:70]))
        for c, v in sorted(d.items()): print("   %-34s %.4g  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
cat gpurun_out/pmc_pipe_summary.txt
rm -rf gpurun_out/pp_0_* gpurun_out/pp_1_*
