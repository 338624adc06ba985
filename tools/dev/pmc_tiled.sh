#!/bin/bash
# memory-side counters of K2t's two kernels on C2-uniform (one pass per counter group; no tracing)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for C in "FETCH_SIZE WRITE_SIZE" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
         "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/pk_$i
  timeout -k 5 300 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pk_$i -- python3 tools/quick_bench.py --cases uniform --only-blocked --tiled uniform > gpurun_out/pk_$i.log 2>&1
  echo "group $i rc=$?"
done
python3 - <<'PY' > gpurun_out/pmc_tiled_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pk_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_t2_expand" not in k and "k_t2_reduce" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k[:70])
    for c, v in sorted(d.items()): print("   %-34s %.5g  (%d launches)" % (c, sum(v) / len(v), len(v)))
    if "TCC_EA_RDREQ_sum" in d:
        rd, rd32 = sum(d["TCC_EA_RDREQ_sum"]) / len(d["TCC_EA_RDREQ_sum"]), sum(d["TCC_EA_RDREQ_32B_sum"]) / len(d["TCC_EA_RDREQ_32B_sum"])
        wr, wr64 = sum(d["TCC_EA_WRREQ_sum"]) / len(d["TCC_EA_WRREQ_sum"]), sum(d["TCC_EA_WRREQ_64B_sum"]) / len(d["TCC_EA_WRREQ_64B_sum"])
        print("   -> read %.3f GB (32-B requests x 32 + the rest x 64), written %.3f GB" % ((rd32 * 32 + (rd - rd32) * 64) / 1e9, (wr64 * 64 + (wr - wr64) * 32) / 1e9))
PY
cat gpurun_out/pmc_tiled_summary.txt
rm -rf gpurun_out/pk_*
