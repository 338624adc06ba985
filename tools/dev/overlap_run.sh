#!/bin/bash
# the exchange beside the product, on one device: bench.py with 4 blocks sharing the GPU (PEER backend), its JSON line, and a
# rocprofv3 kernel trace of the same command summarised by tools/overlap_trace.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SMH_BENCH_SHARE_DEVICES=1
python3 bench.py --gpus 4 --rows 2500000 --steps 20 --warmup 3 > gpurun_out/r03_overlap_bench.json 2> gpurun_out/r03_overlap_bench.err || { tail -20 gpurun_out/r03_overlap_bench.err; exit 1; }
python3 -c "
import json; d = json.load(open('gpurun_out/r03_overlap_bench.json'))
print({k: d.get(k) for k in ('ms_per_step', 'ms_per_step_no_overlap', 'exchange_hidden_ms', 'spmv_kernel_ms', 'step_ms_block0_events', 'interior_rows_block0')})
print(d['config']['workload']); print('exchange_check', d.get('exchange_check'))"
rm -rf gpurun_out/ov_trace
SMH_BENCH_NO_ALLGATHER_LEG=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ov_trace -- python3 bench.py --gpus 4 --rows 2500000 --steps 10 --warmup 2 > /dev/null 2> gpurun_out/ov_trace.err || { tail gpurun_out/ov_trace.err; exit 1; }
python3 tools/overlap_trace.py gpurun_out/ov_trace
rm -rf gpurun_out/ov_trace
