#!/bin/bash
# The committed evidence of a round (GPU box): bench.py's JSON line, the rocprofv3 --kernel-trace --stats summary of the
# same command, the CG benchmark and its kernel stats -> gpurun_out/final_* (copied to profiles/rNN_* by hand).
# Never combines --pmc with tracing; the program itself follows `--`.
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || echo "bench.py failed"
rm -rf gpurun_out/prof_bench gpurun_out/prof_cg
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --no-traffic --no-cpu-baseline \
    > gpurun_out/final_bench_under_rocprof.json 2> gpurun_out/final_bench_under_rocprof.err || echo "profiled bench failed"
find gpurun_out/prof_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/final_bench_kernel_stats.csv \;
timeout -k 10 300 python3 tools/cg_bench.py --iters 100 2>&1 | grep -v amdgpu.ids > gpurun_out/final_cg.txt || echo "cg_bench failed"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cg -- python3 tools/cg_bench.py --iters 100 \
    > gpurun_out/final_cg_under_rocprof.txt 2>&1 || echo "profiled cg_bench failed"
find gpurun_out/prof_cg -name "*kernel_stats.csv" -exec cp {} gpurun_out/final_cg_kernel_stats.csv \;
rm -rf gpurun_out/prof_bench gpurun_out/prof_cg
head -c 600 gpurun_out/final_bench.json; echo; head -5 gpurun_out/final_bench_kernel_stats.csv; tail -2 gpurun_out/final_cg.txt; head -8 gpurun_out/final_cg_kernel_stats.csv
