#!/bin/bash
# rocprofv3 --kernel-trace --stats of K2t's two kernels on C2-uniform and C3 (no counters) -> gpurun_out/tiled_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_t2
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t2 -- python3 tools/quick_bench.py --cases uniform,powerlaw --only-blocked \
    > gpurun_out/tiled_under_rocprof.log 2>&1 || echo "profiled run failed"
find gpurun_out/prof_t2 -name "*kernel_stats.csv" -exec cp {} gpurun_out/tiled_kernel_stats.csv \;
rm -rf gpurun_out/prof_t2
grep "k_t3_\|Name" gpurun_out/tiled_kernel_stats.csv | cut -c1-200
grep -v amdgpu.ids gpurun_out/tiled_under_rocprof.log | cut -c1-200
