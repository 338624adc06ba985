#!/bin/bash
# Round-4 evidence on ONE box: bench.py as the driver runs it (headline + configs C3 / C4 SpMV / C4 CG + inspector costs), the same
# command under `rocprofv3 --kernel-trace --stats` (per-kernel average durations must agree with the line's HIP-event times), and
# calibrated PMC traffic per product of the C3 and C4 kernels (separate --pmc passes, never mixed with tracing).
# Usage (GPU box): bash tools/r04_evidence.sh  -> gpurun_out/r04_evidence/
set -o pipefail
out=gpurun_out/r04_evidence
mkdir -p "$out"
export TMPDIR=/tmp
echo "[1/7] bench.py"; python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || { echo "bench.py failed"; tail -5 "$out/bench.err"; exit 1; }
echo "[2/7] bench.py under rocprofv3 --kernel-trace --stats"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_kt -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-traffic --no-cpu-baseline > "$GRAFT_REPO_ROOT/$out/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$out/bench_under_rocprof.err" ) || { echo "rocprofv3 run failed"; exit 1; }
f=$(find /tmp/r04_kt -name "*kernel_stats.csv" | head -1); cp "$f" "$out/bench_kernel_stats.csv"; head -12 "$out/bench_kernel_stats.csv"
echo "[3/7] PMC traffic: C3 (power law f64, tiled)"; python3 tools/pmc_kernels.py --case powerlaw --variant tiled --match k_t3_expand,k_t3_reduce > "$out/pmc_k2t_powerlaw.txt" 2>&1 || echo "pmc powerlaw failed"
echo "[4/7] PMC traffic: C4 (Laplacian 512^3 f32, stream)"; python3 tools/pmc_kernels.py --case lap512 --variant stream --match k_spmv_stream > "$out/pmc_k1s_xd_lap512.txt" 2>&1 || echo "pmc lap512 failed"
echo "[5/7] PMC traffic: f64 headline shape (window64, K1r)"; python3 tools/pmc_kernels.py --case window64 --variant auto --match k_spmv_ring2 > "$out/pmc_k1r_window64.txt" 2>&1 || echo "pmc window64 failed"
echo "[6/7] C4 CG alone (tools/cg_bench.py, 512^3 f32, 100 iterations), then under rocprofv3 --kernel-trace --stats: the per-kernel times of one iteration at its size"
python3 tools/cg_bench.py --iters 100 > "$out/cg.txt" 2>&1 || echo "cg_bench failed"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_cg -- python3 "$GRAFT_REPO_ROOT/tools/cg_bench.py" --iters 100 > "$GRAFT_REPO_ROOT/$out/cg_under_rocprof.txt" 2>&1 ) || echo "cg under rocprofv3 failed"
f=$(find /tmp/r04_cg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/cg_kernel_stats.csv" && head -8 "$out/cg_kernel_stats.csv" | cut -c1-200
echo "[7/7] the same in f64 (K1s XD-V on persistent workgroups)"
python3 tools/cg_bench.py --iters 100 --dtype f64 > "$out/cg_f64.txt" 2>&1 || echo "cg_bench f64 failed"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_cg64 -- python3 "$GRAFT_REPO_ROOT/tools/cg_bench.py" --iters 100 --dtype f64 > "$GRAFT_REPO_ROOT/$out/cg_f64_under_rocprof.txt" 2>&1 ) || echo "cg f64 under rocprofv3 failed"
f=$(find /tmp/r04_cg64 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/cg_f64_kernel_stats.csv"
for f in pmc_k2t_powerlaw pmc_k1s_xd_lap512 pmc_k1r_window64; do tail -n 4 "$out/$f.txt"; done
