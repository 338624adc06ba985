#!/bin/bash
# Round-4 evidence on ONE box: bench.py as the driver runs it (headline + configs C3 / C4 SpMV / C4 CG + inspector costs), the same
# command under `rocprofv3 --kernel-trace --stats` (per-kernel average durations must agree with the line's HIP-event times), and
# calibrated PMC traffic per product of the C3 and C4 kernels (separate --pmc passes, never mixed with tracing).
# Usage (GPU box): bash tools/r04_evidence.sh  -> gpurun_out/r04_evidence/
set -o pipefail
out=gpurun_out/r04_evidence
mkdir -p "$out"
export TMPDIR=/tmp
echo "[1/5] bench.py"; python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || { echo "bench.py failed"; tail -5 "$out/bench.err"; exit 1; }
echo "[2/5] bench.py under rocprofv3 --kernel-trace --stats"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04_kt -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-traffic --no-cpu-baseline > "$GRAFT_REPO_ROOT/$out/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$out/bench_under_rocprof.err" ) || { echo "rocprofv3 run failed"; exit 1; }
f=$(find /tmp/r04_kt -name "*kernel_stats.csv" | head -1); cp "$f" "$out/bench_kernel_stats.csv"; head -12 "$out/bench_kernel_stats.csv"
echo "[3/5] PMC traffic: C3 (power law f64, tiled)"; python3 tools/pmc_kernels.py --case powerlaw --variant tiled --match k_t3_expand,k_t3_reduce > "$out/pmc_k2t_powerlaw.txt" 2>&1 || echo "pmc powerlaw failed"
echo "[4/5] PMC traffic: C4 (Laplacian 512^3 f32, stream)"; python3 tools/pmc_kernels.py --case lap512 --variant stream --match k_spmv_stream > "$out/pmc_k1s_xd_lap512.txt" 2>&1 || echo "pmc lap512 failed"
echo "[5/5] PMC traffic: f64 headline shape (window64, K1r)"; python3 tools/pmc_kernels.py --case window64 --variant auto --match k_spmv_ring2 > "$out/pmc_k1r_window64.txt" 2>&1 || echo "pmc window64 failed"
for f in pmc_k2t_powerlaw pmc_k1s_xd_lap512 pmc_k1r_window64; do tail -n 4 "$out/$f.txt"; done
