#!/bin/bash
# Every BASELINE config + the widened rows (assembly, transpose, prod, the single-process partition) + bench.py on ONE
# box with the current library -> gpurun_out/all_configs.log (copied to profiles/rNN_final_all_configs.log by hand).
# Usage (GPU box): bash tools/all_configs.sh
set -o pipefail
out=gpurun_out/all_configs.log
mkdir -p gpurun_out
: > "$out"
run() {  # title, command...
    echo "### $1" >> "$out"
    shift
    timeout -k 10 600 "$@" >> "$out" 2>&1 || { echo "FAILED: $*" >> "$out"; return 1; }
}
run "C1" python3 tests/bench/c1_bench.py &&
run "C2 banded (stratified) / window (without replacement) / contiguous band / uniform, C3 power law, C4 Laplacian 512^3 (quick_bench, single launches)" \
    python3 tools/quick_bench.py --cases banded,window,diag,uniform,powerlaw,lap512,banded64 --lanes 8 --cb-shifts 18,19 &&
run "C4 CG" python3 tools/cg_bench.py --iters 100 &&
run "C4 Jacobi PCG (extension) next to CG" python3 tools/pcg_bench.py &&
run "SparseMatrix::inner_prod next to the plain product" python3 tools/inner_prod_bench.py &&
run "DenseVec kernels" python3 tools/blas1_bench.py &&
run "assembly" python3 tests/bench/assemble_bench.py &&
run "transpose, column tables" python3 tests/bench/transpose_bench.py &&
run "prod" python3 tests/bench/prod_bench.py &&
run "single-process SparseMatPar (blocks on one device)" python3 tests/bench/par_local_bench.py &&
run "bench.py" python3 bench.py
grep -v "amdgpu.ids" "$out" > "$out.tmp" && mv "$out.tmp" "$out"
tail -5 "$out"
