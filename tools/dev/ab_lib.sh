#!/bin/bash
# A/B of two builds of the library inside one call: the product .so against sparsemat_amd/libsparsemat_hip_ab.so (built with
# sparsemat_amd.build.build(extra_flags=[...], out=..., objsuffix="_ab")), interleaved.   usage: tools/dev/ab_lib.sh <command ...>
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  echo "== product"; "$@" 2>&1 | grep -E "SpMV auto|per iteration|ms/iter|CG " | head -4
  echo "== ab";      SPARSEMAT_HIP_LIB=$GRAFT_REPO_ROOT/sparsemat_amd/libsparsemat_hip_ab.so "$@" 2>&1 | grep -E "SpMV auto|per iteration|ms/iter|CG " | head -4
done
