#!/bin/bash
# A/B of two libraries on the partitioned step, one process owning 4 and 8 blocks on ONE device (peer backend): old = SPARSEMAT_HIP_LIB
OLD=$PWD/sparsemat_amd/libsparsemat_hip_old.so
for rep in 1 2; do
  for n in 4 8; do
    for lib in old new; do
      if [ $lib = old ]; then export SPARSEMAT_HIP_LIB=$OLD; else unset SPARSEMAT_HIP_LIB; fi
      SMH_BENCH_SHARE_DEVICES=1 python3 bench.py --gpus $n --one-process --rows $((10000000 / n)) --steps 50 --warmup 10 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib N=$n', 'ms_per_step', round(d['ms_per_step'],4), 'no_overlap', round(d.get('ms_per_step_no_overlap') or 0,4), 'hidden', d.get('exchange_hidden_ms'), 'ok', d['exchange_check']['ok'])"
    done
  done
done
