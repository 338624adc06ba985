#!/bin/bash
# host-side cost of one step of the rank path: tiny blocks (the GPU is idle most of the time), 2 and 4 ranks sharing the device through the mock RCCL
MOCK=$(python3 -c "import sys; sys.path.insert(0,'tests'); from util import build_mock_rccl; print(build_mock_rccl())")
for rows in 100000 2000000; do
  python3 bench.py --gpus 1 --rows $rows --steps 200 --warmup 20 --no-traffic --no-cpu-baseline --configs off | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=1 rows', $rows, 'ms_per_step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'])"
  for n in 2 4; do
    LD_PRELOAD=$MOCK SMH_BENCH_SHARE_DEVICES=1 python3 bench.py --gpus $n --rows $rows --steps 200 --warmup 20 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=$n rows', $rows, 'ms_per_step', d['ms_per_step'], 'no_overlap', d.get('ms_per_step_no_overlap'), 'block0 events', d.get('step_ms_block0_events'), d['config'].get('exchange'), d['config'].get('launch'))"
  done
done
