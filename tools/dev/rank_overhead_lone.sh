#!/bin/bash
# the rank path with ONE rank and real RCCL (communicator, smh_par_create_rank, the RCCL call sequence with a lone rank), tiny and mid-size blocks:
# what a step costs on the host when nothing shares the GPU
for rows in 100000 2000000 10000000; do
  for ex in window allgather; do
    SMH_BENCH_FORCE_PAR=1 SMH_PAR_EXCHANGE_SINGLE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --rows $rows --steps 200 --warmup 20 --no-traffic --no-cpu-baseline --exchange $ex 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lone rank rows', $rows, '$ex', 'ms_per_step', round(d['ms_per_step'],4), 'no_overlap', d.get('ms_per_step_no_overlap'), 'events', d.get('step_ms_block0_events'), d['config'].get('exchange_backend'))"
  done
  python3 bench.py --gpus 1 --rows $rows --steps 200 --warmup 20 --no-traffic --no-cpu-baseline --configs off | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('single    rows', $rows, 'ms_per_step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4))"
done
