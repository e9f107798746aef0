#!/bin/bash
# Where does a rank's step time go under torch.distributed.run + RCCL at world size 1?  Same box, same build, alternating:
#   plain            python bench.py
#   plain-omp1       OMP_NUM_THREADS=1 python bench.py               (the launcher's default environment, no process group)
#   nccl             python -m torch.distributed.run ... bench.py    (RCCL all-gather on the communication stream every step)
#   nccl-omp32       the same with OMP_NUM_THREADS=32 exported       (the launcher leaves a set value alone)
#   gloo             launcher, --backend gloo (host-staged exchange)
# Prints frames/s, ms/step, the convolution launches' ms/step (eager event pass) and the difference = everything else.
# usage (GPU box): bash scripts/ab_launcher_overhead.sh [rounds=2] [clips=32]
R=${1:-2}; C=${2:-32}
ARGS="--clips $C --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
show() { python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {}); c = d.get('collective') or {}
        print('%-12s %8.1f frames/s  %7.3f ms/step   conv launches %7.3f ms/step   other %6.3f   %s' % ('$1', d['value'], d['ms_per_step'], r.get('ms_per_step', 0), d['ms_per_step'] - r.get('ms_per_step', 0),
              ('all-gathers %s on comm stream %s, equal %s' % (c.get('all_gathers_on_comm_stream'), c.get('comm_stream'), c.get('last_gather_equals_local_block'))) if c else ''))"; }
P=29530
for r in $(seq $R); do
  python bench.py $ARGS 2>/dev/null | show plain
  OMP_NUM_THREADS=1 python bench.py $ARGS 2>/dev/null | show plain-omp1
  P=$((P+1)); python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 1 $ARGS 2>/dev/null | show nccl
  P=$((P+1)); OMP_NUM_THREADS=32 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 1 $ARGS 2>/dev/null | show nccl-omp32
  P=$((P+1)); python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 1 --backend gloo $ARGS 2>/dev/null | show gloo
done
