#!/bin/bash
# Round 4, GPU session 4: measurement gaps (RCCL branch, the un-optimized drop-in path), host profile of the single-stream step
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04; mkdir -p $OUT; cd $R
echo "== 1. RCCL branch: one rank under torch.distributed.run, backend nccl"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu-baseline --no-extras > $OUT/rccl_world1.json 2> $OUT/rccl_world1.err; echo "exit $?"; python -c "
import json; d=json.load(open('$OUT/rccl_world1.json')); print(d['value'], d['ms_per_step'], d.get('collective'))"; tail -3 $OUT/rccl_world1.err
echo "== 2. the un-optimized drop-in path: reference-shaped modules on the shims, no BN folding, no planar graph (8 clips)"
timeout 900 python bench.py --no-fuse --no-planar --clips 8 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_dropin_nofuse.json 2> $OUT/bench_dropin_nofuse.err; echo "exit $?"; python -c "
import json; d=json.load(open('$OUT/bench_dropin_nofuse.json')); print(d['value'], d['ms_per_step'], d['config']['inference_graph'], d['roofline'].get('frac'), d['roofline'].get('achieved'))"
timeout 900 python bench.py --no-planar --clips 8 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_dropin_fused.json 2> $OUT/bench_dropin_fused.err; echo "exit $?"; python -c "
import json; d=json.load(open('$OUT/bench_dropin_fused.json')); print(d['value'], d['ms_per_step'], d['config']['inference_graph'], d['roofline'].get('frac'), d['roofline'].get('achieved'))"
echo "== 3. host profile, 1 clip and 8 clips"
timeout 600 python scripts/prof_host.py 1 300 2>&1 | grep -v amdgpu.ids > $OUT/prof_host_clips1.txt; head -70 $OUT/prof_host_clips1.txt | cut -c1-180
timeout 600 python scripts/prof_host.py 8 100 2>&1 | grep -v amdgpu.ids > $OUT/prof_host_clips8.txt; head -30 $OUT/prof_host_clips8.txt | cut -c1-180
