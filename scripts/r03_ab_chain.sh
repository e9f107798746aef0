#!/bin/bash
# same-box A/B of the chain kernel in the full step (alternating runs)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
for rep in 1 2 3; do
  for v in 0 1; do
    STM_CONV_CHAIN=$v timeout 600 python bench.py --steps 20 --warmup 4 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('chain=$v', d['value'], d['ms_per_step'], 'frac', r['frac'], 'trunk', r['frac_trunk_only'], 'hbm', r['hbm_bound_launches']['frac'], r['hbm_bound_launches']['ms_per_step'], 'mfma', r['mfma_bound_launches']['frac'])"
  done
done | tee $OUT/ab_chain.txt
