#!/bin/bash
# same-box A/B of TemporalNet's border-class launches in the full step (alternating runs)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -q -x -k "window or border" 2>&1 | tail -3
for rep in 1 2; do
  for v in 0 1; do
    STM_TN_BORDER=$v timeout 600 python bench.py --steps 20 --warmup 4 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('border=$v', d['value'], d['ms_per_step'], 'frac', r['frac'], 'trunk', r['frac_trunk_only'], 'conv ms', r['ms_per_step'], 'tflop/step', r['tflop_per_step'], 'parity', d['parity']['matched_frac'], d['parity']['mask_l2'])"
  done
done | tee $OUT/ab_border.txt
