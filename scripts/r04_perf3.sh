#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04; mkdir -p $OUT; cd $R
echo "== conv tests with the regrouped tile map"
STM_CONV_NSUB=4 timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
echo "== tile map A/B (alternating)"
for rep in 1 2; do for ns in 0 4 2; do echo -n "nsub=$ns: "; STM_CONV_NSUB=$ns timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'conv ms', r['ms_per_step'], 'frac', r['frac'], 'issued', r.get('frac_issued'), 'mfma-bound', r['mfma_bound_launches']['ms_per_step'], r['mfma_bound_launches']['frac'])"; done; done | tee $OUT/nsub_ab.txt
