#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity.py tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -x > $OUT/pytest_gpu3.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu3.log; tail -5 $OUT/pytest_gpu3.log
timeout 900 python bench.py --steps 20 --warmup 4 --layer-table > $OUT/bench2.json 2> $OUT/bench2_layers.txt; echo "bench exit $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench2.json'))
r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'trunk', r['frac_trunk_only'], 'conv ms', r['ms_per_step'])
print('mfma', r['mfma_bound_launches']['frac'], r['mfma_bound_launches']['ms_per_step'], 'hbm', r['hbm_bound_launches']['frac'], r['hbm_bound_launches']['ms_per_step'])
for k,v in d['extras'].items(): print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('frac_trunk_only'))
print(d['parity']['matched_frac'], d['parity']['mask_l2'])
PY
head -24 $OUT/bench2_layers.txt | cut -c1-100
timeout 600 python scripts/bench_kernels.py --batch 32 --what corr 2>&1 | grep -v amdgpu | tail -5
