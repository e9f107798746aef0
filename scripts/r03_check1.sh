#!/bin/bash
# round 3, GPU session 1: the new parity tests, the two-rank-on-one-GPU run, the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py -q -m gpu -p no:cacheprovider -x \
  -k "batched or fp16x1 or config5 or graph" > $OUT/pytest_new.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_new.log; tail -15 $OUT/pytest_new.log
timeout 600 python bench.py --world2-one-gpu --clips 4 --steps 12 --warmup 3 > $OUT/world2.json 2> $OUT/world2.err; echo "world2 exit $?"; tail -c 1500 $OUT/world2.json; tail -5 $OUT/world2.err
timeout 900 python bench.py --steps 20 --warmup 4 > $OUT/bench1.json 2> $OUT/bench1.err; echo "bench exit $?"; tail -3 $OUT/bench1.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench1.json'))
r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'trunk', r['frac_trunk_only'])
print('mfma', r['mfma_bound_launches']); print('hbm', r['hbm_bound_launches'])
for k,v in d['extras'].items(): print(k, {a:b for a,b in v.items() if a not in ('what','context')})
PY
