#!/bin/bash
# Round 4, GPU session 3: what the fixes cost, the z hand-over back on, full GPU tests, bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04; mkdir -p $OUT; cd $R
V=$R/stmask_amd/variants
echo "== 1. chain kernel: shipped (counted + drained + store wait states) vs alternatives, alternating"
for rep in 1 2; do for lib in "" $V/libstmask_hip_safe_lgkm.so $V/libstmask_hip_counted_voff.so $V/libstmask_hip_r03chain.so; do
  for B in 32 8; do echo -n "$(basename ${lib:-shipped}) "; STM_LIBRARY=$lib timeout 300 python scripts/bench_chain.py $B 2>&1 | grep "B="; done; done; done | tee $OUT/perf_chain_variants.txt
echo "== 2. kxr: drained barrier vs round 3's"
for rep in 1 2; do for lib in "" $V/libstmask_hip_kxr_nodrain.so; do echo "-- $(basename ${lib:-shipped})"; STM_LIBRARY=$lib timeout 600 python scripts/bench_kxr.py 32 2>&1 | grep -v amdgpu.ids | tail -12; done; done | tee $OUT/perf_kxr_drain.txt
echo "== 3. kxr / planar rings beside the pipeline hammer, shipped library"
timeout 900 python scripts/ring_stress.py --hammer pipe --launches 1600 --cases kxr,planar --clips 32,4 --json $OUT/stress3_conv_pipe.json > $OUT/stress3_conv_pipe.txt 2>&1; echo "exit $?"; grep -v "amdgpu.ids" $OUT/stress3_conv_pipe.txt | cut -c1-200 | tail -16
echo "== 4. two ranks on one GPU, default configuration (z hand-over on)"
for i in 1 2 3; do timeout 600 python bench.py --world2-one-gpu --clips 4 --steps 12 --warmup 3 > $OUT/world2_default_$i.json 2> $OUT/world2_default_$i.err; echo "exit $?"; python -c "
import json; d=json.load(open('$OUT/world2_default_$i.json')); print('run $i gather_ok', d['gather_ok'], 'max_abs', d['max_abs_diff_vs_solo'], [p['bit_equal_to_solo_run'] for p in d['per_rank']])"; done
echo "== 5. bench"
timeout 1200 python bench.py --steps 20 --warmup 4 > $OUT/bench1.json 2> $OUT/bench1.err; echo "bench exit $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04/bench1.json'))
r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'trunk', r.get('frac_trunk_only'), 'conv ms', r.get('ms_per_step'))
for k,v in (d.get('extras') or {}).items(): print(k, v.get('value'), v.get('ms_per_step'))
print(d['parity'])
PY
echo "== 6. pytest -m gpu"
timeout 2700 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $OUT/pytest_gpu1.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest_gpu1.log
