#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04; mkdir -p $OUT; cd $R
echo "== 1. correlation kernel A/B"; timeout 300 python scripts/bench_corr.py 32 8 1 2>&1 | grep -v amdgpu.ids | tee $OUT/corr_ab.txt
echo "== 2. kernel tests of the temporal ops"; timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -x -k "corr or roi" 2>&1 | tail -3
echo "== 3. schedule: next trunk early vs late"
for rep in 1 2; do for ov in late early; do echo -n "$ov: "; timeout 600 python bench.py --steps 16 --warmup 4 --overlap $ov --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'conv ms', r['ms_per_step'], 'frac', r['frac'], 'issued', r.get('frac_issued'))"; done; done | tee $OUT/overlap_ab.txt
