#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -p no:cacheprovider -x -k "dual" 2>&1 | tail -8
timeout 1800 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x > $OUT/pytest_dual.log 2>&1; tail -4 $OUT/pytest_dual.log
for f in 1 0; do STM_C3DS_FUSED=$f timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras --layer-table 2> $OUT/layers_c3ds$f.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('C3DS_FUSED=$f', d['value'], d['ms_per_step'], r['frac'], r['frac_trunk_only'], r['ms_per_step'])"; done
