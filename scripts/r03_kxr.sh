#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -p no:cacheprovider -x -k "kxr" > $OUT/pytest_kxr.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_kxr.log; tail -15 $OUT/pytest_kxr.log
timeout 600 python scripts/bench_kxr.py 32 > $OUT/bench_kxr.txt 2>&1; cat $OUT/bench_kxr.txt | grep -v amdgpu.ids
