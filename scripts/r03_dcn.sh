#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -x -k "sample or dcn or deform" 2>&1 | tail -6
timeout 600 python scripts/ab_dcn_lds.py 32 1 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_dcn_lds.txt
