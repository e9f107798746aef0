#!/bin/bash
# kernel-choice / ablation timing of the bf16-split conv (results are WRONG under a debug mask; timing only)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for k in 1 2; do
  echo "== STM_CONV_KERNEL=$k"; STM_CONV_KERNEL=$k timeout 200 python scripts/bench_conv.py 8 3 2>&1 | grep -v amdgpu | cut -c1-75
done
