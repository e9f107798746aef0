#!/bin/bash
# timing ablations of conv_planar_kernel (results are WRONG under a debug mask; timing only)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for d in 0 1 2 3 4 5 7; do
  echo "== STM_CONV_DEBUG=$d"; STM_CONV_DEBUG=$d timeout 200 python scripts/bench_conv.py 8 3 2>&1 | grep -E "P3|proto 3x3" | cut -c1-118
done
