#!/bin/bash
# timing ablations of the planar kernels (results are WRONG under a debug mask; timing only)
# bits: 1 no activation DMA in the loop, 2 no barrier, 4 no MFMA (planar kernel only), 8 no weight DMA (kx kernel)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for d in 0 1 8 9 11; do
  echo "== STM_CONV_DEBUG=$d (kx kernel)"; STM_CONV_DEBUG=$d timeout 200 python scripts/bench_conv.py 8 3 2>&1 | grep -E "P3|proto 3x3" | cut -c1-75
done
