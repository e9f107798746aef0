#!/bin/bash
# A/B of the MFMA shape in the planar kernel (both give valid results)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for m in 32 16 32 16; do
  echo "== STM_CONV_MFMA=$m"; STM_CONV_MFMA=$m timeout 200 python scripts/bench_conv.py 8 3 2>&1 | grep -E "P3|proto 3x3|layer1 3x3|layer2 1x1 128|TOTAL" | cut -c1-118
done
