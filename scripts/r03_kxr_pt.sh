#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
for v in 4 3 2; do
  (cd stmask_amd/csrc && touch conv_kxr.hip && make -s EXTRA="-DKX_PT_MAX=$v" 2>&1 | grep -E " error")
  echo "== KX_PT_MAX=$v"; timeout 600 python scripts/bench_kxr.py 32 2>&1 | grep -v amdgpu.ids | cut -c1-160
done > $OUT/kxr_pt.txt 2>&1
(cd stmask_amd/csrc && touch conv_kxr.hip && make -s 2>&1 | grep error)
cat $OUT/kxr_pt.txt
