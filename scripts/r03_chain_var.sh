#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
for v in $VARIANTS; do
  (cd stmask_amd/csrc && touch conv_chain.hip && make -s EXTRA="$(echo $v | tr ',' ' ')" 2>&1 | grep -E "error")
  echo "== $v"; timeout 600 python scripts/bench_chain.py 32 2>&1 | grep -v amdgpu.ids | grep "B=" | cut -c1-150
done > $OUT/bench_chain_var.txt 2>&1
(cd stmask_amd/csrc && touch conv_chain.hip && make -s 2>&1 | grep error)
cat $OUT/bench_chain_var.txt
