#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
for abl in ${ABLS:-0 1 2 3 4 8 12 16 32}; do
  (cd stmask_amd/csrc && touch conv_chain.hip && make -s EXTRA=-DCH_ABL=$abl 2>&1 | grep -E "error")
  echo "== CH_ABL=$abl"; timeout 600 python scripts/bench_chain.py 32 2>&1 | grep -v amdgpu.ids | grep "B=" | cut -c1-150
done > $OUT/bench_chain_abl.txt 2>&1
(cd stmask_amd/csrc && touch conv_chain.hip && make -s 2>&1 | grep error)
cat $OUT/bench_chain_abl.txt
