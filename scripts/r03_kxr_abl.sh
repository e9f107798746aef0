#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
for abl in ${ABLS:-0 1 2}; do
  (cd stmask_amd/csrc && touch conv_kxr.hip && make -s EXTRA=-DKX_ABL=$abl 2>&1 | grep -E "error")
  echo "== KX_ABL=$abl"; timeout 600 python scripts/${SCRIPT:-bench_kxr.py} 32 2>&1 | grep -v amdgpu.ids | cut -c1-150
done > $OUT/bench_kxr_abl.txt 2>&1
(cd stmask_amd/csrc && touch conv_kxr.hip && make -s 2>&1 | grep error)
cat $OUT/bench_kxr_abl.txt
