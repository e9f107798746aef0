#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for abl in 0 1 2 4 7; do
  (cd stmask_amd/csrc && touch deform_im2col.hip && make -s EXTRA=-DSL_ABL=$abl 2>&1 | grep -E "error")
  echo "== SL_ABL=$abl"; timeout 300 python scripts/ab_dcn_lds.py 32 1 2>&1 | grep -v amdgpu.ids | sed -n 1,4p | cut -c1-170
done
(cd stmask_amd/csrc && touch deform_im2col.hip && make -s 2>&1 | grep error)
