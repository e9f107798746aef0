# timing ablations of the ring loops (library built with -DSTM_ABLATE; results of the ablated kernels are wrong by design)
cd $GRAFT_REPO_ROOT
for abl in ${ABLS:-0 16 32 64 1 7}; do
  echo "== ABL $abl"
  STM_LIBRARY=$GRAFT_REPO_ROOT/stmask_amd/variants/libstmask_hip_ablate.so STM_CONV_ABL=$abl python scripts/bench_layers.py --set ${SET:-mfma} 2>&1 | grep -v amdgpu | head -${LINES:-3}
done
