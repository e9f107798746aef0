cd $GRAFT_REPO_ROOT
for abl in 0 16 32 64 1 7; do
  echo "== ABL $abl"
  STM_LIBRARY=$GRAFT_REPO_ROOT/stmask_amd/libstmask_hip_ablate.so STM_CONV_ABL=$abl python scripts/bench_layers.py --set mfma 2>&1 | grep -v amdgpu
done
