# usage: variants_layers.sh "name1 name2 ..." [bench_layers args] -- scripts/bench_layers.py under each variant library (stmask_amd/variants/libstmask_hip_<name>.so;
# "default" = the shipped library), twice in alternation
cd $GRAFT_REPO_ROOT; NAMES=$1; shift
for rep in 1 2; do for n in $NAMES; do
  echo "== $n"
  if [ $n = default ]; then python scripts/bench_layers.py "$@" 2>&1 | grep -v amdgpu | tail -n +2 | head -${LINES:-3}
  else STM_LIBRARY=$GRAFT_REPO_ROOT/stmask_amd/variants/libstmask_hip_$n.so python scripts/bench_layers.py "$@" 2>&1 | grep -v amdgpu | tail -n +2 | head -${LINES:-3}; fi
done; done
