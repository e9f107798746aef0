# usage: ab_env_session.sh VAR "v1 v2 ..." [bench args]  -- alternating bench.py runs (frames/s, ms per step) under VAR=v, twice each; output also in gpurun_out/ab_VAR.txt
cd $GRAFT_REPO_ROOT; VAR=$1; VALS=$2; shift 2; mkdir -p gpurun_out
for rep in 1 2; do for v in $VALS; do
  echo "== $VAR=$v $*"; env $VAR=$v timeout 900 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done 2>&1 | tee gpurun_out/ab_$VAR.txt
