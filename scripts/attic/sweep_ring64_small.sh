R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() { echo "== clips $CLIPS $*"; env "$@" python bench.py --clips $CLIPS --steps $STEPS --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for c in "1 100" "2 80" "4 60" "8 40" "32 20"; do set -- $c; CLIPS=$1; STEPS=$2
run STM_CONV_RING64_SMALL=0
run STM_CONV_RING64_SMALL=128
run STM_CONV_RING64_SMALL=256
run STM_CONV_RING64_SMALL=512
run STM_CONV_RING64_SMALL=1024
done
