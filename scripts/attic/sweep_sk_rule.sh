#!/bin/bash
# split-K rule of the 64-wide tiles, round 1's (STM_CONV_SK_RULE=1) against the current one, alternating on one box
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { echo "== clips $CLIPS $*"; env "$@" python bench.py --clips $CLIPS --steps $STEPS --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for c in "1 100" "2 80" "4 60" "8 40" "32 16"; do set -- $c; CLIPS=$1; STEPS=$2
run STM_CONV_SK_RULE=1
run STM_CONV_SK_RULE=0
run STM_CONV_SK_RULE=1
run STM_CONV_SK_RULE=0
done
