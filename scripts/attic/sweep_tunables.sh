#!/bin/bash
# A/B sweep of the launch heuristics on the benchmark graph (one box, one call: boxes differ by +-2.5 %)
# usage: CLIPS=8 scripts/sweep_tunables.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
CLIPS=${CLIPS:-32}
run() { echo "== $*"; env "$@" python bench.py --clips $CLIPS --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_step'])"; }
run A=0
run STM_TILE64_MAX_SLABS=4
run STM_TILE64_MAX_SLABS=16
run STM_CONV_RING64=2
run STM_CONV_RING64=4
run STM_CONV_SK_TARGET=256
run STM_CONV_SK_TARGET=512
run STM_CONV_SK_TARGET=1024
run STM_CONV_MG=1
run STM_CONV_MG=2
run A=1
