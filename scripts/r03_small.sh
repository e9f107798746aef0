#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do
for t in _old_r02 .; do
  for c in 1 8; do
    (cd $t && timeout 300 python bench.py --clips $c --steps 60 --warmup 6 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', 'clips', $c, d['value'], d['ms_per_step'])")
  done
done
done
