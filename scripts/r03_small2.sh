#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for ex in clips1 clips8,clips1 realistic,clips1; do
  timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --extras $ex 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ex', d['value'], {k:(v.get('value'), v.get('ms_per_step')) for k,v in d['extras'].items()})"
done
