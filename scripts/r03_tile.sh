#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
for v in 8 0; do STM_TILE64_MAX_SLABS=$v timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --layer-table 2> $OUT/layers_t64_$v.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('TILE64_MAX_SLABS=$v', d['value'], d['ms_per_step'], r['frac'], r['ms_per_step'])"; done
python - <<'PY'
def load(p):
    d={}
    for l in open(p):
        f=l.split()
        if len(f)==11 and f[0].isdigit():
            d[tuple(f[:6])]=(f[6], float(f[8]), float(f[10]))
    return d
a=load('gpurun_out/r03/layers_t64_8.txt'); b=load('gpurun_out/r03/layers_t64_0.txt')
for k in a:
    if k in b and a[k][0]!=b[k][0]: print(k, 'tile', a[k][0], a[k][1], 'us ->', b[k][0], b[k][1], 'us')
PY
