#!/bin/bash
# Which DCN layers should take the fused kernel?  Whole step, same box, alternating: the pair everywhere (STM_DCN_FUSED=0), the fused kernel on every layer with a
# large enough grid (STM_DCN_FUSED_RULE=0), only on the layers with one 128-channel tile (or two at stride 2) (RULE=1).  R50 (7 DCN layers) and R101 (11).
# usage (GPU box): bash scripts/ab_dcn_fused_rule.sh [clips=32] [rounds=2]
C=${1:-32}; R=${2:-2}
for cfg in STMask_plus_resnet50_config STMask_plus_base_ali_config; do
for r in $(seq $R); do
  for f in off 0 1; do
    if [ $f = off ]; then export STM_DCN_FUSED=0; else export STM_DCN_FUSED=1 STM_DCN_FUSED_RULE=$f; fi
    python bench.py --config $cfg --clips $C --steps 16 --warmup 4 --no-cpu-baseline --no-extras --no-sampler-pass 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('$cfg clips $C fused rule $f: %.1f frames/s  %.3f ms/step  fused launches/step %s' % (d['value'], d['ms_per_step'], (d.get('roofline_dcn_fused') or {}).get('launches', 0) / d['steps']))"
  done
done; done
