#!/bin/bash
# Same-box alternating A/B of the whole step with the deformable layers fused (STM_DCN_FUSED=1, csrc/dcn_fused.hip) or as the sampler + product pair (0);
# STM_DCN_FUSED_MIN_TILES = smallest grid the fused kernel takes (960 tiles: layer2 at 32 clips, 512: layer3, 256: layer4).
# usage (GPU box): bash scripts/ab_dcn_fused_graph.sh [clips=32] [rounds=3] ["min-tiles values", default "200"]
C=${1:-32}; R=${2:-3}; MT=${3:-200}
for r in $(seq $R); do
  for f in 0 $MT; do
    if [ $f = 0 ]; then export STM_DCN_FUSED=0; else export STM_DCN_FUSED=1 STM_DCN_FUSED_MIN_TILES=$f; fi
    python bench.py --clips $C --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('fused from $f tiles (0 = never), clips $C: %.1f frames/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"
  done
done
