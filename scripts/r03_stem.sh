#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -p no:cacheprovider -x -k "stem" 2>&1 | tail -15
python - <<'PY'
import torch, sys
sys.path.insert(0,'.')
from stmask_amd import ops
import stmask_amd.planar as pl
x=torch.randn(32,384,640,3,device='cuda'); w=torch.randn(64,3,7,7,device='cuda')*0.08; b=torch.randn(64,device='cuda')
pk,osc=ops.stem_pack_weights(w,1)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
print('fused stem, 32 frames 384x640: %.1f us' % t(lambda: ops.stem_fused(x,pk,osc,b,1)))
wr=torch.nn.functional.pad(w.permute(0,2,3,1).reshape(64,7,21),(0,11)).permute(0,2,1).reshape(64,32,7,1).contiguous()
conv=pl.PlanarConv(wr,None,(2,1),(3,0),relu=False,fmt=1)
def old():
    rp,Wo=ops.stem_rows_planes(x,7,2,3,1); y=conv(rp,("img",32,384,Wo),out="f32").view(32,192,Wo,64); return ops.bias_relu_maxpool_planes(y,b,1)
print('three-kernel stem: %.1f us' % t(old))
PY
