for d in 0 1 2; do echo "== STM_IM2COL_DEBUG=$d"; STM_IM2COL_DEBUG=$d STM_IM2COL_VARIANT=3 python - <<'PY'
import os,sys,torch
sys.path.insert(0,'.')
from scripts.bench_kernels import _one, R50_DCN
tot=0;nbt=0
for name,C,H,W,s in R50_DCN:
    ms,nb=_one(8,C,H,W,s,variant=3); tot+=ms; nbt+=nb
    print(f"  {name}: {ms*1e3:7.1f} us {nb/ms/1e6:6.0f} GB/s")
print(f"  TOTAL {tot*1e3:.1f} us {nbt/tot/1e6:.0f} GB/s")
PY
done
