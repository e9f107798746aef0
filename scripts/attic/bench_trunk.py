#!/usr/bin/env python3
"""Where does the dense trunk time go?  forward_single at batch B under different dense-conv library settings."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import synthetic  # noqa: E402
from stmask_amd.config import get_cfg  # noqa: E402
from stmask_amd.model import STMask  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = sys.argv[2] if len(sys.argv) > 2 else "bench"   # bench | nobench
cl = len(sys.argv) > 3 and sys.argv[3] == "cl"
torch.backends.cudnn.benchmark = (mode == "bench")
net = STMask(get_cfg("STMask_plus_resnet50_config"))
net.eval()
synthetic.fill_state_dict(net, seed=0, bg_bias=synthetic.BENCH_BG_BIAS)
net = net.cuda()
x = synthetic.synthetic_clip(B, 384, 640, seed=0).cuda()
if cl:
    net = net.to(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)


def sections(x):
    t = {}
    def tic():
        torch.cuda.synchronize(); return time.perf_counter()
    t0 = tic(); bb = net.backbone(x)
    t1 = tic(); fpn = net.fpn([bb[i] for i in net.backbone_selected])
    t2 = tic(); proto = torch.relu(net.proto_net(fpn[0]))
    t3 = tic()
    for idx, layer in zip(net.selected_layers, net.prediction_layers):
        layer(fpn[idx])
    t4 = tic()
    return dict(backbone=t1 - t0, fpn=t2 - t1, proto=t3 - t2, heads=t4 - t3, total=t4 - t0)


with torch.no_grad():
    t0 = time.perf_counter(); s = sections(x); first = time.perf_counter() - t0
    print(f"B={B} mode={mode} channels_last={cl} MIOPEN_FIND_MODE={os.environ.get('MIOPEN_FIND_MODE')} first pass {first:.1f} s", flush=True)
    sections(x)
    acc = None
    for _ in range(5):
        s = sections(x)
        acc = s if acc is None else {k: acc[k] + s[k] for k in s}
    print("  steady ms:", {k: round(v / 5 * 1e3, 2) for k, v in acc.items()}, f" => {B / (acc['total'] / 5):.1f} frames/s trunk-only",
          f"({155.1e9 * B / (acc['total'] / 5) / 1e12:.1f} TFLOP/s dense-equivalent)", flush=True)
