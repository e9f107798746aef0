#!/usr/bin/env python3
"""Driver for rocprofv3 passes on the bf16-split conv: tower 3x3 at P3 and the proto-net 3x3 at batch 8, REPS launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
planes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for H, W, C, O, k in [(48, 80, 256, 256, 3), (96, 160, 256, 256, 3), (48, 80, 128, 512, 1), (24, 40, 256, 256, 3)]:
    x = torch.randn(8, H, W, C, device="cuda")
    w = torch.randn(O, C, k, k, device="cuda") * 0.02
    b = torch.randn(O, device="cuda")
    pk = ops.conv_pack_weights(w, planes)
    for _ in range(REPS):
        ops.conv2d_planar(ops.split_planes(x), pk, (O, C, k, k), (8, H, W), b, None, padding=k // 2, relu=True, planes=planes)
torch.cuda.synchronize()
print("done")
