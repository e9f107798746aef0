#!/usr/bin/env python3
"""Driver for rocprofv3 passes on the planar split conv: tower 3x3 at P3 and the proto-net 3x3 at batch 8, REPS launches.
usage: prof_conv.py [reps] [planes] [fmt] [big]   (fmt 1: fp16 two-plane format; big: only the 145-GF proto layer)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
planes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
fmt = int(sys.argv[3]) if len(sys.argv) > 3 else 0
big = len(sys.argv) > 4
LAYERS = [(96, 160, 256, 256, 3)] if big else [(48, 80, 256, 256, 3), (96, 160, 256, 256, 3), (48, 80, 128, 512, 1), (24, 40, 256, 256, 3)]
for H, W, C, O, k in LAYERS:
    x = torch.randn(8, H, W, C, device="cuda")
    w = torch.randn(O, C, k, k, device="cuda") * 0.02
    b = torch.randn(O, device="cuda")
    if fmt == 1:
        pk, osc = ops.conv_pack_weights(w, fmt=1)
    else:
        pk, osc = ops.conv_pack_weights(w, planes), 1.0
    for _ in range(REPS):
        ops.conv2d_planar(ops.split_planes(x, fmt), pk, (O, C, k, k), (8, H, W), b, None, padding=k // 2, relu=True, planes=planes, fmt=fmt,
                          out_scale=osc)
torch.cuda.synchronize()
print("done")
