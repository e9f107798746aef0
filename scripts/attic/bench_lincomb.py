#!/usr/bin/env python3
"""lincomb_kernel (prototype linear combination + sigmoid + crop + bit words, mask_ops.hip) alone at the step's shapes: 32 clips x 96x160 prototypes, n rows
sorted by clip, boxes ~ a quarter of the frame.  Prints microseconds and the rate of (masks written + bit words + prototypes once).
usage: bench_lincomb.py [rows=3600] [clips=32]     (STM_LIBRARY=<variant .so> for the LC_ABL builds)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3600
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
H, W, M = 96, 160, 32
g = torch.Generator(device="cuda").manual_seed(0)
protos = [torch.randn(B, H, W, M, device="cuda", generator=g) for _ in range(4)]
coeff = torch.randn(n, M, device="cuda", generator=g)
c = torch.rand(n, 2, device="cuda", generator=g)
wh = torch.rand(n, 2, device="cuda", generator=g) * 0.5 + 0.1
boxes = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 1).contiguous()
clip = (torch.arange(n, device="cuda") * B // n).to(torch.int32)
for k in range(4):
    ops.lincomb_sigmoid_crop_bits(protos[k], coeff, boxes, clip)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 20
e0.record()
for k in range(N):
    ops.lincomb_sigmoid_crop_bits(protos[k & 3], coeff, boxes, clip)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / N * 1e3
by = n * H * W * 4 + n * H * W / 8 + B * H * W * M * 4
print("lincomb %d rows x %dx%d, %d clips: %.1f us = %.2f TB/s of (masks + bit words + prototypes once = %.0f MB)" % (n, H, W, B, us, by / us / 1e6, by / 1e6))
