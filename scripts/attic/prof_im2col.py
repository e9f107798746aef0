#!/usr/bin/env python3
"""Tiny driver for rocprofv3 --pmc passes on the deformable-im2col kernel: the 7 R50 layer shapes at batch B, REPS each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops  # noqa: E402
from scripts.bench_kernels import R50_DCN  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
VARIANT = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for name, C, H, W, s in R50_DCN:
    Ho, Wo = ops.conv_out_hw(H, W, 3, 3, s, s, 1, 1, 1, 1)
    x = torch.randn(B, C, H, W, device="cuda")
    om = torch.randn(B, 27, Ho, Wo, device="cuda")
    om[:, :18] = (torch.rand(B, 18, Ho, Wo, device="cuda") * 4 - 2)   # offsets U(-2,2) like the synthetic DCN biases
    cols = torch.empty(B, C * 9, Ho * Wo, device="cuda")
    for _ in range(REPS):
        ops.deform_im2col(x, None, None, 3, s, 1, 1, 1, variant=VARIANT, fused_om=om, out=cols)
torch.cuda.synchronize()
print("done")
