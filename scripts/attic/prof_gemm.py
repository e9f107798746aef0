#!/usr/bin/env python3
"""Driver for rocprofv3 PMC passes on the fp32 MFMA GEMM: DCN layer shapes at batch 8, REPS launches each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for M, N, K in [(128, 3840, 1152), (256, 960, 2304), (512, 240, 4608), (512, 6272, 5697 // 4 * 4)]:
    A = torch.randn(M, K, device="cuda") * K ** -0.5
    B = torch.randn(8 if N < 5000 else 1, K, N, device="cuda")
    bias = torch.randn(M, device="cuda")
    for _ in range(REPS):
        ops.gemm_bias(A, B, bias)
torch.cuda.synchronize()
print("done")
