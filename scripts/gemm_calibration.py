#!/usr/bin/env python3
"""What does the vendor GEMM reach on this board?  (calibration of the 2 500 TFLOP/s the roofline prices against)

hipBLASLt through torch.matmul, fp16 and bf16 inputs with fp32 accumulation, random normal data (the power a matrix pipe draws depends on its operands:
zeros are cheap), back-to-back launches for about a second per shape, HIP events.  Shapes: cubes, and the two GEMM shapes of the step's largest 3x3 layers
(M = pixels, N = output channels, K = 9 x input channels: what conv_planar_kernel computes per launch, three times over for the two-plane format).
usage: gemm_calibration.py [seconds per shape]"""
import sys
import torch

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device("cuda", 0)
SHAPES = [(8192, 8192, 8192), (16384, 16384, 8192), (491520, 256, 2304), (163840, 1024, 2304), (122880, 256, 2304)]
for dt in (torch.float16, torch.bfloat16):
    for M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev, dtype=dt)
        b = torch.randn(N, K, device=dev, dtype=dt)
        for _ in range(3):
            c = a @ b.t()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        c = a @ b.t()
        e1.record()
        torch.cuda.synchronize()
        n = max(3, int(secs * 1e3 / max(e0.elapsed_time(e1), 1e-3)))
        e0.record()
        for _ in range(n):
            c = a @ b.t()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        tf = 2.0 * M * N * K / us / 1e6
        print("%-9s M %7d N %6d K %5d  %9.1f us  %7.1f TFLOP/s = %.3f of 2500  (%d launches)" % (str(dt).split(".")[1], M, N, K, us, tf, tf / 2500, n), flush=True)
        del a, b, c
