#!/usr/bin/env python3
"""TemporalNet (3x 3x3 conv on 7x7 RoI tiles) -- MIOpen vs unfold + GEMM formulations, per 128-RoI block."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops
from stmask_amd.layers.modules import TemporalNet
from scripts.bench_kernels import timeit
torch.backends.cudnn.benchmark = False
net = TemporalNet(633, 32).cuda().eval()
for n in (64, 128, 256):
    x = torch.randn(n, 633, 7, 7, device="cuda")
    xcl = x.contiguous(memory_format=torch.channels_last)
    netcl = TemporalNet(633, 32).cuda().eval().to(memory_format=torch.channels_last)
    gf = n * 0.98
    with torch.no_grad():
        ms = timeit(lambda: net(x)); mscl = timeit(lambda: netcl(xcl))
        print(f"n={n}: MIOpen NCHW {ms*1e3:8.1f} us ({gf/ms:6.1f} TF)  NHWC {mscl*1e3:8.1f} us ({gf/mscl:6.1f} TF)", flush=True)
        # unfold + GEMM formulation of the three convs
        w1 = net.conv1.weight.reshape(512, -1); w2 = net.conv2.weight.reshape(512, -1); w3 = net.conv3.weight.reshape(1024, -1)
        def unf(t):  # [n,C,7,7] -> [C*9, n*49]
            return F.unfold(t, 3, padding=1).permute(1, 0, 2).reshape(t.shape[1] * 9, -1)
        def via_matmul():
            a = torch.relu(w1 @ unf(x) + net.conv1.bias[:, None]).view(512, n, 7, 7).permute(1, 0, 2, 3)
            a = torch.relu(w2 @ unf(a) + net.conv2.bias[:, None]).view(512, n, 7, 7).permute(1, 0, 2, 3)
            a = torch.relu(w3 @ unf(a) + net.conv3.bias[:, None])
            return a
        def via_own_gemm():
            a = ops.gemm_bias(w1, unf(x), net.conv1.bias, relu=True).view(512, n, 7, 7).permute(1, 0, 2, 3)
            a = ops.gemm_bias(w2, unf(a), net.conv2.bias, relu=True).view(512, n, 7, 7).permute(1, 0, 2, 3)
            return ops.gemm_bias(w3, unf(a), net.conv3.bias, relu=True)
        ref = torch.relu(net.conv3(torch.relu(net.conv2(torch.relu(net.conv1(x))))))
        got = via_own_gemm().view(1024, n, 7, 7).permute(1, 0, 2, 3)
        print("   max diff own-gemm formulation vs MIOpen:", (got - ref).abs().max().item())
        m1 = timeit(via_matmul); m2 = timeit(via_own_gemm)
        u = timeit(lambda: unf(x))
        g1 = timeit(lambda: w1 @ unf.__call__(x)) if False else 0
        B1 = unf(x)
        tg = timeit(lambda: w1 @ B1); to = timeit(lambda: ops.gemm_bias(w1, B1, net.conv1.bias, relu=True))
        print(f"   unfold+rocBLAS {m1*1e3:8.1f} us ({gf/m1:6.1f} TF) | unfold+own GEMM {m2*1e3:8.1f} us ({gf/m2:6.1f} TF) | unfold alone {u*1e3:.1f} us | conv1 GEMM rocBLAS {tg*1e3:.1f} us own {to*1e3:.1f} us", flush=True)
