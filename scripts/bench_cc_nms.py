#!/usr/bin/env python3
"""cc_nms through ops.detect_cc (row statistics + cross-class Fast NMS, detection_TF.py:85-134) at batch 32 by candidate count.
usage: bench_cc_nms.py [batch=32]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = 15345
g = torch.Generator(device="cuda").manual_seed(0)
priors = torch.rand(N, 4, device="cuda", generator=g) * 0.5 + 0.1
for n_hot in (200, 600, 1000, 1500, 3000, 6000, 12000):
    logits = torch.randn(B, N, 41, device="cuda", generator=g)
    logits[..., 0] += 8.0
    for b in range(B):
        hot = torch.randperm(N, device="cuda", generator=g)[:n_hot]
        logits[b, hot, 1 + (hot % 40)] += 10.0 + torch.rand(n_hot, device="cuda", generator=g) * 4
    loc = torch.randn(B, N, 4, device="cuda", generator=g)
    cen = torch.tanh(torch.randn(B, N, 1, device="cuda", generator=g) + 1.0)
    f = lambda: ops.detect_cc(loc, priors, logits, cen, 0.05, 0.5, 200, logits=True)
    for _ in range(3):
        out = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    idx, cls, sc, bx, cnt = out
    chk = int(idx.sum().item()) ^ int(cls.sum().item())
    print("candidates/frame %6d: %7.1f us per launch pair (row stats + NMS), kept %s, checksum %d" % (n_hot, e0.elapsed_time(e1) * 50, cnt[:3].tolist(), chk), flush=True)
