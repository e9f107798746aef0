#!/usr/bin/env python3
"""HBM calibration on this box: what do plain fill / copy / read-reduce kernels reach (torch built-ins, float4 paths)?"""
import torch
def t(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for mb in (142, 566, 2264):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device="cuda"); b = torch.randn(n, device="cuda")
    f = t(lambda: a.fill_(1.0)); c = t(lambda: a.copy_(b)); r = t(lambda: b.sum())
    m = t(lambda: torch.mul(b, 2.0, out=a))
    print(f"{mb:5d} MB: fill {mb / f:6.0f} GB/s (write only) | copy {2 * mb / c:6.0f} GB/s (r+w) | sum {mb / r:6.0f} GB/s (read only) | mul {2*mb/m:6.0f} GB/s", flush=True)
