#!/usr/bin/env python3
"""What plain streaming kernels reach on this box for the byte mix of the bottleneck's expanding 1x1 convolution (read residual
planes + write output planes, 4 B per element each): torch elementwise ops over fp16 tensors of the same sizes, rotating buffers."""
import torch
dev = "cuda"
M, O = 491520, 256
n = M * O * 2      # fp16 elements of two planes
bufs = [(torch.randn(n, device=dev, dtype=torch.float16), torch.empty(n, device=dev, dtype=torch.float16)) for _ in range(3)]


def t(fn, reps=20):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


us = t(lambda i: torch.clamp_min(bufs[i % 3][0], 0.0, out=bufs[i % 3][1]))
print(f"relu copy  {2 * n * 2 / 1e6:.0f} MB in {us:.1f} us = {2 * n * 2 / us / 1e6:.2f} TB/s")
us = t(lambda i: bufs[i % 3][1].copy_(bufs[i % 3][0]))
print(f"copy_      {2 * n * 2 / 1e6:.0f} MB in {us:.1f} us = {2 * n * 2 / us / 1e6:.2f} TB/s")
us = t(lambda i: bufs[i % 3][1].fill_(1.0))
print(f"fill_      {n * 2 / 1e6:.0f} MB in {us:.1f} us = {n * 2 / us / 1e6:.2f} TB/s")
us = t(lambda i: bufs[i % 3][0].sum())
print(f"sum (read) {n * 2 / 1e6:.0f} MB in {us:.1f} us = {n * 2 / us / 1e6:.2f} TB/s")
