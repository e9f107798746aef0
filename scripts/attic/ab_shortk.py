#!/usr/bin/env python3
"""Memory-bound 1x1 layers of the bottlenecks (short K, wide output, planar residual): tile width / loop variant A/B.
usage: python scripts/ab_shortk.py [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops, planar
from stmask_amd.planar import PlanarConv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
planar.set_format(1)
LAYERS = [("layer1 c3 64->256 96x160", 96, 160, 64, 256, 1, True), ("layer2 c3 128->512 48x80", 48, 80, 128, 512, 1, True),
          ("layer3 c3 256->1024 24x40", 24, 40, 256, 1024, 1, True), ("layer1 c1 256->64", 96, 160, 256, 64, 1, False),
          ("layer1 c2 3x3 64->64", 96, 160, 64, 64, 3, False), ("layer2 c1 512->128", 48, 80, 512, 128, 1, False)]


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, H, W, C, O, k, res in LAYERS:
    x = torch.randn(B, H, W, C, device="cuda")
    w = torch.randn(O, C, k, k, device="cuda") * (C * k * k) ** -0.5
    b = torch.randn(O, device="cuda")
    xp = ops.split_planes(x, 1)
    rp = ops.split_planes(torch.randn(B, H, W, O, device="cuda"), 1) if res else None
    mb = (B * H * W * (C + O * (2 if res else 1)) * 4) / 1e6
    row = []
    for tile, env in [(128, {}), (128, {"STM_CONV_RING": "2"}), (64, {"STM_CONV_RING64": "2"}), (64, {"STM_CONV_RING64": "4"})]:
        conv = PlanarConv(w, b, 1, k // 2, relu=True, tile_n=tile)
        for kk, vv in env.items():
            os.environ[kk] = vv
        us = timeit(lambda: conv(xp, ("img", B, H, W), out="planes", residual=rp))
        for kk in env:
            del os.environ[kk]
        row.append("t%d%s %7.1f us %4.2f TB/s" % (tile, "".join("/" + v for v in env.values()), us, mb / us))
    print("%-28s %7.1f MB  " % (name, mb) + " | ".join(row))
