#!/usr/bin/env python3
"""Tile width (64 / 128) and split-K of the 3x3 layers whose grids are about one 128 x 128 workgroup per CU at small batch (proto-net at 96x160,
head towers over the levels, FPN 3x3), replayed from a HIP graph (launch gaps as in the step).  usage: sweep_tile_small.py [clips]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops, _lib, planar
from stmask_amd.planar import PlanarConv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = "cuda"
planar.set_format(1)
LEVELS = [(48, 80), (24, 40), (12, 20), (6, 10), (3, 5)]
CASES = [("proto 256->256 3x3 @96x160", ("img", B, 96, 160), 256, 256, 1),
         ("proto/fpn 256->256 3x3 @48x80", ("img", B, 48, 80), 256, 256, 1),
         ("tower1 256->1024 3x3 @levels", ("levels", B, LEVELS), 256, 1024, 1),
         ("tower2 4x(256->256) 3x3 @levels", ("levels", B, LEVELS), 256, 1024, 4),
         ("up 256->256 3x3 @levels", ("levels", B, LEVELS), 256, 256, 1)]


def gpu_us(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        f()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                f()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


for name, shape, C, O, groups in CASES:
    M = B * sum(h * w for h, w in shape[2]) if shape[0] == "levels" else B * shape[2] * shape[3]
    w = torch.randn(O, C, 3, 3, device=dev) * (C * 9) ** -0.5
    b = torch.randn(O, device=dev)
    x = ops.split_planes(torch.randn(M, C * groups, device=dev), 1)
    row = []
    for tile in (64, 128):
        for sk in (0, 2, 3):
            os.environ["STM_CONV_SPLITK"] = str(sk)
            _lib.lib().stm_debug_reload_tunables()
            conv = PlanarConv(w, b, 1, 1, relu=True, groups=groups, fmt=1, tile_n=tile)
            try:
                us = gpu_us(lambda: conv(x, shape, out="planes"))
                row.append("t%d/sk%d %6.1f" % (tile, sk, us))
            except Exception as e:
                row.append("t%d/sk%d  err" % (tile, sk))
    os.environ.pop("STM_CONV_SPLITK", None)
    _lib.lib().stm_debug_reload_tunables()
    conv = PlanarConv(w, b, 1, 1, relu=True, groups=groups, fmt=1)
    us = gpu_us(lambda: conv(x, shape, out="planes"))
    print("%-34s M=%6d  rule: tile %3d %6.1f us | %s" % (name, M, conv.pick_tile(M), us, " | ".join(row)), flush=True)
