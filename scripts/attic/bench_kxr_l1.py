#!/usr/bin/env python3
"""layer1's 64 -> 64 3x3 at 96x160 on the kx-reuse kernel (four channel tiles), batch from argv."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops
from stmask_amd.planar import PlanarConv
from bench_kxr import timeit, DEV
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for C in (64, 128, 256):
    w = torch.randn(64, C, 3, 3, device=DEV) * (C * 9) ** -0.5
    M = B * 96 * 160
    xp = (torch.randn(2, C // 32, M, 32, device=DEV) * 0.5).half()
    for kxr in (False, True):
        conv = PlanarConv(w, None, 1, 1, relu=True, tile_n=64, fmt=1)
        conv.kxr = kxr and ops.conv_kxr_supported(64, C, 3, 3, 1, 1, 1, None, 1)
        conv.kxr_min_pixels = 0
        out = torch.empty(2, 2, M, 32, device=DEV, dtype=torch.float16)
        t = timeit(lambda: conv(xp, ("img", B, 96, 160), out="planes", out_planes=out))
        print(f"C={C:4d} stages={C // 32 * 3:3d} kxr={kxr}  {t:8.1f} us", flush=True)
