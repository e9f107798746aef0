#!/usr/bin/env python3
"""Fixed cost vs per-stage cost of the kx-reuse kernel: one job (41 of 64 channels, 3x3) over the head's pixel axis with C = 64 .. 512."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd.planar import PlanarConv
from bench_kxr import timeit, LEVELS, B, DEV
for real in ((41,), (32,), (5,)):
    for C in (64, 128, 256, 512):
        w = torch.randn(64, C, 3, 3, device=DEV) * (C * 9) ** -0.5
        M = sum(B * h * w_ for h, w_ in LEVELS)
        xp = (torch.randn(2, C // 32, M, 32, device=DEV) * 0.5).half()
        conv = PlanarConv(w, None, 1, 1, groups=1, group_cout=list(real), tile_n=64, fmt=1)
        conv.kxr_min_pixels = 0
        out = torch.empty(M, 64, device=DEV)
        t = timeit(lambda: conv(xp, ("levels", B, LEVELS), out="f32", out_f32=out))
        print(f"real={real[0]:3d} C={C:4d} stages={C // 32 * 3:3d}  {t:8.1f} us", flush=True)
