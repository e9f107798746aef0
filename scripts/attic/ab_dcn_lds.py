#!/usr/bin/env python3
"""A/B of the planar deformable sampler: LDS-staged form (STM_DCN_LDS=1, csrc/deform_im2col.hip dcn_sample_planar_lds_kernel) against
the register-gather form, on the R50 DCN layer shapes at batch 32 with COLD inputs (eight input sets cycled: 60-250 MB each, more
than the Infinity Cache holds), offsets as the benchmark's synthetic weights give them (bias U(-2, 2) + small noise) and N(0, 1.5).
usage: python scripts/ab_dcn_lds.py [batch] [fmt]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fmt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
LAYERS = [("layer2.0 s2 128ch 96x160", 128, 96, 160, 2), ("layer2.2 s1 128ch 48x80", 128, 48, 80, 1), ("layer3.0 s2 256ch 48x80", 256, 48, 80, 2),
          ("layer3.2 s1 256ch 24x40", 256, 24, 40, 1), ("layer4.0 s2 512ch 24x40", 512, 24, 40, 2), ("layer4.2 s1 512ch 12x20", 512, 12, 20, 1)]
NSETS = 8


def timeit(f, n=16):
    for i in range(NSETS):
        f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        f(i % NSETS)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for kind in ("bench-like offsets (|d| <= ~2.1)", "N(0, 1.5) offsets"):
    print("#", kind)
    for name, C, H, W, s in LAYERS:
        xs = [torch.randn(B, H, W, C, device="cuda") for _ in range(NSETS)]
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        if kind.startswith("bench"):
            om = torch.cat([torch.rand(1, 18, device="cuda") * 4 - 2 + 0.05 * torch.randn(B * Ho * Wo, 18, device="cuda"), torch.randn(B * Ho * Wo, 9, device="cuda")], 1).contiguous()
        else:
            om = torch.randn(B * Ho * Wo, 27, device="cuda") * 1.5
        nbytes = 4 * B * C * H * W + 4 * 27 * B * Ho * Wo + (4 if fmt == 1 else 6) * 9 * C * B * Ho * Wo
        row, outs = [], []
        for lds in ("0", "1"):
            os.environ["STM_DCN_LDS"] = lds
            _lib.lib().stm_debug_reload_tunables()
            us = timeit(lambda i: ops.dcn_sample_planar(xs[i], om, s, 1, 1, fmt=fmt))
            outs.append(ops.dcn_sample_planar(xs[0], om, s, 1, 1, fmt=fmt))
            row.append("%s %7.1f us %5.2f TB/s (%.3f of 8)" % ("lds" if lds == "1" else "reg", us, nbytes / us / 1e6, nbytes / us / 8e6))
        print("%-26s %6.1f MB  " % (name, nbytes / 1e6) + " | ".join(row) + ("  bit-equal" if torch.equal(outs[0], outs[1]) else "  DIFFERENT"), flush=True)
        del xs
