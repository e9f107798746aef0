#!/usr/bin/env python3
"""A/B of the planar deformable sampler's forms on the R50 DCN layer shapes at batch 32, cold inputs (eight input sets cycled), bench-like offsets:
STM_DCN_VARIANT 0 = run-time-format kernel (rounds 1-3), 1 = straight-line fp16x2 kernel, 2 = + one tap of look-ahead, -1 = the library's default
rule.  Every form must be bit-equal to form 0.  usage: ab_dcn_v2.py [batch] [variants, e.g. 0,1,2]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
VARS = (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,-1").split(",")
LAYERS = [("layer2.0 s2 128ch 96x160", 128, 96, 160, 2), ("layer2.2 s1 128ch 48x80", 128, 48, 80, 1), ("layer3.0 s2 256ch 48x80", 256, 48, 80, 2),
          ("layer3.2 s1 256ch 24x40", 256, 24, 40, 1), ("layer3.4 s1 256ch 24x40", 256, 24, 40, 1), ("layer4.0 s2 512ch 24x40", 512, 24, 40, 2),
          ("layer4.2 s1 512ch 12x20", 512, 12, 20, 1)]
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


tot = {v: 0.0 for v in VARS}
totb = 0.0
g = torch.Generator(device="cuda").manual_seed(0)
for name, C, H, W, s in LAYERS:
    xs = [torch.randn(B, H, W, C, device="cuda", generator=g) for _ in range(NSETS)]
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    om = torch.cat([torch.rand(1, 18, device="cuda", generator=g) * 4 - 2 + 0.05 * torch.randn(B * Ho * Wo, 18, device="cuda", generator=g),
                    torch.randn(B * Ho * Wo, 9, device="cuda", generator=g)], 1).contiguous()
    nbytes = 4 * B * C * H * W + 4 * 27 * B * Ho * Wo + 4 * 9 * C * B * Ho * Wo
    totb += nbytes
    row, ref = [], None
    for v in VARS:
        os.environ["STM_DCN_VARIANT"] = v
        _lib.lib().stm_debug_reload_tunables()
        us = timeit(lambda i: ops.dcn_sample_planar(xs[i], om, s, 1, 1, fmt=1))
        out = ops.dcn_sample_planar(xs[0], om, s, 1, 1, fmt=1)
        if ref is None:
            ref = out
        eq = torch.equal(out, ref)
        tot[v] += us
        row.append("v%s %6.1f us %.3f%s" % (v, us, nbytes / us / 8e6, "" if eq else " DIFFERENT"))
    print("%-26s %6.1f MB  " % (name, nbytes / 1e6) + " | ".join(row), flush=True)
    del xs
print("sum of the seven launches of a step: " + " | ".join("v%s %6.1f us = %.3f of 8 TB/s" % (v, tot[v], totb / tot[v] / 8e6) for v in VARS))
