"""Planar deformable sampler with the input resident in the Infinity Cache (same tensors every launch) and cold (eight input
sets cycled, 8 x 60-250 MB): the isolated figures of profiles/r01_dcn_sample_isolated.txt are the resident case."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from stmask_amd import ops
B, fmt = 32, 1
for name, C, H, W, s in [("layer2 s1 128ch 48x80", 128, 48, 80, 1), ("layer3 256ch 24x40", 256, 24, 40, 1)]:
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    sets = [(torch.randn(B, H, W, C, device="cuda"), torch.randn(B * Ho * Wo, 27, device="cuda") * 1.5) for _ in range(8)]
    nbytes = 4 * B * C * H * W + 4 * 27 * B * Ho * Wo + 4 * 9 * C * B * Ho * Wo
    for mode in ("same", "cycle", "cycle-noprefetch", "cycle"):
        os.environ["STM_DCN_PREFETCH"] = "0" if mode.endswith("noprefetch") else "1"
        for i in range(3):
            ops.dcn_sample_planar(*sets[0], s, 1, 1, fmt=fmt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(16):
            x, om = sets[i % 8 if mode.startswith("cycle") else 0]
            ops.dcn_sample_planar(x, om, s, 1, 1, fmt=fmt)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 16 * 1e3
        print(name, mode, "%.1f us %.2f TB/s" % (us, nbytes / us / 1e6))
