"""A/B of the tiled correlation kernel (csrc/temporal.hip corr_patch_tiled) on the pipeline's own entry: channels-last P4 maps in, channels-last
correlation volume out.  STM_CORR_VARIANT=2 = round 3's form (188 VGPRs: 2 waves per SIMD, the 768 workgroups of a 32-clip step in 1.5 rounds),
default = packed staging plan at 158 VGPRs (3 waves per SIMD, one round).  Results must be bit-equal.  usage: bench_corr.py [batches ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import _lib, ops

DEV = "cuda"


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in [int(b) for b in (sys.argv[1:] or ["32", "8", "1"])]:
    g = torch.Generator(device=DEV).manual_seed(B)
    # eight input sets cycled: 63 MB per set at batch 32, so nothing is served from a cache a real step would not have
    sets = [(torch.randn(B, 24, 40, 256, device=DEV, generator=g).permute(0, 3, 1, 2), torch.randn(B, 24, 40, 256, device=DEV, generator=g).permute(0, 3, 1, 2))
            for _ in range(8)]
    nbytes = 4 * B * (2 * 256 * 960 + 121 * 960)
    outs = {}
    for rep in range(2):
        for var in ("2", "0"):
            os.environ["STM_CORR_VARIANT"] = var
            _lib.lib().stm_debug_reload_tunables()
            k = [0]

            def run():
                f1, f2 = sets[k[0] % 8]
                k[0] += 1
                return ops.corr_patch_nhwc(f1, f2, 11, scale=1.0 / 256, leaky_slope=0.1)

            us = timeit(run)
            outs[var] = ops.corr_patch_nhwc(sets[0][0], sets[0][1], 11, scale=1.0 / 256, leaky_slope=0.1).clone()
            print(f"corr B={B} {'round-3 form (2 waves/SIMD)' if var == '2' else 'packed plan (3 waves/SIMD)  '}: {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s algorithmic = "
                  f"{nbytes / us / 1e6 / 8:.3f} of 8 TB/s", flush=True)
    print(f"corr B={B}: the two forms are bit-equal: {torch.equal(outs['0'][..., :121], outs['2'][..., :121])}", flush=True)
os.environ.pop("STM_CORR_VARIANT", None)
