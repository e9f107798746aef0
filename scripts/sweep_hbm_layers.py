#!/usr/bin/env python3
"""Re-sweep of the launch rules of the HBM-bound 1x1 layers (scripts/bench_layers.py shapes, batch 32, cold buffers): tile width x pixel-tile group x K parts x loop
variant, every candidate in one process (the library re-reads its switches between candidates).  The rules date from round 2 (sweep_small_m / sweep_tile_small in
scripts/attic); the kernels under them changed in rounds 3-5.  usage: sweep_hbm_layers.py [batch=32]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench_layers as bl  # noqa: E402
from stmask_amd import _lib, planar  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
planar.set_format(1)
LAYERS = [l for l in bl.HBM if l[0].startswith(("l2", "l3", "l4"))]
CANDS = [("rule", None, {})]
for tile in (64, 128):
    for mg in ((1, 2) if tile == 128 else (0,)):
        for sk in (1, 2, 4):
            for ring in ((2, 4) if tile == 64 else (2, 3)):
                env = {"STM_CONV_SPLITK": sk, ("STM_CONV_RING64" if tile == 64 else "STM_CONV_RING"): ring}
                if mg:
                    env["STM_CONV_MG"] = mg
                CANDS.append((f"tile{tile} mg{mg} sk{sk} ring{ring}", tile, env))
print(torch.cuda.get_device_name(0), "batch", B, flush=True)


def run_dual(name, Ho, Wo, C1, C2, O, tile_n, reps=12, nbuf=4):
    """conv3 + projection shortcut of a stage's first block as one two-source product (stm_conv2d_planar_dual_f32): [mid (C1 ch, Ho x Wo) ; x (C2 ch, 2Ho x 2Wo, stride 2)]."""
    from stmask_amd import ops
    from stmask_amd.planar import PlanarConv
    g = torch.Generator(device="cuda").manual_seed(0)
    w = torch.randn(O, C1 + C2, 1, 1, device="cuda", generator=g) * (C1 + C2) ** -0.5
    conv = PlanarConv(w, torch.randn(O, device="cuda", generator=g), 1, 0, relu=True, fmt=1, tile_n=tile_n)
    x1 = [ops.split_planes(torch.randn(B, Ho, Wo, C1, device="cuda", generator=g), 1) for _ in range(nbuf)]
    x2 = [ops.split_planes(torch.randn(B, 2 * Ho, 2 * Wo, C2, device="cuda", generator=g), 1) for _ in range(nbuf)]
    call = lambda i: conv(x1[i % nbuf], ("img", B, Ho, Wo), x2=(x2[i % nbuf], 2 * Ho, 2 * Wo, 2))
    for i in range(3):
        call(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        call(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


DUAL = [("l2.0 conv3+ds 128+256->512", 48, 80, 128, 256, 512), ("l3.0 conv3+ds 256+512->1024", 24, 40, 256, 512, 1024), ("l4.0 conv3+ds 512+1024->2048", 12, 20, 512, 1024, 2048)]
for name, *shape in [(d[0], "dual", *d[1:]) for d in DUAL] + LAYERS:
    best = None
    rows = []
    for label, tile, env in CANDS:
        for k in ("STM_CONV_SPLITK", "STM_CONV_RING64", "STM_CONV_RING", "STM_CONV_MG"):
            os.environ.pop(k, None)
        for k, v in env.items():
            os.environ[k] = str(v)
        _lib.lib().stm_debug_reload_tunables()
        try:
            sys.stdout = open(os.devnull, "w")
            us = run_dual(name, *shape[1:], tile_n=tile) if shape[0] == "dual" else bl.run(name, B, *shape, tile_n=tile, fmt=1, reps=12)
        except Exception as e:  # a combination the kernel refuses
            us = float("inf")
        finally:
            sys.stdout = sys.__stdout__
        rows.append((us, label))
    rule = rows[0][0]
    rows.sort()
    print("%-26s rule %7.1f us | best: %s" % (name, rule, "  ".join("%s %.1f" % (l, u) for u, l in rows[:4])), flush=True)
