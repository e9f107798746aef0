#!/usr/bin/env python3
"""Isolated timings of the tracker-tail kernels at the 32-clip shapes of bench.py (164 detections and 112 tracked instances per clip, 96x160 masks):
mask_iou_bits (same-clip pairs only) and temporal_pool_fc.  Inside the step these launches run beside the next frame's trunk, where a kernel's duration
says little about its cost; alone they show their own time.  usage: bench_tail_kernels.py [clips]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32     # (fewer than 25 clips: under 4096 detection rows, the small-problem kernel)
dev = "cuda"


def timeit(f, n=20):
    """GPU time per call: n calls captured into a HIP graph and replayed (a Python -> ctypes launch costs more host time than these kernels run)."""
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
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


hw = 96 * 160
words = hw // 64
nd, npv = 164, 112
g = torch.Generator(device=dev).manual_seed(0)
b1 = torch.randint(-2**62, 2**62, (B * nd, words), device=dev, dtype=torch.int64, generator=g)
b2 = torch.randint(-2**62, 2**62, (B * npv, words), device=dev, dtype=torch.int64, generator=g)
g1 = torch.arange(B, device=dev, dtype=torch.int32).repeat_interleave(nd)
g2 = torch.arange(B, device=dev, dtype=torch.int32).repeat_interleave(npv)
us = timeit(lambda: ops.mask_iou_bits(b1, b2, hw, group1=g1, group2=g2))
out = ops.mask_iou_bits(b1, b2, hw, group1=g1, group2=g2)
# reference: popcounts on the CPU for one clip
a, b = b1[:nd].cpu(), b2[:npv].cpu()
pc = lambda x: torch.tensor([[bin(int(v) & (2**64 - 1)).count("1") for v in row] for row in x.tolist()]).sum(1)
inter = torch.tensor([[sum(bin((int(x) & int(y)) & (2**64 - 1)).count("1") for x, y in zip(ra, rb)) for rb in b[:4].tolist()] for ra in a[:4].tolist()])
uni = pc(a[:4]).view(-1, 1) + pc(b[:4]).view(1, -1) - inter
assert torch.equal(out[:4, :4].cpu(), inter.float() / uni.float()), "mask_iou_bits differs from the host popcounts"
assert B == 1 or float(out[:nd, npv:].abs().max()) == 0.0
print("mask_iou_bits %d x %d masks of %d px, %d clips: %.1f us" % (B * nd, B * npv, hw, B, us))
n, C = B * npv, 1024
pool = torch.randint(0, 2**40, (n, C), device=dev, dtype=torch.int64, generator=g)
w, bias = torch.randn(36, C, device=dev), torch.randn(36, device=dev)
us = timeit(lambda: ops.temporal_pool_fc(pool, n, 49, w, bias, n_first=4, clear=False))
print("temporal_pool_fc %d RoIs x %d channels -> 4 + 32: %.1f us" % (n, C, us))

# RoI features of CandidateShift as TemporalNet's input planes (stm_roi_align_planes_nhwc_f32): the step's shapes -- P4 24 x 40, 256-channel T2S maps,
# 121 correlation channels in rows of 128, 112 tracked instances per clip, boxes of ~0.1-0.4 of the frame -- both kernel forms (STM_ROI_TILED)
from stmask_amd import _lib
H, W, C1, Cc = 24, 40, 256, 121
prev, cur = torch.randn(B, H, W, C1, device=dev, generator=g), torch.randn(B, H, W, C1, device=dev, generator=g)
corr = torch.randn(B, H, W, 128, device=dev, generator=g)
n = B * npv
cx, cy = torch.rand(n, device=dev, generator=g) * W, torch.rand(n, device=dev, generator=g) * H
bw, bh = (0.1 + 0.3 * torch.rand(n, device=dev, generator=g)) * W, (0.1 + 0.3 * torch.rand(n, device=dev, generator=g)) * H
rois = torch.stack([torch.arange(B, device=dev).repeat_interleave(npv).float(), (cx - bw / 2).clamp(0, W), (cy - bh / 2).clamp(0, H),
                    (cx + bw / 2).clamp(0, W), (cy + bh / 2).clamp(0, H)], 1).contiguous()
outs = {}
for tiled in ("0", "1", "2"):
    os.environ["STM_ROI_TILED"] = tiled
    _lib.lib().stm_debug_reload_tunables()
    us = timeit(lambda: ops.roi_align_planes(prev, cur, corr, rois, 7, fmt=1, corr_nhwc=Cc))
    outs[tiled] = ops.roi_align_planes(prev, cur, corr, rois, 7, fmt=1, corr_nhwc=Cc)
    nb = outs[tiled].numel() * 2 + (prev.numel() + cur.numel() + corr.numel()) * 4
    print("roi_align_planes %s: %d RoIs -> %d pixels x 640 channels: %.1f us = %.2f TB/s of (inputs once + planes written)" % (
        {"0": "one pixel's channel groups per wave", "1": "tiled (16 pixels x all slabs per workgroup)", "2": "the RoI's patch in LDS (one workgroup per RoI)"}[tiled],
        n, n * 49, us, nb / us / 1e6))
del os.environ["STM_ROI_TILED"]
_lib.lib().stm_debug_reload_tunables()
assert torch.equal(outs["0"], outs["1"]) and torch.equal(outs["0"], outs["2"]), "the roi_align_planes forms differ"
