import os, sys, torch
sys.path.insert(0, os.getcwd())
from stmask_amd import ops, _lib
exec(open("scripts/bench_tail_kernels.py").read().split("hw = 96 * 160")[0].split("B = int")[0])
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        f(); g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): f()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
dev="cuda"; hw=96*160; words=hw//64
gen = torch.Generator(device=dev).manual_seed(0)
for B in (4, 8, 12, 16, 24, 32):
    for nd, npv in ((110, 112),):
        b1 = torch.randint(-2**62, 2**62, (B * nd, words), device=dev, dtype=torch.int64, generator=gen)
        b2 = torch.randint(-2**62, 2**62, (B * npv, words), device=dev, dtype=torch.int64, generator=gen)
        g1 = torch.arange(B, device=dev, dtype=torch.int32).repeat_interleave(nd)
        g2 = torch.arange(B, device=dev, dtype=torch.int32).repeat_interleave(npv)
        r = []
        for big in ("100000", "1"):
            os.environ["STM_MIOU_BIG_ROWS"] = big; _lib.lib().stm_debug_reload_tunables()
            r.append(timeit(lambda: ops.mask_iou_bits(b1, b2, hw, group1=g1, group2=g2)))
        print("clips %2d: %5d x %5d rows: small kernel %6.1f us, row-block kernel %6.1f us" % (B, B*nd, B*npv, r[0], r[1]))
