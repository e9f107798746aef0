#!/usr/bin/env python3
"""The fused deformable convolution (csrc/dcn_fused.hip: sampler -> plane split -> MFMA product, no column buffer) against the kernel pair it
replaces (planar sampler + planar 1x1 product over 9C channels) on the 7 DCN layer shapes of R50 @384x640, cold inputs (eight input sets cycled),
bench-like offsets (bias ~ U(-2, 2) + N(0, 0.05)); every launch replayed from HIP events on the launch stream.  Prints per layer: microseconds of
the sampler, of the product, of the pair, of the fused kernel; the fused kernel's rate against SURVEY 8(d)'s fused byte formula and against the
2.5 PFLOP/s fp16 peak (three plane products per reference product); the largest difference between the two results.
usage: bench_dcn_fused.py [batch=32] [fmt=1]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
FMT = int(sys.argv[2]) if len(sys.argv) > 2 else 1
LAYERS = [("layer2.0 s2 128ch 96x160", 128, 96, 160, 2), ("layer2.2 s1 128ch 48x80", 128, 48, 80, 1), ("layer3.0 s2 256ch 48x80", 256, 48, 80, 2),
          ("layer3.2 s1 256ch 24x40", 256, 24, 40, 1), ("layer3.4 s1 256ch 24x40", 256, 24, 40, 1), ("layer4.0 s2 512ch 24x40", 512, 24, 40, 2),
          ("layer4.2 s1 512ch 12x20", 512, 12, 20, 1)]
NSETS = int(os.environ.get("NSETS", "8"))     # input sets cycled (8 x 60-250 MB: cold; 1: the input stays in the Infinity Cache)
if os.environ.get("LAYER"):
    LAYERS = [LAYERS[int(os.environ["LAYER"])]]
FUSED_ONLY = os.environ.get("FUSED_ONLY", "0") != "0"     # ablation builds (STM_LIBRARY=...): only the fused kernel's time, no comparison


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


NP = ops.plane_layout(FMT)[0]
prod = {1: 3, 2: 1}[FMT]
tot = dict(s=0.0, g=0.0, f=0.0)
g = torch.Generator(device="cuda").manual_seed(0)
ops.planar_range_flag()
for name, C, H, W, s in LAYERS:
    O = C
    xs = [torch.randn(B, H, W, C, device="cuda", generator=g).clamp_min(0) for _ in range(NSETS)]
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    M = B * Ho * Wo
    om = torch.cat([torch.rand(1, 18, device="cuda", generator=g) * 4 - 2 + 0.05 * torch.randn(M, 18, device="cuda", generator=g),
                    torch.randn(M, 9, device="cuda", generator=g), torch.zeros(M, 5, device="cuda")], 1).contiguous()
    w = torch.randn(O, C, 3, 3, device="cuda", generator=g) * (9 * C) ** -0.5
    bias = torch.randn(O, device="cuda", generator=g)
    pk3, sc3 = ops.conv_pack_weights(w, tile_n=128, fmt=FMT)
    wk = w.permute(0, 2, 3, 1).reshape(O, 9 * C, 1, 1).contiguous()
    pk1, sc1 = ops.conv_pack_weights(wk, tile_n=128, fmt=FMT)
    pk1_64, _ = ops.conv_pack_weights(wk, tile_n=64, fmt=FMT)
    if FUSED_ONLY:
        from stmask_amd import planar
        fz = planar.PlanarConv(w, bias, s, 1, relu=True, fmt=FMT)
        x2 = [x.view(B * H * W, C) for x in xs]
        XPAD = int(os.environ.get("XPAD", "0"))            # pixel stride of x = C + XPAD floats (a channel slice of a wider tensor)
        if XPAD:
            x2 = [torch.cat([x, torch.zeros(B * H * W, XPAD, device="cuda")], 1)[:, :C] for x in x2]
        t_f = timeit(lambda i: fz.deform(x2[i], B, H, W, om, s, 1, 1, has_mask=True))
        tot["f"] += t_f
        print("%-26s fused %6.1f us" % (name, t_f), flush=True)
        del xs, x2
        continue
    cols = [ops.dcn_sample_planar(xs[i], om, s, 1, 1, fmt=FMT) for i in range(2)]
    t_s = timeit(lambda i: ops.dcn_sample_planar(xs[i], om, s, 1, 1, fmt=FMT))
    # the product as the inference graph runs it (PlanarConv picks the tile and split-K rule)
    from stmask_amd import planar
    pc = planar.PlanarConv(wk, bias, 1, 0, relu=True, fmt=FMT)
    t_g = timeit(lambda i: pc(cols[i & 1], ("img", B, Ho, Wo)))
    fz = planar.PlanarConv(w, bias, s, 1, relu=True, fmt=FMT)
    x2 = [x.view(B * H * W, C) for x in xs]
    t_f = timeit(lambda i: fz.deform(x2[i], B, H, W, om, s, 1, 1, has_mask=True))
    a = ops.planes_to_f32(fz.deform(x2[0], B, H, W, om, s, 1, 1, has_mask=True))
    b = ops.planes_to_f32(pc(cols[0], ("img", B, Ho, Wo)))
    diff = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-9)
    fbytes = 4 * B * C * H * W + 4 * 27 * M + 2 * NP * O * M + 2 * NP * O * C * 9
    flops = 2.0 * M * O * C * 9
    tot["s"] += t_s; tot["g"] += t_g; tot["f"] += t_f
    print("%-26s sampler %6.1f + product %6.1f = %6.1f us | fused %6.1f us (x%.2f)  %6.1f MB fused-algorithmic = %.2f TB/s, %5.0f TF-eq = %.3f of peak  tiles %d  rel diff %.1e"
          % (name, t_s, t_g, t_s + t_g, t_f, (t_s + t_g) / t_f, fbytes / 1e6, fbytes / t_f / 1e6, flops / t_f / 1e6, flops * prod / t_f / 1e6 / 2500e0,
             ops.deform_conv_fused_tiles(B, Ho, Wo, O), diff), flush=True)
    del xs, x2, cols
print("sum over the seven layers of a step: sampler %.1f + product %.1f = %.1f us | fused %.1f us" % (tot["s"], tot["g"], tot["s"] + tot["g"], tot["f"]))
