"""Microbench of stm_conv2d_nhwc_f32 (bf16-split implicit GEMM) against torch/MIOpen fp32 conv on the layer shapes of
the R50 trunk at 384x640, batch 8.  usage: python scripts/bench_conv.py [batch] [planes]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from stmask_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
planes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.backends.cudnn.benchmark = False
dev = "cuda"
SHAPES = [
    # name, H, W, C, Cout, k, stride
    ("head/proto 3x3 P3", 48, 80, 256, 256, 3, 1),
    ("head 3x3 P4", 24, 40, 256, 256, 3, 1),
    ("head 3x3 P5", 12, 20, 256, 256, 3, 1),
    ("proto 3x3 up", 96, 160, 256, 256, 3, 1),
    ("layer1 1x1 64->256", 96, 160, 64, 256, 1, 1),
    ("layer1 1x1 256->64", 96, 160, 256, 64, 1, 1),
    ("layer1 3x3 64", 96, 160, 64, 64, 3, 1),
    ("layer2 1x1 512->128", 48, 80, 512, 128, 1, 1),
    ("layer2 1x1 128->512", 48, 80, 128, 512, 1, 1),
    ("layer3 1x1 1024->256", 24, 40, 1024, 256, 1, 1),
    ("layer3 1x1 256->1024", 24, 40, 256, 1024, 1, 1),
    ("layer4 1x1 2048->512", 12, 20, 2048, 512, 1, 1),
    ("layer4 1x1 512->2048", 12, 20, 512, 2048, 1, 1),
    ("fpn lat 2048->256", 12, 20, 2048, 256, 1, 1),
    ("offset_mask 3x3 128->27", 48, 80, 128, 27, 3, 1),
]


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot_o = tot_t = 0.0
for name, H, W, C, O, k, s in SHAPES:
    x = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(O, C, k, k, device=dev) * (C * k * k) ** -0.5
    b = torch.randn(O, device=dev)
    pk = ops.conv_pack_weights(w, planes)
    out = torch.empty(B, (H - 1) // s + 1, (W - 1) // s + 1, O, device=dev)
    f_ours = lambda: ops.conv2d_nhwc(x, pk, (O, C, k, k), b, None, stride=s, padding=k // 2, relu=True, planes=planes, out=out)
    xc = x.permute(0, 3, 1, 2)   # channels_last view
    wc = w.contiguous(memory_format=torch.channels_last)
    f_torch = lambda: F.conv2d(xc, wc, b, stride=s, padding=k // 2)
    xn = xc.contiguous()
    f_nchw = lambda: F.conv2d(xn, w, b, stride=s, padding=k // 2)
    xp = ops.split_planes(x)
    f_pl = lambda: ops.conv2d_planar(xp, pk, (O, C, k, k), (B, H, W), b, None, stride=s, padding=k // 2, relu=True, planes=planes, out="planes")
    pk64 = ops.conv_pack_weights(w, planes, 64)
    f_p64 = lambda: ops.conv2d_planar(xp, pk64, (O, C, k, k), (B, H, W), b, None, stride=s, padding=k // 2, relu=True, planes=planes, out="planes", tile_n=64)
    pk16, osc = ops.conv_pack_weights(w, tile_n=128, fmt=1)
    xp16 = ops.split_planes(x, fmt=1)
    f_h16 = lambda: ops.conv2d_planar(xp16, pk16, (O, C, k, k), (B, H, W), b, None, stride=s, padding=k // 2, relu=True, out="planes", fmt=1, out_scale=osc)
    pk16n, _ = ops.conv_pack_weights(w, tile_n=64, fmt=1)
    f_h16n = lambda: ops.conv2d_planar(xp16, pk16n, (O, C, k, k), (B, H, W), b, None, stride=s, padding=k // 2, relu=True, out="planes", fmt=1, out_scale=osc, tile_n=64)
    f_sp = lambda: ops.split_planes(x)
    to, tt, tn, tp, tsp, tp64 = timeit(f_ours), timeit(f_torch), timeit(f_nchw), timeit(f_pl), timeit(f_sp), timeit(f_p64)
    th16, th16n = timeit(f_h16), timeit(f_h16n)
    gf = 2.0 * B * out.shape[1] * out.shape[2] * O * C * k * k / 1e9
    err = (f_ours().permute(0, 3, 1, 2) - torch.relu(f_torch())).abs().max().item()
    tot_o += tp
    tot_t += min(tt, tn)
    print(f"{name:26s} {gf:7.1f} GF | fp16x2 {th16:7.1f} us {gf / th16 * 1e3:6.1f} TF n64 {th16n:7.1f} us {gf / th16n * 1e3:6.1f} TF | planar {tp:8.1f} us {gf / tp * 1e3:7.1f} TF | n64 {tp64:8.1f} us {gf / tp64 * 1e3:7.1f} TF (split {tsp:6.1f} us) | f32-in {to:8.1f} us {gf / to * 1e3:7.1f} TF | MIOpen NHWC {tt:8.1f} us {gf / tt * 1e3:6.1f} TF | "
          f"NCHW {tn:8.1f} us {gf / tn * 1e3:6.1f} TF | maxdiff {err:.2e}")
print(f"TOTAL ours {tot_o:.1f} us  best-MIOpen {tot_t:.1f} us")
