"""conv_chain.hip against the three planar launches it replaces, at layer1's shape (96 x 160, 64 / 256 channels)."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops, _lib
from stmask_amd.planar import PlanarConv

DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, W = 96, 160
g = torch.Generator().manual_seed(0)
w2 = torch.randn(64, 64, 3, 3, generator=g) / 24
w3 = torch.randn(256, 64, 1, 1, generator=g) / 8
w1 = torch.randn(64, 256, 1, 1, generator=g) / 16
b2, b3, b1 = torch.randn(64, generator=g), torch.randn(256, generator=g), torch.randn(64, generator=g)
mid1 = torch.randn(B * H * W, 64, generator=g).abs()
x = torch.randn(B * H * W, 256, generator=g).abs()
geo = _lib.ConvGeom()
geo.C, geo.Cout, geo.kh, geo.kw, geo.sh, geo.sw, geo.ph, geo.pw, geo.groups, geo.fmt = 64, 64, 3, 3, 1, 1, 1, 1, 1, 1
ops.planar_range_flag()
w2p, s2 = ops.conv_pack_weights_kxr(w2.to(DEV), geo)
tail, s3, s1 = ops.chain_pack_tail(w3.to(DEV), w1.to(DEV))
m1p, xp = ops.split_planes(mid1.to(DEV), 1), ops.split_planes(x.to(DEV), 1)
b2d, b3d, b1d = b2.to(DEV), b3.to(DEV), b1.to(DEV)
y = torch.empty(2, 8, B * H * W, 32, device=DEV, dtype=torch.float16)
z = torch.empty(2, 2, B * H * W, 32, device=DEV, dtype=torch.float16)
l2 = PlanarConv(w2.to(DEV), b2d, 1, 1, relu=True, fmt=1)
l3 = PlanarConv(w3.to(DEV), b3d, 1, 0, relu=True, fmt=1)
l1 = PlanarConv(w1.to(DEV), b1d, 1, 0, relu=True, fmt=1)
m2 = torch.empty(2, 2, B * H * W, 32, device=DEV, dtype=torch.float16)
y3 = torch.empty_like(y)
z3 = torch.empty_like(z)


def chain(want_z=True):
    ops.bottleneck_chain(m1p, xp, w2p, tail, b2d, b3d, b1d, (s2, s3, s1), B, H, W, y=y, z=z, want_z=want_z)


def three():
    l2(m1p, ("img", B, H, W), out_planes=m2)
    l3(m2, ("img", B, H, W), residual=xp, out_planes=y3)
    l1(y3, ("img", B, H, W), out_planes=z3)


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


three()
chain()
torch.cuda.synchronize()
ya, yb = ops.planes_to_f32(y), ops.planes_to_f32(y3)
za, zb = ops.planes_to_f32(z), ops.planes_to_f32(z3)
print("max |y - y3| %.3g (max %.3g)   max |z - z3| %.3g (max %.3g)" % ((ya - yb).abs().max().item(), yb.abs().max().item(), (za - zb).abs().max().item(), zb.abs().max().item()))
t3 = timeit(three)
tc = timeit(chain)
tn = timeit(lambda: chain(False))
gb = B * H * W * (64 * 4 + 256 * 4 + 256 * 4 + 64 * 4) / 1e9
print(f"B={B}: three launches {t3:.1f} us   chain {tc:.1f} us ({gb / tc * 1e6:.0f} GB/s algorithmic)   chain without z {tn:.1f} us")
