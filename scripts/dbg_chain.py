import sys
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from stmask_amd import ops, _lib
DEV = "cuda"
def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)
B, H, W = 2, 6, 9
mid1, x = rnd(B, H, W, 64, seed=3).abs(), rnd(B, H, W, 256, seed=4) * 0
g = _lib.ConvGeom()
g.C, g.Cout, g.kh, g.kw, g.sh, g.sw, g.ph, g.pw, g.groups, g.fmt = 64, 64, 3, 3, 1, 1, 1, 1, 1, 1
ops.planar_range_flag()
w2 = torch.zeros(64, 64, 3, 3); w2[:, :, 1, 1] = torch.eye(64)
w3 = torch.zeros(256, 64)
for k in range(4):
    w3[64 * k:64 * (k + 1)] = torch.eye(64) * (k + 1)
w1 = torch.zeros(64, 256); w1[:, 64:128] = torch.eye(64)
w2p, s2 = ops.conv_pack_weights_kxr(w2.to(DEV), g)
tail, s3, s1 = ops.chain_pack_tail(w3.to(DEV), w1.to(DEV))
y, z = ops.bottleneck_chain(ops.split_planes(mid1.to(DEV), 1), ops.split_planes(x.to(DEV), 1), w2p, tail, None, None, None, (s2, s3, s1), B, H, W)
got = ops.planes_to_f32(y).cpu().view(B * H * W, 256)
exp = torch.cat([mid1.view(-1, 64) * (k + 1) for k in range(4)], -1)
err = (got - exp).abs()
print("y err by 16-ch tile:", [round(err[:, 16 * i:16 * i + 16].max().item(), 4) for i in range(16)])
print("y err by pixel block of 16:", [round(err[16 * i:16 * i + 16].max().item(), 4) for i in range(7)])
p = 5
print("px", p, "exp", exp[p, :8], "\n got", got[p, :8])
print("got ch 0..64 step: ", got[p, :64:4])
print("exp ch 0..64 step: ", exp[p, :64:4])
gz = ops.planes_to_f32(z).cpu().view(B * H * W, 64)
print("z err", (gz - got[:, 64:128]).abs().max().item(), "z vs exp", (gz - exp[:, 64:128]).abs().max().item())
# x only
x = rnd(B, H, W, 256, seed=4)
y, z = ops.bottleneck_chain(ops.split_planes(mid1.to(DEV) * 0, 1), ops.split_planes(x.to(DEV), 1), w2p, tail, None, None, None, (s2, s3, s1), B, H, W)
got = ops.planes_to_f32(y).cpu().view(B * H * W, 256)
err = (got - x.view(-1, 256).relu()).abs()
print("shortcut err by 16-ch tile:", [round(err[:, 16 * i:16 * i + 16].max().item(), 4) for i in range(16)])
xe = x.view(-1, 256).relu()
for p in (0, 5, 17, 40):
    print("px", p, "exp", xe[p, :12].numpy().round(3), "\n      got", got[p, :12].numpy().round(3))
# where does got[p, c] come from?
xf = x.view(-1, 256)
for p in (0, 5):
    for c in (0, 1, 4, 8, 16):
        v = got[p, c].item()
        hits = ((xf - v).abs() < 1e-3).nonzero()
        print(p, c, round(v, 4), hits[:4].tolist())
