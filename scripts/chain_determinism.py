"""Run-to-run determinism of conv_chain_kernel: same inputs, repeated launches (optionally with a second stream keeping the GPU busy)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops, _lib

DEV = "cuda"
H, W = 96, 160
g = torch.Generator().manual_seed(0)
w2 = torch.randn(64, 64, 3, 3, generator=g) / 24
w3 = torch.randn(256, 64, 1, 1, generator=g) / 8
wds = torch.randn(256, 64, 1, 1, generator=g) / 8
w1 = torch.randn(64, 256, 1, 1, generator=g) / 16
b2, b3, b1 = torch.randn(64, generator=g).to(DEV), torch.randn(256, generator=g).to(DEV), torch.randn(64, generator=g).to(DEV)
geo = _lib.ConvGeom()
geo.C, geo.Cout, geo.kh, geo.kw, geo.sh, geo.sw, geo.ph, geo.pw, geo.groups, geo.fmt = 64, 64, 3, 3, 1, 1, 1, 1, 1, 1
ops.planar_range_flag()
w2p, s2 = ops.conv_pack_weights_kxr(w2.to(DEV), geo)
from stmask_amd.planar import PlanarConv
side_conv = PlanarConv(torch.randn(256, 256, 3, 3, device=DEV) / 48, None, 1, 1, relu=True, fmt=1)
side_x = ops.split_planes(torch.randn(8 * 48 * 80, 256, device=DEV), 1)
for B in (4, 32):
    for proj in (False, True):
        for want_z in (True, False):
            tail, s3, s1 = ops.chain_pack_tail(w3.to(DEV), w1.to(DEV) if want_z else None, wds.to(DEV) if proj else None)
            mid1 = ops.split_planes(torch.randn(B * H * W, 64, generator=g).abs().to(DEV), 1)
            x = ops.split_planes(torch.randn(B * H * W, 64 if proj else 256, generator=g).abs().to(DEV), 1)
            ref = None
            bad = 0
            side = torch.cuda.Stream()
            junk = torch.randn(4096, 4096, device=DEV)
            ew = torch.randn(32 << 20, device=DEV)
            for it in range(16):
                if 4 <= it < 8:
                    with torch.cuda.stream(side):
                        for _ in range(4):
                            junk = junk @ junk * 1e-4
                elif 8 <= it < 12:
                    with torch.cuda.stream(side):           # zero-LDS elementwise kernels: the only ones that fit beside a chain workgroup on a CU
                        for _ in range(40):
                            ew.mul_(1.0001).add_(1e-6)
                elif it >= 12:
                    with torch.cuda.stream(side):           # the planar convolution's ring kernel (144 KB of LDS)
                        for _ in range(3):
                            side_conv(side_x, ("img", 8, 48, 80))
                y, z = ops.bottleneck_chain(mid1, x, w2p, tail, b2, b3, b1 if want_z else None, (s2, s3, s1), B, H, W, want_z=want_z, proj=proj)
                torch.cuda.synchronize()
                cur = (y.clone(), z.clone() if z is not None else None)
                if ref is None:
                    ref = cur
                else:
                    dy = (cur[0].view(torch.int16) != ref[0].view(torch.int16)).sum().item()
                    dz = (cur[1].view(torch.int16) != ref[1].view(torch.int16)).sum().item() if want_z else 0
                    if dy or dz:
                        bad += 1
                        print(f"  B={B} proj={proj} z={want_z} run {it}: {dy} y elements, {dz} z elements differ; first y idx", (cur[0].view(torch.int16) != ref[0].view(torch.int16)).nonzero()[:3].tolist())
            print(f"B={B} proj={proj} want_z={want_z}: {bad} of 15 repeats differ")
