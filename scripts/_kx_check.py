import os, sys, torch
sys.path.insert(0, "/root/repo")
from stmask_amd import ops
for (H, W, C, O, k) in [(48, 80, 256, 256, 3), (96, 160, 64, 64, 3), (41, 123, 512, 1024, 3), (48, 80, 256, 128, 5)]:
    x = torch.randn(8, H, W, C, device="cuda"); w = torch.randn(O, C, k, k, device="cuda") * (C * k * k) ** -0.5; b = torch.randn(O, device="cuda")
    pk, osc = ops.conv_pack_weights(w, fmt=1); xp = ops.split_planes(x, 1)
    os.environ["STM_CONV_KX"] = "0"
    y0 = ops.conv2d_planar(xp, pk, (O, C, k, k), (8, H, W), b, None, padding=k // 2, relu=True, out="f32", fmt=1, out_scale=osc)
    os.environ["STM_CONV_KX"] = "1"
    y1 = ops.conv2d_planar(xp, pk, (O, C, k, k), (8, H, W), b, None, padding=k // 2, relu=True, out="f32", fmt=1, out_scale=osc)
    os.environ["STM_CONV_KX"] = "0"
    print((H, W, C, O, k), "max diff kx vs tap", (y0 - y1).abs().max().item(), "max", y0.abs().max().item())
