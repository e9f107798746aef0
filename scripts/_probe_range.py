import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from stmask_amd import synthetic
from stmask_amd.fuse import optimize_for_inference
from test_gpu_model import build
for scale in (1.0, 1e5, 1e7):
    frames = (synthetic.synthetic_clip(1, 128, 192, seed=2) * scale).cuda().contiguous(memory_format=torch.channels_last)
    for planes in ("fp16x2", "bf16x3"):
        net = build("STMask_plus_resnet50_config")
        optimize_for_inference(net, planar=True, planes=planes)
        net = net.to(memory_format=torch.channels_last)
        with torch.no_grad():
            fe, p = net.forward_single(frames)
        print(scale, planes, "in", frames.abs().max().item(), "conf finite", torch.isfinite(p["conf"]).all().item(), "conf max", p["conf"].abs().max().item(),
              "proto max", p["proto"].abs().max().item(), [f.abs().max().item() for f in fe if f is not None])
