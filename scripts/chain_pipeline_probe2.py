"""Keep every step's layer1 output (and the stem's) alive and compare them across two passes, plus the detections."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stmask_amd import planar

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 32
args = bench.parse_args(["--clips", str(clips), "--steps", "10", "--warmup", "2"])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
net = bench.build_net(args, dev)
kept = []
orig = planar.PlanarBackbone._stages


def stages(self, xp, B, C, H, W):
    out = orig(self, xp, B, C, H, W)
    kept.append((xp, self.out_planes[0][0], self.out_planes[1][0]))      # stem planes, layer1 output, layer2 output (references only)
    return out


planar.PlanarBackbone._stages = stages
res = []
for rep in range(10):
    kept.clear()
    run = bench.Runner(args, dev, 0, 1, clips, net=net)
    run.keep = []
    run.timed(args.warmup, args.steps)
    torch.cuda.synchronize()
    cur = [k[1] for k in kept]
    if rep == 0:
        ref = [t.clone() for t in cur]
    else:
        for i, (x, y) in enumerate(zip(ref, cur)):
            if not torch.equal(x, y):
                d = (x.view(torch.int16) != y.view(torch.int16))        # [2 planes, 8 slabs, N, 32]
                idx = d.nonzero()
                print("pass", rep, "trunk", i, "differing elements", idx.shape[0], "planes", idx[:, 0].unique().tolist(), "slabs", idx[:, 1].unique().tolist())
                px = idx[:, 2].unique()
                print("  pixels", px.numel(), "min", int(px.min()), "max", int(px.max()), "tiles of 128:", (px // 128).unique().tolist()[:24])
                print("  pixel offsets within tile:", (px % 128).unique().tolist()[:40])
                print("  channels in slab", idx[:, 3].unique().tolist())
                p0 = int(px[0])
                print("  first pixel", p0, "image", p0 // 15360, "y", (p0 % 15360) // 160, "x", p0 % 160)
                va = planar.ops.planes_to_f32(x[:, :, p0:p0 + 1].contiguous()); vb = planar.ops.planes_to_f32(y[:, :, p0:p0 + 1].contiguous())
                print("  max abs diff at that pixel", float((va - vb).abs().max()), "max value", float(va.abs().max()))
                dv = (va - vb).flatten()
                print("  channels differing at that pixel:", dv.nonzero().flatten().tolist()[:40])
                sys.exit(0)
    del run
print("no difference in 5 passes")
