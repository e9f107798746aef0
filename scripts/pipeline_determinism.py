"""Two Runner passes over the same clips on one net, in one process: are the kept detection blocks bit-equal?  usage: [clips] [steps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
args = bench.parse_args(["--clips", str(clips), "--steps", str(steps), "--warmup", "3"])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
net = bench.build_net(args, dev)
keeps = []
for rep in range(3):
    run = bench.Runner(args, dev, 0, 1, clips, net=net)
    run.keep = []
    run.timed(args.warmup, args.steps)
    torch.cuda.synchronize()
    keeps.append([k.clone() for k in run.keep])
    del run
    torch.cuda.empty_cache()
for rep in (1, 2):
    neq = [t for t in range(len(keeps[0])) if not torch.equal(keeps[0][t], keeps[rep][t])]
    mx = max(float((keeps[0][t] - keeps[rep][t]).abs().max()) for t in range(len(keeps[0])))
    print(f"clips {clips}: pass {rep} vs pass 0: {len(neq)} of {len(keeps[0])} steps differ, first {neq[:3]}, max abs diff {mx}")
