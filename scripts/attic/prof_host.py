"""Host-side profile of the single-stream (or small-batch) step: where the Python time of BatchedClipPipeline.step goes once the trunk replays
from HIP graphs.  usage: prof_host.py [clips] [steps]   -> cProfile table (tottime) + wall time per step + GPU-busy share from two events"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
args = bench.parse_args(["--clips", str(clips), "--steps", str(steps), "--warmup", "8", "--no-cpu-baseline", "--no-extras"])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
net = bench.build_net(args, dev)
run = bench.Runner(args, dev, 0, 1, clips, net=net)
run.timed(8, 16)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for t in range(24, 24 + steps):
    run.step(t)
pr.disable()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"clips {clips}: {dt / steps * 1e3:.3f} ms per step under the profiler ({clips * steps / dt:.0f} frames/s)")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(35)
