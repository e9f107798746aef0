#!/usr/bin/env python3
"""What one per-step all-gather of detections costs a rank (world size 1, RCCL): host time of the call, device time of the collective on the
communication stream, and what a busy compute stream beside it loses.  Forms: DetectionGatherer.gather as the pipeline calls it; the same into a
PERSISTENT output buffer; a plain device copy (what the collective degenerates to at world size 1).
run: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 scripts/probe_allgather_cost.py"""
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import dist as sdist
fd = os.dup(1); os.dup2(2, 1)
def say(*a):
    os.write(fd, (" ".join(str(x) for x in a) + "\n").encode())
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]), device_id=dev)
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 32
packed = torch.randn(clips, 200, 40, device=dev)
g = sdist.DetectionGatherer(dev)
for _ in range(5):
    g.gather(packed); g.wait()
torch.cuda.synchronize()
N = 50
# 1. host time of the call, device time of the collective
host = 0.0
e0s, e1s = [], []
for _ in range(N):
    t0 = time.perf_counter(); out = g.gather(packed); host += time.perf_counter() - t0
    g.wait()
torch.cuda.synchronize()
say("DetectionGatherer.gather: host %.1f us per call" % (host / N * 1e6))
comm = torch.cuda.Stream(device=dev)
outp = torch.empty(clips * dist.get_world_size(), 200, 40, device=dev)
with torch.cuda.stream(comm):
    for _ in range(5):
        dist.all_gather_into_tensor(outp, packed)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.record()
    for _ in range(N):
        dist.all_gather_into_tensor(outp, packed)
    e1.record(); th = time.perf_counter() - t0
    torch.cuda.synchronize()
say("all_gather_into_tensor, persistent output, back to back: host %.1f us, device %.1f us per call" % (th / N * 1e6, e0.elapsed_time(e1) / N * 1e3))
# 2. a busy compute stream beside it: matmuls of ~1 ms each; with / without a collective per matmul
a = torch.randn(8192, 8192, device=dev, dtype=torch.float16); b = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
def busy(with_coll, form):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        c = a @ b
        if with_coll:
            if form == "gatherer":
                g.gather(packed)
            elif form == "persistent":
                comm.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(comm):
                    dist.all_gather_into_tensor(outp, packed)
            else:
                comm.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(comm):
                    outp.copy_(packed)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N
base = busy(False, None)
for form in ("gatherer", "persistent", "copy"):
    say("compute stream: %.3f ms per matmul alone, %.3f with one %s per matmul beside it (+%.3f ms)" % (base, busy(True, form), form, busy(True, form) - base))
    base = busy(False, None)
dist.destroy_process_group()
