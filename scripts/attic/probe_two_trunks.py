#!/usr/bin/env python3
"""What frame-level pipelining of the trunk could give small batches: the trunk (forward_single) of `clips` frames captured into HIP graphs with their own
memory pools and workspaces, replayed (a) back to back on one stream, (b) alternately on two streams, (c) on three -- launches per second of one
trunk.  At 1-8 clips a replayed trunk is a chain of ~110-155 dependent small launches (GPU-side launch latency, not work): independent chains overlap.
usage: probe_two_trunks.py [clips=1]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stmask_amd import ops
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 1
args = bench.parse_args(["--clips", str(clips), "--no-cpu-baseline", "--no-extras"])
dev = torch.device("cuda:0")
net = bench.build_net(args, dev)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(clips, 3, args.height, args.width, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(3):
        net.forward_single(x)
torch.cuda.synchronize()
NG = 3
graphs = []
for i in range(NG):
    ws = {}
    xin = x.clone(memory_format=torch.preserve_format)
    gr = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream(device=dev)
    cap.wait_stream(torch.cuda.current_stream())
    with torch.no_grad(), ops.workspace_scope(ws):
        with torch.cuda.stream(cap):
            net.forward_single(xin)
        with torch.cuda.graph(gr, stream=cap):
            out = net.forward_single(xin)
    torch.cuda.current_stream().wait_stream(cap)
    graphs.append((gr, xin, out, ws))
torch.cuda.synchronize()
streams = [torch.cuda.Stream(device=dev) for _ in range(NG)]


def run(nstreams, n=120):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(streams[i % nstreams]):
            graphs[i % NG if nstreams > 1 else 0][0].replay()
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


for ns in (1, 2, 3, 1, 2, 3):
    r = run(ns)
    print("clips %d: %d stream(s): %.1f trunks/s = %.1f frames/s (trunk only), %.3f ms per trunk" % (clips, ns, r, r * clips, 1e3 / r), flush=True)
