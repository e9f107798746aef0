#!/usr/bin/env python3
"""Does what ran EARLIER in a process change the small-batch step?  (It did: HIP maps streams onto a few hardware queues, torch hands out pool streams
round-robin, and a pipeline whose two trunk side streams land on one queue -- or on the main stream's -- loses the overlap it was built for, silently:
780 -> 560 -> 440 frames/s single-stream.  stmask_amd.pipeline.concurrent_side_streams picks the streams by test since.)
usage: probe_stream_order.py order | big8 | wsclear      (single-stream runs before / after a 32-clip (or 8- / 16-clip) run in the same process)"""
import sys, os, torch, time, gc
sys.path.insert(0, os.getcwd())
import bench
args = bench.parse_args(["--no-cpu-baseline", "--no-extras"])
dev = torch.device("cuda:0")
net = bench.build_net(args, dev)
mode = sys.argv[1]
def small(tag):
    r2 = bench.Runner(args, dev, 0, 1, 1, net=net)
    for k in range(2):
        el, *_ = r2.timed(4, 60)
        print(mode, tag, "clips 1 pass", k, round(60 / el, 1), flush=True)
    del r2; torch.cuda.empty_cache()
def big(n=32):
    run = bench.Runner(args, dev, 0, 1, n, net=net)
    el, *_ = run.timed(4, 10)
    print(mode, "%d clips" % n, n * 10 / el, flush=True)
    del run; gc.collect(); torch.cuda.empty_cache()
if mode == "order":
    small("before"); big(); small("after"); time.sleep(8); small("after+8s")
elif mode == "big8":
    small("before"); big(8); small("after big8"); big(16); small("after big16")
elif mode == "wsclear":
    small("before"); big()
    from stmask_amd import ops
    ops._ws_cache.clear(); gc.collect(); torch.cuda.empty_cache()
    small("after ws clear")
