"""The benchmark's model / pipeline driver: build_net, Runner (step, timed)."""
import time

import torch

from .launch import barrier


def backbone_tag(cfg):
    depth = {(3, 4, 6, 3): "R50", (3, 4, 23, 3): "R101"}.get(tuple(cfg.backbone_layers), "ResNet")
    return depth + ("-DCN" if any(cfg.backbone_dcn_layers) else "") + "-FPN"


def heads_tag(cfg):
    fcb = ("+FCB(ada)" if cfg.use_pred_offset else "+FCB(ali)") if cfg.use_dcn_class else ""
    return "FCA" + fcb + (" + temporal fusion" if cfg.temporal_fusion_module else "")


def image_tag(h, w):
    """Tensor size -> the image size it is the /32 padding of (360x640 -> 384x640, 720x1280 -> 736x1280)."""
    known = {(384, 640): "360x640", (736, 1280): "720x1280"}
    return known.get((h, w), f"{h}x{w}")


def build_net(args, dev, planes=None):
    from stmask_amd import synthetic
    from stmask_amd.config import get_cfg
    from stmask_amd.model import STMask
    planes = planes or args.planes
    net = STMask(get_cfg(args.config))
    net.eval()
    synthetic.fill_state_dict(net, seed=0, bg_bias=synthetic.BENCH_BG_BIAS)
    net = net.to(dev)
    if args.fuse:
        from stmask_amd.fuse import optimize_for_inference
        # BN folded into conv / DCN weights, bias (+residual) + ReLU as one epilogue pass; every dense convolution on
        # stm_conv2d_planar_f32 (split-operand MFMA convolution, all FPN levels per launch)
        optimize_for_inference(net, planar=args.planar and args.channels_last, planes=planes)
    if args.channels_last:
        net = net.to(memory_format=torch.channels_last)
        net.TemporalNet = net.TemporalNet.to(memory_format=torch.contiguous_format)
    return net


class Runner:
    """One pipeline over resident synthetic clips; step(t) = every local clip advances one frame + the detection all-gather."""

    def __init__(self, args, dev, rank, world, clips, planes=None, net=None, max_instances=None):
        from stmask_amd import synthetic
        from stmask_amd.pipeline import BatchedClipPipeline, ClipPipeline
        self.args, self.dev, self.clips_n, self.T = args, dev, clips, args.frames
        self.net = net if net is not None else build_net(args, dev, planes)
        # clip c of this rank = global clip rank + c*world (stmask_amd.dist.shard_clips); inputs resident in HBM
        clip_t = torch.stack([synthetic.synthetic_clip(self.T, args.height, args.width, seed=rank + c * world)
                              for c in range(clips)]).to(dev)                      # [clips, T, 3, H, W]
        fmt = torch.channels_last if args.channels_last else torch.contiguous_format
        self.frames_t = [clip_t[:, t].contiguous(memory_format=fmt) for t in range(self.T)]   # in the trunk's layout
        del clip_t
        self.batched = args.pipeline == "batched"
        self.pipe = BatchedClipPipeline(self.net, clips) if self.batched else ClipPipeline(self.net, clips)
        if self.batched:
            self.pipe.max_instances = args.max_instances if max_instances is None else max_instances
            self.pipe.prefetch_early = args.overlap == "early"
            gm = getattr(args, "graph", "auto")
            # (round 6: graphs at every batch size -- at 32 clips two replayed trunks in flight are +2 % over the eager trunk, pipeline.BatchedClipPipeline.LARGE_BATCH)
            self.pipe.use_graph = gm in ("on", "auto") and args.fuse and args.planar and args.channels_last
        self.tracked_sum = 0.0
        self.tracked_steps = 0
        from stmask_amd.dist import DetectionGatherer
        self.gatherer = DetectionGatherer(dev)
        self.keep = None             # a list: the gathered detections of every step are kept (the two-rank check compares them)

    def step(self, t):
        from stmask_amd import dist as sdist
        T, pipe = self.T, self.pipe
        if self.batched and self.args.overlap != "off":
            # the next frame's trunk starts on a second stream while this frame's tracker logic (tiny launches, two host
            # reads) runs; every step still enqueues exactly one trunk
            # the frames of the next two calls: under graph replay (small batches) two trunks run ahead on two side streams (BatchedClipPipeline._prefetch_trunk)
            out = pipe.step(self.frames_t[t % T], is_first=(t % T == 0), next_frames=[self.frames_t[(t + k) % T] for k in range(1, 1 + max(2, pipe.PREFETCH_DEPTH))])
        else:
            out = pipe.step(self.frames_t[t % T], is_first=(t % T == 0))
        if self.batched:
            self.tracked_sum += sum(pipe.prev_n) / max(self.clips_n, 1)
            self.tracked_steps += 1
        packed = out if self.batched else sdist.pack_detections(out, top_k=self.net.cfg.nms_top_k, device=self.dev)
        # the all-gather rides on its own stream (stmask_amd.dist.DetectionGatherer): neither this step's tail nor the next trunk waits
        full = self.gatherer.gather(packed)
        if self.keep is not None:
            self.keep.append(full)
        return full

    def timed(self, warmup, steps, use_dist=False, collect=False):
        """W untimed steps, then exactly K steps bracketed by barrier + synchronize; returns (seconds, last output, timings)."""
        from stmask_amd import ops
        for t in range(warmup):
            self.step(t)
        t_first = warmup
        if self.batched and self.pipe.use_graph and not collect:
            # the trunk graphs are captured lazily, one slot per trunk call (two eager calls first): keep the captures out of the timed region
            while len(self.pipe._graphs) < self.pipe.n_graph_slots and t_first < warmup + self.pipe.n_graph_slots + 4:
                self.step(t_first)
                t_first += 1
        torch.cuda.synchronize()
        if use_dist:
            barrier()
        if collect:
            ops.im2col_timing(True)
            ops.conv_timing(True)
            ops.fused_dcn_timing(True)
        self.tracked_sum, self.tracked_steps = 0.0, 0
        tm = getattr(self.pipe, "timer", None)
        if tm is not None and tm.on:
            tm.acc.clear()   # diagnosis runs: stage times of the timed steps only
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for t in range(t_first, t_first + steps):
            out = self.step(t)
        self.gatherer.wait()
        torch.cuda.synchronize()
        if use_dist:
            barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        timing = ops.im2col_timing(False) if collect else None
        conv_t = (ops.conv_timing(False) or []) if collect else None
        if collect:
            self.fused_t = ops.fused_dcn_timing(False) or []       # launches of the fused deformable convolution (csrc/dcn_fused.hip) of this pass
        return elapsed, out, timing, conv_t

