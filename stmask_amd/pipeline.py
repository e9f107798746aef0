"""Batched clip pipeline: many independent video clips advance one frame per step on one GPU.

forward_single has no cross-frame dependence, so frame t of all local clips goes through the trunk
and heads as ONE batch; the temporal-fusion / tracker stage is per clip in the reference (previous-frame features and
tracker state, track_TF.py:52-54,96-100).  Two drivers:

* ``ClipPipeline``        -- reference-shaped: one ``Track_TF`` per clip, the layer API of ``stmask_amd.layers``
                             (what a user of the reference's eval loop gets; ~4 host syncs per clip and frame).
* ``BatchedClipPipeline`` -- same arithmetic, but the state of all clips lives in concatenated tensors and every stage
                             is ONE launch for all clips: fused decode + threshold + Fast NMS (no sync), one lincomb
                             launch for all detections (per-row prototype index), one correlation / RoIAlign /
                             TemporalNet / decode / lincomb chain for all tracked instances, one cross matrix for the
                             matching scores.  Two small device->host reads per STEP (detection counts, match ids).
"""
import os

import torch
import torch.nn.functional as F

from . import ops
from .layers import Track_TF, generate_candidate
from .layers.box_utils import center_size, sanitize_coordinates_hw

ROI_CHUNKS = (256, 128, 64)  # TemporalNet only ever sees these RoI batch sizes: three dense-conv shapes in total


class _StageTimer:
    """Optional per-stage wall-clock breakdown (STM_PIPE_TIMING=1): synchronises around every stage, so only for
    diagnosis -- never enabled in a timed benchmark run."""

    def __init__(self):
        import os
        self.on = os.environ.get("STM_PIPE_TIMING", "0") == "1"
        self.acc, self.t0 = {}, None

    def tic(self):
        if self.on:
            import time
            torch.cuda.synchronize()
            self.t0 = time.perf_counter()

    def toc(self, name):
        if self.on:
            import time
            torch.cuda.synchronize()
            t = time.perf_counter()
            self.acc[name] = self.acc.get(name, 0.0) + (t - self.t0)
            self.t0 = t


class ClipPipeline:
    def __init__(self, net, n_clips):
        self.net = net
        self.cfg = net.cfg
        if not self.cfg.temporal_fusion_module:
            raise NotImplementedError("ClipPipeline drives the temporal-fusion configs (all STMask_plus_* configs)")
        self.trackers = [Track_TF(cfg=self.cfg) for _ in range(n_clips)]
        self.t = 0

    @torch.no_grad()
    def step(self, frames, is_first=None):
        """frames [n_clips,3,H,W] = the next frame of every clip -> list of per-clip detection dicts."""
        net, cfg = self.net, self.cfg
        first = (self.t == 0) if is_first is None else is_first
        fpn_outs, pred = net.forward_single(frames)
        pred["conf"] = F.softmax(pred["conf"], -1)
        pred["fpn_feat"] = fpn_outs[net.correlation_selected_layer]
        pred["T2S_feat"] = pred["T2S_feat"][net.correlation_selected_layer]
        candidates = generate_candidate(pred, cfg=cfg)
        out = []
        for i, cand in enumerate(candidates):
            det = net.Detect_TF.detect(cand, is_output_candidate=True)
            meta = {"is_first": first, "video_id": i, "frame_id": self.t}
            out.append(self.trackers[i].track(net, det, meta, img=None))
        self.t += 1
        return out


_ROW_KEYS = ("box", "mask_coeff", "track", "class", "score", "centerness", "mask")


class _TrackedRows(dict):
    """The tracked set's row tensors.  After a tracker update the soft masks of the merged set are not gathered (61 KB per row,
    rewritten from the prototypes at the next step anyway; the keep rule reads their bit words): `rows["mask"]` gathers them on
    first use from the two sources and the gather plan of that update."""

    def defer_mask(self, prev_mask, det_mask, plan, n_prev):
        self._deferred = (prev_mask, det_mask, plan, n_prev)
        dict.pop(self, "mask", None)

    def _materialize(self):
        if not dict.__contains__(self, "mask") and getattr(self, "_deferred", None) is not None:
            a, b, plan, n_prev = self._deferred
            dict.__setitem__(self, "mask", ops.gather_rows2([a], [b], plan, n_prev)[0])
            self._deferred = None

    def __getitem__(self, key):
        if key == "mask":
            self._materialize()
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        self._materialize()
        return dict.keys(self)

    def values(self):
        self._materialize()
        return dict.values(self)

    def items(self):
        self._materialize()
        return dict.items(self)

    def __iter__(self):
        self._materialize()
        return dict.__iter__(self)

    def __setitem__(self, key, value):
        if key == "mask":
            self._deferred = None
        dict.__setitem__(self, key, value)

    def __contains__(self, key):
        return dict.__contains__(self, key) or (key == "mask" and getattr(self, "_deferred", None) is not None)


_SIDE_STREAMS = {}


def concurrent_side_streams(dev, n=2):
    """n streams that really run BESIDE the current stream and beside each other.  HIP maps streams onto a few hardware queues; two streams that land on
    one queue run their work one after the other, silently -- measured: the same pipeline gives 780 frames/s single-stream with two trunk graphs in
    flight, 560 when its side streams happen to share a queue (after another pipeline in the same process had used up some streams of torch's pool) and
    440 when one of them shares the main stream's queue; GPU_MAX_HW_QUEUES only moves the collisions.  So the streams are picked by test, once per
    process and main stream: a 0.5-ms spin kernel on the main stream, on the streams chosen so far and on the candidate -- the candidate is taken when
    all of them finish in the time of one."""
    import time
    dev = torch.device(dev)
    main = torch.cuda.current_stream(dev)
    key = (dev.index, main.cuda_stream)
    have = _SIDE_STREAMS.get(key, [])
    if len(have) >= n:
        return have[:n]                              # (the list only grows: the first trunk_stream_count() are the trunk streams, the one after them serves the detection gather)
    cands = [torch.cuda.Stream(device=dev) for _ in range(16)]
    chosen = list(have)
    spin = getattr(torch.cuda, "_sleep", None)
    if spin is not None and not torch.cuda.is_current_stream_capturing():
        cycles = 1_000_000

        def run(streams):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for s_ in streams:
                with torch.cuda.stream(s_):
                    spin(cycles)
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0

        run([main])
        base = min(run([main]) for _ in range(3))
        for c in cands:
            if len(chosen) == n:
                break
            if min(run([main] + chosen + [c]) for _ in range(2)) < 1.4 * base:
                chosen.append(c)
    for c in cands:                                  # (no spin kernel, or fewer independent queues than asked for: any streams will do -- results never depend on it)
        if len(chosen) == n:
            break
        if c not in chosen:
            chosen.append(c)
    _SIDE_STREAMS[key] = chosen
    return chosen[:n]


def trunk_stream_count():
    """How many of concurrent_side_streams()'s streams the pipelines rotate their prefetched trunks over (dist.DetectionGatherer takes the next one)."""
    return max(2, BatchedClipPipeline.PREFETCH_DEPTH)


class BatchedClipPipeline:
    """All clips' tracker state concatenated (rows sorted by clip); per-clip row ranges are host integers."""

    def __init__(self, net, n_clips):
        self.net, self.cfg, self.B = net, net.cfg, n_clips
        # temporal-fusion configs: Detect_TF + Track_TF (CandidateShift, soft masks, the keep rule); without the module the reference runs
        # Detect + Track (detection.py:98-137, track.py:56-179: binary masks, the (mask_ious > 0.3).sum() < 2 update gate, the frame's own
        # detections as output) -- _step_nontf
        self.tf = bool(self.cfg.temporal_fusion_module)
        self.range_fallback = True  # an fp16 plane graph that leaves its range is replaced by the bf16x3 graph and the step repeated (see step)
        self.fell_back = False
        self._last = None           # non-TF: the last step's detections (rows, ids, clip ranges) for detections()
        self.t = 0
        self.prev = None            # dict of concatenated row tensors
        self.prev_n = [0] * n_clips  # tracked instances per clip
        self.prev_feat = None       # (P4 [B,256,h,w], T2S [B,256,h,w]) of the previous frame
        self.has_prev = [False] * n_clips
        self.tracked = [[] for _ in range(n_clips)]  # host-side "frames since last match" counters
        self.timer = _StageTimer()
        self._pending = []          # FIFO of (frames, (fpn_outs, pred), event): trunks of the NEXT frame(s), running on the side stream(s)
        self._sides = []            # side streams of the prefetched trunks (two under graph replay: see _prefetch_trunk)
        self._side_next = 0
        self.prefetch_early = True   # start the next trunk at the beginning of step() (measured best at every batch size: +0.5 % at 32 clips, +5.7 % at 8, +13 % at 1);
                                     # False: after the TF convolutions are enqueued (the two big kernel groups then never share the GPU: clean per-kernel timings)
        self.use_graph = False       # replay the trunk (forward_single) from captured HIP graphs: see _trunk
        self.graph_active = False
        self._graphs = []            # round-robin slots: (static input, graph, outputs)
        self._graph_next = 0
        self._graph_warm = 0
        self._graph_ws = []          # per slot: the workspaces its captured graph writes into (kept alive here)
        self._graph_planes = None    # plane format of the net's inference graph when the slots were captured (a net may serve several pipelines: _trunk)
        # Workload knob of the benchmark ("a max_instances cap to study n ~ 5-10, the realistic regime"), NOT
        # a reference semantic: the reference's tracker never prunes (track_TF.py:132-165).  n > 0: at most n detections per frame
        # (the best-scoring ones: Fast NMS returns them sorted) and at most n tracked instances per clip (an unmatched detection
        # opens a new track only while the clip holds fewer).  0 = the reference's behaviour.
        self.max_instances = 0

    # -- stage helpers ------------------------------------------------------------------------------------------------
    def _shift_prev(self, P4, T2S, proto, dev):
        """CandidateShift (TF_utils.py:12-51) for every tracked instance of every clip in one chain."""
        net, cfg, prev = self.net, self.cfg, self.prev
        clip_of_row = prev["clip"]
        P4_prev, T2S_prev = self.prev_feat
        P = cfg.correlation_patch_size
        fh, fw = P4.shape[2:]
        box_ref = prev["box"]
        rois = ops.shift_rois(box_ref, clip_of_row, fh, fw)          # (clip, sanitised box in feature-map pixels)
        ptn = getattr(net, "_planar_temporal", None)
        a_prev, a_cur = T2S_prev.permute(0, 2, 3, 1), T2S.permute(0, 2, 3, 1)
        fused = (ptn is not None and ptn.ncorr == P * P and a_prev.is_contiguous() and a_cur.is_contiguous()
                 and 2 * T2S.shape[1] + P * P == ptn.cin)
        if fused:
            # ReLU + concatenation + RoIAlign + channel padding + split in one kernel: the RoI features leave as the planes
            # TemporalNet's first convolution reads (the feature maps are channels_last views of the head's fp32 output; the
            # correlation volume is written channels-last too, so a sample's 121 displacements are 4 cache lines, not 121)
            n = rois.shape[0]
            # (P4 is a channels_last view of the FPN's fp32 output: the correlation kernel reads it in place)
            corr = ops.corr_patch_nhwc(P4_prev, P4, P, scale=1.0 / P4.shape[1], leaky_slope=0.1)
            xp = ops.roi_align_planes(a_prev, a_cur, corr, rois, 7, fmt=ptn.fmt, corr_nhwc=P * P)
            self.timer.toc("tf_corr_roi")
            loc_shift, coeff_shift = ptn.forward_planes(xp, n)
        else:
            corr = ops.corr_patch(P4_prev, P4, P, 1, scale=1.0 / P4.shape[1], leaky_slope=0.1)
            corr = corr.view(P4.shape[0], P * P, P4.shape[2], P4.shape[3])
            feats = F.relu(torch.cat([corr, T2S_prev, T2S], dim=1))
            roi_feats = ops.roi_align(feats, rois, 7)
            self.timer.toc("tf_corr_roi")
            n = roi_feats.shape[0]
        if fused:
            pass
        elif ptn is not None:      # planar convolution: any RoI count, one launch per layer
            loc_shift, coeff_shift = ptn(roi_feats)
        else:
            n_pad = -(-n // ROI_CHUNKS[-1]) * ROI_CHUNKS[-1]
            if n_pad != n:  # rows are independent: zero rows change nothing
                roi_feats = torch.cat([roi_feats, roi_feats.new_zeros(n_pad - n, *roi_feats.shape[1:])], 0)
            # fixed-shape blocks (greedy 256 / 128 / 64): the dense-conv library (MIOpen) selects / builds kernels per
            # shape and the tracked set changes size every frame -- three shapes mean that cost is paid three times
            outs, i = [], 0
            while i < n_pad:
                c = next(c for c in ROI_CHUNKS if c <= n_pad - i)
                outs.append(net.TemporalNet(roi_feats[i:i + c]))
                i += c
            loc_shift = torch.cat([o[0] for o in outs], 0)[:n]
            coeff_shift = torch.cat([o[1] for o in outs], 0)[:n]
        self.timer.toc("tf_temporalnet")
        # decode(loc_shift, center_size(box)) / coeff += shift / score *= 0.95 in one launch, in place on the tracked rows
        ops.shift_apply_(loc_shift, coeff_shift, prev["box"], prev["mask_coeff"], prev["score"], 0.95)
        # masks of the shifted instances on the CURRENT prototypes, and their > 0.5 bits for this step's mask IoU (one pass)
        prev["mask"], self._prev_bits = ops.lincomb_sigmoid_crop_bits(proto, prev["mask_coeff"], prev["box"], clip_of_row)
        self.timer.toc("tf_masks")
        for b in range(self.B):
            self.tracked[b] = [v + 1 for v in self.tracked[b]]

    # -- trunk, eager or from HIP graphs ---------------------------------------------------------------------------------
    # Outputs of frame t-1 (previous-frame features of the temporal fusion) and t are live while t+1 .. t+D are produced; a prefetched trunk that is
    # dropped (the caller changed its mind about the next frames) still used up its slot: 2 D + 2 slots cover D prefetched frames with D drops (the
    # slot of frame f is replayed again by the (2 D + 2)-th trunk after it; at most 2 D + 1 start before step f + 1 has read its features).
    # D = trunks in flight ahead of the current frame under graph replay (eager trunks: 1).  Single stream: depth 1 543 frames/s, 2 777, 3 878, 4 793; 8 clips: 1 401 /
    # 1 491 / 1 488 / 1 454 (profiles/r05_trunk_depth2_ab.txt)
    PREFETCH_DEPTH = int(os.environ.get("STM_PREFETCH_DEPTH", "3"))
    N_GRAPH_SLOTS = 2 * PREFETCH_DEPTH + 2
    # Large batches (round 6): a 32-clip trunk fills the GPU by itself, but not at its two ends (the stem and layer1 ramp up, the small FPN levels, P6 / P7 and the head's
    # last launches run on few workgroups) -- two replayed trunks in flight overlap those: 1 599 (eager, one frame ahead) -> 1 587-1 593 (graphs, depth 1) -> 1 630-1 633
    # (depth 2) -> 1 627-1 628 (depth 3) frames/s at 32 clips, same box, alternating.  Above LARGE_BATCH clips the depth is capped at 2 (a slot's private pool is ~10 GB there).
    LARGE_BATCH = 8

    @property
    def prefetch_depth(self):
        return self.PREFETCH_DEPTH if self.B <= self.LARGE_BATCH else min(self.PREFETCH_DEPTH, 2)

    @property
    def n_graph_slots(self):
        return 2 * self.prefetch_depth + 2

    def _trunk(self, frames):
        """forward_single(frames).  With use_graph the ~110 launches of the trunk (every one a Python -> ctypes call: ~25 us of
        host time each, i.e. more than the GPU needs for them at 1-8 clips) are captured once per slot into a HIP graph and
        replayed: one copy of the frames into the slot's static input + one graph launch per step.  Slots in round-robin,
        because a step still reads the previous frame's P4 / T2S while the next frames' trunks are already running on the side
        streams; a slot's outputs stay valid until it is replayed again, N_GRAPH_SLOTS trunks later.  Every slot has its own memory pool
        and its own workspaces: replays may run CONCURRENTLY on different side streams (a single-frame trunk is a chain of ~155 dependent
        small launches -- 1.57 ms of GPU-side launch latency for half that in work; two chains overlap almost completely:
        profiles/r05_two_trunks_probe.txt)."""
        net = self.net
        if not (self.use_graph and getattr(net, "_planar", None) is not None and not self.timer.on and ops._conv_timing is None
                and ops._im2col_timing is None):
            return net.forward_single(frames)
        planes = getattr(net, "_planar_planes", None)
        if self._graphs and self._graph_planes != planes:
            # another pipeline on the same net fell back to bf16x3 planes (_fall_back swaps the net's inference graph): this pipeline's captured trunks
            # still replay the fp16 graph they were captured from -- drop them (with their private pools) and capture again on the net's current graph
            self._pending = []
            torch.cuda.synchronize()
            self._graphs, self._graph_next, self.graph_active, self._graph_warm, self._graph_ws = [], 0, False, 0, []
        if self._graph_warm < 2:
            # eager first: packs the weights, sizes the workspaces, fills the prior cache, reserves the kernels' LDS
            self._graph_warm += 1
            return net.forward_single(frames)
        if len(self._graphs) < self.n_graph_slots:
            reserved0 = torch.cuda.memory_reserved(frames.device)
            static_in = frames.clone(memory_format=torch.preserve_format)
            graph = torch.cuda.CUDAGraph()
            ws = {}
            self._graph_ws.append(ws)
            cur = torch.cuda.current_stream()
            cap = torch.cuda.Stream(device=frames.device)
            cap.wait_stream(cur)
            # scratch buffers whose addresses the graph bakes in are owned by this pipeline (ops.workspace_scope), not by the
            # capture stream's slot of the global cache
            with ops.workspace_scope(ws):
                with torch.cuda.stream(cap):
                    net.forward_single(static_in)             # once more on the capture stream: sizes this scope's workspaces
                with torch.cuda.graph(graph, stream=cap):
                    out = net.forward_single(static_in)
            cur.wait_stream(cap)
            if not self._graphs:
                # every slot keeps a private pool the size of a trunk's activations (~8 GB at 32 clips of 384x640): before the ring is built, make sure
                # the other 2 D + 1 slots fit beside what the process holds -- else this pipeline keeps the eager trunk (one frame of look-ahead)
                slot_bytes = max(torch.cuda.memory_reserved(frames.device) - reserved0, 0)
                free = torch.cuda.mem_get_info(frames.device)[0] + torch.cuda.memory_reserved(frames.device) - torch.cuda.memory_allocated(frames.device)
                if (self.n_graph_slots - 1) * slot_bytes > 0.9 * free:
                    import sys
                    sys.stderr.write(f"stmask_amd: {self.n_graph_slots} trunk-graph slots of {slot_bytes / 1e9:.1f} GB do not fit in {free / 1e9:.1f} GB of free HBM: "
                                     "this pipeline keeps the eager trunk\n")
                    del graph, out, static_in
                    self._graph_ws.pop()
                    self.use_graph = False
                    torch.cuda.synchronize()
                    torch.cuda.empty_cache()
                    return net.forward_single(frames)
            self._graphs.append((static_in, graph, out))
            self._graph_planes = planes
            self.graph_active = True
        static_in, graph, out = self._graphs[self._graph_next]
        self._graph_next = (self._graph_next + 1) % self.n_graph_slots
        if static_in.shape != frames.shape:
            raise ops.StmError("BatchedClipPipeline: the frame batch changed shape under a captured trunk graph")
        static_in.copy_(frames)
        graph.replay()
        return out

    def _prefetch_trunk(self, next_frames):
        """Enqueue the trunk(s) of the next frame(s) on side streams.  The trunk does not depend on the tracker, and the rest
        of this step is ~200 tiny launches around two host reads (latency-bound: the GPU idles 10-17 % of the step without
        this).  next_frames: the frames of the next call, or a list [next, the one after, ...] -- under graph replay up to PREFETCH_DEPTH of
        them are started (those not in flight yet), rotating over as many side streams, so that the trunk graphs run beside each other and
        beside this step's tracker tail; eager trunks (large batches fill the GPU by themselves) keep one frame of look-ahead.  A side stream
        waits for everything enqueued on the main stream so far."""
        if next_frames is None or self.timer.on:
            return
        nxt = list(next_frames) if isinstance(next_frames, (list, tuple)) else [next_frames]
        depth = self.prefetch_depth if (self.use_graph and self.graph_active) else 1
        main = torch.cuda.current_stream()
        for f in nxt[:depth]:
            if f is None or any(p[0] is f for p in self._pending):
                continue
            if not self._sides:
                self._sides = concurrent_side_streams(f.device, trunk_stream_count())
            side = self._sides[self._side_next]
            self._side_next = (self._side_next + 1) % len(self._sides)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                out = self._trunk(f)
                ev = torch.cuda.Event()
                ev.record()
            self._pending.append((f, out, ev))

    def _take_trunk(self, frames):
        """(fpn_outs, pred) of `frames`: the trunk started for them on a side stream by an earlier step, or a fresh one.  Third value: a
        prefetched trunk of OTHER frames was dropped."""
        net = self.net
        dropped = False
        while self._pending and self._pending[0][0] is not frames:
            torch.cuda.current_stream().wait_event(self._pending.pop(0)[2])   # a trunk nobody asked for: let it finish, drop it
            dropped = True
        if self._pending:
            _, (fpn_outs, pred), ev = self._pending.pop(0)
            torch.cuda.current_stream().wait_event(ev)
            if not self.graph_active:                            # (graph outputs live in the graphs' own pools)
                for t_ in list(pred.values()) + list(fpn_outs):  # allocated on the side stream, consumed on this one
                    if torch.is_tensor(t_):
                        t_.record_stream(torch.cuda.current_stream())
                t2s_ = pred["T2S_feat"][net.correlation_selected_layer] if isinstance(pred.get("T2S_feat"), (list, tuple)) else None
                if torch.is_tensor(t2s_):
                    t2s_.record_stream(torch.cuda.current_stream())
        else:
            fpn_outs, pred = self._trunk(frames)
        return fpn_outs, pred, dropped

    def _detect(self, pred):
        """Decode + confidence threshold + Fast NMS for every frame of the batch, no host sync -> (prior_idx [B, cap], cls, score, box, count [B]).
        Cross-class Fast NMS (the default, detection_TF.py:85-134: cap = nms_top_k) or -- Detect_TF.use_cross_class_nms = False, the reference's
        per-class variant (detection_TF.py:136-204, README "mAP*" column: cap = max_num_detections) -- one launch pair for all frames."""
        cfg, net = self.cfg, self.net
        priors = pred["priors"].squeeze(0)
        det = net.Detect_TF if self.tf else net.detect
        if getattr(det, "use_cross_class_nms", True):
            # (the softmax of STMask.py:314 is taken per row inside the candidate pass: no pass over the [B, N, 41] logits of its own)
            return ops.detect_cc(pred["loc"], priors, pred["conf"], pred["centerness"], cfg.eval_conf_thresh, cfg.nms_thresh, cfg.nms_top_k, logits=True)
        # (per-class ranking: conf * centerness in Detect_TF.fast_nms, detection_TF.py:139-141; the raw confidences in the non-TF Detect.fast_nms,
        # detection.py:211-212)
        return ops.detect_pc(pred["loc"], priors, F.softmax(pred["conf"], -1), pred["centerness"] if self.tf else None, cfg.eval_conf_thresh, cfg.nms_thresh,
                             cfg.nms_top_k, cfg.max_num_detections)

    @torch.no_grad()
    def step(self, frames, is_first=None, next_frames=None):
        """frames [B,3,H,W] -> packed detections [B, top_k, 40] (stmask_amd.dist layout) without a final sync, plus the
        per-clip tracked-instance counts (host ints).  next_frames (optional): the frames the NEXT call will be given -- or a list: those of the
        next call, of the one after it, ... (the same tensor OBJECTS the later calls pass as `frames`); their trunks are started on side streams
        while this step's tracker logic runs (_prefetch_trunk).

        fp16 plane graphs carry |activation| <= 65504 only; their producers raise a sticky device flag beyond that, which arrives with the step's
        first host read (ops.RangeError).  The step is then NOT lost: the tracker state it had touched is put back, the inference graph is rebuilt
        with bf16x3 planes (fp32's range; weights repacked from the same modules, in-process), the step is repeated on it and the pipeline stays
        there (`fell_back`; logged once on stderr).  range_fallback = False restores the raise."""
        first = (self.t == 0) if is_first is None else is_first
        if first:
            self.prev, self.prev_n, self.prev_feat = None, [0] * self.B, None
            self.tracked = [[] for _ in range(self.B)]
        snap = self._snapshot() if (self.range_fallback and self._range_guarded()) else None
        try:
            return self._step(frames, first, next_frames)
        except ops.RangeError:
            if snap is None:
                raise
            self._fall_back(snap)
            return self._step(frames, first, next_frames)

    def _range_guarded(self):
        g = getattr(self.net, "_planar", None)
        return g is not None and getattr(self.net, "_planar_planes", "bf16x3") != "bf16x3"

    def _snapshot(self):
        """What a step mutates IN PLACE before its first host read (CandidateShift's decode / coefficient shift / score decay on the tracked rows)
        plus the host-side counters: enough to repeat the step."""
        prev = self.prev
        rows = None
        if self.tf and prev is not None and sum(self.prev_n):
            rows = tuple(prev[k].clone() for k in ("box", "mask_coeff", "score"))
        return rows, [list(t) for t in self.tracked], self.t

    def _fall_back(self, snap):
        import sys
        from . import fuse
        net = self.net
        self._pending = []
        torch.cuda.synchronize()                      # nothing enqueued on the fp16 graph may raise the flag after it has been cleared
        flag = ops._range_flags.get(torch.cuda.current_device())
        if flag is not None:
            flag.zero_()
        sys.stderr.write(f"stmask_amd: step {self.t}: an activation left the range of the {getattr(net, '_planar_planes', 'fp16')} planar format (|x| > 65504); "
                         "rebuilding the inference graph with bf16x3 planes, repeating the step and staying there (half the convolution rate)\n")
        if getattr(net, "_planar_bf16x3", None) is None:
            net._planar_bf16x3 = fuse.build_planar(net, "bf16x3")
        fuse.attach_planar(net, net._planar_bf16x3)
        self._graphs, self._graph_next, self.graph_active, self._graph_warm, self._graph_ws = [], 0, False, 0, []
        rows, tracked, t = snap
        if rows is not None:
            for k, v in zip(("box", "mask_coeff", "score"), rows):
                self.prev[k].copy_(v)
        self.tracked, self.t = tracked, t
        self.fell_back = True

    def _step(self, frames, first, next_frames):
        net, cfg, B = self.net, self.cfg, self.B
        dev = frames.device
        tmr = self.timer
        tmr.tic()
        if getattr(net, "_planar", None) is not None and tmr.on:
            net._planar.timer = tmr      # finer stages inside the trunk
        if not self.tf:
            return self._step_nontf(frames, first, next_frames)
        fpn_outs, pred, dropped = self._take_trunk(frames)
        tmr.toc("trunk")
        # A dropped prefetch under graph replay leaves the round-robin one slot ahead: the slot the NEXT replay overwrites is then
        # the one holding the previous frame's P4 / T2S, which CandidateShift below still reads -- so in that step the next trunk
        # must not start before _shift_prev is enqueued (the late position), whatever prefetch_early says.
        if self.prefetch_early and not (dropped and self.graph_active):
            # start the next trunk right away: it then also shares the GPU with this step's temporal-fusion convolutions
            # (more throughput, but kernels of the two streams stretch each other: per-kernel timings stop being clean)
            self._prefetch_trunk(next_frames)
            next_frames = None
        P4 = fpn_outs[net.correlation_selected_layer]
        T2S = pred["T2S_feat"][net.correlation_selected_layer]
        proto = pred["proto"]
        # CandidateShift of the tracked set needs this frame's features but not its detections: enqueue it first, then
        # start the next frame's trunk on the second stream -- everything that follows in this step (detection, two host
        # reads, matching, tracker update: ~200 tiny launches) then runs beside that trunk, while the two big kernel groups
        # (temporal-fusion convolutions, trunk) never share the GPU
        Pn = sum(self.prev_n) if self.prev is not None else 0
        if Pn:
            self._shift_prev(P4, T2S, proto, dev)
        self._prefetch_trunk(next_frames)
        idx, cls, score, box, cnt = self._detect(pred)
        if self.max_instances > 0:
            cnt = torch.clamp(cnt, max=self.max_instances)
        counts, host_scores = ops.counts_to_host(cnt, extra=score)  # host read 1: B counts + the fp16 range flag + the NMS scores
        tmr.toc("detect")
        D = sum(counts)
        top_k = idx.shape[1]                              # slots per frame of the detector's outputs (nms_top_k, or max_num_detections per class-wise NMS)
        # ---- detections of all clips, concatenated (rows sorted by clip): one gather kernel ------------------------------
        det = ops.gather_detections(idx, cls, score, box, cnt, pred["mask_coeff"], pred["track"], pred["centerness"], D)
        if D:
            det["mask"], det_bits = ops.lincomb_sigmoid_crop_bits(proto, det["mask_coeff"], det["box"], det["clip"])
        else:
            det["mask"], det_bits = proto.new_zeros(0, proto.shape[1], proto.shape[2]), None
        tmr.toc("det_gather_masks")
        det_scores = [float(host_scores[b * top_k + j]) for b in range(B) for j in range(counts[b])]   # row order of det

        if self.prev is None:
            # first frame of every clip (track_TF.py:88-93): the detections become the tracked set
            self.prev, self._bits = det, det_bits
            self.prev_n = list(counts)
            self.tracked = [[0] * k for k in counts]
            self._upload_meta(dev, None)
        else:
            prev = self.prev
            if D and Pn:
                # matching scores for all clips at once; pairs from different clips can never match
                miou = ops.mask_iou_bits(det_bits, self._prev_bits, proto.shape[1] * proto.shape[2], group1=det["clip"],
                                         group2=prev["clip"])                               # same-clip pairs only
                # (the embedding dot products of the same-clip pairs are taken inside the kernel)
                match = ops.match_scores_embed(det["track"], prev["track"], miou, det["box"], prev["box"], det["score"], det["class"],
                                               prev["class"], det["clip"], self._off_dev, cfg.match_coeff, 0.3)
                ids = match.tolist()  # host read 2
                scores = det_scores
                tmr.toc("match_scores")
            else:
                ids, scores = [0] * D, [0.0] * D
            # greedy resolution (track_TF.py:132-156) per clip on host scalars -> one gather plan for all clips
            plan, new_n, new_tracked = [], [], []
            p0 = d0 = 0
            cap = self.max_instances
            for b in range(B):
                pn, dn = self.prev_n[b], counts[b]
                src = list(range(p0, p0 + pn))
                tm = list(self.tracked[b])
                best = [-1.0] * pn
                for j in range(dn):
                    mid = ids[d0 + j]
                    if mid == 0:
                        if cap and len(src) >= cap:
                            continue                     # benchmark-only cap on the tracked set (see max_instances)
                        src.append(Pn + d0 + j)
                        tm.append(0)
                    else:
                        obj = mid - 1 - p0
                        if scores[d0 + j] > best[obj]:
                            best[obj] = scores[d0 + j]
                            src[obj] = Pn + d0 + j
                            tm[obj] = 0
                plan += src
                new_n.append(len(src))
                new_tracked.append(tm)
                p0 += pn
                d0 += dn
            self.prev_n, self.tracked = new_n, new_tracked
            plan_dev = self._upload_meta(dev, plan if D else None)
            if D:
                # prev <- cat(prev, det)[plan] for every row tensor but the soft masks (see _TrackedRows); the masks' bit words ride
                # along instead: the keep rule of _pack_outputs counts pixels on them
                keys = tuple(k for k in _ROW_KEYS if k != "mask") + ("clip",)
                pbits = self._prev_bits if Pn else det_bits[:0]
                rows = ops.gather_rows2([prev[k] for k in keys] + [pbits], [det[k] for k in keys] + [det_bits], plan_dev, Pn)
                merged = _TrackedRows()
                for k, t in zip(keys, rows[:-1]):
                    merged[k] = t
                merged.defer_mask(prev["mask"], det["mask"], plan_dev, Pn)
                self.prev, self._bits = merged, rows[-1]
            else:
                self._bits = self._prev_bits if Pn else None
            tmr.toc("tracker_update")
        self.prev_feat = (P4, T2S)
        self.t += 1
        out = self._pack_outputs(dev)
        tmr.toc("pack")
        return out

    def _step_nontf(self, frames, first, next_frames):
        """One frame of every clip through Detect + Track (reference detection.py:98-137, track.py:56-179; STMask.py:323-325): the frame's own
        detections leave with their object ids; the tracker keeps BINARY masks (their bit words here) and replaces a matched object's row only while
        (mask_ious > 0.3).sum() < 2 (track.py:162).  All clips per launch; two host reads per step as the temporal-fusion path."""
        from .dist import DET_COLS
        net, cfg, B = self.net, self.cfg, self.B
        dev = frames.device
        if first:
            self.prev, self.prev_n, self._bits = None, [0] * B, None
        fpn_outs, pred, _ = self._take_trunk(frames)
        self._prefetch_trunk(next_frames)
        proto = pred["proto"]
        mc = torch.tanh(pred["mask_coeff"])                               # STMask.py:324 (generate_mask applies tanh AGAIN on this path: reproduced)
        idx, cls, score, box, cnt = self._detect(pred)
        counts, host_scores = ops.counts_to_host(cnt, extra=score)       # host read 1
        D, cap = sum(counts), idx.shape[1]
        det = ops.gather_detections(idx, cls, score, box, cnt, mc, pred["track"], pred["centerness"], D)
        if not cfg.train_track:
            det["track"] = F.normalize(det["mask_coeff"], dim=1)
        out = torch.zeros(B, cfg.nms_top_k, DET_COLS, device=dev)
        if D == 0:
            self._last = None
            self.t += 1
            return out
        det_mask, det_bits = ops.lincomb_sigmoid_crop_bits(proto, det["mask_coeff"], det["box"], det["clip"])
        det_scores = [float(host_scores[b * cap + j]) for b in range(B) for j in range(counts[b])]
        Pn = sum(self.prev_n)
        if Pn:
            prev = self.prev
            self._upload_offsets(dev)
            miou = ops.mask_iou_bits(det_bits, self._bits, proto.shape[1] * proto.shape[2], group1=det["clip"], group2=prev["clip"])
            match = ops.match_scores_embed(det["track"], prev["track"], miou, det["box"], prev["box"], det["score"], det["class"], prev["class"],
                                           det["clip"], self._off_dev, cfg.match_coeff, 0.3)
            host = torch.stack([match, (miou > 0.3).sum(1).to(torch.int32)]).tolist()   # host read 2: match ids + the update gate's counts
            ids, n_over = host
        else:
            ids, n_over = [0] * D, [0] * D
        plan, new_n, obj_ids = [], [], [-1] * D
        p0 = d0 = 0
        for b in range(B):
            pn, dn = self.prev_n[b], counts[b]
            src = list(range(p0, p0 + pn))
            if pn == 0:
                # (track.py:92-97: the first frame with detections -- they become the objects)
                for j in range(dn):
                    obj_ids[d0 + j] = j
                    src.append(Pn + d0 + j)
            else:
                best_score, best_idx = [-1.0] * pn, [-1] * pn
                for j in range(dn):
                    mid = ids[d0 + j]
                    if mid == 0:
                        obj_ids[d0 + j] = len(src)
                        src.append(Pn + d0 + j)
                    else:
                        obj = mid - 1 - p0
                        if det_scores[d0 + j] > best_score[obj]:
                            if best_idx[obj] != -1:
                                obj_ids[d0 + best_idx[obj]] = -1
                            obj_ids[d0 + j] = obj
                            best_score[obj], best_idx[obj] = det_scores[d0 + j], j
                            if n_over[d0 + j] < 2:                        # track.py:162
                                src[obj] = Pn + d0 + j
            plan += src
            new_n.append(len(src))
            p0 += pn
            d0 += dn
        # output rows: the frame's detections with an object id (remove_false_inst, track.py:172-179), in detection order
        rows, dst_b, dst_j = [], [], []
        d0 = 0
        for b in range(B):
            j_out = 0
            for j in range(counts[b]):
                if obj_ids[d0 + j] >= 0 or not cfg.remove_false_inst:
                    rows.append(d0 + j); dst_b.append(b); dst_j.append(j_out)
                    j_out += 1
            d0 += counts[b]
        meta = torch.tensor(plan + rows + dst_b + dst_j + [obj_ids[r] for r in rows], dtype=torch.int32).to(dev, non_blocking=True)
        nP, nR = len(plan), len(rows)
        plan_dev = meta[:nP]
        r_dev, b_dev, j_dev, id_dev = (meta[nP + k * nR:nP + (k + 1) * nR].long() for k in range(4))
        keys = ("box", "mask_coeff", "track", "class", "score", "clip")
        pbits = self._bits if Pn else det_bits[:0]
        a_rows = [self.prev[k] for k in keys] if Pn else [det[k][:0] for k in keys]
        merged = ops.gather_rows2(a_rows + [pbits], [det[k] for k in keys] + [det_bits], plan_dev, Pn)
        self.prev = dict(zip(keys, merged[:-1]))
        self._bits = merged[-1]
        self.prev_n = new_n
        if nR:
            out[b_dev, j_dev, 0:4] = det["box"][r_dev]
            out[b_dev, j_dev, 4] = det["score"][r_dev]
            out[b_dev, j_dev, 5] = det["class"][r_dev].float()
            out[b_dev, j_dev, 6] = id_dev.float()
            out[b_dev, j_dev, 7] = 1.0
            out[b_dev, j_dev, 8:8 + det["mask_coeff"].shape[1]] = det["mask_coeff"][r_dev]
        self._last = (det, det_mask, r_dev, b_dev, id_dev)
        self.t += 1
        return out

    def _upload_offsets(self, dev):
        off = [0]
        for n in self.prev_n:
            off.append(off[-1] + n)
        self._off_dev = torch.tensor(off, dtype=torch.int32).to(dev, non_blocking=True)

    def _upload_meta(self, dev, plan):
        """One host -> device copy per step: [clip row offsets (B + 1) | frames-since-last-match counters | gather plan]."""
        off = [0]
        for n in self.prev_n:
            off.append(off[-1] + n)
        tm = [v for t in self.tracked for v in t]
        meta = torch.tensor(off + tm + (plan or []), dtype=torch.int32).to(dev, non_blocking=True)
        nb = len(off)
        self._off_dev, self._tm_dev = meta[:nb], meta[nb:nb + len(tm)]
        return meta[nb + len(tm):]

    def _pack_outputs(self, dev):
        """keep rule of track_TF.py:158-165 on device, scattered into [B, top_k, 40] without a host sync (two launches)."""
        from .dist import DET_COLS
        cfg, B, prev = self.cfg, self.B, self.prev
        if prev is None or sum(self.prev_n) == 0:
            return torch.zeros(B, cfg.nms_top_k, DET_COLS, device=dev)
        return ops.pack_tracked_bits(self._bits, prev["score"], self._tm_dev, self._off_dev, prev["box"], prev["class"], prev["mask_coeff"], B,
                                     cfg.nms_top_k, DET_COLS, 10, cfg.eval_conf_thresh)

    def detections(self):
        """Reference-shaped per-clip detection dicts of the last step (host sync; for tests and users who want them)."""
        cfg, prev = self.cfg, self.prev
        outs = []
        if not self.tf:
            # the frame's own detections with their object ids; masks binary (track.py:88)
            if self._last is None:
                return [{} for _ in range(self.B)]
            det, det_mask, r_dev, b_dev, id_dev = self._last
            for b in range(self.B):
                sel = r_dev[b_dev == b]
                d = {k: det[k].index_select(0, sel) for k in ("box", "mask_coeff", "track", "class", "score")}
                d["mask"] = det_mask.index_select(0, sel).gt(0.5).float()
                d["box_ids"] = id_dev[b_dev == b]
                outs.append(d)
            return outs
        if prev is None:
            return [{} for _ in range(self.B)]
        dev = prev["box"].device
        tm = torch.tensor([v for t in self.tracked for v in t], device=dev)
        keep = (tm <= 10) & (prev["mask"].gt(0.5).sum([1, 2]) > 1) & (prev["score"] > cfg.eval_conf_thresh)
        p0 = 0
        for b in range(self.B):
            n = self.prev_n[b]
            k = torch.nonzero(keep[p0:p0 + n]).view(-1)
            d = {key: prev[key][p0:p0 + n].index_select(0, k) for key in _ROW_KEYS}
            d["box_ids"] = k
            outs.append(d)
            p0 += n
        return outs
