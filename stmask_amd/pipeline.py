"""Batched clip pipeline: many independent video clips advance one frame per step on one GPU.

forward_single has no cross-frame dependence, so frame t of all local clips goes through the trunk
and heads as ONE batch; only the temporal-fusion / tracker stage is per clip (previous-frame features and tracker
state, track_TF.py:52-54,96-100).  This is the unit bench.py times and dist.py shards over GPUs.
"""
import torch
import torch.nn.functional as F

from .layers import Track_TF, generate_candidate


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
