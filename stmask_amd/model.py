"""STMask network (eval path) -- host-side mirror of the reference's STMask.py:19-329 with identical module names and
state-dict keys, built from the MI355X layers in this package.

``forward(x, img_meta)`` follows the reference's eval branch (STMask.py:310-329):
forward_single -> softmax -> generate_candidate -> Detect_TF -> Track_TF (temporal fusion configs) or
tanh -> Detect -> Track.  Training (STMask.py:285-309) is outside the hot path.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .backbone import construct_backbone
from .config import cfg as _default_cfg
from .layers import FPN, Detect, Detect_TF, PredictionModule_FC, TemporalNet, Track, Track_TF, generate_candidate, \
    make_net


class STMask(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        cfg = (cfg or _default_cfg).copy()
        self.cfg = cfg
        self.backbone = construct_backbone(cfg)
        nf = cfg.fpn_num_features
        self.proto_src = cfg.mask_proto_src
        self.proto_net, cfg.mask_dim = make_net(nf, cfg.mask_proto_net, include_last_relu=False)
        self.backbone_selected = list(cfg.selected_layers)
        self.fpn = FPN([self.backbone.channels[i] for i in self.backbone_selected], cfg=cfg)
        self.selected_layers = list(range(len(self.backbone_selected) + cfg.fpn_num_downsample))
        cfg.num_heads = len(self.selected_layers)

        self.prediction_layers = nn.ModuleList()
        for idx in self.selected_layers:
            parent = self.prediction_layers[0] if (cfg.share_prediction_module and idx > 0) else None
            self.prediction_layers.append(PredictionModule_FC(nf, nf, deform_groups=1,
                                                              pred_aspect_ratios=cfg.pred_aspect_ratios[idx],
                                                              pred_scales=cfg.pred_scales[idx], parent=parent, cfg=cfg))
        if cfg.temporal_fusion_module:
            corr_channels = 2 * nf + cfg.correlation_patch_size ** 2
            self.TemporalNet = TemporalNet(corr_channels, cfg.mask_proto_n)
            self.correlation_selected_layer = cfg.correlation_selected_layer
            self.Detect_TF = Detect_TF(cfg.num_classes, bkg_label=0, top_k=cfg.nms_top_k,
                                       conf_thresh=cfg.nms_conf_thresh, nms_thresh=cfg.nms_thresh, cfg=cfg)
            self.Track_TF = Track_TF(cfg=cfg)
        self.detect = Detect(cfg.num_classes, bkg_label=0, top_k=cfg.nms_top_k, conf_thresh=cfg.nms_conf_thresh,
                             nms_thresh=cfg.nms_thresh, cfg=cfg)
        self.Track = Track(cfg=cfg)

    # -- checkpoints (reference STMask.py:127-155) ------------------------------------------------------------------
    def save_weights(self, path):
        torch.save(self.state_dict(), path)

    def load_weights(self, path):
        state = torch.load(path, map_location="cpu")
        for key in list(state.keys()):
            if key.startswith("backbone.layer") and not key.startswith("backbone.layers"):
                del state[key]
            elif key.startswith("fpn.downsample_layers.") and int(key.split(".")[2]) >= self.cfg.fpn_num_downsample:
                del state[key]
        own = self.state_dict()
        own.update({k: v for k, v in state.items() if k in own})
        self.load_state_dict(own)

    # -- trunk + heads (reference STMask.py:205-282) ----------------------------------------------------------------
    def forward_single(self, x):
        planes = None      # planar form of the selected backbone outputs (PlanarBackbone only)
        if getattr(self, "_planar_backbone", None) is not None:
            bb_outs = self._planar_backbone(x)
            planes = [self._planar_backbone.out_planes[i] for i in self.backbone_selected]
        else:
            bb_outs = self.backbone(x)
        if getattr(self, "_planar", None) is not None:     # fuse.optimize_for_inference(net, planar=True)
            return self._planar.run([bb_outs[i] for i in self.backbone_selected], planes=planes)
        fpn_outs = self.fpn([bb_outs[i] for i in self.backbone_selected])
        proto = F.relu(self.proto_net(fpn_outs[self.proto_src]))
        proto = proto.permute(0, 2, 3, 1).contiguous()
        keys = ("mask_coeff", "priors", "loc", "T2S_feat", "centerness", "conf", "track")
        pred = {k: [] for k in keys}
        for idx, layer in zip(self.selected_layers, self.prediction_layers):
            p = layer(fpn_outs[idx])
            for k in keys:
                pred[k].append(p[k])
        for k in keys:
            if k != "T2S_feat":
                pred[k] = torch.cat(pred[k], 1)
        pred["proto"] = proto
        return fpn_outs, pred

    def forward(self, x, img_meta=None):
        if self.training:
            raise NotImplementedError("training is outside the MI355X hot path (SURVEY.md §2 row 18)")
        cfg = self.cfg
        fpn_outs, pred = self.forward_single(x)
        pred["conf"] = F.softmax(pred["conf"], -1)
        if cfg.temporal_fusion_module:
            pred["fpn_feat"] = fpn_outs[self.correlation_selected_layer]
            pred["T2S_feat"] = pred["T2S_feat"][self.correlation_selected_layer]
            candidates = generate_candidate(pred, cfg=cfg)
            after_nms = self.Detect_TF(self, candidates, is_output_candidate=True)
            return self.Track_TF(self, after_nms, img_meta, imgs=x)
        pred["mask_coeff"] = torch.tanh(pred["mask_coeff"])
        return self.Track(self.detect(pred, self), img_meta)
