"""nn.Modules of the hot path -- host-side mirror of the reference's layers/modules/{FPN,make_net,prediction_head_FC,
Featurealign,track_to_segment_head}.py with identical module / parameter names (state-dict compatible, SURVEY.md
Appendix B).  Dense convs, bilinear interpolation and activations stay torch (MIOpen); deformable convs, correlation
and RoIAlign are the hand-written kernels.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..config import cfg as _default_cfg
from ..mmcv_ops import DeformConv2d, roi_align
from .box_utils import sanitize_coordinates_hw


class InterpolateModule(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.args, self.kwargs = args, kwargs

    def forward(self, x):
        return F.interpolate(x, *self.args, **self.kwargs)


def make_net(in_channels, conf, include_last_relu=True):
    """(channels, kernel, kwargs) list -> nn.Sequential (reference make_net.py:5-59).  kernel < 0 with channels None is a
    bilinear x(-kernel) upsample; every layer is followed by ReLU (index layout matters for checkpoint keys:
    proto_net.{0,2,4,8,10})."""
    layers = []
    for ch, k, kw in conf:
        if k > 0:
            layers.append(nn.Conv2d(in_channels, ch, k, **kw))
            in_channels = ch
        elif ch is None:
            layers.append(InterpolateModule(scale_factor=-k, mode="bilinear", align_corners=False, **kw))
        else:
            layers.append(nn.ConvTranspose2d(in_channels, ch, -k, **kw))
            in_channels = ch
        layers.append(nn.ReLU(inplace=True))
    if not include_last_relu:
        layers = layers[:-1]
    return nn.Sequential(*layers), in_channels


class FPN(nn.Module):
    """Reference FPN.py:40-108.  lat_layers are stored in REVERSE level order (checkpoint compatibility)."""

    def __init__(self, in_channels, cfg=None):
        super().__init__()
        cfg = cfg or _default_cfg
        nf = cfg.fpn_num_features
        self.lat_layers = nn.ModuleList([nn.Conv2d(c, nf, kernel_size=1) for c in reversed(in_channels)])
        self.pred_layers = nn.ModuleList([nn.Conv2d(nf, nf, kernel_size=3, padding=1 if cfg.fpn_pad else 0)
                                          for _ in in_channels])
        self.downsample_layers = nn.ModuleList([nn.Conv2d(nf, nf, kernel_size=3, padding=1, stride=2)
                                                for _ in range(cfg.fpn_num_downsample)])
        self.interpolation_mode = cfg.fpn_interpolation_mode
        self.pred_relu_fused = False  # set by fuse.optimize_for_inference: the pred conv applies bias + ReLU itself

    def forward(self, convouts):
        n = len(convouts)
        out = [None] * n
        x = None
        for i, lat in enumerate(self.lat_layers):
            j = n - 1 - i
            lateral = lat(convouts[j])
            if x is None:
                x = lateral  # reference adds to torch.zeros(1): identical values
            else:
                h, w = convouts[j].shape[2:]
                x = F.interpolate(x, size=(h, w), mode=self.interpolation_mode, align_corners=False) + lateral
            out[j] = x
        for i, pred in enumerate(self.pred_layers):
            j = n - 1 - i
            out[j] = pred(out[j]) if self.pred_relu_fused else F.relu(pred(out[j]))
        for ds in self.downsample_layers:
            out.append(ds(out[-1]))
        return out


class FeatureAlign(nn.Module):
    """FCB head (reference Featurealign.py:6-74): box regression -> sampling offsets -> DeformConv2d -> ReLU -> conv."""

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), deformable_groups=4, use_pred_offset=True):
        super().__init__()
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.kernel_size = ks
        self.padding = ((ks[0] - 1) // 2, (ks[1] - 1) // 2)
        self.use_pred_offset = use_pred_offset
        if use_pred_offset:
            self.conv_offset = nn.Conv2d(4, deformable_groups * ks[0] * ks[1] * 2, 1, bias=False)
        self.conv_adaption = DeformConv2d(in_channels, in_channels, kernel_size=ks, padding=self.padding,
                                          deform_groups=deformable_groups)
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=ks, padding=self.padding)

    def forward(self, x, shape):
        if self.use_pred_offset:
            offset = self.conv_offset(shape)
        else:
            offset = ops.fcb_ali_offsets(shape, self.kernel_size[0], self.kernel_size[1])
        x = ops.deform_conv(x, offset, None, self.conv_adaption.weight, None, 1, self.padding, 1,
                            self.conv_adaption.deform_groups, relu=True)  # ReLU fused into the GEMM epilogue
        return self.conv(x)


class PredictionModule_FC(nn.Module):
    """FCA prediction head shared over the 5 FPN levels (reference prediction_head_FC.py:12-247): three kernel shapes
    3x3 / 3x5 / 5x3 <-> three anchor shapes."""

    _prior_cache = {}

    def __init__(self, in_channels, out_channels=1024, deform_groups=1, pred_aspect_ratios=None, pred_scales=None,
                 parent=None, cfg=None):
        super().__init__()
        cfg = cfg or _default_cfg
        self.cfg_ = cfg
        self.num_classes, self.mask_dim, self.embed_dim = cfg.num_classes, cfg.mask_dim, cfg.embed_dim
        self.num_priors = len(pred_scales)
        self.pred_aspect_ratios, self.pred_scales = pred_aspect_ratios, pred_scales
        self.parent = [parent]  # list: keeps the parent out of the state dict
        if parent is not None:
            return
        self.upfeature, self.out_channels = make_net(in_channels, cfg.extra_head_net)
        oc, npri = self.out_channels, self.num_priors
        self.bbox_layer, self.track_layer, self.mask_layer = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.centerness_layer, self.conf_layer = nn.ModuleList(), nn.ModuleList()

        def head(out_ch, params, fcb):
            if fcb:
                return FeatureAlign(oc, out_ch, kernel_size=params["kernel_size"], deformable_groups=deform_groups,
                                    use_pred_offset=cfg.use_pred_offset)
            return nn.Conv2d(oc, out_ch, **params)

        for params in cfg.head_layer_params:
            self.centerness_layer.append(nn.Conv2d(oc, npri, **params))
            self.bbox_layer.append(nn.Conv2d(oc, npri * 4, **params))
            self.conf_layer.append(head(npri * self.num_classes, params, cfg.use_dcn_class))
            self.track_layer.append(head(npri * self.embed_dim, params, cfg.use_dcn_track))
            self.mask_layer.append(head(npri * self.mask_dim, params, cfg.use_dcn_mask))

        def extra(n):
            return nn.Sequential(*sum([[nn.Conv2d(oc, oc, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                                       for _ in range(n)], []))

        self.track_extra = extra(cfg.extra_layers[2])
        self.conf_extra = extra(cfg.extra_layers[0])
        self.bbox_extra, self.mask_extra = extra(cfg.extra_layers[0]), extra(cfg.extra_layers[1])

    def forward(self, x):
        src = self if self.parent[0] is None else self.parent[0]
        cfg = src.cfg_
        B = x.size(0)
        x = src.upfeature(x)
        conf_x, bbox_x, mask_x, track_x = src.conf_extra(x), src.bbox_extra(x), src.mask_extra(x), src.track_extra(x)
        conf, bbox, cen, mask, track = [], [], [], [], []

        def nhwc(t):
            return t.permute(0, 2, 3, 1).contiguous()

        for k in range(len(cfg.head_layer_params)):
            cen.append(nhwc(src.centerness_layer[k](bbox_x)))
            bbox_cur = src.bbox_layer[k](bbox_x)
            bbox.append(nhwc(bbox_cur))
            conf.append(nhwc(src.conf_layer[k](conf_x, bbox_cur) if cfg.use_dcn_class else src.conf_layer[k](conf_x)))
            track.append(nhwc(src.track_layer[k](track_x, bbox_cur) if cfg.use_dcn_track else src.track_layer[k](track_x)))
            mask.append(nhwc(src.mask_layer[k](mask_x, bbox_cur) if cfg.use_dcn_mask else src.mask_layer[k](mask_x)))

        preds = {
            "mask_coeff": torch.cat(mask, dim=-1).view(B, -1, src.mask_dim),
            "priors": self.make_priors(x.size(2), x.size(3), x.device),
            "loc": torch.cat(bbox, dim=-1).view(B, -1, 4),
            # NB: the reference concatenates centerness along dim=1 (H), not the channel dim
            # (prediction_head_FC.py:189) -- reproduced as is
            "centerness": torch.tanh(torch.cat(cen, dim=1).view(B, -1, 1)),
            "T2S_feat": x,
            "conf": torch.cat(conf, dim=-1).view(B, -1, src.num_classes),
            "track": F.normalize(torch.cat(track, dim=-1).view(B, -1, src.embed_dim), dim=-1),
        }
        return preds

    def make_priors(self, conv_h, conv_w, device):
        """Anchors (cx, cy, w, h) in Python float64, rounded to fp32 once (reference :224-247), cached per
        (h, w, device) -- the reference rebuilds them with a Python triple loop every frame (13-23 ms)."""
        key = (conv_h, conv_w, str(device), str(self.pred_aspect_ratios), str(self.pred_scales))
        hit = PredictionModule_FC._prior_cache.get(key)
        if hit is not None:
            return hit
        data = []
        for j in range(conv_h):
            for i in range(conv_w):
                x = (i + 0.5) / conv_w
                y = (j + 0.5) / conv_h
                for ars in self.pred_aspect_ratios:
                    for arh, arw in ars:
                        for scale in self.pred_scales:
                            ratio = scale / self.pred_scales[0]
                            data += [x, y, ratio * arw / conv_w, ratio * arh / conv_h]
        priors = torch.tensor(data, dtype=torch.float64).to(torch.float32).view(1, -1, 4).to(device)
        PredictionModule_FC._prior_cache[key] = priors
        return priors


class TemporalNet(nn.Module):
    """Reference track_to_segment_head.py:10-37: 3x(3x3 conv + ReLU) on 7x7 RoI features -> avg-pool -> (dbox, dcoeff)."""

    def __init__(self, corr_channels, mask_proto_n=32):
        super().__init__()
        self.conv1 = nn.Conv2d(corr_channels, 512, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(512, 512, kernel_size=3, padding=1)
        self.conv3 = nn.Conv2d(512, 1024, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.pool = nn.AvgPool2d((7, 7), stride=1)
        self.fc = nn.Linear(1024, 4)
        self.fc_coeff = nn.Linear(1024, mask_proto_n)

    def forward(self, x):
        x = self.relu(self.conv1(x))
        x = self.relu(self.conv2(x))
        x = self.relu(self.conv3(x))
        x = self.pool(x).flatten(1)
        return self.fc(x), self.fc_coeff(x)


def correlate(x1, x2, patch_size=11, dilation_patch=1):
    """Reference track_to_segment_head.py:40-62: patch correlation, / C, leaky_relu(0.1) -> [B, P*P, H, W].
    The division and the activation are fused into the correlation kernel's epilogue."""
    b, c, h, w = x1.shape
    out = ops.corr_patch(x1, x2, patch_size, dilation_patch, scale=1.0 / c, leaky_slope=0.1)
    return out.view(b, patch_size * patch_size, h, w)


def bbox_feat_extractor(feature_maps, boxes_w_norm, h, w, pool_size):
    """Reference track_to_segment_head.py:65-88: relative boxes -> feature pixels -> RoIAlign(pool_size)."""
    boxes = sanitize_coordinates_hw(boxes_w_norm, h, w)
    if feature_maps.dim() == 3:
        feature_maps = feature_maps.unsqueeze(0)
    rois = torch.cat([boxes.new_zeros(boxes.size(0), 1), boxes], dim=1)
    return roi_align(feature_maps, rois, pool_size)
