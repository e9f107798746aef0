"""Hyper-parameters of the benchmark configs, restated as data.

The reference keeps a mutable global ``cfg`` built from ~1000 lines of nested dict copies (datasets/config.py:68-106,
975-995).  Only the values the inference hot path reads are kept here, as one flat immutable-by-convention object per
config; ``set_cfg(name)`` swaps the module-level default like the reference does (config.py:978-987).
"""
import copy


class Config:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def copy(self, **kw):
        c = copy.deepcopy(self)
        c.__dict__.update(kw)
        return c

    def __repr__(self):
        return f"Config({self.name})"


_ASPECT = [[3, 3], [3, 5], [5, 3]]  # [h, w] per kernel shape (config.py:642)

_base = Config(
    name="STMask_base",
    # dataset-level (config.py:389-400, 717-722)
    num_classes=41, max_num_detections=100, nms_top_k=200, nms_conf_thresh=0.05, nms_thresh=0.5,
    eval_conf_thresh=0.05, nms_as_miou=False, remove_false_inst=True,
    # backbone (config.py:288,307): (blocks, dcn_layers, dcn_interval)
    backbone_layers=[3, 4, 23, 3], backbone_dcn_layers=[0, 0, 0, 0], backbone_dcn_interval=1,
    selected_layers=[1, 2, 3],
    pred_aspect_ratios=[[_ASPECT]] * 5, pred_scales=[[24], [48], [96], [192], [384]],
    # FPN (config.py:362-384,647-651)
    fpn_num_features=256, fpn_interpolation_mode="bilinear", fpn_num_downsample=2, fpn_use_conv_downsample=True,
    fpn_pad=True, fpn_relu_downsample_layers=False, fpn_relu_pred_layers=True,
    # prediction module (config.py:654-659)
    share_prediction_module=True, extra_head_net=[(256, 3, {"padding": 1})], extra_layers=(2, 2, 2, 2),
    head_layer_params=[{"kernel_size": [3, 3], "padding": (1, 1)}, {"kernel_size": [3, 5], "padding": (1, 2)},
                       {"kernel_size": [5, 3], "padding": (2, 1)}],
    # masks (config.py:445-447, 662-667)
    mask_proto_src=0, mask_proto_n=32, mask_dim=32,
    mask_proto_net=[(256, 3, {"padding": 1})] * 3 + [(None, -2, {}), (256, 3, {"padding": 1})] + [(32, 1, {})],
    # heads (config.py:681-686)
    train_boxes=True, train_class=True, train_centerness=True, train_track=True, embed_dim=128,
    match_coeff=[0, 1, 2, 0],
    # temporal fusion (config.py:689-691)
    temporal_fusion_module=True, correlation_patch_size=11, correlation_selected_layer=1,
    # FCB (config.py:698-701)
    use_pred_offset=False, use_dcn_class=False, use_dcn_track=False, use_dcn_mask=False,
    use_sipmask=False, use_yolo_regressors=False,
)

_R50 = dict(backbone_layers=[3, 4, 6, 3])
_R50_DCN = dict(backbone_layers=[3, 4, 6, 3], backbone_dcn_layers=[0, 4, 6, 3], backbone_dcn_interval=2)
_R101_DCN = dict(backbone_layers=[3, 4, 23, 3], backbone_dcn_layers=[0, 4, 23, 3], backbone_dcn_interval=3)
_ADA = dict(use_pred_offset=True, use_dcn_class=True)
_ALI = dict(use_pred_offset=False, use_dcn_class=True)

CONFIGS = {
    "STMask_base_config": _base,
    "STMask_resnet50_config": _base.copy(name="STMask_resnet50", **_R50),
    "STMask_plus_base_config": _base.copy(name="STMask_plus_base", **_R101_DCN),
    "STMask_plus_base_ada_config": _base.copy(name="STMask_plus_base_ada", **_R101_DCN, **_ADA),
    "STMask_plus_base_ali_config": _base.copy(name="STMask_plus_base_ali", **_R101_DCN, **_ALI),
    "STMask_plus_resnet50_config": _base.copy(name="STMask_plus_resnet50", **_R50_DCN),
    "STMask_plus_resnet50_ada_config": _base.copy(name="STMask_plus_resnet50_ada", **_R50_DCN, **_ADA),
    "STMask_plus_resnet50_ali_config": _base.copy(name="STMask_plus_resnet50_ali", **_R50_DCN, **_ALI),
}

cfg = CONFIGS["STMask_plus_base_config"].copy()


def get_cfg(name):
    if name not in CONFIGS:
        raise KeyError(f"unknown config {name!r}; known: {sorted(CONFIGS)}")
    return CONFIGS[name].copy()


def set_cfg(name):
    """Replace the module-level default config in place (reference: datasets/config.py:978-987)."""
    cfg.__dict__.clear()
    cfg.__dict__.update(get_cfg(name).__dict__)
    return cfg
