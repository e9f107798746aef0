"""Inference-time graph surgery on the trunk: fold eval-mode BatchNorm into the preceding convolution.

The reference runs conv -> BN -> ReLU as three kernels per layer (backbone.py:38-58); in eval mode BN is the affine map
y = (x - mean) * gamma / sqrt(var + eps) + beta, which folds exactly into the conv's weights and bias
(w' = w * s, b' = (b - mean) * s + beta with s = gamma / sqrt(var + eps)).  53 BN launches per frame batch disappear.
Works for the DCN 3x3 convs too (their GEMM epilogue adds the bias).  Results change only by fp32 rounding (~1e-7
relative); tests/test_host_model_cpu.py checks the folded model against the reference goldens.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .backbone import Bottleneck
from .dcn_v2 import DCN


def _fold(conv, bn):
    with torch.no_grad():
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        conv.weight.mul_(s.view(-1, 1, 1, 1))
        old = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
        new_bias = (old - bn.running_mean) * s + bn.bias
        if conv.bias is None:
            conv.bias = nn.Parameter(new_bias.clone())
        else:
            conv.bias.copy_(new_bias)


def fold_batchnorm(net):
    """In place; returns the number of BatchNorm layers removed.  Call after load_weights(), in eval mode."""
    assert not net.training, "fold_batchnorm is an inference-time transformation"
    n = 0
    bb = net.backbone
    if isinstance(bb.bn1, nn.BatchNorm2d):
        _fold(bb.conv1, bb.bn1)
        bb.bn1 = nn.Identity()
        n += 1
    for layer in bb.layers:
        for blk in layer:
            for cname, bname in (("conv1", "bn1"), ("conv2", "bn2"), ("conv3", "bn3")):
                conv, bn = getattr(blk, cname), getattr(blk, bname)
                if isinstance(bn, nn.BatchNorm2d):
                    assert isinstance(conv, (nn.Conv2d, DCN))
                    _fold(conv, bn)
                    setattr(blk, bname, nn.Identity())
                    n += 1
            if blk.downsample is not None and isinstance(blk.downsample[1], nn.BatchNorm2d):
                _fold(blk.downsample[0], blk.downsample[1])
                blk.downsample[1] = nn.Identity()
                n += 1
    return n


# ---------------------------------------------------------------------------------------------------------------
# Fused epilogues: after BN folding every trunk conv is followed by (bias add [+ residual add] + ReLU).  MIOpen runs
# the bias add as its own kernel and torch runs the add / ReLU as further kernels; the modules below run the conv
# without bias and finish with ONE in-place HIP pass (stm_bias_act_f32).  Classes are swapped in place
# (instance.__class__), so parameter names / state-dict keys do not change.
def _epilogue(y, bias, residual=None, relu=True):
    B, C, H, W = y.shape
    nhwc = (not y.is_contiguous()) and y.is_contiguous(memory_format=torch.channels_last)
    if y.is_cuda and ((nhwc and C % 4 == 0) or (y.is_contiguous() and (H * W) % 4 == 0)):
        return ops.bias_act_(y, bias, residual, relu)
    y = y + bias.view(1, -1, 1, 1)          # shapes the kernel does not cover (e.g. 3x5 maps in NCHW)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


class _ConvBiasReLU(nn.Conv2d):
    def forward(self, x):
        y = F.conv2d(x, self.weight, None, self.stride, self.padding, self.dilation, self.groups)
        return _epilogue(y, self.bias, None, True)


class _FusedBottleneck(Bottleneck):
    def forward(self, x):
        c1, c2, c3 = self.conv1, self.conv2, self.conv3
        out = _epilogue(F.conv2d(x, c1.weight, None, c1.stride, c1.padding), c1.bias)
        if isinstance(c2, DCN):
            out = c2(out)                    # bias + ReLU already in the GEMM epilogue (c2.fuse_relu)
        else:
            out = _epilogue(F.conv2d(out, c2.weight, None, c2.stride, c2.padding, c2.dilation), c2.bias)
        out = F.conv2d(out, c3.weight, None, c3.stride, c3.padding)
        if self.downsample is not None:
            d = self.downsample[0]
            res = F.conv2d(x, d.weight, None, d.stride, d.padding)
        else:
            res = x
        return _epilogue(out, self.bias3, res, True)


def optimize_for_inference(net, planar=False, planes="fp16x2"):
    """fold_batchnorm + fused conv epilogues on the backbone, FPN prediction layers, proto-net and the head towers.
    In place, eval mode only; call after the weights are loaded and the model is on its device.
    planar=True (GPU only) additionally routes FPN prediction/downsample layers, proto-net and the shared head through
    the split-operand matrix-core convolution (stmask_amd/planar.py); planes = "fp16x2" (two fp16 planes, three MFMA
    products per fp32 product; activations must stay inside fp16's range, the pipeline checks) or "bf16x3" (three bf16
    planes, six products; any range) -- both carry fp32-level error (tests/test_gpu_conv.py) -- or "fp16x1": the
    backbone in genuine fp16 (one plane, one product, fp32 accumulation; ~1e-3 relative), everything else fp16x2."""
    n_bn = fold_batchnorm(net)
    n_fused = 0
    bb = net.backbone
    for layer in bb.layers:
        for blk in layer:
            b3 = blk.conv3.bias.detach().clone()
            if blk.downsample is not None:
                b3 = b3 + blk.downsample[0].bias.detach()
            blk.register_buffer("bias3", b3, persistent=False)
            if isinstance(blk.conv2, DCN):
                blk.conv2.fuse_relu = True
            blk.__class__ = _FusedBottleneck
            n_fused += 3
    # stem: conv1 (+ folded bn1) + ReLU
    bb.conv1.__class__ = _ConvBiasReLU
    bb.relu = nn.Identity()
    n_fused += 1

    def fuse_sequential(seq):
        nonlocal n_fused
        mods = list(seq.children())
        for i in range(len(mods) - 1):
            if type(mods[i]) is nn.Conv2d and isinstance(mods[i + 1], nn.ReLU) and mods[i].bias is not None:
                mods[i].__class__ = _ConvBiasReLU
                seq[i + 1] = nn.Identity()
                n_fused += 1

    fuse_sequential(net.proto_net)
    head = net.prediction_layers[0]
    for name in ("upfeature", "bbox_extra", "conf_extra", "mask_extra", "track_extra"):
        fuse_sequential(getattr(head, name))
    for pred in net.fpn.pred_layers:
        pred.__class__ = _ConvBiasReLU
        n_fused += 1
    net.fpn.pred_relu_fused = True
    if planar:
        attach_planar(net, build_planar(net, planes))
    return n_bn, n_fused


def build_planar(net, planes):
    """The planar inference graph of a net that optimize_for_inference has folded and fused, in the plane format `planes`: (PlanarGraph,
    PlanarBackbone, PlanarTemporalNet or None, planes).  Leaves the module-level format of stmask_amd.planar set to it (the objects capture their
    format at construction; helpers that split tensors read the module's)."""
    from . import planar as _planar
    from .planar import PlanarBackbone, PlanarGraph, PlanarTemporalNet
    if planes not in ("fp16x2", "bf16x3", "fp16x1"):
        raise ValueError("planes must be 'fp16x2', 'bf16x3' or 'fp16x1'")
    # fp16x1 (BASELINE config 5, "fp16 MFMA backbone convs"): the ResNet backbone's convolutions -- bottleneck 1x1 / 3x3,
    # downsample projections, DCN offset convs and DCN GEMMs -- run on ONE fp16 plane (one MFMA product, fp32 accumulate);
    # FPN, proto-net, heads and TemporalNet keep the fp32-equivalent fp16x2 format
    _planar.set_format(0 if planes == "bf16x3" else 1, backbone_fmt=2 if planes == "fp16x1" else None)
    graph = PlanarGraph(net)
    bb = PlanarBackbone(net.backbone, selected=net.backbone_selected)
    bb.planes_only = True                           # the planar FPN laterals read the stage outputs as planes
    tn = PlanarTemporalNet(net.TemporalNet) if getattr(net, "TemporalNet", None) is not None else None
    return graph, bb, tn, planes


def attach_planar(net, built):
    """Make `built` (build_planar) the graph net.forward_single runs."""
    from . import planar as _planar
    graph, bb, tn, planes = built
    _planar.set_format(0 if planes == "bf16x3" else 1, backbone_fmt=2 if planes == "fp16x1" else None)
    net._planar, net._planar_backbone = graph, bb
    if tn is not None:
        net._planar_temporal = tn
    net._planar_planes = planes
