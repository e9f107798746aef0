"""Inference-time graph surgery on the trunk: fold eval-mode BatchNorm into the preceding convolution.

The reference runs conv -> BN -> ReLU as three kernels per layer (backbone.py:38-58); in eval mode BN is the affine map
y = (x - mean) * gamma / sqrt(var + eps) + beta, which folds exactly into the conv's weights and bias
(w' = w * s, b' = (b - mean) * s + beta with s = gamma / sqrt(var + eps)).  53 BN launches per frame batch disappear.
Works for the DCN 3x3 convs too (their GEMM epilogue adds the bias).  Results change only by fp32 rounding (~1e-7
relative); tests/test_host_model_cpu.py checks the folded model against the reference goldens.
"""
import torch
import torch.nn as nn

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
