"""Drop-in for the ``dcn_v2`` package (CharlesShang/DCNv2 @ pytorch_1.0) on MI355X.

The reference imports ``from dcn_v2 import DCN`` (backbone.py:5; ``DCN, DCNv2`` at FPN.py:8) and builds
``DCN(planes, planes, kernel_size=3, stride=stride, padding=dilation, dilation=dilation, deformable_groups=1)``
(backbone.py:21-22), then touches ``.bias`` and ``.conv_offset_mask.{weight,bias}`` (backbone.py:24-26).  Parameter
names, shapes and state-dict keys are therefore part of the contract.

Forward = one hand-written gfx950 launch pair (deformable im2col + fp32 MFMA GEMM with fused bias); the chunk / cat /
sigmoid that dcn_v2 does in torch is folded into the im2col kernel (it reads the raw ``conv_offset_mask`` output).
Inference only: no backward kernels (training is outside the hot path).
"""
import math

import torch
import torch.nn as nn

from . import ops


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class DCNv2(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        stdv = 1.0 / math.sqrt(n)
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)
            self.bias.zero_()

    def forward(self, input, offset, mask):
        K = self.kernel_size[0] * self.kernel_size[1]
        if offset.shape[1] != 2 * self.deformable_groups * K or mask.shape[1] != self.deformable_groups * K:
            raise ValueError("DCNv2: offset / mask channel count does not match kernel size and deformable_groups")
        return ops.deform_conv(input, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                               self.deformable_groups)


class DCN(DCNv2):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups)
        ch = self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset_mask = nn.Conv2d(in_channels, ch, kernel_size=self.kernel_size, stride=self.stride,
                                          padding=self.padding, bias=True)
        self.fuse_relu = False  # fuse.optimize_for_inference: apply the following ReLU in the GEMM epilogue
        self.init_offset()

    def init_offset(self):
        with torch.no_grad():
            self.conv_offset_mask.weight.zero_()
            self.conv_offset_mask.bias.zero_()

    def forward(self, input):
        om = self.conv_offset_mask(input)
        # offsets = om[:, :2*dg*K], mask = sigmoid(om[:, 2*dg*K:]) -- read in place by the kernel
        return ops.deform_conv(input, None, None, self.weight, self.bias, self.stride, self.padding, self.dilation,
                               self.deformable_groups, relu=self.fuse_relu, fused_om=om)
