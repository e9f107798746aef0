"""Drop-in for the two ``mmcv.ops`` entry points the reference uses (mmcv-full 1.1.2, README.md:37):

* ``DeformConv2d(C, C, kernel_size=(kh,kw), padding=(ph,pw), deform_groups=dg)`` -- Featurealign.py:3,27-31,72 and
  prediction_head_FC.py:10.  ``forward(x, offset)``, ``offset [B, dg*2*kh*kw, H, W]`` with channel 2k = dy, 2k+1 = dx.
  Implements the INTENDED semantics ((padH, padW) = (padding[0], padding[1]); 3x5 / 5x3 keep HxW) -- the reference's
  README patch (README.md:63-88) exists only to work around an argument-order quirk of that mmcv release.
* ``roi_align(input, rois, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True)`` --
  track_to_segment_head.py:6,86.
"""
import math

import torch
import torch.nn as nn

from . import ops


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class DeformConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deform_groups=1, bias=False):
        super().__init__()
        assert not bias, "mmcv DeformConv2d has no bias"
        assert in_channels % groups == 0 and out_channels % groups == 0
        if groups != 1:
            raise NotImplementedError("groups != 1 is outside the STMask hot path")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.groups, self.deform_groups = groups, deform_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        stdv = 1.0 / math.sqrt(n)
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)

    def forward(self, x, offset):
        K = self.kernel_size[0] * self.kernel_size[1]
        assert offset.shape[1] == self.deform_groups * 2 * K, \
            f"offset has {offset.shape[1]} channels, expected {self.deform_groups * 2 * K}"
        return ops.deform_conv(x, offset, None, self.weight, None, self.stride, self.padding, self.dilation,
                               self.deform_groups)


def roi_align(input, rois, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode="avg", aligned=True):
    if pool_mode != "avg":
        raise NotImplementedError("only pool_mode='avg' is on the STMask hot path")
    assert rois.size(1) == 5, "RoI must be (idx, x1, y1, x2, y2)!"
    return ops.roi_align(input, rois, output_size, spatial_scale, sampling_ratio, aligned)


class RoIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode="avg", aligned=True):
        super().__init__()
        self.output_size, self.spatial_scale = _pair(output_size), float(spatial_scale)
        self.sampling_ratio, self.pool_mode, self.aligned = int(sampling_ratio), pool_mode, aligned

    def forward(self, input, rois):
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.pool_mode,
                         self.aligned)
