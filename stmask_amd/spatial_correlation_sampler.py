"""Drop-in for ``spatial_correlation_sampler`` (README.md:50-53) as called at track_to_segment_head.py:4,53-59:
``spatial_correlation_sample(x1, x2, kernel_size=1, patch_size=11, stride=1, padding=0, dilation_patch=1)``
-> ``[B, patch, patch, H, W]``.  Only the configuration on the hot path is implemented on the MI355X."""
import torch.nn as nn

from . import ops


def spatial_correlation_sample(input1, input2, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1,
                               dilation_patch=1):
    if kernel_size != 1 or stride != 1 or padding != 0 or dilation != 1:
        raise NotImplementedError("only kernel_size=1, stride=1, padding=0, dilation=1 is on the STMask hot path")
    return ops.corr_patch(input1, input2, patch_size, dilation_patch)


class SpatialCorrelationSampler(nn.Module):
    def __init__(self, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1, dilation_patch=1):
        super().__init__()
        self.args = dict(kernel_size=kernel_size, patch_size=patch_size, stride=stride, padding=padding,
                         dilation=dilation, dilation_patch=dilation_patch)

    def forward(self, input1, input2):
        return spatial_correlation_sample(input1, input2, **self.args)
