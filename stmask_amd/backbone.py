"""ResNet-50/101 (v1.5) backbone with DCNv2 3x3 convs -- mirrors backbone.py:8-186 of the reference.

Dense convs / BN / ReLU / max-pool stay on PyTorch-ROCm (MIOpen, MFMA); the deformable 3x3 convs run on the
hand-written kernels through ``stmask_amd.dcn_v2.DCN``.  Module and parameter names match the reference so its
checkpoints load unchanged: ``backbone.layers.{s}.{b}.conv2.conv_offset_mask.weight`` ...
"""
import torch.nn as nn

from .dcn_v2 import DCN


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1, use_dcn=False):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        if use_dcn:
            self.conv2 = DCN(planes, planes, kernel_size=3, stride=stride, padding=dilation, dilation=dilation,
                             deformable_groups=1)
        else:
            self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=dilation, bias=False,
                                   dilation=dilation)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.use_dcn = use_dcn

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        residual = x if self.downsample is None else self.downsample(x)
        out += residual
        return self.relu(out)


class ResNetBackbone(nn.Module):
    """layers: blocks per stage; dcn_layers: how many trailing blocks of each stage are deformable, thinned by
    dcn_interval (backbone.py:124,130): block 0 is DCN iff dcn_layers >= blocks; block i>0 iff
    i + dcn_layers >= blocks and i % dcn_interval == 0."""

    def __init__(self, layers, dcn_layers=(0, 0, 0, 0), dcn_interval=1):
        super().__init__()
        self.layers = nn.ModuleList()
        self.channels = []
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for planes, blocks, stride, ndcn in zip((64, 128, 256, 512), layers, (1, 2, 2, 2), dcn_layers):
            self._make_layer(planes, blocks, stride, ndcn, dcn_interval)

    def _make_layer(self, planes, blocks, stride, dcn_layers, dcn_interval):
        downsample = None
        if stride != 1 or self.inplanes != planes * Bottleneck.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * Bottleneck.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * Bottleneck.expansion))
        blocks_ = [Bottleneck(self.inplanes, planes, stride, downsample, use_dcn=(dcn_layers >= blocks))]
        self.inplanes = planes * Bottleneck.expansion
        for i in range(1, blocks):
            use_dcn = ((i + dcn_layers) >= blocks) and (i % dcn_interval == 0)
            blocks_.append(Bottleneck(self.inplanes, planes, use_dcn=use_dcn))
        self.channels.append(planes * Bottleneck.expansion)
        self.layers.append(nn.Sequential(*blocks_))

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        outs = []
        for layer in self.layers:
            x = layer(x)
            outs.append(x)
        return tuple(outs)


def construct_backbone(cfg):
    return ResNetBackbone(cfg.backbone_layers, cfg.backbone_dcn_layers, cfg.backbone_dcn_interval)
