"""ResNet-50 body with torchvision's parameter names, written here because torchvision is not
part of this stack.  The reference instantiates ``torchvision.models.resnet50(
replace_stride_with_dilation=[False, False, dilation], norm_layer=FrozenBatchNorm2d)``
(/root/reference/models/backbone_scratch.py:156-159); this module reproduces that architecture
(v1.5 bottleneck: stride on the 3x3 conv; a dilated stage keeps dilation 1 in its first block and
uses the new dilation afterwards) so reference checkpoints load key-for-key:
``conv1, bn1, layer{1..4}.{i}.conv{1,2,3}, .bn{1,2,3}, .downsample.{0,1}``.
torchvision itself is third-party and absent from /root/reference: PARITY UNPINNED for the
architecture; numerics are plain conv2d and pinned by torch.
"""
import torch
from torch import nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, dilation, downsample, norm_layer):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        skip = x if self.downsample is None else self.downsample(x)
        return self.relu(out + skip)


class ResNet50(nn.Module):
    def __init__(self, norm_layer, replace_stride_with_dilation=(False, False, False)):
        super().__init__()
        self.inplanes, self.dilation = 64, 1
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._stage(64, 3, 1, False, norm_layer)
        self.layer2 = self._stage(128, 4, 2, replace_stride_with_dilation[0], norm_layer)
        self.layer3 = self._stage(256, 6, 2, replace_stride_with_dilation[1], norm_layer)
        self.layer4 = self._stage(512, 3, 2, replace_stride_with_dilation[2], norm_layer)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _stage(self, planes, blocks, stride, dilate, norm_layer):
        first_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                       norm_layer(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, first_dilation, downsample, norm_layer)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, 1, self.dilation, None, norm_layer))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))
