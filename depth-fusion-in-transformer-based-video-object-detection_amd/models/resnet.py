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

import os

from dfx import ops as _ops

_PAIR_SHORTCUT = os.environ.get("DFX_PAIR_SHORTCUT", "1") == "1"      # A/B switch of the fused conv3 + shortcut product


def _fold(conv, bn):
    """Conv weight with the frozen-BN scale folded in, and the remaining per-channel shift."""
    scale, shift = bn.scale_shift()
    return (conv.weight.detach() * scale.reshape(-1, 1, 1, 1)).contiguous(), shift.detach().contiguous()


def _versions(*mods):
    out = []
    for m in mods:
        for t in list(m.parameters(recurse=False)) + list(m.buffers(recurse=False)):
            out.append((t.data_ptr(), t._version))
    return tuple(out)


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
        self._folded = None

    def _folded_params(self):
        mods = [self.conv1, self.bn1, self.conv2, self.bn2, self.conv3, self.bn3]
        if self.downsample is not None:
            mods += [self.downsample[0], self.downsample[1]]
        key = _versions(*mods)
        if self._folded is None or self._folded[0] != key:
            c2 = self.conv2
            scale2, shift2 = self.bn2.scale_shift()
            # conv2 on the hand-written kernels (Winograd F(2x2,3x3) for stride 1 incl. the dilated stage,
            # implicit GEMM for stride 2): folded-BN scale in the weights, shift + ReLU in the epilogue
            plan2 = _ops.ConvPlan(c2.weight, shift2.detach(), c2.stride, c2.padding, c2.dilation, "relu",
                                  scale=scale2.detach(), groups=c2.groups, padding_mode=c2.padding_mode)
            f = [_fold(self.conv1, self.bn1), plan2, _fold(self.conv3, self.bn3)]
            f.append(_fold(self.downsample[0], self.downsample[1]) if self.downsample is not None else None)
            # a stride-1 projection shortcut (layer1[0]; layer4[0] of the dilated DC5 stage) joins conv3 in ONE product over
            # the concatenated input channels: W = [W3 | Wd], bias = shift3 + shiftd (dfx.ops.conv1x1_pair)
            pair = None
            if self.downsample is not None and self.downsample[0].stride[0] == 1 and _PAIR_SHORTCUT:
                (w3, b3), (wd, bd) = f[2], f[3]
                if w3.shape[1] % 16 == 0 and wd.shape[1] % 16 == 0:
                    pair = (torch.cat([w3.flatten(1), wd.flatten(1)], 1).contiguous(), (b3 + bd).contiguous())
            f.append(pair)
            self._folded = (key, f)
        return self._folded[1]

    def forward_fused(self, x):
        """Inference on the GPU: frozen BN folded into the convolutions.  The 3x3 convolutions run on the
        hand-written Winograd / implicit-GEMM kernels (dfx.ops.ConvPlan, csrc/conv_wino.hip,
        csrc/conv_igemm.hip), the 1x1 convolutions on the hand-written MFMA GEMM (dfx.ops.conv1x1), each
        with bias / residual / ReLU in its epilogue.  No CPU route."""
        (w1, b1), plan2, (w3, b3), down, pair = self._folded_params()
        out = self._conv1x1(0, x, w1, b1, relu=True)
        out = plan2(out)
        if pair is not None and (out.shape[2] * out.shape[3]) % 4 == 0 and out.shape[2:] == x.shape[2:]:
            return _ops.conv1x1_pair(out, x.contiguous(), pair[0], pair[1], relu=True)
        if down is not None:
            x = self._conv1x1(1, x, down[0], down[1], relu=False, stride=self.downsample[0].stride[0])
        return self._conv1x1(2, out, w3, b3, relu=True, residual=x)

    def _conv1x1(self, slot, x, w, b, relu, residual=None, stride=1):
        """1x1 convolution + bias (+ residual) (+ ReLU): the MFMA GEMM over [Ci] x [H*W] when its 16-byte
        operand rows allow (H*W a multiple of 4: every map of an 800x1333 frame), else the implicit-GEMM
        convolution (scalar gathers, any geometry) followed by the fused bias / residual / ReLU pass."""
        ho, wo = (x.shape[2] + stride - 1) // stride, (x.shape[3] + stride - 1) // stride
        if (ho * wo) % 4 == 0 and stride == 1:
            return _ops.conv1x1(x, w, b, residual=residual, relu=relu, stride=stride)
        # strided shortcut (the implicit GEMM gathers every other pixel itself: no subsampled copy of the map) and
        # maps whose rows are not 16-byte aligned
        # keyed on the version key of the parameters the folded tensors were made from (the folded temporaries' own
        # addresses / versions say nothing: the allocator hands an address out again after a refold)
        key = self._folded[0]
        plans = self.__dict__.setdefault("_plans1x1", {})
        fused = residual is None
        if slot not in plans or plans[slot][0] != key:
            plans[slot] = (key, _ops.ConvPlan(w.reshape(w.shape[0], -1, 1, 1), b if fused else None, stride, 0, 1,
                                              "relu" if (fused and relu) else None))
        y = plans[slot][1](x)
        return y if fused else _ops.bias_act_(y, b, residual=residual, relu=relu)

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
        # torchvision's resnet50 also owns the ImageNet classifier; the detectors never call it, but a reference checkpoint
        # carries `backbone.0.body.fc.{weight,bias}` and a strict load (benchmark.py:58) needs somewhere to put them
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, 1000)
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

    def stem(self, x, fused=False):
        if not fused:
            return self.maxpool(self.relu(self.bn1(self.conv1(x))))
        key = _versions(self.conv1, self.bn1)
        if getattr(self, "_stem_folded", None) is None or self._stem_folded[0] != key:
            scale, shift = self.bn1.scale_shift()
            self._stem_folded = (key, (_ops.ConvPlan(self.conv1.weight, None, 2, 3, 1, None, scale=scale.detach()),
                                       shift.detach().contiguous()))
        plan, b = self._stem_folded[1]
        return _ops.bias_relu_maxpool(plan(x), b)

    def run_stage(self, stage, x, fused=False):
        if not fused:
            return stage(x)
        for block in stage:
            x = block.forward_fused(x)
        return x

    def forward(self, x):
        x = self.stem(x)
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))
