"""Dilated ResNet-50/101 backbone on the HIP path -- mirror of models/backbone/resnet.py:6-169.

Same constructor signatures, attribute names and state_dict keys as the reference; every
conv+BN(+ReLU)(+residual) site is one fused dass_hip call (implicit-GEMM MFMA conv, BN statistics /
apply kernels) and the stem pool is the HIP max-pool.
"""
import torch.nn as nn

from dass_hip import ops
from models._common import channels_last_weights, init_weights, load_local_pretrained


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, batchnorm=None):
        super(Bottleneck, self).__init__()
        # the block's three conv + BN pairs as data (resnet.py:9-20): (in, out, kernel, stride, dilation); registration order = the
        # reference's state_dict order conv1, bn1, conv2, bn2, conv3, bn3
        plan = ((inplanes, planes, 1, 1, 1), (planes, planes, 3, stride, dilation), (planes, planes * self.expansion, 1, 1, 1))
        for i, (cin, cout, k, st, dil) in enumerate(plan, start=1):
            setattr(self, "conv%d" % i, nn.Conv2d(cin, cout, kernel_size=k, stride=st, dilation=dil, padding=dil * (k // 2), bias=False))
            setattr(self, "bn%d" % i, batchnorm(cout))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride, self.dilation = stride, dilation

    def forward(self, x):
        if self.downsample:
            xa, xb = ops.fanout(x, 2)
            out = ops.conv_bn_act(xa, self.conv1, self.bn1, ops.ACT_RELU, consumer=self.conv2, sole_consumer=True)
            residual = ops.conv_bn_act(xb, self.downsample[0], self.downsample[1], ops.ACT_NONE)
        else:
            # identity block: conv1 hands x back for the skip connection, so that the skip gradient is added inside
            # conv1's input-gradient launch (no separate accumulation pass over the block input)
            out, residual = ops.conv_bn_act(x, self.conv1, self.bn1, ops.ACT_RELU, fork=True, consumer=self.conv2, sole_consumer=True)
        # (conv1's and conv2's outputs have exactly one reader each -- the next conv of the block -- so only their split rows are written)
        out = ops.conv_bn_act(out, self.conv2, self.bn2, ops.ACT_RELU, consumer=self.conv3, sole_consumer=True)
        # bn3 -> += residual -> relu (resnet.py:36-43) in one epilogue
        return ops.conv_bn_act(out, self.conv3, self.bn3, ops.ACT_RELU, residual=residual)


# (stride, dilation) of the four stages per output stride (resnet.py:50-57) and the multi-grid rates of the last one (resnet.py:48)
_STAGE_GEOMETRY = {16: ((1, 1), (2, 1), (2, 1), (1, 2)), 8: ((1, 1), (2, 1), (1, 2), (1, 4))}
_MULTI_GRID = (1, 2, 4)
_STAGE_PLANES = (64, 128, 256, 512)


class ResNet(nn.Module):

    def __init__(self, block, layers, output_stride, batchnorm, pretrained=True):
        super(ResNet, self).__init__()
        if output_stride not in _STAGE_GEOMETRY:
            raise NotImplementedError
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = batchnorm(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for i, ((stride, dilation), planes) in enumerate(zip(_STAGE_GEOMETRY[output_stride], _STAGE_PLANES)):
            # stages 1-3: layers[i] blocks at one rate; stage 4: the multi-grid unit, one block per rate whatever layers[3] says
            rates = [dilation] * layers[i] if i < 3 else [m * dilation for m in _MULTI_GRID]
            setattr(self, "layer%d" % (i + 1), self._stage(block, planes, stride, rates, batchnorm))
        self._layers = layers
        self._init_weight()
        if pretrained:
            self._load_pretrained_model()
        channels_last_weights(self)

    @ops.bn_counter_scope
    def forward(self, input):
        x = ops.conv_bn_act(input, self.conv1, self.bn1, ops.ACT_RELU, image_input=True, emit_x3=False)  # consumer: the max-pool
        x = ops.maxpool3x3s2(x)
        x = self.layer1(x)
        x, low_level_feat = ops.fanout(x, 2)   # consumers: layer 2 and the decoder
        x = self.layer2(x)
        x = self.layer3(x)
        x = self.layer4(x)
        return x, low_level_feat

    def _stage(self, block, planes, stride, rates, batchnorm):
        """one residual stage: the first block carries the stride and -- when the shape changes -- the 1x1 projection of the skip path
        (`downsample.0` / `.1` in the state_dict), the others keep stride 1; block j runs at dilation rates[j]"""
        out_planes = planes * block.expansion
        project = None
        if stride != 1 or self.inplanes != out_planes:
            project = nn.Sequential(nn.Conv2d(self.inplanes, out_planes, kernel_size=1, stride=stride, bias=False), batchnorm(out_planes))
        blocks = []
        for j, rate in enumerate(rates):
            blocks.append(block(self.inplanes, planes, stride if j == 0 else 1, rate, project if j == 0 else None, batchnorm))
            self.inplanes = out_planes
        return nn.Sequential(*blocks)

    def _init_weight(self):
        init_weights(self)

    def _load_pretrained_model(self):
        load_local_pretrained(self, 'resnet101-5d3b4d8f.pth' if self._layers[2] == 23 else 'resnet50-19c8e357.pth')


def ResNet101(output_stride, batchnorm, pretrained=True):
    return ResNet(Bottleneck, [3, 4, 23, 3], output_stride, batchnorm, pretrained=pretrained)


def ResNet50(output_stride, batchnorm, pretrained=True):
    return ResNet(Bottleneck, [3, 4, 6, 3], output_stride, batchnorm, pretrained=pretrained)
