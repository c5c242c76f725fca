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
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = batchnorm(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, dilation=dilation, padding=dilation, bias=False)
        self.bn2 = batchnorm(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = batchnorm(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.dilation = dilation

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


class ResNet(nn.Module):

    def __init__(self, block, layers, output_stride, batchnorm, pretrained=True):
        self.inplanes = 64
        super(ResNet, self).__init__()
        blocks = [1, 2, 4]
        if output_stride == 16:
            strides = [1, 2, 2, 1]
            dilations = [1, 1, 1, 2]
        elif output_stride == 8:
            strides = [1, 2, 1, 1]
            dilations = [1, 1, 2, 4]
        else:
            raise NotImplementedError

        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = batchnorm(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)

        self.layer1 = self._make_layer(block, 64, layers[0], stride=strides[0], dilation=dilations[0], batchnorm=batchnorm)
        self.layer2 = self._make_layer(block, 128, layers[1], stride=strides[1], dilation=dilations[1], batchnorm=batchnorm)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=strides[2], dilation=dilations[2], batchnorm=batchnorm)
        self.layer4 = self._make_MG_unit(block, 512, blocks=blocks, stride=strides[3], dilation=dilations[3], batchnorm=batchnorm)
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

    def _downsample(self, planes, block, stride, batchnorm):
        if stride != 1 or self.inplanes != planes * block.expansion:
            return nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                batchnorm(planes * block.expansion))
        return None

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1, batchnorm=None):
        downsample = self._downsample(planes, block, stride, batchnorm)
        layers = [block(self.inplanes, planes, stride, dilation, downsample, batchnorm)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, dilation=dilation, batchnorm=batchnorm))
        return nn.Sequential(*layers)

    def _make_MG_unit(self, block, planes, blocks, stride=1, dilation=1, batchnorm=None):
        downsample = self._downsample(planes, block, stride, batchnorm)
        layers = [block(self.inplanes, planes, stride, dilation=blocks[0] * dilation, downsample=downsample, batchnorm=batchnorm)]
        self.inplanes = planes * block.expansion
        for i in range(1, len(blocks)):
            layers.append(block(self.inplanes, planes, dilation=blocks[i] * dilation, batchnorm=batchnorm))
        return nn.Sequential(*layers)

    def _init_weight(self):
        init_weights(self)

    def _load_pretrained_model(self):
        load_local_pretrained(self, 'resnet101-5d3b4d8f.pth' if self._layers[2] == 23 else 'resnet50-19c8e357.pth')


def ResNet101(output_stride, batchnorm, pretrained=True):
    return ResNet(Bottleneck, [3, 4, 23, 3], output_stride, batchnorm, pretrained=pretrained)


def ResNet50(output_stride, batchnorm, pretrained=True):
    return ResNet(Bottleneck, [3, 4, 6, 3], output_stride, batchnorm, pretrained=pretrained)
