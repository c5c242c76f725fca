"""Dilated MobileNetV2 backbone on the HIP path -- mirror of models/backbone/mobilenet.py:12-169.

Quirk kept on purpose (mobilenet.py:23-30,70-77): `fixed_padding` zero-pads the block INPUT, so the
1x1 expand conv + BN + ReLU6 run over the padded (H+2d)x(W+2d) map -- the border of the hidden tensor
is relu6(bn(0)), not 0, and train-mode BN statistics include it.  Here that is the implicit-GEMM conv
with R=S=1 and pad=d (output larger than input); the depthwise 3x3 then runs with pad 0.
"""
import torch.nn as nn

import constants
from dass_hip import ops
from models._common import channels_last_weights, dropout_mask_for, init_weights, load_local_pretrained


def conv_bn(inplanes, outplanes, stride, batchnorm):
    return nn.Sequential(nn.Conv2d(inplanes, outplanes, 3, stride, 1, bias=False), batchnorm(outplanes), nn.ReLU6(inplace=True))


def _conv_bn_act(cin, cout, k, stride, dilation, groups, batchnorm, act):
    """[conv, BN(, ReLU6)] -- the unit every entry of an inverted-residual block's `conv` Sequential is made of"""
    unit = [nn.Conv2d(cin, cout, k, stride, 0, dilation, groups=groups, bias=False), batchnorm(cout)]
    return unit + [nn.ReLU6(inplace=True)] if act else unit


class InvertedResidual(nn.Module):

    def __init__(self, inplanes, outplanes, stride, dilation, expand_ratio, batchnorm):
        super(InvertedResidual, self).__init__()
        assert stride in (1, 2)
        hidden = round(inplanes * expand_ratio)
        self.stride, self.dilation, self.kernel_size = stride, dilation, 3
        self.expand = expand_ratio != 1
        self.use_res_connect = stride == 1 and inplanes == outplanes
        # (mobilenet.py:33-66) pointwise expand (absent at ratio 1) -> depthwise 3x3, pad 0 (the block pads its own input) -> linear pointwise
        units = ([(inplanes, hidden, 1, 1, 1, 1, True)] if self.expand else []) + [(hidden, hidden, 3, stride, dilation, hidden, True),
                                                                                   (hidden, outplanes, 1, 1, 1, 1, False)]
        self.conv = nn.Sequential(*[m for cin, cout, k, st, dil, g, act in units for m in _conv_bn_act(cin, cout, k, st, dil, g, batchnorm, act)])

    def forward(self, x):
        d = self.dilation  # fixed_padding for k=3: d on every side
        c = self.conv
        res = None
        if self.use_res_connect:
            x, res = ops.fanout(x, 2)  # block input read by the expand conv and the skip connection: one gradient sum pass
        if self.expand:
            h = ops.conv_bn_act(x, c[0], c[1], ops.ACT_RELU6, extra_pad=d, emit_x3=False)   # 1x1 over the zero-padded input (feeds the depthwise conv)
            h = ops.conv_bn_act(h, c[3], c[4], ops.ACT_RELU6)                 # depthwise, pad 0
            return ops.conv_bn_act(h, c[6], c[7], ops.ACT_NONE, residual=res)
        h = ops.conv_bn_act(x, c[0], c[1], ops.ACT_RELU6, extra_pad=d)       # depthwise over the padded input
        return ops.conv_bn_act(h, c[3], c[4], ops.ACT_NONE, residual=res)


class MobileNetV2(nn.Module):

    # (expansion t, channels c, blocks n, stride s) of the seven stages (mobilenet.py:92-101)
    STAGES = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))

    def __init__(self, input_channels=3, output_stride=8, batchnorm=None, width_mult=1., pretrained=True, mc_dropout=False):
        super(MobileNetV2, self).__init__()
        width = int(32 * width_mult)
        features = [conv_bn(input_channels, width, 2, batchnorm)]
        reached, rate = 2, 1          # stride reached so far; dilation that replaces further striding once output_stride is reached
        for t, c, n, s in self.STAGES:
            if reached == output_stride:
                stride, dilation, rate = 1, rate, rate * s
            else:
                stride, dilation, reached = s, 1, reached * s
            for i in range(n):
                features.append(InvertedResidual(width, int(c * width_mult), stride if i == 0 else 1, dilation, t, batchnorm))
                width = int(c * width_mult)
        if mc_dropout:
            features.append(nn.Dropout2d(p=constants.MC_DROPOUT_RATE))
        self.features = nn.Sequential(*features)
        self._initialize_weights()
        if pretrained:
            self._load_pretrained_model()
        self.low_level_features = self.features[0:4]
        self.high_level_features = self.features[4:]
        self.dropout = nn.Dropout2d(p=constants.MC_DROPOUT_RATE)
        self.mc_dropout = mc_dropout
        channels_last_weights(self)

    @staticmethod
    def _run(seq, x, first_is_image):
        for i, m in enumerate(seq):
            if isinstance(m, nn.Sequential):  # the stem conv_bn
                x = ops.conv_bn_act(x, m[0], m[1], ops.ACT_RELU6, image_input=first_is_image and i == 0, emit_x3=not (first_is_image and i == 0))
            elif isinstance(m, nn.Dropout2d):
                mask = dropout_mask_for(m, x.shape[0], x.shape[1], x.device)
                if mask is not None:
                    x = ops.channel_scale(x, mask)
            else:
                x = m(x)
        return x

    @ops.bn_counter_scope
    def forward(self, x):
        low_level_feat = self._run(self.low_level_features, x, True)
        low_level_feat, hi_in = ops.fanout(low_level_feat, 2)   # consumers: the decoder and the high-level features
        x = self._run(self.high_level_features, hi_in, False)
        if self.mc_dropout:
            mask = dropout_mask_for(self.dropout, low_level_feat.shape[0], low_level_feat.shape[1], low_level_feat.device)
            if mask is not None:
                low_level_feat = ops.channel_scale(low_level_feat, mask)
        return x, low_level_feat

    def _load_pretrained_model(self):
        load_local_pretrained(self, 'mobilenet_v2-6a65762b.pth')

    def _initialize_weights(self):
        init_weights(self)
