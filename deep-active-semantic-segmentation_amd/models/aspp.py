"""ASPP on the HIP path -- mirror of models/aspp.py:8-101 (same names, signatures, state_dict keys,
error behaviour).  Four conv+BN+ReLU branches (1x1 and three dilated 3x3, each one implicit-GEMM
launch that skips the taps falling entirely in the padding), the image-pool branch
(avg-pool -> 1x1 -> ReLU -> broadcast -> BN, the reference's ReLU-before-BN / BN-after-upsample order),
channel concat, 1x1 merge conv + BN + ReLU with the Dropout2d(0.5) mask fused into the BN-apply pass.
"""
import torch.nn as nn

from dass_hip import ops
from models._common import channels_last_weights, dropout_mask_for, init_weights


class ASPPModule(nn.Module):

    def __init__(self, inplanes, planes, kernel_size, padding, dilation, batchnorm):
        super(ASPPModule, self).__init__()
        self.atrous_conv = nn.Conv2d(inplanes, planes, kernel_size=kernel_size, stride=1, padding=padding, dilation=dilation, bias=False)
        self.bn = batchnorm(planes)
        self.relu = nn.ReLU()
        self._init_weight()

    def forward(self, x, out_into=None):
        # consumer: the channel concat -- out_into = (the 1280-wide buffer, this branch's channel offset): written in place there
        return ops.conv_bn_act(x, self.atrous_conv, self.bn, ops.ACT_RELU, emit_x3=False, out_into=out_into)

    def _init_weight(self):
        init_weights(self)
        channels_last_weights(self)


class ASPP(nn.Module):

    INPLANES = {'resnet': 2048, 'resnet101': 2048, 'mobilenet': 320}
    RATES = {16: (1, 6, 12, 18), 8: (1, 12, 24, 36)}

    def __init__(self, backbone, output_stride, batchnorm):
        super(ASPP, self).__init__()
        if backbone not in self.INPLANES:
            raise Exception('Unknown backbone')
        if output_stride not in self.RATES:
            raise NotImplementedError
        inplanes = self.INPLANES[backbone]
        # aspp1: the 1x1 branch; aspp2-4: 3x3 at the three larger rates, padding = rate (aspp.py:46-59)
        for i, rate in enumerate(self.RATES[output_stride], start=1):
            k = 1 if i == 1 else 3
            setattr(self, "aspp%d" % i, ASPPModule(inplanes, 256, k, padding=rate * (k // 2), dilation=rate, batchnorm=batchnorm))
        self.global_average_pool = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Conv2d(inplanes, 256, 1, stride=1, bias=False), nn.ReLU())
        self.bn_global_average_pool = batchnorm(256)
        self.conv1 = nn.Conv2d(5 * 256, 256, 1, bias=False)
        self.bn1 = batchnorm(256)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout2d(0.5)
        self._init_weight()

    @ops.bn_counter_scope
    def forward(self, x, dropout_mask=None, apply_dropout=True):
        """dropout_mask: optional explicit [N,256] multipliers (reproducible stochastic passes);
        apply_dropout=False returns the pre-dropout activation (used by the hoisted MC-dropout tail)."""
        h, w = x.shape[2], x.shape[3]
        xa, xb, xc, xd, xe = ops.fanout(x, 5)  # five consumers: their gradients are added in one pass
        # torch.cat((x1, x2, x3, x4, x5), dim=1) of aspp.py:83 without the copies: every branch's BN-apply pass (and the image-pool
        # branch's broadcast) writes its 256 channels straight into the 1280-wide buffer the merge conv reads
        wide = ops.new_act(x.shape[0], 1280, h, w, ops.compute_dtype(), x.device)
        x1 = self.aspp1(xa, (wide, 0))
        x2 = self.aspp2(xb, (wide, 256))
        x3 = self.aspp3(xc, (wide, 512))
        x4 = self.aspp4(xd, (wide, 768))
        x5 = ops.global_avgpool(xe)
        x5 = ops.conv_bn_act(x5, self.global_average_pool[1], None, ops.ACT_RELU, emit_x3=False)
        x5 = ops.broadcast_bn(x5, self.bn_global_average_pool, h, w, out_into=(wide, 1024))
        cat = ops.concat_shared(wide, x1, x2, x3, x4, x5)
        mask = None
        if apply_dropout:
            mask = dropout_mask if dropout_mask is not None else dropout_mask_for(self.dropout, x.shape[0], 256, x.device)
        return ops.conv_bn_act(cat, self.conv1, self.bn1, ops.ACT_RELU, nc_scale=mask, emit_x3=False)  # consumer: the decoder upsample

    def _init_weight(self):
        init_weights(self)
        channels_last_weights(self)
