"""CPU oracle: DeepLab-v3+ (ResNet-50/101 or MobileNetV2 backbone, ASPP, decoder) in stock PyTorch fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  A restatement -- not a copy -- of what the reference
computes, keeping its state_dict keys so one set of weights drives reference, oracle and the HIP product:
  backbone  models/backbone/resnet.py:6-169, models/backbone/mobilenet.py:12-169
  ASPP      models/aspp.py:8-101     (image-pool branch: ReLU before BN, BN after the 1x1->HxW upsample)
  decoder   models/decoder.py:9-60   (returns (low-res logits, 304-ch concat features))
  assembly  models/deeplab.py:11-62
Dropout2d sites take explicit [N,C] multiplier masks so stochastic passes are reproducible.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

MC_DROPOUT_RATE = 0.25  # constants.py:5


def _bilinear(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


def _mask(x, m):
    return x if m is None else x * m[:, :, None, None].to(x.dtype)


# ------------------------------------------------------------------ ResNet (resnet.py:6-157)
class OBottleneck(nn.Module):
    def __init__(self, cin, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        skip = x if self.downsample is None else self.downsample(x)
        return F.relu(y + skip)


class OResNet(nn.Module):
    def __init__(self, layers, output_stride=16):
        super().__init__()
        if output_stride == 16:
            strides, dils = [1, 2, 2, 1], [1, 1, 1, 2]
        elif output_stride == 8:
            strides, dils = [1, 2, 1, 1], [1, 1, 2, 4]
        else:
            raise NotImplementedError
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self._cin = 64
        self.layer1 = self._stage(64, [dils[0]] * layers[0], strides[0])
        self.layer2 = self._stage(128, [dils[1]] * layers[1], strides[1])
        self.layer3 = self._stage(256, [dils[2]] * layers[2], strides[2])
        self.layer4 = self._stage(512, [m * dils[3] for m in (1, 2, 4)], strides[3])  # multi-grid unit

    def _stage(self, planes, dilations, stride):
        down = None
        if stride != 1 or self._cin != planes * 4:
            down = nn.Sequential(nn.Conv2d(self._cin, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        blocks = [OBottleneck(self._cin, planes, stride, dilations[0], down)]
        self._cin = planes * 4
        for d in dilations[1:]:
            blocks.append(OBottleneck(self._cin, planes, 1, d))
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        low = self.layer1(x)
        return self.layer4(self.layer3(self.layer2(low))), low


# ------------------------------------------------------------------ MobileNetV2 (mobilenet.py:12-144)
class OInvertedResidual(nn.Module):
    def __init__(self, cin, cout, stride, dilation, expand):
        super().__init__()
        hid = round(cin * expand)
        self.res = stride == 1 and cin == cout
        self.dilation = dilation
        layers = []
        if expand != 1:
            layers += [nn.Conv2d(cin, hid, 1, bias=False), nn.BatchNorm2d(hid), nn.ReLU6()]
        layers += [nn.Conv2d(hid, hid, 3, stride, 0, dilation, groups=hid, bias=False), nn.BatchNorm2d(hid), nn.ReLU6(),
                   nn.Conv2d(hid, cout, 1, bias=False), nn.BatchNorm2d(cout)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        d = self.dilation  # fixed_padding(k=3): pad d on every side, BEFORE the expand 1x1 (mobilenet.py:23-30,72)
        y = self.conv(F.pad(x, (d, d, d, d)))
        return x + y if self.res else y


class OMobileNetV2(nn.Module):
    SETTING = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]

    def __init__(self, output_stride=16, input_channels=3, mc_dropout=False):
        super().__init__()
        feats = [nn.Sequential(nn.Conv2d(input_channels, 32, 3, 2, 1, bias=False), nn.BatchNorm2d(32), nn.ReLU6())]
        cin, cur, rate = 32, 2, 1
        for t, c, n, s in self.SETTING:
            if cur == output_stride:
                stride, dil = 1, rate
                rate *= s
            else:
                stride, dil = s, 1
                cur *= s
            for i in range(n):
                feats.append(OInvertedResidual(cin, c, stride if i == 0 else 1, dil, t))
                cin = c
        if mc_dropout:
            feats.append(nn.Dropout2d(MC_DROPOUT_RATE))
        self.features = nn.Sequential(*feats)
        self.low_level_features = self.features[0:4]
        self.high_level_features = self.features[4:]
        self.mc_dropout = mc_dropout

    def forward(self, x):
        low = self.low_level_features(x)
        return self.high_level_features(low), low


# ------------------------------------------------------------------ ASPP (aspp.py:8-89)
class OASPPModule(nn.Module):
    def __init__(self, cin, k, d):
        super().__init__()
        self.atrous_conv = nn.Conv2d(cin, 256, k, 1, 0 if k == 1 else d, d, bias=False)
        self.bn = nn.BatchNorm2d(256)

    def forward(self, x):
        return F.relu(self.bn(self.atrous_conv(x)))


class OASPP(nn.Module):
    def __init__(self, backbone, output_stride):
        super().__init__()
        cin = {"resnet": 2048, "resnet101": 2048, "mobilenet": 320}[backbone]
        d = {16: [1, 6, 12, 18], 8: [1, 12, 24, 36]}[output_stride]
        self.aspp1, self.aspp2 = OASPPModule(cin, 1, d[0]), OASPPModule(cin, 3, d[1])
        self.aspp3, self.aspp4 = OASPPModule(cin, 3, d[2]), OASPPModule(cin, 3, d[3])
        self.global_average_pool = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Conv2d(cin, 256, 1, bias=False), nn.ReLU())
        self.bn_global_average_pool = nn.BatchNorm2d(256)
        self.conv1 = nn.Conv2d(1280, 256, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(256)

    def forward(self, x, mask=None):
        x5 = self.bn_global_average_pool(_bilinear(self.global_average_pool(x), x.shape[2:]))
        cat = torch.cat((self.aspp1(x), self.aspp2(x), self.aspp3(x), self.aspp4(x), x5), 1)
        return _mask(F.relu(self.bn1(self.conv1(cat))), mask)  # Dropout2d(0.5) as an explicit mask


# ------------------------------------------------------------------ decoder (decoder.py:9-48)
class ODecoder(nn.Module):
    def __init__(self, num_classes, backbone):
        super().__init__()
        low = {"resnet": 256, "resnet101": 256, "mobilenet": 24}[backbone]
        self.conv1 = nn.Conv2d(low, 48, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(48)
        self.last_conv = nn.Sequential(nn.Conv2d(304, 256, 3, 1, 1, bias=False), nn.BatchNorm2d(256), nn.ReLU(),
                                       nn.Conv2d(256, 256, 3, 1, 1, bias=False), nn.BatchNorm2d(256), nn.ReLU(),
                                       nn.Dropout2d(MC_DROPOUT_RATE), nn.Conv2d(256, num_classes, 1))

    def forward(self, x, low, mask=None):
        low = F.relu(self.bn1(self.conv1(low)))
        feats = torch.cat((_bilinear(x, low.shape[2:]), low), 1)
        lc = self.last_conv
        y = F.relu(lc[1](lc[0](feats)))
        y = _mask(F.relu(lc[4](lc[3](y))), mask)
        return lc[7](y), feats


# ------------------------------------------------------------------ assembly (deeplab.py:11-62)
class ODeepLab(nn.Module):
    """backbone: 'mobilenet' | 'resnet' (=ResNet-50, as the reference wires it) | 'resnet101'."""

    def __init__(self, backbone="mobilenet", output_stride=16, num_classes=19):
        super().__init__()
        if backbone == "mobilenet":
            self.backbone = OMobileNetV2(output_stride)
        elif backbone == "resnet":
            self.backbone = OResNet([3, 4, 6, 3], output_stride)
        elif backbone == "resnet101":
            self.backbone = OResNet([3, 4, 23, 3], output_stride)
        else:
            raise NotImplementedError
        self.aspp = OASPP(backbone, output_stride)
        self.decoder = ODecoder(num_classes, backbone)
        self.return_features = False

    def forward(self, x, masks=None, noise=None):
        """masks: None (dropout inactive) or (aspp_mask [N,256], decoder_mask [N,256]) multipliers.
        noise: None or draw(shape, scale) -> tensor: the gaussian feature noise of deeplab.py:39-56
        (input: |mean| * 0.05; backbone output, low-level features and ASPP output: |mean| * 0.5), drawn in that order."""
        m1, m2 = masks if masks is not None else (None, None)
        if noise is not None:
            x = x + noise(tuple(x.shape), abs(float(x.mean()) * 0.05))
        hi, low = self.backbone(x)
        if noise is not None:
            hi = hi + noise(tuple(hi.shape), abs(float(hi.mean()) * 0.5))
            low = low + noise(tuple(low.shape), abs(float(low.mean()) * 0.5))
        a = self.aspp(hi, m1)
        if noise is not None:
            a = a + noise(tuple(a.shape), abs(float(a.mean()) * 0.5))
        low_res, feats = self.decoder(a, low, m2)
        out = _bilinear(low_res, x.shape[2:])
        return (out, feats) if self.return_features else out


# ------------------------------------------------------------------ deterministic weights
def _hash_uniform(n, seed):
    """n reproducible uniforms in [0,1): 64-bit integer mix (splitmix64), no libm involved."""
    import numpy as np

    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def fill_state_dict(module, seed=0, randomize_bn_stats=True):
    """Closed-form weights: He-uniform convs, BN gamma in [0.5,1.5), small beta / running_mean,
    running_var in [0.5,1.5).  Identical on every machine, so fixtures only store outputs."""
    import zlib

    sd = module.state_dict()
    new = {}
    for name, t in sd.items():
        s = (zlib.crc32(name.encode()) + 1000003 * seed) & 0x7FFFFFFF
        u = torch.from_numpy(_hash_uniform(t.numel(), s)).reshape(t.shape) if t.numel() else t.clone()
        if name.endswith("num_batches_tracked"):
            new[name] = torch.zeros_like(t)
        elif t.dim() == 4:
            fan_in = t.shape[1] * t.shape[2] * t.shape[3]
            new[name] = (u * 2 - 1) * (6.0 / fan_in) ** 0.5
        elif name.endswith("running_var"):
            new[name] = (0.5 + u) if randomize_bn_stats else torch.ones_like(t)
        elif name.endswith("running_mean"):
            new[name] = ((u - 0.5) * 0.2) if randomize_bn_stats else torch.zeros_like(t)
        elif name.endswith("bn3.weight"):
            new[name] = 0.1 + 0.2 * u  # keep the 16/33-block residual stacks of ResNet-50/101 at O(1) magnitude
        elif name.endswith("weight"):
            new[name] = 0.5 + u
        else:  # BN beta, conv bias
            new[name] = (u - 0.5) * 0.2
    module.load_state_dict(new)
    return module


def dropout_masks(n, t, seed):
    """explicit Dropout2d multipliers for T passes: ([T,N,256] in {0,2}, [T,N,256] in {0,4/3})"""
    u1 = torch.from_numpy(_hash_uniform(t * n * 256, seed * 2 + 11)).reshape(t, n, 256)
    u2 = torch.from_numpy(_hash_uniform(t * n * 256, seed * 2 + 12)).reshape(t, n, 256)
    m1 = (u1 >= 0.5).float() * 2.0
    m2 = (u2 >= MC_DROPOUT_RATE).float() * (1.0 / (1.0 - MC_DROPOUT_RATE))
    return m1, m2


def synthetic_batch(n, h, w, num_classes, first_index=0):
    """per-image seeded synthetic Cityscapes-shaped tensors: content is independent of
    how the pool is sharded.  Top h//10 rows of the label are 255 (ignore)."""
    imgs, labs = [], []
    for i in range(n):
        g = torch.Generator().manual_seed(1000 + first_index + i)
        imgs.append(torch.randn(3, h, w, generator=g))
        lab = torch.randint(0, num_classes, (h, w), generator=g).float()
        lab[: h // 10] = 255
        labs.append(lab)
    return torch.stack(imgs), torch.stack(labs)
