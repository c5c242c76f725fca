"""CPU oracle for BASELINE config 0: U-Net(3, 4) on 128 x 128 synthetic tensors, batch 2, stock PyTorch fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Config 0 is the reference's CPU-runnable plumbing case
(`train.py` with `--architecture unet`; SURVEY.md 8d "config 0"): it has no GPU path, so there is nothing for the HIP
library to replace -- it is kept here as the CPU leg (iv) of bench.py's `cpu_baseline` and as a pinned fixture
(tests/golden/unet_config0.npz, written by oracle/make_goldens_r3.py from the REFERENCE's models/unet.py).

Restates models/unet.py:7-71 with its state_dict keys:
  four encoder stages of (3x3 conv + bias -> BN -> ReLU) x 2 at 32 / 64 / 128 / 256 channels, 2x2 max-pool between them,
  three decoder stages: bilinear (align_corners) upsample to the skip's size, concat [up, skip], double conv,
  1x1 classifier, bilinear upsample to the input size (a no-op resize when the sizes already match).
  The reference's `self.dropout` is constructed but never called (unet.py:28), so it does not appear here either.
  Init (unet.py:58-71): kaiming-normal conv weights, zero biases, BN gamma 1 / beta 0.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _stage(cin, cout):
    # indices 0..5 of the Sequential give the reference's parameter names (…​.0.weight, ….1.running_mean, ….3.bias, …)
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(),
                         nn.Conv2d(cout, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU())


def _up(x, like):
    return F.interpolate(x, size=like.shape[2:], mode="bilinear", align_corners=True)


class OUNet(nn.Module):
    WIDTHS = (32, 64, 128, 256)

    def __init__(self, in_channels=3, num_classes=4):
        super().__init__()
        w = self.WIDTHS
        self.dconv_down1 = _stage(in_channels, w[0])
        self.dconv_down2 = _stage(w[0], w[1])
        self.dconv_down3 = _stage(w[1], w[2])
        self.dconv_down4 = _stage(w[2], w[3])
        self.dconv_up3 = _stage(w[2] + w[3], w[2])
        self.dconv_up2 = _stage(w[1] + w[2], w[1])
        self.dconv_up1 = _stage(w[1] + w[0], w[0])
        self.conv_last = nn.Conv2d(w[0], num_classes, 1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        s1 = self.dconv_down1(x)
        s2 = self.dconv_down2(F.max_pool2d(s1, 2))
        s3 = self.dconv_down3(F.max_pool2d(s2, 2))
        y = self.dconv_down4(F.max_pool2d(s3, 2))
        y = self.dconv_up3(torch.cat((_up(y, s3), s3), 1))
        y = self.dconv_up2(torch.cat((_up(y, s2), s2), 1))
        y = self.dconv_up1(torch.cat((_up(y, s1), s1), 1))
        return _up(self.conv_last(y), x)


def config0_batch(seed=0, n=2, hw=128, num_classes=4):
    """the synthetic batch of config 0: N(0,1) images, uniform labels with the top hw//10 rows ignored (255)"""
    g = torch.Generator().manual_seed(4000 + seed)
    x = torch.randn(n, 3, hw, hw, generator=g)
    y = torch.randint(0, num_classes, (n, hw, hw), generator=g).float()
    y[:, : hw // 10] = 255
    return x, y


def config0_steps(model, steps=3, lr=0.01, seed=0):
    """`steps` SGD steps (momentum 0.9, weight decay 5e-4, the reference trainer's optimizer settings, train.py:56-57,261-266)
    on one fixed config-0 batch with the reference's CE loss (utils/loss.py:39-51).  -> list of loss values"""
    from oracle import selection_cpu as S

    x, y = config0_batch(seed)
    opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=5e-4, nesterov=False)
    model.train()
    losses = []
    for _ in range(steps):
        opt.zero_grad()
        loss = S.ce_loss(model(x), y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses
