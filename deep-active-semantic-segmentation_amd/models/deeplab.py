"""DeepLab-v3+ on the MI355X HIP path -- drop-in mirror of models/deeplab.py:11-89.

Same constructor, attributes (backbone / aspp / decoder / return_features / noisy_features /
model_name), methods (set_return_features, set_noisy_features, freeze_bn, get_1x_lr_params,
get_10x_lr_params) and state_dict keys; forward(input[N,C,H,W] f32) -> logits[N,classes,H,W] f32
(or (logits, feats[N,304,H/4,W/4])).  Everything between the NCHW input and the NCHW logits runs in
NHWC through libdass_hip.

Additions (do not change the reference surface):
  * backbone='resnet101' (the benchmark config; 'resnet' stays ResNet-50 like the reference),
  * forward(..., dropout_masks=(m_aspp[N,256], m_dec[N,256])) for reproducible stochastic passes,
  * mc_dropout_votes(): the T-pass scoring tail with the deterministic prefix computed once
    (SURVEY.md 8a notes i-iii), used by active_selection.mc_dropout.
"""
import os

import torch
import torch.nn as nn

from dass_hip import ops
from models.aspp import ASPP
from models.backbone import build_backbone
from models.decoder import Decoder
from models.sync_batchnorm.batchnorm import SynchronizedBatchNorm2d


class DeepLab(nn.Module):

    def __init__(self, backbone='mobilenet', output_stride=16, num_classes=19, sync_bn=True, freeze_bn=False,
                 mc_dropout=False, input_channels=3, pretrained=True):
        super(DeepLab, self).__init__()
        if sync_bn == True:  # noqa: E712  (reference idiom, deeplab.py:17)
            batchnorm = SynchronizedBatchNorm2d
        else:
            batchnorm = nn.BatchNorm2d

        self.backbone = build_backbone(backbone, output_stride, batchnorm, mc_dropout, input_channels, pretrained)
        self.aspp = ASPP(backbone, output_stride, batchnorm)
        self.decoder = Decoder(num_classes, backbone, batchnorm, mc_dropout)
        self.return_features = False
        self.noisy_features = False
        self.model_name = 'deeplab'
        self.num_classes = num_classes
        self.noise_source = None  # optional draw(shape, sigma) -> tensor replacing the device RNG (reproducible noise passes)
        if freeze_bn:
            self.freeze_bn()

    def set_return_features(self, return_features):
        self.return_features = return_features

    def set_noisy_features(self, noisy_features):
        self.noisy_features = noisy_features

    def _noise_like(self, t, frac):
        # deeplab.py:39-56 draws numpy gaussians on the host (sigma = |mean| * frac); the same distribution is drawn on
        # the device here, or taken from `noise_source(shape, sigma)` when one is set (tests replay numpy's stream)
        sigma = abs(float(t.float().mean()) * frac)
        if self.noise_source is not None:
            return self.noise_source(tuple(t.shape), sigma).to(device=t.device, dtype=torch.float32)
        return torch.randn_like(t.float()) * sigma

    @ops.bn_counter_scope
    def forward(self, input, dropout_masks=None):
        m_aspp, m_dec = dropout_masks if dropout_masks is not None else (None, None)
        if self.noisy_features is True:
            input = input + self._noise_like(input, 0.05)

        x, low_level_feat = self.backbone(input)

        if self.noisy_features is True:
            x = x + self._noise_like(x, 0.5).to(x.dtype)
            low_level_feat = low_level_feat + self._noise_like(low_level_feat, 0.5).to(low_level_feat.dtype)

        x = self.aspp(x, dropout_mask=m_aspp)

        if self.noisy_features is True:
            x = x + self._noise_like(x, 0.5).to(x.dtype)

        low_res_x, features = self.decoder(x, low_level_feat, dropout_mask=m_dec)
        x = ops.upsample_to_nchw(low_res_x, input.shape[2], input.shape[3])
        if self.return_features:
            return x, features
        return x

    # ------------------------------------------------------------------ scoring fast paths (inference only)
    def _bn_all_eval(self):
        return not any(m.training for m in self.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm))

    @torch.no_grad()
    def encoder_features(self, input):
        """the 304-channel decoder feature map without the dead last_conv / final upsample
        (core_set.py:60-61 only consumes the features; SURVEY.md 8a note ii)"""
        x, low = self.backbone(input)
        x = self.aspp(x, apply_dropout=False)
        return self.decoder.features(x, low)

    @torch.no_grad()
    def mc_dropout_votes(self, input, steps, masks=None, generator=None):
        """uint8 votes [N, steps, H, W] = argmax of `steps` stochastic forwards (mc_dropout.py:37-40).

        Only Dropout2d after aspp.bn1 (p=0.5) and decoder.last_conv.6 (p=MC_DROPOUT_RATE) are stochastic
        for models built the way active_train.py:48 builds them, so backbone + ASPP + decoder.conv1 +
        the upsample/concat run ONCE; each pass re-runs last_conv with its masks folded into the conv
        loaders (upsample(m*x) == m*upsample(x)) and ends in the fused upsample+argmax kernel.
        masks: optional ([T,N,256], [T,N,256]) multipliers; default: Bernoulli draws like nn.Dropout2d."""
        return self.mc_tail(self.mc_prefix(input), steps, masks=masks, generator=generator)

    @torch.no_grad()
    def mc_prefix(self, input):
        """the deterministic part of mc_dropout_votes, run once per batch: backbone + ASPP + decoder features, and the share
        of last_conv[0] that no mask touches.  -> state for mc_tail.  (Split out so that a caller scoring many batches can run
        the prefix of batch i + 1 on another HIP stream while the T passes of batch i run: active_selection/mc_dropout.py)"""
        assert self._bn_all_eval(), "mc_dropout_votes needs eval-mode BN (model.eval() + dropout switched on)"
        feats = self.encoder_features(input)
        # f32 tensors: the unmasked low-level channels' share of last_conv[0] is hoisted out of the T passes as well
        prep = self.decoder.head_mc_prepare(feats) if feats.dtype == torch.float32 and feats.shape[1] == 304 else None
        return feats, prep, tuple(input.shape)

    @torch.no_grad()
    def mc_tail(self, state, steps, masks=None, generator=None):
        """the `steps` stochastic passes over a mc_prefix state -> uint8 votes [N, steps, H, W]"""
        feats, prep, (n, _, hh, ww) = state
        dev = feats.device
        p1, p2 = self.aspp.dropout.p, self.decoder.last_conv[6].p
        votes = torch.empty((n, steps, hh, ww), dtype=torch.uint8, device=dev)
        ones48 = torch.ones((n, 48), dtype=torch.float32, device=dev)
        if masks is None:
            # all T x 2 Bernoulli draws in one launch sequence (8 tiny kernels per pass otherwise: ~40 us of a 1.4 ms pass)
            draws = torch.rand((2, steps, n, 256), device=dev, generator=generator)
            masks = ((draws[0] >= p1).to(torch.float32) * (1.0 / (1.0 - p1)), (draws[1] >= p2).to(torch.float32) * (1.0 / (1.0 - p2)))
        masks = (masks[0].to(dev).float().contiguous(), masks[1].to(dev).float().contiguous())
        if prep is not None and os.environ.get("DASS_MC_BATCHED", "1") == "1":
            # all T passes as ONE launch per conv (decoder.head_mc_all); T is cut only where the operand would outgrow 32-bit offsets
            rows = n * feats.shape[2] * feats.shape[3]
            tc = max(1, min(steps, ((1 << 32) - (1 << 21)) // 1024 // rows))
            done = True
            for t0 in range(0, steps, tc):
                t1 = min(steps, t0 + tc)
                low = self.decoder.head_mc_all(feats, prep, masks[0][t0:t1], masks[1][t0:t1])
                if low is None:
                    done = False
                    break
                for t in range(t0, t1):
                    ops.upsample_argmax(low[(t - t0) * n:(t - t0 + 1) * n], hh, ww, votes, t)
            if done:
                return votes
        packs = self.decoder.head_mc_pack(prep, masks[0][:steps]) if prep is not None else None

        def one_pass(t):
            m1, m2 = masks[0][t], masks[1][t]
            if prep is not None:
                low_res = self.decoder.head_mc_pass(feats, prep, m1, m2, packs[t] if packs is not None else None)
            else:
                low_res = self.decoder.head(feats, in_scale=torch.cat((m1, ones48), dim=1), mask_as_in_scale=m2)
            ops.upsample_argmax(low_res, hh, ww, votes, t)

        # The T passes are independent of each other.  Their two 3x3 convs are 1056 tiles of 256 x 128 on 256 CUs -- 4.1 rounds --
        # so every launch ends with most of the chip idle; dealing the passes over DASS_MC_STREAMS (default 2) HIP streams lets another
        # pass's launch fill that tail.  Pass 0 runs on the caller's stream (it also fills the operand caches the others read).
        sides = ops.mc_side_streams(dev) if steps > 1 else []
        one_pass(0)
        if not sides:
            for t in range(1, steps):
                one_pass(t)
            return votes
        main = torch.cuda.current_stream(dev)
        fork = torch.cuda.Event()
        fork.record(main)
        for st in sides:
            st.wait_event(fork)
        lanes = [None] + sides  # None = the caller's stream
        for t in range(1, steps):
            st = lanes[t % len(lanes)]
            if st is None:
                one_pass(t)
            else:
                with torch.cuda.stream(st):
                    one_pass(t)
        for st in sides:
            join = torch.cuda.Event()
            join.record(st)
            main.wait_event(join)
        return votes

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, SynchronizedBatchNorm2d):
                m.eval()
            elif isinstance(m, nn.BatchNorm2d):
                m.eval()

    def _lr_params(self, modules):
        for mod in modules:
            for m in mod.named_modules():
                if isinstance(m[1], nn.Conv2d) or isinstance(m[1], SynchronizedBatchNorm2d) \
                        or isinstance(m[1], nn.BatchNorm2d):
                    for p in m[1].parameters():
                        if p.requires_grad:
                            yield p

    def get_1x_lr_params(self):
        return self._lr_params([self.backbone])

    def get_10x_lr_params(self):
        return self._lr_params([self.aspp, self.decoder])
