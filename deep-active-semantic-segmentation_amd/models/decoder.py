"""DeepLab-v3+ decoder on the HIP path -- mirror of models/decoder.py:9-60.

forward(x, low_level_feat) -> (low-res logits [N,classes,h,w], concat features [N,304,h,w]) as the
reference.  The bilinear(align_corners) upsample of the ASPP output is written straight into the
304-channel concat buffer; Dropout2d(MC_DROPOUT_RATE) is a per-(n,c) scale fused into the BN-apply
pass of the second 3x3; the classifier's bias rides in the conv epilogue.
"""
import torch.nn as nn

import constants
from dass_hip import ops
from models._common import channels_last_weights, dropout_mask_for, init_weights


class Decoder(nn.Module):

    def __init__(self, num_classes, backbone, batchnorm, mc_dropout):
        super(Decoder, self).__init__()
        if backbone in ('resnet', 'resnet101'):
            low_level_inplanes = 256
        elif backbone == 'xception':
            low_level_inplanes = 128
        elif backbone == 'mobilenet':
            low_level_inplanes = 24
        else:
            raise NotImplementedError

        self.conv1 = nn.Conv2d(low_level_inplanes, 48, 1, bias=False)
        self.bn1 = batchnorm(48)
        self.relu = nn.ReLU()
        # aspp always gives out 256 planes + 48 from conv1
        self.last_conv = nn.Sequential(nn.Conv2d(304, 256, kernel_size=3, stride=1, padding=1, bias=False),
                                       batchnorm(256),
                                       nn.ReLU(),
                                       nn.Conv2d(256, 256, kernel_size=3, stride=1, padding=1, bias=False),
                                       batchnorm(256),
                                       nn.ReLU(),
                                       nn.Dropout2d(p=constants.MC_DROPOUT_RATE),
                                       nn.Conv2d(256, num_classes, kernel_size=1, stride=1))
        self._init_weight()

    def features(self, x, low_level_feat):
        """[upsampled ASPP output | 48-ch low-level] -- the core-set feature tensor (decoder.py:41-46)"""
        low = ops.conv_bn_act(low_level_feat, self.conv1, self.bn1, ops.ACT_RELU, emit_x3=False)
        return ops.upsample_cat(x, low)

    def head(self, feats, dropout_mask=None, in_scale=None, mask_as_in_scale=None):
        lc = self.last_conv
        h = ops.conv_bn_act(feats, lc[0], lc[1], ops.ACT_RELU, in_scale=in_scale, consumer=lc[3])
        if mask_as_in_scale is not None:  # inference: fold the Dropout2d mask into the classifier's loader
            h = ops.conv_bn_act(h, lc[3], lc[4], ops.ACT_RELU, emit_x3=False)
            return ops.conv_bn_act(h, lc[7], in_scale=mask_as_in_scale)
        mask = dropout_mask if dropout_mask is not None else dropout_mask_for(lc[6], feats.shape[0], 256, feats.device)
        h = ops.conv_bn_act(h, lc[3], lc[4], ops.ACT_RELU, nc_scale=mask, emit_x3=False)  # consumer: the 19-class 1x1
        return ops.conv_bn_act(h, lc[7])

    # ---- MC-dropout scoring: last_conv[0] split along its input channels.  Only the 256 upsampled-ASPP channels carry a
    # Dropout2d mask; the 48 low-level channels' share of the 3x3 304->256 conv (16 % of it) is the same in every pass,
    # so it is computed once per batch -- already multiplied by the eval-BN scale -- and enters each pass as the
    # residual of the 256-channel conv's epilogue:  relu((Wa*(m.xa) + Wb*xb)*s + b) = relu(Wa*(m.xa)*s + b + (Wb*xb)*s).
    def _split_first_conv(self):
        conv = self.last_conv[0]
        wt = conv.weight
        key = (wt.data_ptr(), wt._version, ops.f32_mma(), ops.x3_pipeline())
        hit = self.__dict__.get("_dass_split_w")
        if hit is None or hit[0] != key:
            krsc = wt.detach().float().permute(0, 2, 3, 1)
            wa = ops.prepare_conv_weight(krsc[..., :256].contiguous(), x3=ops.x3_pipeline())
            wb = ops.prepare_conv_weight(krsc[..., 256:].contiguous(), x3=ops.x3_pipeline())
            la = ops.weight_l1(krsc[..., :256].contiguous()) if ops.x3_pipeline() and ops.x3_parts() == 2 else None
            hit = self.__dict__["_dass_split_w"] = (key, wa, wb, la)
        return hit[1], hit[2], hit[3]

    def head_mc_prepare(self, feats):
        """-> state for head_mc_pass: operands + eval-BN vectors + the deterministic low-level share (f32 tensors only)"""
        import torch

        n, c, h, w = feats.shape
        assert c == 304 and feats.dtype == torch.float32
        wa, wb, la = self._split_first_conv()
        st = ops.bn_eval_state(self.last_conv[1], 256, feats.device)
        xb, ldb = ops.rows(feats[:, 256:])
        yb = ops.new_act(n, 256, h, w, torch.float32, feats.device)
        yb_amax = torch.zeros((4,), dtype=torch.float32, device=feats.device) if la is not None else None  # max |yb|: part of every pass's output bound
        if ops.x3_pipeline():
            ops.conv_x3_launch(ops.split3_rows(xb, ldb, n * h * w, 48), wb, yb, 256, (n, h, w, 48, h, w, 256, 3, 3, 1, 1, 1), scale=st.scale,
                               y_amax=yb_amax)
        else:
            ops.conv_launch(xb, ldb, wb, yb, 256, (n, h, w, 48, h, w, 256, 3, 3, 1, 1, 1), scale=st.scale)
        # max |xa| of the 256 masked channels BEFORE masking: with the largest Dropout2d multiplier it bounds every pass's operand
        xa, lda = ops.rows(feats[:, :256])
        xa_amax = ops.absmax_rows(xa, lda, n * h * w, 256) if la is not None else None
        return wa, st, yb, la, yb_amax, xa_amax

    def head_mc_pack(self, prep, masks1):
        """the Dropout2d-sparse operands of ALL T passes in two launches: masks1 [T, N, 256] -> per pass (order, limit,
        per-image weight copies); None when the sparse pre-split path is off"""
        if not (ops.x3_pipeline() and ops.mc_sparse()):
            return None
        wa = prep[0]
        t, n, c = masks1.shape
        order, lim = ops.dropout_pack(masks1.reshape(t * n, c))
        if ops.x3_parts() == 2:  # every operand ends in its own trailer (the weights' scale): one buffer per pass
            # one bound for the masked operand of ALL T passes: max |xa| (head_mc_prepare) x the largest mask multiplier, on the device
            bnd = (prep[5] * masks1.amax()).reshape(1) if prep[5] is not None else None
            return [(order[i * n:(i + 1) * n], lim[i * n:(i + 1) * n],
                     ops.w3_pack_per_image(wa, 256 * 9, 256, order[i * n:(i + 1) * n], lim[i * n:(i + 1) * n]), bnd) for i in range(t)]
        wan = ops.w3_pack_per_image(wa, 256 * 9, 256, order, lim)
        per = (wan.numel() - 16) // t
        return [(order[i * n:(i + 1) * n], lim[i * n:(i + 1) * n], wan[i * per:(i + 1) * per], None) for i in range(t)]

    def head_mc_all(self, feats, prep, masks1, masks2):
        """ALL T stochastic passes of last_conv as one launch per conv (two-part pre-split engine with the Dropout2d-sparse operands):
        masks1 / masks2 [T, N, 256] -> low-resolution logits [T * N, classes, h, w], pass-major.  The T x N (pass, image) pairs are
        the "images" of the per-image conv -- each with its own packed operand rows and weight copy, all sharing the batch's
        deterministic low-level share as residual -- and of the dense second conv and the classifier: T x the tiles of one pass
        fill the chip for ~40 rounds, where every single pass ended in a partly filled round + a stream-K fix-up launch.
        None when that path is off (the caller then runs head_mc_pass per pass)."""
        import torch

        wa, st, yb, la, yb_amax, xa_amax = prep
        if not (ops.x3_pipeline() and ops.mc_sparse() and ops.x3_parts() == 2 and la is not None and xa_amax is not None):
            return None
        n, c, h, w = feats.shape
        t = masks1.shape[0]
        v, rpi = t * n, h * w
        m = v * rpi
        if m * 1024 >= (1 << 32) - (1 << 20):   # the operand's 32-bit byte offsets (8 slabs x 128 B per row): the caller cuts T
            return None
        lc = self.last_conv
        m1 = masks1.reshape(v, 256).contiguous()
        order, lim = ops.dropout_pack(m1)
        wan = ops.w3_pack_per_image(wa, 256 * 9, 256, order, lim)
        bnd = (xa_amax * masks1.amax()).reshape(1)          # one bound for the masked operand of all passes, on the device
        xa, lda = ops.rows(feats[:, :256])
        xa3 = ops.split3_rows_packed_rep(xa, lda, m, 256, m1, order, lim, rpi, n, bnd)
        h1_3 = ops.x3_alloc(m, 256, feats.device)
        ops.x3_prepare_out(h1_3, m, 256, la, st.scale, st.shift, xa3, m, 256, ops._p(yb_amax), ops.ACT_RELU)
        dims = (v, h, w, 256, h, w, 256, 3, 3, 1, 1, 1)
        ops.conv_x3_per_image_rep_launch(xa3, wan, lim, None, 256, dims, n, y3=h1_3, scale=st.scale, shift=st.shift, residual=yb, ldr=256,
                                         act=ops.ACT_RELU)
        st2 = ops.bn_eval_state(lc[4], 256, feats.device)
        h2 = ops.new_act(v, 256, h, w, torch.float32, feats.device)
        ops.conv_x3_launch(h1_3, ops.weight_operand(lc[3].weight, 0, torch.float32, cpad=256, x3=True), h2, 256, dims,
                           scale=st2.scale, shift=st2.shift, act=ops.ACT_RELU)
        return ops.conv_bn_act(h2, lc[7], in_scale=masks2.reshape(v, 256).contiguous())

    def head_mc_pass(self, feats, prep, m1, m2, packed=None):
        """one stochastic pass of last_conv: masks m1 (ASPP Dropout2d, [N,256]) and m2 (last_conv[6]) folded into loaders;
        packed: this pass's entry of head_mc_pack (else the operands are packed here)"""
        import torch

        wa, st, yb, la, yb_amax, xa_amax = prep
        n, c, h, w = feats.shape
        lc = self.last_conv
        xa, lda = ops.rows(feats[:, :256])
        dims = (n, h, w, 256, h, w, 256, 3, 3, 1, 1, 1)
        if ops.x3_pipeline():
            # pre-split engine: the ASPP mask rides in the conversion pass of the 256 masked channels (exact: the
            # multipliers are 0 and 2), the first conv hands its result to the second as x3 rows (no f32 round trip)
            # A channel the mask drops contributes exact zeros: the survivors of every image are packed to the front
            # (operand rows and a per-image copy of the weights alike) and the reduction stops after them -- about half
            # of this conv's multiplications are never issued.
            m = n * h * w
            m1 = m1.contiguous()
            # the first conv hands its result to the second as x3 rows from its epilogue (two-part format: the rows' scale is fixed
            # before the launch by x3_prepare_out from a bound of the result -- L1 norms of wa x max |masked xa| + shift + max |yb|)
            h1_3 = ops.x3_alloc(m, 256, feats.device)
            h1 = None

            def prepare(xa3):
                if la is not None:
                    ops.x3_prepare_out(h1_3, m, 256, la, st.scale, st.shift, xa3, m, 256, ops._p(yb_amax), ops.ACT_RELU)

            if ops.mc_sparse():
                bnd = None
                if packed is not None:
                    order, lim, wan, bnd = packed
                else:
                    order, lim = ops.dropout_pack(m1)
                    wan = ops.w3_pack_per_image(wa, 256 * 9, 256, order, lim)
                xa3 = ops.split3_rows_packed(xa, lda, m, 256, m1, order, lim, h * w, bound=bnd)
                prepare(xa3)
                ops.conv_x3_per_image_launch(xa3, wan, lim, h1, 256, dims, y3=h1_3, scale=st.scale, shift=st.shift, residual=yb,
                                             ldr=256, act=ops.ACT_RELU)
            else:
                xa3 = ops.split3_rows(xa, lda, m, 256, nc_scale=m1, rows_per_image=h * w)
                prepare(xa3)
                ops.conv_x3_launch(xa3, wa, h1, 256, dims, y3=h1_3, scale=st.scale, shift=st.shift, residual=yb, ldr=256,
                                   act=ops.ACT_RELU)
            st2 = ops.bn_eval_state(lc[4], 256, feats.device)
            h2 = ops.new_act(n, 256, h, w, torch.float32, feats.device)
            ops.conv_x3_launch(h1_3, ops.weight_operand(lc[3].weight, 0, torch.float32, cpad=256, x3=True), h2, 256, dims,
                               scale=st2.scale, shift=st2.shift, act=ops.ACT_RELU)
            return ops.conv_bn_act(h2, lc[7], in_scale=m2)
        h1 = ops.new_act(n, 256, h, w, torch.float32, feats.device)
        ops.conv_launch(xa, lda, wa, h1, 256, dims, scale=st.scale, shift=st.shift,
                        residual=yb, ldr=256, in_scale=m1.contiguous(), act=ops.ACT_RELU)
        h2 = ops.conv_bn_act(h1, lc[3], lc[4], ops.ACT_RELU, emit_x3=False)
        return ops.conv_bn_act(h2, lc[7], in_scale=m2)

    @ops.bn_counter_scope
    def forward(self, x, low_level_feat, dropout_mask=None):
        second_to_last_features = self.features(x, low_level_feat)
        return self.head(second_to_last_features, dropout_mask), second_to_last_features

    def _init_weight(self):
        init_weights(self)
        channels_last_weights(self)
