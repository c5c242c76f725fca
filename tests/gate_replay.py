"""Gate replay for gradient parity tests: the HIP forward's own activation gates (ReLU: out > 0; ReLU6: 0 < out < 6), recorded per
conv + BN + activation site, are replayed inside the f64 oracle (forward values the oracle's own, backward g * gate), so both
sides differentiate the SAME piecewise-linear function: a pre-activation within rounding of a kink cannot move anything, and
every parameter gradient can be compared at the rounding level."""
import torch


def activation_values(ops, t):
    """f32 values of an activation handed out by conv_bn_act: the tensor itself, or -- for a rows-only output (sole_consumer=True:
    its f32 copy is never written) -- decoded from its attached two-part split rows: [rows + 1][ceil(C/32)][2 parts][32] f16, then a
    trailer whose first float is 1 / scale; x = (h0 + h1) / scale."""
    if not (hasattr(t, "__dict__") and t.__dict__.get("_dass_rows_only")):
        return t.detach()
    n, c, h, w = t.shape
    m, cc = n * h * w, (c + 31) // 32
    buf = ops.attached_x3(t, m, c)
    assert buf is not None and ops.x3_parts() == 2
    rows = buf[: (m + 1) * cc * 128].view(torch.float16).view(m + 1, cc, 2, 32)[:m]
    inv = buf[(m + 1) * cc * 128:(m + 1) * cc * 128 + 4].view(torch.float32)
    vals = (rows[:, :, 0].float() + rows[:, :, 1].float()) * inv          # [m, cc, 32]
    return vals.reshape(n, h, w, cc * 32)[..., :c].permute(0, 3, 1, 2)


class GateReplay(object):
    """records the gates of every ReLU / ReLU6 site of the HIP forward; replays them, matched by shape in call order, in place
    of torch.nn.functional.relu / hardtanh during the oracle's forward"""

    def __init__(self, ops):
        self.ops, self.gates, self.used = ops, [], []
        self._orig_cba, self._orig_relu, self._orig_ht = ops.conv_bn_act, torch.nn.functional.relu, torch.nn.functional.hardtanh

    def record(self):
        ops, rec = self.ops, self

        def wrapped(x, conv, bn=None, act=ops.ACT_NONE, **kw):
            out = rec._orig_cba(x, conv, bn, act, **kw)
            if act in (ops.ACT_RELU, ops.ACT_RELU6):
                first = out[0] if isinstance(out, tuple) else out  # fork=True: (out, the input again)
                d = activation_values(ops, first)
                rec.gates.append(((d > 0) if act == ops.ACT_RELU else ((d > 0) & (d < 6))).cpu())
            return out

        ops.conv_bn_act = wrapped

    def stop_recording(self):
        self.ops.conv_bn_act = self._orig_cba
        self.used = [False] * len(self.gates)

    def replay(self):
        rec = self

        class Gate(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, gate, hi):
                ctx.save_for_backward(gate)
                return x * gate if hi is None else x.clamp(0.0, hi)

            @staticmethod
            def backward(ctx, g):
                return g * ctx.saved_tensors[0], None, None

        def pick(x):
            for i, gt in enumerate(rec.gates):
                if not rec.used[i] and tuple(gt.shape) == tuple(x.shape):
                    rec.used[i] = True
                    return gt.to(x.dtype)
            raise AssertionError("no recorded gate of shape %s left" % (tuple(x.shape),))

        def relu(x, inplace=False):
            return Gate.apply(x, pick(x), None)

        def hardtanh(x, min_val=-1.0, max_val=1.0, inplace=False):
            assert min_val == 0.0 and max_val == 6.0, "only ReLU6 is replayed"
            return Gate.apply(x, pick(x), 6.0)

        torch.nn.functional.relu = relu
        torch.nn.functional.hardtanh = hardtanh

    def restore(self):
        torch.nn.functional.relu = self._orig_relu
        torch.nn.functional.hardtanh = self._orig_ht
        self.ops.conv_bn_act = self._orig_cba
