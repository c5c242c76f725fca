#!/usr/bin/env python
"""Soak of the multi-stream train step: the same 60 steps of DeepLab-R101 513^2 batch 8 on a learnable target with the chunked
weight gradients on the side stream (default) and on the caller's stream.  A cross-stream race (a block recycled while the
other stream still reads it, a missing event) would show as NaN, a loss that stops falling, or early steps that disagree;
the two runs otherwise differ only by the order of f32 atomic additions.   python tools/soak_streams.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import torch  # noqa: E402
from dass_hip import ops  # noqa: E402
from dass_hip.optim import SGD  # noqa: E402
from models.deeplab import DeepLab  # noqa: E402
from utils.loss import SegmentationLosses  # noqa: E402


def run(side, steps=60):
    ops.set_wgrad_chunk(16, side)
    torch.manual_seed(0)
    m = DeepLab(backbone="resnet101", output_stride=16, num_classes=19, sync_bn=False, freeze_bn=False, pretrained=False).cuda().train()
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    opt = SGD([{"params": m.get_1x_lr_params(), "lr": 0.01}, {"params": m.get_10x_lr_params(), "lr": 0.1}], momentum=0.9, weight_decay=5e-4)
    g = torch.Generator(device="cuda").manual_seed(1)
    y = (torch.arange(513, device="cuda") * 19 // 513).float().view(1, 1, 513).expand(8, 513, 513).contiguous()
    x = torch.randn(8, 3, 513, 513, device="cuda", generator=g) * 0.3 + (y / 9.0 - 1.0).unsqueeze(1)
    losses = []
    for i in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = crit(m(x), y)
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    torch.cuda.synchronize()
    out = [float(v) for v in losses]
    print("side=%s: loss %s ... %s | reserved %.2f GB" % (side, " ".join("%.4f" % v for v in out[:4]), " ".join("%.4f" % v for v in out[-3:]),
                                                        torch.cuda.memory_reserved() / 2 ** 30), flush=True)
    return out


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    a, b, c = run(True, steps), run(False, steps), run(False, steps)
    assert all(v == v and v < 1e3 for v in a + b + c), "non-finite loss"
    assert a[-1] < 0.5 * a[0] and b[-1] < 0.5 * b[0], "the loss must fall on the learnable target"
    rel = lambda u, v: [abs(p - q) / max(abs(q), 1e-6) for p, q in zip(u[:4], v[:4])]  # noqa: E731
    print("steps 0-3, side vs main: %s" % " ".join("%.1e" % r for r in rel(a, b)))
    print("steps 0-3, main vs main: %s   (run-to-run: the order of f32 atomic additions)" % " ".join("%.1e" % r for r in rel(c, b)))
    assert rel(a, b)[0] <= 1e-5, "step 0 is a forward pass of identical weights"
    assert max(rel(a, b)) <= 5 * max(max(rel(c, b)), 1e-4), "the side stream must not add to the run-to-run spread"
    print("soak ok: final losses %.4f / %.4f / %.4f" % (a[-1], b[-1], c[-1]))
