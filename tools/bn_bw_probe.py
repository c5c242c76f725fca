#!/usr/bin/env python
"""HBM rates of the train step's BN passes per tensor shape, next to a plain device copy of the same bytes (the streaming rate this chip
gives a trivially simple kernel at that size): dass_bn_apply_train (rows only / f32 + rows / with a residual and gate bits) and the lean
dass_bn_bwd_apply_sums (gate from the conv output, split rows out).   python tools/bn_bw_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import torch  # noqa: E402
from dass_hip import ops  # noqa: E402
from dass_hip._lib import check, lib  # noqa: E402

ops.set_f32_mma("f16x3")
_p, st = ops._p, ops._stream


def timeit(f, reps=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


print("%-16s | %-22s | %-24s | %-24s | %-26s | %-24s" % ("tensor", "copy in->out", "apply: rows only", "apply: f32 + rows", "apply: + residual + gates", "bwd apply (lean)"))
for m, k in ((8712, 256), (8712, 1024), (8712, 2048), (33800, 128), (33800, 512), (133128, 64), (133128, 256)):
    dev = "cuda"
    x = torch.randn((m, k), device=dev)
    out = torch.empty_like(x)
    res = torch.randn_like(x)
    sums = torch.stack((x.double().sum(0), (x.double() ** 2).sum(0))).contiguous()
    g, b = torch.rand(k, device=dev) + 0.5, torch.randn(k, device=dev)
    mean, inv, sc, sh = (torch.empty(k, device=dev) for _ in range(4))
    out3 = ops.x3_alloc_for(m, k, x.device)
    gates = torch.empty((m, k // 4), dtype=torch.uint8, device=dev)
    rb = torch.full((1,), 6.0, device=dev)
    e = m * k * 4 / 1e6  # MB per f32 pass over the tensor

    def apply(o, r, gt):
        check(lib.dass_bn_apply_train(_p(x), k, _p(o), k, _p(sums), float(m), _p(g), _p(b), None, None, -1.0, 1e-5, _p(mean), _p(inv), _p(sc), _p(sh),
                                      _p(r), k if r is not None else 0, None, m, k, m // 8, ops.ACT_RELU, 0, _p(out3), _p(gt), gt.numel() if gt is not None else 0,
                                      _p(rb) if r is not None else None, st()), "apply")

    t_copy = timeit(lambda: out.copy_(x))
    t_a = timeit(lambda: apply(None, None, None))
    t_b = timeit(lambda: apply(out, None, None))
    t_c = timeit(lambda: apply(out, res, gates))
    apply(out, None, None)
    dy = torch.randn_like(x)
    dy3 = ops.x3_alloc_for(m, k, x.device)
    bs = torch.zeros((3, k), dtype=torch.float64, device=dev)
    dbeta, dgamma = torch.empty(k, device=dev), torch.empty(k, device=dev)
    check(lib.dass_bn_bwd_reduce_sums(_p(dy), k, None, k, _p(x), k, _p(mean), _p(inv), _p(sc), _p(sh), None, m, k, m // 8, ops.ACT_RELU, _p(bs), None, 0, 0, st()), "reduce")

    def bwd():
        check(lib.dass_bn_bwd_apply_sums(_p(dy), k, None, k, _p(x), k, _p(mean), _p(inv), _p(g), _p(bs), _p(dbeta), _p(dgamma), _p(sc), _p(sh), None,
                                         None, k, None, k, m, k, m // 8, float(m), ops.ACT_RELU, None, 0, 0, _p(dy3), st()), "bwd")

    try:
        t_d = timeit(bwd)
    except RuntimeError as exc:
        t_d = float("nan")
        print("   (bwd probe failed: %s)" % exc)
    f = lambda mb, t: "%6.1f us %5.2f TB/s" % (t, mb / t)  # noqa: E731
    print("%6d x %4d     | %s | %s | %s | %s | %s" % (m, k, f(2 * e, t_copy), f(2 * e, t_a), f(3 * e, t_b), f(4.25 * e, t_c), f(3 * e, t_d)), flush=True)
