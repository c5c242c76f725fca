#!/usr/bin/env python
"""Per-family kernel time and HBM fraction of a DeepLab-MobileNetV2 train step (BASELINE config C) from a rocprofv3 rocpd file:
    python tools/mbv2_families.py results.db STEPS [batch size classes] [--after sgd_multi N]
Families: depthwise 3x3 (forward / input gradient / weight gradient), dense convs (pointwise 1x1 + the ASPP / decoder 3x3: pre-split
and classic kernels, grouped weight gradients), BN passes, everything else.  The ALGORITHMIC bytes of a family are its tensors
moved once per pass in f32 (mobilenet.py:33-79: per InvertedResidual an expand 1x1 over the zero-padded input, a depthwise 3x3, a
linear 1x1): depthwise = hidden tensor in + out per pass; BN = the SURVEY 8d four passes per train-mode tensor; dense = in + out
per pass of every groups=1 conv.  GB/s = bytes / family time; fraction of the 8 TB/s HBM3E peak (MI355X_MICROARCH.md)."""
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
after = None
argv = list(sys.argv)
if "--after" in argv:
    i = argv.index("--after")
    after = (argv[i + 1], int(argv[i + 2]))
    del argv[i:i + 3]
db = sqlite3.connect(argv[1])
steps = float(argv[2])
batch, size, classes = (int(v) for v in argv[3:6]) if len(argv) > 5 else (16, 513, 21)
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
where = ""
if after is not None:
    marks = [r[0] for r in cur.execute("select start from kernels where %s like ? order by start" % name_col, ("%" + after[0] + "%",))]
    where = " where start > %d" % marks[after[1] - 1]
if steps <= 0:  # 0: count the steps in the window (one ce_fwd_kernel launch per train step)
    steps = float(cur.execute("select count(*) from kernels%s%s like '%%ce_fwd_kernel%%'" % (where, (" and " if where else " where ") + name_col)).fetchone()[0]) or 1.0
rows = cur.execute("select %s, count(*), sum(end - start) from kernels%s group by %s" % (name_col, where, name_col)).fetchall()


def out(h, k, s, p, d):
    return (h + 2 * p - d * (k - 1) - 1) // s + 1


# tensors of the network (elements per step), as bench.py:mobilenet_train_bytes walks them
h = out(size, 3, 2, 1, 1)
dense_e = batch * (3 * size * size + 32 * h * h)
bn_e = batch * 32 * h * h
dw_e = 0
cin, cur_s, rate, low_h = 32, 2, 1, None
for t, c, n, s_ in [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]:
    if cur_s == 16:
        stride, dil = 1, rate
        rate *= s_
    else:
        stride, dil = s_, 1
        cur_s *= s_
    for i in range(n):
        st = stride if i == 0 else 1
        hid, hp = cin * t, h + 2 * dil
        if t != 1:
            dense_e += batch * (cin + hid) * hp * hp
            bn_e += batch * hid * hp * hp
        ho = out(hp, 3, st, 0, dil)
        dw_e += batch * hid * (hp * hp + ho * ho)
        dense_e += batch * (hid + c) * ho * ho
        bn_e += batch * (hid + c) * ho * ho
        cin, h = c, ho
    if c == 24:
        low_h = h
dense_e += 4 * batch * (320 + 256) * h * h + batch * (1280 + 256) * h * h
bn_e += 5 * batch * 256 * h * h
dense_e += batch * ((24 + 48) + (304 + 256) + 512 + (256 + classes)) * low_h * low_h
bn_e += batch * (48 + 512) * low_h * low_h
alg = {"depthwise fwd": dw_e * 4, "depthwise input gradient": dw_e * 4, "depthwise weight gradient": dw_e * 4,
       "dense convs (fwd + input gradient + weight gradient)": 3 * dense_e * 4, "BN passes (apply, backward reduce, backward apply, sums)": 4 * bn_e * 4}
fam_of = [("dw_fwd", "depthwise fwd"), ("dw_bwd_data", "depthwise input gradient"), ("dw_bwd_weight", "depthwise weight gradient"),
          ("conv_x3", "dense convs (fwd + input gradient + weight gradient)"), ("conv_igemm", "dense convs (fwd + input gradient + weight gradient)"),
          ("wgrad", "dense convs (fwd + input gradient + weight gradient)"), ("bn_", "BN passes (apply, backward reduce, backward apply, sums)"),
          ("colstat", "BN passes (apply, backward reduce, backward apply, sums)"), ("scale_shift", "BN passes (apply, backward reduce, backward apply, sums)")]
tot, total = {}, 0.0
for name, n, t in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    strip = re.search(r"dw_conv_strip_kernel<\d+, \d+, (true|false)", name)   # (round 4: forward and input gradient are one kernel, FLIP = third argument)
    if strip:
        fam = "depthwise input gradient" if strip.group(1) == "true" else "depthwise fwd"
    else:
        fam = next((f for key, f in fam_of if key in name), "other (splits, loss, pooling, resampling, optimizer, torch fills)")
    tot[fam] = tot.get(fam, 0.0) + t / 1e6 / steps
    total += t / 1e6 / steps
print("DeepLab-MobileNetV2 %d-class %dx%d batch %d train step: %.2f ms of kernels per step" % (classes, size, size, batch, total))
print("| family | ms/step | algorithmic GB/step | GB/s | of 8 TB/s |\n|---|---|---|---|---|")
for fam, ms in sorted(tot.items(), key=lambda kv: -kv[1]):
    if fam in alg:
        gbs = alg[fam] / 1e9 / (ms * 1e-3)
        print("| %s | %.2f | %.2f | %.0f | %.3f |" % (fam, ms, alg[fam] / 1e9, gbs, gbs / 8000.0))
    else:
        print("| %s | %.2f | -- | -- | -- |" % (fam, ms))
print("| whole step | %.2f | %.2f | %.0f | %.3f |" % (total, sum(alg.values()) / 1e9, sum(alg.values()) / 1e9 / (total * 1e-3), sum(alg.values()) / 1e9 / (total * 1e-3) / 8000.0))
