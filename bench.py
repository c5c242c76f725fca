#!/usr/bin/env python
"""bench.py -- DeepLab-v3+ ResNet-101 513x513 training + MC-dropout pool scoring on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic batch: zero_grad -> forward -> CE loss -> backward
-> (N>1: RCCL gradient all-reduce) -> SGD step on DeepLab-R101 os16, 19 classes, 513x513, per-GPU batch 8
(the configuration BASELINE.json's metric is quoted on).  Inputs are generated
from per-image seeds and are resident in HBM before the timed region.  W untimed warm-up steps, then
EXACTLY K timed steps between barrier + synchronize; the max over ranks is reported by rank 0 as ONE
JSON line of < 4 KB (`compact_line`); every table behind it (per-kernel, per-pass, per-mode) goes to `bench_detail.json` (under gpurun_out/
when that directory exists, else the repo root) and the progress log to stderr.  The headline (`value`, `dtype`) is the f32 PARITY mode -- the mode every parity test runs in
(f32 tensors end to end, logits within 1e-3 of the reference, argmax bit-exact).  Its dense convs run the "f16x3" engine
(`config.f32_mma`): every operand tensor is scaled by a power of two and split into two f16 parts (23 significant bits), three
MFMA products per f32 product on the f16 pipe, f32 accumulation (dass_hip/ops.py:set_f32_mma) -- the f32 product to 2^-22, and
the roofline peak is priced accordingly (2500 / 3 TFLOP/s); `--f32-mma bf16x6` selects the exact three-part bf16 engine.
`--gpus N` with N > 1 and no torchrun environment starts N ranks itself (before anything touches the GPU) and relays rank 0's
line; a WORLD_SIZE that contradicts --gpus is an error.  The same line carries
  mc_dropout  : pool-images/s of the T=10 MC-dropout vote-entropy scoring call on the same model over 376 pool images
                per rank (config D's per-GPU share of the 2975-image pool is 372),
  core_set    : config E -- pooled decoder features of the same pool shard (images/s) and the k-center greedy selection
                (k = 125, 50 pre-selected) on a [2975, 2736] feature matrix (seconds),
  roofline    : `kernel` / `achieved` / `frac` = the DOMINANT conv kernel of the TIMED train step, named by its symbol (the one a rocprofv3
                kernel table names): after the timed region three more steps run with a start / stop HIP event pair bound to every dispatch
                of the library, on the stream it is launched on (dass_hip/_lib.py:KernelTimer); achieved = algorithmic GFLOP of that kernel's
                launches / the sum of their durations (= GFLOP per launch / average launch duration), peak = the dense MFMA peak of the
                engine per algorithmic flop, `traffic` = its HBM bytes per launch from the committed --pmc passes.  `conv_family_frac` is the
                same quotient over every conv / weight-gradient launch, `train_step_frac` the whole step against the peak, `t_lb_ms` /
                `mixed_frac` SURVEY 8(d)'s mixed per-layer bound sum max(FLOPs / peak, bytes / 8 TB/s) and its share of the measured step,
  cpu_baseline: the CPU oracle (stock PyTorch fp32 restatement, oracle/) timed on this box's host cores on bounded
                samples (rank 0, N=1 only): (i) train steps, (ii) 10-pass MC-dropout the reference way (T full forwards)
                and with the deterministic prefix hoisted, (iii) core-set features + sklearn fp64 k-center on the SAME
                [2975, 2736] matrix as the GPU leg (`core_set.kcenter.picks_equal_sklearn_on_same_matrix`), (iv) BASELINE
                config 0 (U-Net 128x128 batch 2, 3 SGD steps, CPU only),
  f32_mfma_mode: the same legs with the convs on v_mfma_f32_32x32x2_f32 (the plain f32 fma chain), for comparison,
  bf16_perf_mode: the same legs with bf16 storage / f32 accumulate (NOT parity-grade: deviation from the
                f32 reference is measured in tests/test_bf16_gpu.py), reported for information only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# MI355X_MICROARCH.md dense peaks, per ALGORITHMIC flop of each conv engine: the split engines execute 6 (3) bf16
# MFMA products per f32 product, so their ceiling in algorithmic TFLOP/s is the bf16 peak / 6 (/ 3)
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16x6": 2500.0 / 6, "bf16x3": 2500.0 / 3, "f16x3": 2500.0 / 3, "bf16": 2500.0, "bf16x1": 2500.0}
PEAK_NOTE = {"f32": "f32 MFMA dense", "bf16x6": "bf16 MFMA dense 2500 / 6 products per f32 product",
             "bf16x3": "bf16 MFMA dense 2500 / 3 products per f32 product",
             "f16x3": "f16 MFMA dense 2500 / 3 products per f32 product (two scaled f16 parts per operand)", "bf16": "bf16 MFMA dense",
             "bf16x1": "bf16 MFMA dense (one bf16 part per operand, one product: a perf engine on f32 tensors, not parity-grade)"}
HBM_PEAK_TBS = 8.0
DETAIL_FILE = "bench_detail.json"
LINE_LIMIT = 4096  # the driver keeps ~8 KB of stdout: the final JSON line stays below half of that (tests/test_cpu.py)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=513)
    ap.add_argument("--backbone", default="resnet101")
    ap.add_argument("--classes", type=int, default=19)
    ap.add_argument("--mc-steps", type=int, default=10, help="T of the MC-dropout scoring leg")
    ap.add_argument("--mc-batch", type=int, default=0, help="images per scoring forward (0 = --batch, what the reference's driver passes to its selectors)")
    ap.add_argument("--mc-batches", type=int, default=47, help="timed scoring batches per rank (47 x 8 = 376 >= config D's 372 per GPU)")
    ap.add_argument("--no-coreset", action="store_true")
    ap.add_argument("--no-pool-reader", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="headline mode (f32 = parity mode)")
    ap.add_argument("--f32-mma", default=os.environ.get("DASS_F32_MMA", "f16x3"), choices=["f16x3", "bf16x6", "f32", "bf16x3", "bf16x1"],
                    help="conv engine of the f32 headline (f16x3 = two scaled f16 parts per operand, three products: the default parity "
                         "engine; bf16x6 = three bf16 parts, six products)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="capture the train step into a hipGraph (dass_hip/graph.py) and time K replays: 'auto' = on for the single-GPU f16x3 / "
                         "bf16x1 legs (the step's ~550 launches cost the host 26-29 ms, at or above the GPU time), falling back to eager steps "
                         "if the capture fails; 'off' = eager")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mc", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-second-dtype", action="store_true", help="skip the informational legs in the other modes")
    ap.add_argument("--only", default="", choices=["", "mc", "coreset"],
                    help="profiling aid: run ONLY the MC-dropout leg / the core-set leg (no train steps; the JSON line then carries no headline)")
    return ap.parse_args()


def synthetic_batch(n, h, w, num_classes, first_index):
    imgs, labs = [], []
    for i in range(n):
        g = torch.Generator().manual_seed(1000 + first_index + i)
        imgs.append(torch.randn(3, h, w, generator=g))
        lab = torch.randint(0, num_classes, (h, w), generator=g).float()
        lab[: h // 10] = 255
        labs.append(lab)
    return torch.stack(imgs), torch.stack(labs)


def log(msg):
    """progress to stderr and (when present) gpurun_out/: a silent run is taken for a hung one"""
    line = "[bench %s] %s" % (time.strftime("%H:%M:%S"), msg)
    print(line, file=sys.stderr, flush=True)
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "bench_progress.log"), "a") as f:
            f.write(line + "\n")


class Env(object):
    pass


def r101_conv_layers(batch, size, blocks3=23):
    """(count, N, H, W, C, K, ksize, stride, pad, dil) of every groups=1 conv of DeepLab-R101 os16, stem excluded
    (3 input channels: a separate row-tap kernel, 0.7 % of the FLOPs) and without the 19-class classifier; blocks3 = 6: ResNet-50"""
    s2 = (size + 1) // 2
    s4 = (s2 + 1) // 2
    s8 = (s4 + 1) // 2
    s16 = (s8 + 1) // 2
    b = batch
    return [(1, b, s4, s4, 64, 64, 1, 1, 0, 1), (2, b, s4, s4, 256, 64, 1, 1, 0, 1), (3, b, s4, s4, 64, 64, 3, 1, 1, 1),
            (3, b, s4, s4, 64, 256, 1, 1, 0, 1), (1, b, s4, s4, 64, 256, 1, 1, 0, 1),
            (1, b, s4, s4, 256, 128, 1, 1, 0, 1), (1, b, s4, s4, 128, 128, 3, 2, 1, 1), (1, b, s4, s4, 256, 512, 1, 2, 0, 1),
            (3, b, s8, s8, 512, 128, 1, 1, 0, 1), (3, b, s8, s8, 128, 128, 3, 1, 1, 1), (4, b, s8, s8, 128, 512, 1, 1, 0, 1),
            (1, b, s8, s8, 512, 256, 1, 1, 0, 1), (1, b, s8, s8, 256, 256, 3, 2, 1, 1), (1, b, s8, s8, 512, 1024, 1, 2, 0, 1),
            (blocks3 - 1, b, s16, s16, 1024, 256, 1, 1, 0, 1), (blocks3 - 1, b, s16, s16, 256, 256, 3, 1, 1, 1), (blocks3, b, s16, s16, 256, 1024, 1, 1, 0, 1),
            (1, b, s16, s16, 1024, 512, 1, 1, 0, 1), (2, b, s16, s16, 2048, 512, 1, 1, 0, 1), (1, b, s16, s16, 512, 512, 3, 1, 2, 2),
            (1, b, s16, s16, 512, 512, 3, 1, 4, 4), (1, b, s16, s16, 512, 512, 3, 1, 8, 8), (3, b, s16, s16, 512, 2048, 1, 1, 0, 1),
            (1, b, s16, s16, 1024, 2048, 1, 1, 0, 1),
            (1, b, s16, s16, 2048, 256, 1, 1, 0, 1), (1, b, s16, s16, 2048, 256, 3, 1, 6, 6), (1, b, s16, s16, 2048, 256, 3, 1, 12, 12),
            (1, b, s16, s16, 2048, 256, 3, 1, 18, 18), (1, b, s16, s16, 1280, 256, 1, 1, 0, 1),
            (1, b, s4, s4, 256, 48, 1, 1, 0, 1), (1, b, s4, s4, 304, 256, 3, 1, 1, 1), (1, b, s4, s4, 256, 256, 3, 1, 1, 1)]


def mobilenet_train_bytes(batch, size, classes):
    """ALGORITHMIC HBM bytes of one DeepLab-MobileNetV2 (os16) train step in f32 -- the network is bandwidth-bound (SURVEY.md 8a/8d:
    5 of 61 layers MFMA-bound): every conv reads its input and writes its output once per pass (forward, input gradient, weight
    gradient: 3 x (E_in + E_out) x 4 B) and every train-mode BN moves its tensor four more times (statistics / apply, reduce /
    apply backward: SURVEY 8d "BN train traffic 2 x (2 x elems)").  -> (bytes, conv elements in + out, BN elements)"""
    def o(h, k, s, p, d):
        return (h + 2 * p - d * (k - 1) - 1) // s + 1
    h = o(size, 3, 2, 1, 1)
    conv_e = batch * (3 * size * size + 32 * h * h)
    bn_e = batch * 32 * h * h
    cin, cur, rate = 32, 2, 1
    low_h = None
    for t, c, n, s_ in [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]:
        if cur == 16:
            stride, dil = 1, rate
            rate *= s_
        else:
            stride, dil = s_, 1
            cur *= s_
        for i in range(n):
            st = stride if i == 0 else 1
            hid = cin * t
            hp = h + 2 * dil  # fixed_padding before the expand conv (mobilenet.py:70-77)
            if t != 1:
                conv_e += batch * (cin * hp * hp + hid * hp * hp)
                bn_e += batch * hid * hp * hp
            ho = o(hp, 3, st, 0, dil)
            conv_e += batch * (hid * hp * hp + hid * ho * ho) + batch * (hid * ho * ho + c * ho * ho)
            bn_e += batch * (hid + c) * ho * ho
            cin, h = c, ho
        if c == 24:
            low_h = h
    # ASPP (320 -> 256 x4, image pool, merge 1280 -> 256) and decoder (24 -> 48, 304 -> 256, 256 -> 256, 256 -> classes)
    conv_e += 4 * batch * (320 + 256) * h * h + batch * (1280 + 256) * h * h
    bn_e += 5 * batch * 256 * h * h
    conv_e += batch * (24 + 48) * low_h * low_h + batch * (304 + 256) * low_h * low_h + batch * 512 * low_h * low_h + batch * (256 + classes) * low_h * low_h
    bn_e += batch * (48 + 512) * low_h * low_h
    return 3 * conv_e * 4 + 4 * bn_e * 4, conv_e, bn_e


def _out(h, k, s, p, d):
    return (h + 2 * p - d * (k - 1) - 1) // s + 1


def mobilenet_gmac(size, classes):
    """forward GMAC per image of DeepLab-MobileNetV2 os16 (27.07 at 513^2, 21 classes)"""
    h = _out(size, 3, 2, 1, 1)
    m = h * h * 32 * 27
    cin, cur, rate, low_h = 32, 2, 1, None
    for t, c, n, s_ in [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]:
        if cur == 16:
            stride, dil = 1, rate
            rate *= s_
        else:
            stride, dil = s_, 1
            cur *= s_
        for i in range(n):
            st = stride if i == 0 else 1
            hid, hp = cin * t, h + 2 * dil
            if t != 1:
                m += hp * hp * cin * hid
            ho = _out(hp, 3, st, 0, dil)
            m += ho * ho * hid * 9 + ho * ho * hid * c
            cin, h = c, ho
        if c == 24:
            low_h = h
    m += h * h * 320 * 256 * 28 + 320 * 256 + h * h * 1280 * 256
    return (m + low_h * low_h * (24 * 48 + 304 * 256 * 9 + 256 * 256 * 9 + 256 * classes)) / 1e9


def work_model(backbone, size, classes, T=10):
    """ALGORITHMIC work per image, derived from --backbone / --size / --classes (SURVEY.md 8d's figures at its sizes: R101 513^2
    92.81 GMAC forward = 556.9 GFLOP per trained image, 769^2: 1233.8; MobileNetV2 513^2 21 classes: 162.4; MC-dropout = prefix once
    + T x last_conv tail = 573.6 GFLOP at R101 513^2 T=10; core-set features = the prefix = 142.5)"""
    s4 = ((size + 1) // 2 + 1) // 2
    tail = s4 * s4 * (304 * 256 * 9 + 256 * 256 * 9 + 256 * classes) / 1e9
    if backbone == "mobilenet":
        fwd = mobilenet_gmac(size, classes)
    else:
        s2 = (size + 1) // 2
        fwd = s2 * s2 * 64 * 147 / 1e9 + s4 * s4 * 256 * classes / 1e9
        for cnt, n, h, w, c, k, ks, st, pad, dil in r101_conv_layers(1, size, 6 if backbone == "resnet" else 23):
            oh = _out(h, ks, st, pad, dil)
            fwd += cnt * oh * oh * k * ks * ks * c / 1e9
    # what the MC-dropout tail EXECUTES (DASS_MC_SPARSE, the default): of last_conv.0's 304 input channels the 256 that come from the ASPP carry
    # the Dropout2d(0.5) mask of aspp.py:70; the surviving ones are packed to the front and the slab loop stops behind them -- Binomial(256, 0.5)
    # + 48 channels = 6 of 10 32-channel slabs in 95 % of the draws (5 in 3 %, 7 in 2 %)
    conv0 = s4 * s4 * 304 * 256 * 9 / 1e9
    tail_exec = tail - 0.4 * conv0
    return {"forward_gflop": 2 * fwd, "train_gflop": 6 * fwd, "mc_gflop": 2 * ((fwd - tail) + T * tail), "coreset_gflop": 2 * (fwd - tail),
            "mc_executed_gflop": 2 * ((fwd - tail) + T * tail_exec)}


def mixed_roofline(args, peak_tflops):
    """SURVEY.md 8(d)'s bounding roofline of one train step: t_lb = sum over layers and passes of max(FLOPs / peak_MFMA, bytes / peak_HBM).
    Per conv layer three passes (forward, input gradient, weight gradient), each 2 M K R S C FLOP against (E_in + E_out) x 4 B + the
    weights; per train-mode BN tensor four passes of 2 x 4 B per element (SURVEY 8d "BN train traffic 2 x (2 x elems)").  f32 tensors.
    -> {t_lb_ms, conv_ms, bn_ms, conv_gb, bn_gb, mfma_bound_layers, layers}"""
    b, size = args.batch, args.size
    if args.backbone == "mobilenet":
        nbytes, conv_e, bn_e = mobilenet_train_bytes(b, size, args.classes)
        # depthwise / pointwise layers are HBM-bound one and all at these widths except the ASPP / decoder 3x3s: price the whole net
        # by bytes and add the MFMA time of the dense 3x3 layers where it exceeds their bytes
        h16, h4 = ((((size + 1) // 2 + 1) // 2 + 1) // 2 + 1) // 2, ((size + 1) // 2 + 1) // 2
        dense = [(3, b * h16 * h16, 320 * 9, 256), (1, b * h4 * h4, 304 * 9, 256), (1, b * h4 * h4, 256 * 9, 256)]
        extra, nm = 0.0, 0
        for cnt, m, red, k in dense:
            t_f = 2.0 * m * red * k / (peak_tflops * 1e12)
            t_b = (m * (red // 9 + k) * 4.0 + red * k * 4.0) / (HBM_PEAK_TBS * 1e12)
            if t_f > t_b:
                extra += 3 * cnt * (t_f - t_b)
                nm += cnt
        conv_ms = 3 * conv_e * 4 / (HBM_PEAK_TBS * 1e9) + extra * 1e3
        bn_ms = 4 * bn_e * 4 / (HBM_PEAK_TBS * 1e9)
        return {"t_lb_ms": round(conv_ms + bn_ms, 3), "conv_ms": round(conv_ms, 3), "bn_ms": round(bn_ms, 3), "conv_gb": round(3 * conv_e * 4 / 1e9, 2),
                "bn_gb": round(4 * bn_e * 4 / 1e9, 2), "mfma_bound_layers": nm, "layers": 61}
    s2 = (size + 1) // 2
    s4 = (s2 + 1) // 2
    layers = [(1, b, size, size, 3, 64, 7, 2, 3, 1)] + r101_conv_layers(b, size, 6 if args.backbone == "resnet" else 23) + [(1, b, s4, s4, 256, args.classes, 1, 1, 0, 1)]
    conv_s = conv_b = bn_e = 0.0
    nm = nl = 0
    for cnt, n, h, w, c, k, ks, st, pad, dil in layers:
        oh = _out(h, ks, st, pad, dil)
        flops = 2.0 * n * oh * oh * k * ks * ks * c
        nbytes = (n * h * h * c + n * oh * oh * k + k * ks * ks * c) * 4.0
        t_f, t_b = flops / (peak_tflops * 1e12), nbytes / (HBM_PEAK_TBS * 1e12)
        conv_s += 3 * cnt * max(t_f, t_b)
        conv_b += 3 * cnt * nbytes
        nm += cnt * (t_f > t_b)
        nl += cnt
        if k != args.classes:
            bn_e += cnt * n * oh * oh * k
    bn_b = 4 * bn_e * 4.0
    return {"t_lb_ms": round((conv_s + bn_b / (HBM_PEAK_TBS * 1e12)) * 1e3, 3), "conv_ms": round(conv_s * 1e3, 3),
            "bn_ms": round(bn_b / (HBM_PEAK_TBS * 1e9), 3), "conv_gb": round(conv_b / 1e9, 2), "bn_gb": round(bn_b / 1e9, 2),
            "mfma_bound_layers": nm, "layers": nl}


def conv_aggregate(args, ops, tdt):
    """times forward, input-gradient and weight-gradient launch of every conv layer shape of the train step through the
    C-ABI (events on the launch stream, 5 reps) and weights them by their count: the time-weighted conv roofline"""
    from dass_hip._lib import check, lib

    dev = "cuda"
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    gflop = 0.0
    group = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timeit(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for cnt, n, h, w, c, k, ks, st, pad, dil in r101_conv_layers(args.batch, args.size):
        oh, ow = ops.conv_out_size(h, ks, st, pad, dil), ops.conv_out_size(w, ks, st, pad, dil)
        x = torch.randn((n, h, w, c), device=dev).to(tdt)
        wt = (torch.randn((k, ks, ks, c), device=dev) * 0.05)
        y = torch.empty((n, oh, ow, k), device=dev, dtype=tdt)
        dy = torch.randn((n, oh, ow, k), device=dev).to(tdt)
        dx = torch.empty((n, h, w, c), device=dev, dtype=tdt)
        dw = torch.empty((k, ks, ks, c), device=dev)
        stream = ops._stream()
        pad_t = dil * (ks - 1) - pad
        # the engine the train step uses for this layer (bf16x6, DASS_X3=select: pre-split kernels on the long 3x3 reductions, the
        # forward pays the conversion pass of its input, dy's split rows come out of the BN backward; f16x3: pre-split kernels on
        # every dense layer, both operands of all three launches written by the BN passes -- no conversion pass in the step)
        x3_fwd = tdt == torch.float32 and ops._x3_train_layer(ks * ks, c) and k > 32
        x3_dg = tdt == torch.float32 and ops._x3_train_layer(ks * ks, k) and c > 32
        x3_wg = x3_fwd and ops.x3_pipeline(training=True)
        wop = ops.prepare_conv_weight(wt.to(tdt) if tdt != torch.float32 else wt, x3=x3_fwd)
        wop_t = ops.prepare_conv_weight((wt.permute(3, 1, 2, 0).flip(1, 2).contiguous()).to(tdt) if tdt != torch.float32
                                        else wt.permute(3, 1, 2, 0).flip(1, 2).contiguous(), x3=x3_dg)
        if x3_fwd and ops.x3_parts() <= 2:
            x3_ = ops.split3_rows(x, c, n * h * w, c)      # (written by the producer's BN-apply pass in the step)
            tot["fwd"] += cnt * timeit(lambda: ops.conv_x3_launch(x3_, wop, y, k, (n, h, w, c, oh, ow, k, ks, ks, st, pad, dil)))
        elif x3_fwd:
            tot["fwd"] += cnt * timeit(lambda: ops.conv_x3_launch(ops.split3_rows(x, c, n * h * w, c), wop, y, k,
                                                                   (n, h, w, c, oh, ow, k, ks, ks, st, pad, dil)))
        else:
            tot["fwd"] += cnt * timeit(lambda: ops.conv_launch(x, c, wop, y, k, (n, h, w, c, oh, ow, k, ks, ks, st, pad, dil)))
        if x3_dg:
            dy3 = ops.split3_rows(dy, k, n * oh * ow, k)
            tot["dgrad"] += cnt * timeit(lambda: ops.conv_x3_launch(dy3, wop_t, dx, c, (n, oh, ow, k, h, w, c, ks, ks, 1, pad_t, dil), ustride=st))
            del dy3
        else:
            tot["dgrad"] += cnt * timeit(lambda: check(lib.dass_conv2d_igemm(ops._p(dy), k, ops._p(wop_t), ops._p(dx), c, None, None, None, 0, None,
                                                                            n, oh, ow, k, h, w, c, ks, ks, 1, pad_t, dil, st, 0, ops._cdt(dx), stream), "dgrad"))
        if x3_wg and ops.deferred_wgrad():
            # the train step computes these in ONE grouped launch per tile class at the end of backward: collect, time below
            x3w, dy3w = ops.split3_rows(x, c, n * h * w, c), ops.split3_rows(dy, k, n * oh * ow, k)
            for _ in range(cnt):
                group.append((x3w, dy3w, torch.zeros((k, ks, ks, c), device=dev), (n, h, w, c, oh, ow, k, ks, ks, st, pad, dil)))
        elif x3_wg:
            x3w, dy3w = ops.split3_rows(x, c, n * h * w, c), ops.split3_rows(dy, k, n * oh * ow, k)
            tot["wgrad"] += cnt * timeit(lambda: check(lib.dass_conv2d_wgrad_x3(ops._p(x3w), ops._p(dy3w), ops._p(dw), n, h, w, c, oh, ow, k, ks, ks,
                                                                               st, pad, dil, 1, stream), "wgrad_x3"))
            del x3w, dy3w
        else:
            tot["wgrad"] += cnt * timeit(lambda: check(lib.dass_conv2d_wgrad(ops._p(x), c, ops._p(dy), k, ops._p(dw), n, h, w, c, oh, ow, k, ks, ks,
                                                                            st, pad, dil, ops._cdt(dy), stream), "wgrad"))
        gflop += cnt * 2.0 * n * oh * ow * k * ks * ks * c / 1e9
    if group:
        import ctypes

        import numpy as np

        items = np.zeros((len(group), 16), dtype=np.int64)
        for i, (x3w, dy3w, dwg, dims) in enumerate(group):
            items[i, :3] = (x3w.data_ptr(), dy3w.data_ptr(), dwg.data_ptr())
            items[i, 3:15] = dims
        scratch = torch.empty((lib.dass_conv2d_wgrad_x3_group_scratch_bytes(len(group)) + 128,), dtype=torch.uint8, device=dev)
        tot["wgrad"] += timeit(lambda: check(lib.dass_conv2d_wgrad_x3_group(items.ctypes.data_as(ctypes.c_void_p), len(group), ops._p(scratch),
                                                                           scratch.numel(), ops._stream()), "wgrad_x3_group"))
        del group[:]
    ms = tot["fwd"] + tot["dgrad"] + tot["wgrad"]
    return {"ms_per_step": round(ms, 3), "fwd_ms": round(tot["fwd"], 3), "dgrad_ms": round(tot["dgrad"], 3), "wgrad_ms": round(tot["wgrad"], 3),
            "gflop": round(3 * gflop, 1), "achieved": round(3 * gflop / ms, 2)}


def _tile_tag(name, tag):
    if not tag:
        return name.replace("dass_conv2d_", "")
    bm, bn = tag >> 16, (tag >> 4) & 0xfff
    form = "whole-tile" if tag & 2 else ("stream-K + fix-up" if tag & 1 else "general")
    return "%s: conv_x3_kernel<%d,%d> %s" % (name.replace("dass_conv2d_", ""), bm, bn, form)


def step_conv_times(train_step, reps=3):
    """`reps` real train steps with the library's launch profile open (dass_hip/_lib.py:KernelTimer -> csrc/prof.hip): every kernel
    carries a start / stop HIP event pair bound to its own dispatch on its own stream -> per-step totals of the conv / weight-
    gradient entry points (GFLOP, ms), a per-(entry point, tile class) table, the train-mode BN passes (GB, ms) and the kernel
    table of the step (what `rocprofv3 --kernel-trace --stats` lists for the same command, profiles/r04_train_summary.md)"""
    import re

    from dass_hip._lib import BN_ENTRY_POINTS, KernelTimer

    with KernelTimer() as kt:
        train_step()                      # (first profiled step untimed: event pool, allocator)
        torch.cuda.synchronize()
        kt.restart()
        for _ in range(reps):
            train_step()
        torch.cuda.synchronize()
        kernels, calls = kt.results()
        call_kernel_ms = {id(c[4]): [k[1] for k in kernels[i0:i1]] for c, (_, _, _, i0, i1) in zip(calls, kt.calls)}
    by, bn = {}, {}
    for name, tag, work, ms, knames in calls:
        d = bn if name in BN_ENTRY_POINTS else by
        e = d.setdefault(name.replace("dass_", "") if name in BN_ENTRY_POINTS else _tile_tag(name, tag), [0, 0.0, 0.0])
        e[0] += 1
        e[1] += work
        e[2] += ms
    table = [{"kernel": k, "launches_per_step": round(v[0] / reps, 1), "gflop_per_step": round(v[1] / reps, 1),
              "ms_per_step": round(v[2] / reps, 3), "avg_us": round(1e3 * v[2] / v[0], 1),
              "tflops": round(v[1] / v[2], 1) if v[2] > 0 else None} for k, v in by.items()]
    table.sort(key=lambda r: -r["ms_per_step"])
    bn_table = [{"pass": k, "launches_per_step": round(v[0] / reps, 1), "gb_per_step": round(v[1] / reps, 3), "ms_per_step": round(v[2] / reps, 3),
                 "tb_per_s": round(v[1] / v[2], 3) if v[2] > 0 else None, "frac_of_8_tb_per_s": round(v[1] / v[2] / 8.0, 4) if v[2] > 0 else None}
                for k, v in bn.items()]
    bn_table.sort(key=lambda r: -r["ms_per_step"])
    conv = [c for c in calls if c[0] not in BN_ENTRY_POINTS]

    def short_name(kname):
        return re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", kname)).split("(")[0].replace(", ", ",")

    # the same conv calls by KERNEL SYMBOL: a call's GFLOP and time go to the kernel that took most of the call's time (its fix-up /
    # phase launches ride along), so that "dominant" names what a rocprofv3 kernel table names
    sym = {}
    for name, tag, work, ms, knames in conv:
        durs = {}
        for kn, kms in zip(knames, call_kernel_ms[id(knames)]):
            durs[kn] = durs.get(kn, 0.0) + kms
        main = short_name(max(durs, key=durs.get)) if durs else name
        e = sym.setdefault(main, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += work
        e[2] += ms
    sym_table = [{"kernel": k, "launches_per_step": round(v[0] / reps, 1), "gflop_per_step": round(v[1] / reps, 1), "ms_per_step": round(v[2] / reps, 3),
                  "avg_us": round(1e3 * v[2] / v[0], 1), "gflop_per_launch": round(v[1] / v[0], 3), "tflops": round(v[1] / v[2], 1) if v[2] > 0 else None}
                 for k, v in sym.items()]
    sym_table.sort(key=lambda r: -r["ms_per_step"])
    kt_by = {}
    for kname, ms, grid, stream in kernels:
        e = kt_by.setdefault(short_name(kname), [0, 0.0])
        e[0] += 1
        e[1] += ms
    ktable = [{"kernel": k, "launches_per_step": round(v[0] / reps, 1), "ms_per_step": round(v[1] / reps, 3), "avg_us": round(1e3 * v[1] / v[0], 1)}
              for k, v in sorted(kt_by.items(), key=lambda kv: -kv[1][1])]
    return {"gflop": sum(c[2] for c in conv) / reps, "ms": sum(c[3] for c in conv) / reps, "launches": len(conv) / reps, "table": table,
            "by_symbol": sym_table,
            "bn": bn_table, "bn_ms": sum(r["ms_per_step"] for r in bn_table), "bn_gb": sum(r["gb_per_step"] for r in bn_table),
            "kernels": ktable[:16], "kernel_ms": sum(k[1] for k in kernels) / reps, "kernel_launches": len(kernels) / reps,
            "streams": len({k[3] for k in kernels})}


def run_mode(args, env, dtype_name, steps, warmup, mma="bf16x6"):
    """train leg + MC-dropout leg + dominant-kernel timing in one numerics mode (mma: conv engine for f32 tensors)"""
    from dass_hip import ops
    from models.deeplab import DeepLab
    from utils.loss import SegmentationLosses
    from active_selection.base import shard_bounds
    from active_selection.mc_dropout import ActiveSelectionMCDropout
    from dass_hip.dist import GradientAverager, average_gradients

    rank, world, dev, dist = env.rank, env.world, env.dev, env.dist
    wm = work_model(args.backbone, args.size, args.classes, args.mc_steps)
    ops.set_compute_dtype(torch.float32 if dtype_name == "f32" else torch.bfloat16)
    engine = mma if dtype_name == "f32" else "bf16"
    ops.set_f32_mma(mma)
    dtype_name = dtype_name if dtype_name == "bf16" or mma == "bf16x6" else "f32/" + mma  # log label
    torch.manual_seed(1234)  # identical random-init weights on every rank
    model = DeepLab(backbone=args.backbone, output_stride=16, num_classes=args.classes, sync_bn=False,
                    freeze_bn=False, pretrained=False).to(dev)
    crit_obj = SegmentationLosses(cuda=True, global_batch=True)   # DataParallel loss semantics under N > 1
    criterion = crit_obj.build_loss("ce")
    lr = 0.01
    from dass_hip.optim import SGD  # torch.optim.SGD surface and state, update arithmetic in dass_sgd_step_multi

    if os.environ.get("DASS_TORCH_SGD", "0") == "1":  # A/B knob: the stock foreach implementation
        SGD = torch.optim.SGD
    optimizer = SGD([{"params": model.get_1x_lr_params(), "lr": lr}, {"params": model.get_10x_lr_params(), "lr": lr * 10}],
                    momentum=0.9, weight_decay=5e-4, nesterov=False)
    params = [p for g in optimizer.param_groups for p in g["params"]]
    b, s = args.batch, args.size
    image, target = synthetic_batch(b, s, s, args.classes, first_index=rank * b)
    image, target = image.to(dev), target.to(dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def tmax(dt):
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.only:
        steps = warmup = 0
    # 'auto': every f32-tensor leg on the pipelined engines.  The parity step is ~27 ms of GPU time against 26-32 ms of host time for its
    # ~550 launches (the host's share differs from box to box: 285 img/s eager on one, 249 on another), the perf engine 20 ms against the
    # same host time: a graph replay takes the host out of the step.  N > 1: zero_grad + forward + loss + backward replay as graph A, the
    # gradient all-reduce runs eagerly behind it on the same stream, the SGD step replays as graph B (dass_hip/graph.py) -- the loss's
    # own exchange of denominators depends on the labels only and runs ahead of graph A (utils/loss.py:use_static_global).
    # The bf16-storage mode keeps its eager loop.
    use_graph = steps > 0 and dtype_name.startswith("f32") and mma in ("f16x3", "bf16x1") and args.graph in ("on", "auto")
    if args.graph == "on" and steps > 0:
        use_graph = True
    ddp_graph = use_graph and dist is not None
    static = crit_obj.use_static_global(dev) if ddp_graph else None
    # N > 1, eager: bucketed RCCL all-reduce of the gradients, overlapped with backward through grad hooks
    # (DASS_DDP_OVERLAP=0: the plain after-backward form)
    averager = GradientAverager(params) if dist is not None and not ddp_graph and os.environ.get("DASS_DDP_OVERLAP", "1") == "1" else None

    def before_step():
        if crit_obj.static is not None:
            crit_obj.static.exchange(target)

    def compute_step():
        # (DASS_BENCH_GRAPH_FAULT=1: a fault injected INSIDE the capture, to rehearse the fallback below on a real GPU)
        if os.environ.get("DASS_BENCH_GRAPH_FAULT") == "1" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("injected capture fault (DASS_BENCH_GRAPH_FAULT=1)")
        optimizer.zero_grad(set_to_none=True)
        out = model(image)
        loss = criterion(out, target)
        loss.backward()
        return loss

    def reduce_step():
        if averager is not None:
            averager.finish()
        elif dist is not None:
            average_gradients(params)

    def train_step():
        before_step()
        loss = compute_step()
        reduce_step()
        optimizer.step()
        return loss

    def whole_step():
        loss = compute_step()
        optimizer.step()
        return loss

    model.train()
    loss = torch.zeros((), device=dev)
    for i in range(warmup):
        tw = time.perf_counter()
        train_step()
        torch.cuda.synchronize()
        if rank == 0:
            log("[%s] warm-up step %d: %.1f ms" % (dtype_name, i, (time.perf_counter() - tw) * 1e3))
    timed_step = train_step
    if use_graph:
        from dass_hip.graph import GraphedStep

        try:
            if ddp_graph:
                timed_step = GraphedStep(compute_step, warmup=2, reduce=reduce_step, finish=optimizer.step, before=before_step)
            else:
                timed_step = GraphedStep(whole_step, warmup=2)   # zero_grad + forward + loss + backward + SGD as ONE graph launch
            for _ in range(2):
                timed_step()
            torch.cuda.synchronize()
            if rank == 0:
                log("[%s] train step captured into %s" % (dtype_name, "two hipGraphs around the eager gradient all-reduce" if ddp_graph
                                                          else "a hipGraph (two streams inside)"))
        except Exception as exc:  # noqa: BLE001  (a capture the runtime refuses must not cost the run: eager steps instead)
            log("[%s] hipGraph capture FAILED (%r): timing eager steps" % (dtype_name, exc))
            use_graph, timed_step = False, train_step
            torch.cuda.synchronize()
            if ddp_graph:   # (every rank captures the same step: they fail, and fall back, together)
                crit_obj.static = None
                ddp_graph = False
                if os.environ.get("DASS_DDP_OVERLAP", "1") == "1":
                    averager = GradientAverager(params)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = timed_step()
    barrier()
    dt = max(tmax(time.perf_counter() - t0), 1e-9)
    loss = loss.detach().clone()
    if dist is not None and crit_obj.static is not None:   # pre-exchanged denominators: a rank's value is world x its share of the global loss
        dist.all_reduce(loss)
        loss /= world
    res = {"train_ips": b * world * steps / dt, "ms_per_step": dt / max(steps, 1) * 1e3, "final_loss": float(loss), "graph": bool(use_graph)}
    if dist is not None and steps > 0:
        # every rank must hold bit-identical weights after the same averaged gradients: a wrapping int64 sum over the raw bits
        h = torch.stack([p.detach().view(torch.int32).sum(dtype=torch.int64) for p in params]).sum().reshape(1)
        hs = [torch.zeros_like(h) for _ in range(world)]
        dist.all_gather(hs, h)
        res["replicas_identical"] = bool(all(int(x) == int(hs[0]) for x in hs))
        res["ddp"] = "graph A (fwd+bwd) -> eager flat all-reduce -> graph B (SGD)" if ddp_graph else (
            "eager, bucketed all-reduce overlapped with backward" if averager is not None else "eager, all-reduce after backward")
    if rank == 0 and steps > 0:
        log("[%s] train: %.2f images/s (%.1f ms/step)" % (dtype_name, res["train_ips"], res["ms_per_step"]))

    # ------------------------------------------------------------------ MC-dropout pool scoring (T passes)
    res["mc"] = None
    res["coreset"] = None
    if not args.no_mc:
        model.eval()
        b_train, b = b, (args.mc_batch or b)
        pool_keys = [("pool_%06d" % i).encode("ascii") for i in range(world * args.mc_batches * b_train)]
        # this rank's shard, resident in HBM before timing (per-image seeds: content independent of sharding)
        s0, s1 = shard_bounds(len(pool_keys), rank, world)
        shard = {}
        for gi in range(s0, s1):
            im, lb = synthetic_batch(1, s, s, args.classes, first_index=100000 + gi)
            shard[pool_keys[gi]] = (im.to(dev), lb.to(dev))

        def factory(images, include_labels):
            for i in range(0, len(images), b):
                chunk = images[i:i + b]
                yield {"image": torch.cat([shard[k][0] for k in chunk]), "label": torch.cat([shard[k][1] for k in chunk])}

        selector = ActiveSelectionMCDropout(args.classes, None, s, b, loader_factory=factory)
        # warm-up pool: the first batch of every rank's shard (a contiguous split of this list hands each rank its own keys)
        warm = [pool_keys[shard_bounds(len(pool_keys), r, world)[0] + i] for r in range(world) for i in range(b)]
        selector.get_vote_entropy_for_images(model, warm, 1, steps=args.mc_steps)
        barrier()
        t0 = time.perf_counter()
        if args.only == "coreset":
            selected = []
        else:
            selected = selector.get_vote_entropy_for_images(model, pool_keys, max(1, len(pool_keys) // 8), steps=args.mc_steps)
        barrier()
        dts = max(tmax(time.perf_counter() - t0), 1e-9)
        pool_ips = len(pool_keys) / dts
        if rank == 0 and args.only != "coreset":
            log("[%s] mc-dropout T=%d: %.2f pool images/s (%d images)" % (dtype_name, args.mc_steps, pool_ips, len(pool_keys)))
        res["mc"] = {"metric": "mc_dropout_pool_images_per_s", "value": round(pool_ips, 3), "unit": "images/s",
                     "T": args.mc_steps, "pool_images": len(pool_keys),
                     "pool_note": "per-rank pool = config D's 8-GPU share (2975 / 8 = 372, rounded up to whole batches); throughput is per image, "
                                  "so one GPU scores the whole 2975-image pool in 2975 / value seconds",
                     "scoring_batch": b, "seconds": round(dts, 4), "selected": len(selected),
                     "frac_of_mfma_peak": round(pool_ips * wm["mc_gflop"] / 1e3 / (MFMA_PEAK_TFLOPS[engine] * world), 4),
                     "gflop_per_image": {"algorithmic": round(wm["mc_gflop"], 1), "executed": round(wm["mc_executed_gflop"], 1),
                                         "note": "algorithmic = prefix once + T x last_conv (SURVEY 8d); executed = the same with the channel slabs Dropout2d "
                                                 "zeroed skipped in last_conv.0 (6 of 10 slabs on average): work avoided, not matrix-pipe utilisation"},
                     "executed_frac_of_mfma_peak": round(pool_ips * wm["mc_executed_gflop"] / 1e3 / (MFMA_PEAK_TFLOPS[engine] * world), 4),
                     "sharding": "contiguous key shards per rank + RCCL all_gather of per-image scores" if world > 1 else "single rank"}

        # -------------------------------------------------------------- config E: core-set features + k-center greedy
        # (the reference's core-set feature is 2736-wide = 304 channels x 3 x 3 cells of avg_pool2d(64, 32) over the 129 x 129 decoder
        #  map, core_set.py:45-47: defined at the 513 x 513 crop only, so the leg is skipped at other sizes)
        if not args.no_coreset and args.only != "mc" and (s + 3) // 4 == 129:
            from active_selection.core_set import ActiveSelectionCoreSet
            from dass_hip.dist import ModuleWrapper

            def factory_img(images, include_labels):
                for i in range(0, len(images), b):
                    yield torch.cat([shard[k][0] for k in images[i:i + b]])

            cs = ActiveSelectionCoreSet(None, s, b, loader_factory=factory_img)
            wrapped = ModuleWrapper(model)
            cs._features(wrapped, warm)
            barrier()
            t0 = time.perf_counter()
            feats = cs._features(wrapped, pool_keys)          # sharded feature pass + all-gather of [n, 2736]
            barrier()
            dtf = tmax(time.perf_counter() - t0)
            # the greedy selection at the full pool size (2975 x 2736, 50 already selected, k = 125: the authors' setting)
            # -- on the SAME matrix the CPU leg hands to sklearn (kcenter_matrix), so the two pick lists can be compared
            full = torch.from_numpy(kcenter_matrix()).to(dev)
            ops.kcenter_greedy(full, list(range(50)), 4)      # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            picks, _ = ops.kcenter_greedy(full, list(range(50)), 125)
            picks = picks.cpu()
            dtk = time.perf_counter() - t0
            feat_ips = len(pool_keys) / dtf
            if rank == 0:
                log("[%s] core-set: features %.1f pool images/s, k-center k=125 on 2975x2736: %.3f s" % (dtype_name, feat_ips, dtk))
            res["coreset"] = {"metric": "core_set_feature_images_per_s", "value": round(feat_ips, 2), "unit": "images/s",
                              "pool_images": len(pool_keys), "feature_seconds": round(dtf, 4),
                              "kcenter": {"n": 2975, "d": 2736, "preselected": 50, "k": 125, "seconds": round(dtk, 4),
                                          "picks": [int(i) for i in picks.tolist()],
                                          "hbm_gb_per_s": round(175 * 2975 * 2736 * 4 / dtk / 1e9, 1)},
                              "selection_2975_pool_seconds_at_this_gpu_count": round(2975.0 / feat_ips + dtk, 3),
                              "frac_of_mfma_peak": round(feat_ips * wm["coreset_gflop"] / 1e3 / (MFMA_PEAK_TFLOPS[engine] * world), 4)}

    b = args.batch  # (the scoring legs may have used --mc-batch)
    # ------------------------------------------------------------------ roofline of the dominant kernel
    res["roofline"] = None
    inst = None
    if not args.no_roofline and args.backbone != "mobilenet" and steps > 0:
        model.train()
        inst = step_conv_times(train_step)   # on EVERY rank: the step holds collectives when N > 1
    if rank == 0 and not args.no_roofline and args.backbone == "mobilenet":
        nbytes, conv_e, bn_e = mobilenet_train_bytes(b, s, args.classes)
        gbs = nbytes / (res["ms_per_step"] * 1e-3) / 1e9
        res["roofline"] = {"bound": "hbm", "kernel": "whole train step (61 conv layers: depthwise 3x3, pointwise 1x1, BN passes)", "achieved": round(gbs, 1),
                           "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "traffic": None,
                           "algorithmic_bytes_per_step": nbytes, "mixed": dict(mixed_roofline(args, MFMA_PEAK_TFLOPS[engine]), frac_of_step=None),
                           "train_step_frac": round(res["train_ips"] * wm["train_gflop"] / 1e3 / (MFMA_PEAK_TFLOPS[engine] * world), 4),
                           "note": "f32 tensors; 3 x (in + out) x 4 B per conv layer and pass + 4 BN passes per train-mode BN tensor (SURVEY 8d); "
                                   "per-family kernel times: profiles/r04_train_C_mbv2_summary.md"}
        res["roofline"]["mixed"]["frac_of_step"] = round(res["roofline"]["mixed"]["t_lb_ms"] / res["ms_per_step"], 4)
    elif rank == 0 and inst is not None:
        tdt = torch.bfloat16 if engine == "bf16" else torch.float32
        n_, h_, c_, k_ = b, (s + 3) // 4, 304, 256  # decoder.last_conv.0: 3x3 304->256 @129^2, the largest single layer
        x = torch.randn((n_, h_, h_, c_), device=dev).to(tdt)
        w = (torch.randn((k_, 3, 3, c_), device=dev) * 0.02).to(tdt)
        y = torch.empty((n_, h_, h_, k_), device=dev, dtype=tdt)
        dims = (n_, h_, h_, c_, h_, h_, k_, 3, 3, 1, 1, 1)
        x3_best = tdt == torch.float32 and ops._x3_train_layer(9, c_)  # this layer runs on the pre-split kernel in the train step
        w = ops.prepare_conv_weight(w, x3=x3_best)  # the split engines multiply pre-split weights (once per optimizer step in training)
        if x3_best:
            x3_ = ops.split3_rows(x, c_, n_ * h_ * h_, c_)
            launch = lambda: ops.conv_x3_launch(x3_, w, y, k_, dims)  # noqa: E731
        else:
            launch = lambda: ops.conv_launch(x, c_, w, y, k_, dims)  # noqa: E731
        for _ in range(3):
            launch()
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flops = 2.0 * n_ * h_ * h_ * k_ * 9 * c_
        achieved = flops / (ms * 1e-3) / 1e12
        peak = round(MFMA_PEAK_TFLOPS[engine], 1)
        log("[%s] dominant conv kernel: %.3f ms/launch = %.1f TFLOP/s" % (dtype_name, ms, achieved))
        traffic = None
        tfile = os.path.join(ROOT, "profiles", ("r03_traffic_x3_f16.json" if engine == "f16x3" else "r02_traffic_x3.json") if x3_best
                             else "r01_traffic_%s.json" % engine)
        if os.path.exists(tfile):  # HBM bytes per launch from the rocprofv3 --pmc passes (collected offline, see profiles/)
            traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
        kname = {"f32": "float,128,128,2,2", "bf16": "bf16,128,128,2,2", "bf16x6": "float,128,128,4,1,split=3",
                 "bf16x3": "float,128,128,2,2,split=2", "f16x3": "float,128,128,4,1,split=3", "bf16x1": "float,128,128,4,1,split=3"}[engine]
        if x3_best:
            kname_full = "conv_x3_kernel<256,128,4,2,2,%s> + fix-up, pre-split operands (3x3 304->256 @%dx%d, batch %d)" % (
                {"f16x3": "NP=2: two f16 parts", "bf16x1": "NP=1: one bf16 part"}.get(engine, "NP=3: three bf16 parts"), h_, h_, n_)
        else:
            kname_full = "conv_igemm_kernel<%s> (3x3 304->256 @%dx%d, batch %d)" % (kname, h_, h_, n_)
        best = {"kernel": kname_full, "achieved": round(achieved, 2),
                "frac": round(achieved / peak, 4), "launch_ms": round(ms, 4), "flops_per_launch": flops, "traffic": traffic}
        launch = x3_ = None
        del x, w, y
        agg = conv_aggregate(args, ops, tdt)
        log("[%s] isolated: all conv launches of one train step back to back on an idle chip: %.2f ms (fwd %.2f, dgrad %.2f, wgrad %.2f) = %.1f TFLOP/s"
            % (dtype_name, agg["ms_per_step"], agg["fwd_ms"], agg["dgrad_ms"], agg["wgrad_ms"], agg["achieved"]))
        in_step = inst["gflop"] / inst["ms"]
        log("[%s] in-step: %d conv launches per train step, %.2f ms of launch time (streams overlap) for %.0f GFLOP = %.1f TFLOP/s"
            % (dtype_name, inst["launches"], inst["ms"], inst["gflop"], in_step))
        for r in inst["table"][:6]:
            log("    %-70s %5.1f x %7.1f us = %6.2f ms  %s TFLOP/s" % (r["kernel"], r["launches_per_step"], r["avg_us"], r["ms_per_step"], r["tflops"]))
        log("[%s] in-step: %d kernels of the library per step on %d streams, %.2f ms of kernel time; train-mode BN passes %.2f ms for %.2f GB = %.2f TB/s"
            % (dtype_name, inst["kernel_launches"], inst["streams"], inst["kernel_ms"], inst["bn_ms"], inst["bn_gb"], inst["bn_gb"] / max(inst["bn_ms"], 1e-9)))
        for r in inst["bn"]:
            log("    %-30s %5.1f launches, %6.3f GB, %6.2f ms = %s TB/s" % (r["pass"], r["launches_per_step"], r["gb_per_step"], r["ms_per_step"], r["tb_per_s"]))
        sustained = None
        cfile = os.path.join(ROOT, "profiles", "r02_clock_probe.json")
        if os.path.exists(cfile) and engine != "f32":
            cp = json.load(open(cfile))
            div = {"bf16x6": 6.0, "bf16x3": 3.0, "f16x3": 3.0, "bf16": 1.0, "bf16x1": 1.0}[engine]
            speak = cp["sustained_bf16_mfma_peak_tflops"] / div
            sustained = {"clock_ghz": cp["sustained_clock_ghz_all_cus_lds_fed"], "peak": round(speak, 1),
                         "frac": round(in_step / speak, 4), "isolated_frac": round(agg["achieved"] / speak, 4),
                         "best_launch_frac": round(achieved / speak, 4),
                         "note": "bf16 MFMA rate at the shader clock the chip holds with all 256 CUs in an LDS-fed MFMA loop "
                                 "(tools/clock_probe.py, profiles/r02_clock_probe.json); the nominal peak assumes 2.4 GHz"}
        dom = inst["by_symbol"][0]
        mixed = mixed_roofline(args, peak)
        mixed["frac_of_step"] = round(mixed["t_lb_ms"] / res["ms_per_step"], 4)
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this command (tools/pmc_bytes.py ->
        # profiles/r05_pmc_dominant.json: {engine: {kernel symbol: bytes per launch}}); null when no pass exists for this engine / kernel
        dom_traffic, dfile = None, os.path.join(ROOT, "profiles", "r05_pmc_dominant.json")
        if os.path.exists(dfile):
            dom_traffic = json.load(open(dfile)).get(engine, {}).get(dom["kernel"])
        dom_ach = dom["tflops"] or 0.0
        log("[%s] dominant kernel by symbol: %s, %.1f launches x %.1f us = %.2f ms/step, %.1f TFLOP/s = %.3f of %.1f; mixed roofline bound of the step %.2f ms "
            "(conv %.2f + BN %.2f) = %.3f of the measured step" % (dtype_name, dom["kernel"], dom["launches_per_step"], dom["avg_us"], dom["ms_per_step"],
                                                                   dom_ach, dom_ach / peak, peak, mixed["t_lb_ms"], mixed["conv_ms"], mixed["bn_ms"], mixed["frac_of_step"]))
        res["roofline"] = {"bound": "mfma",
                           "kernel": dom["kernel"],
                           "kernel_note": "the conv kernel SYMBOL with the most time in the train step itself (3 steps after the timed region with a start / "
                                          "stop HIP event pair bound to each dispatch on its own stream, csrc/prof.hip); achieved = the algorithmic GFLOP of "
                                          "its launches / the sum of their durations = GFLOP per launch / average launch duration",
                           "achieved": round(dom_ach, 2), "peak": peak, "peak_note": PEAK_NOTE[engine], "unit": "TFLOP/s",
                           "frac": round(dom_ach / peak, 4), "traffic": dom_traffic,
                           "traffic_note": "HBM bytes per launch of the dominant kernel (FETCH_SIZE / WRITE_SIZE passes, profiles/r05_pmc_dominant.json)",
                           "dominant": {"kernel": dom["kernel"], "launches_per_step": dom["launches_per_step"], "ms_per_step": dom["ms_per_step"],
                                        "avg_us": dom["avg_us"], "gflop_per_launch": dom["gflop_per_launch"], "achieved": dom["tflops"],
                                        "frac": round(dom_ach / peak, 4)},
                           "conv_family": {"note": "every conv / weight-gradient launch of the step, durations of overlapping launches summed as in the rocprofv3 "
                                                   "kernel table (profiles/r05_train_summary.md)",
                                           "achieved": round(in_step, 2), "frac": round(in_step / peak, 4)},
                           "mixed": mixed,
                           "best_launch_traffic": traffic,
                           "best_launch_traffic_note": "HBM bytes of the best_launch shape from the committed rocprofv3 --pmc passes (%s)" % os.path.basename(tfile),
                           "conv_ms_per_step": round(inst["ms"], 3), "conv_gflop_per_step": round(inst["gflop"], 1),
                           "conv_launches_per_step": round(inst["launches"], 1),
                           "conv_ceiling_ms_per_step": round(inst["gflop"] / peak, 3),
                           "by_symbol": inst["by_symbol"],
                           "by_kernel": inst["table"],
                           "bn_passes": {"note": "train-mode BN apply / backward passes of the same steps: ALGORITHMIC bytes (every [M][K] tensor a call "
                                                 "reads or writes, 4 B per element; gate bits 1 B per 4) / kernel time, against HBM3E 8 TB/s; counter "
                                                 "bytes: profiles/r05_pmc_bytes.txt (r04_pmc_bytes.txt for round 4)",
                                         "ms_per_step": round(inst["bn_ms"], 3), "gb_per_step": round(inst["bn_gb"], 3),
                                         "tb_per_s": round(inst["bn_gb"] / max(inst["bn_ms"], 1e-9), 3), "passes": inst["bn"]},
                           "step_kernels": {"kernel_ms_per_step": round(inst["kernel_ms"], 3), "launches_per_step": round(inst["kernel_launches"], 1),
                                            "streams": inst["streams"], "top": inst["kernels"]},
                           "isolated": {"note": "every layer shape launched 5x back to back on an idle chip, operands prepared outside the timing: a "
                                                "ceiling for the kernels, NOT what the step achieves",
                                        "achieved": agg["achieved"], "frac": round(agg["achieved"] / peak, 4), "conv_ms_per_step": agg["ms_per_step"],
                                        "conv_gflop_per_step": agg["gflop"],
                                        "split_ms": {"fwd": agg["fwd_ms"], "dgrad": agg["dgrad_ms"], "wgrad": agg["wgrad_ms"]}},
                           "best_launch": best, "sustained": sustained,
                           "train_step_frac": round(res["train_ips"] * wm["train_gflop"] / 1e3 / (peak * world), 4)}
    del model, optimizer
    torch.cuda.empty_cache()
    return res


def kcenter_matrix():
    """config E's selection input at full size: [2975, 2736] f32, non-negative like the pooled post-ReLU decoder features
    (the same array for the GPU loop and for sklearn, so that their pick lists are comparable)"""
    import numpy as np

    return np.abs(np.random.RandomState(5).randn(2975, 2736)).astype(np.float32)


def pool_reader_leg(args, dev):
    """1024 x 2048 Cityscapes-shaped records (pickled uint8 [H, W, 4], the LMDB wire format) -> normalised
    513 x 513 crops + labels on the GPU through dataloaders.dataset.paths_dataset.pool_loader (unpickle, 8 MB host->device
    copy per frame, PIL-exact resize / crop / normalise kernels), against the reference's host pipeline on ONE core
    (DataLoader(num_workers=0), mc_dropout.py:180-181: PIL bilinear + nearest resize, numpy Normalize)"""
    import pickle

    import numpy as np
    from dataloaders.dataset.paths_dataset import DictEnv, pool_loader

    rng = np.random.RandomState(0)
    base = rng.randint(0, 256, (1024, 2048, 4)).astype(np.uint8)
    recs = {}
    for i in range(32):
        r = np.roll(base, 37 * i, axis=1)
        r[:, :, 3] = (r[:, :, 3] % 20)
        recs[("frame_%03d" % i).encode("ascii")] = pickle.dumps(r, protocol=3)
    keys = sorted(recs)
    env = DictEnv(recs)
    for _ in pool_loader(env, keys[:8], args.size, True, args.batch):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for batch in pool_loader(env, keys, args.size, True, args.batch):
        n += batch["image"].shape[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"metric": "pool_reader_images_per_s", "value": round(n / dt, 2), "unit": "images/s", "records": n,
           "record": "1024x2048x4 uint8 (8 MB) -> 3x%dx%d f32 + label" % (args.size, args.size),
           "note": "host->device copy of the raw record included (PCIe); resize / crop / normalise on the GPU"}
    try:
        from PIL import Image

        torch.set_num_threads(1)
        t0 = time.perf_counter()
        m = 0
        while m < 8 and time.perf_counter() - t0 < 5.0:
            rec = pickle.loads(recs[keys[m]])
            img = np.asarray(Image.fromarray(rec[:, :, :3]).resize((1026, 513), resample=Image.BILINEAR))[:, 256:769]
            lab = np.asarray(Image.fromarray(rec[:, :, 3]).resize((1026, 513), resample=Image.NEAREST))[:, 256:769]
            x = img.astype(np.float32)
            x /= 255.0
            x -= (0.485, 0.456, 0.406)
            x /= (0.229, 0.224, 0.225)
            torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1))), torch.from_numpy(lab.astype(np.float32))
            m += 1
        out["cpu_reference_style"] = {"value": round(m / (time.perf_counter() - t0), 2), "unit": "images/s", "cores": 1,
                                      "sample": "%d frames, PIL resize + numpy Normalize on one core (the reference's num_workers=0 loader)" % m}
    except ImportError:
        out["cpu_reference_style"] = None
    log("pool reader: %.1f images/s on the GPU path%s" % (out["value"], (", %.1f on one host core" % out["cpu_reference_style"]["value"]) if out["cpu_reference_style"] else ""))
    return out


def cpu_baseline(args):
    from oracle import deeplab_cpu as O
    from oracle import selection_cpu as S

    # the box's CPU share, not the host's core count: oversubscribing a cgroup-limited box stalls for minutes
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores_box = cores
    # stock PyTorch's intra-op pool stops scaling long before 256 threads on these layer sizes: the train leg is timed with 16
    # threads and with min(available, 64), the better figure is `value` (with its thread count in `cores`), both are kept
    cores_small, cores = max(1, min(cores, 16)), max(1, min(cores, 64))
    log("cpu baseline on %d / %d of the box's %d usable cores (host reports %s) ..." % (cores_small, cores, cores_box, os.cpu_count()))
    torch.set_num_threads(cores)
    s = args.size
    om = O.ODeepLab(args.backbone, 16, args.classes)
    om.train()
    oopt = torch.optim.SGD(om.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    xs, ls = synthetic_batch(1, 65, 65, args.classes, 0)
    S.ce_loss(om(xs), ls).backward()  # thread-pool / allocator warm-up on a tiny input
    xc, lc = synthetic_batch(2, s, s, args.classes, 0)

    def cpu_step():
        oopt.zero_grad()
        lo = S.ce_loss(om(xc), lc)
        lo.backward()
        oopt.step()

    cpu_step()  # untimed: oneDNN primitive creation
    by_threads = {}
    for nthreads in sorted({cores_small, cores}):
        torch.set_num_threads(nthreads)
        cpu_step()
        nsteps, t0 = 0, time.perf_counter()
        while nsteps < 8 and time.perf_counter() - t0 < 8.0:  # ~8 s of CPU work per thread count
            cpu_step()
            nsteps += 1
        dtc = time.perf_counter() - t0
        by_threads[nthreads] = (2 * nsteps / dtc, nsteps, dtc)
        log("cpu train leg, %d threads: %.3f images/s" % (nthreads, 2 * nsteps / dtc))
    cores = max(by_threads, key=lambda t: by_threads[t][0])
    _, nsteps, dtc = by_threads[cores]
    torch.set_num_threads(cores)
    out = {"value": round(2 * nsteps / dtc, 4), "unit": "images/s", "cores": cores, "cores_available": cores_box,
           "cores_host": os.cpu_count(), "kind": "port",
           "by_threads": {str(t): round(v[0], 4) for t, v in by_threads.items()},
           "threads_note": "SURVEY 8d asks for os.cpu_count() threads; the box's cgroup grants %d of the host's %s cores, and stock PyTorch's "
                           "intra-op pool gets SLOWER beyond ~16 threads on these layer sizes (see by_threads), so the leg is timed at 16 and at "
                           "min(granted, 64) threads and the faster one is `value` / `cores`" % (cores_box, os.cpu_count()),
           "sample": "%d train steps (fwd+CE+bwd+SGD) of batch 2 = %d images, %s %dx%d, stock PyTorch CPU fp32 "
                     "(oracle/deeplab_cpu.py), %.1f s" % (nsteps, 2 * nsteps, args.backbone, s, s, dtc)}
    del oopt
    # ---- (ii) MC-dropout scoring, T passes: the reference way (T full forwards, mc_dropout.py:39-40) and with the
    # deterministic prefix (backbone + ASPP) computed once per batch; argmax votes + vote entropy included
    T = args.mc_steps
    om.eval()
    with torch.no_grad():
        m1, m2 = O.dropout_masks(2, T, seed=3)
        for style in ("reference", "hoisted"):
            nimg, t0 = 0, time.perf_counter()
            while nimg < 8 and time.perf_counter() - t0 < 10.0:
                xi, li = synthetic_batch(2, s, s, args.classes, 100000 + nimg)
                if style == "reference":
                    votes = S.mc_votes(om, xi, (m1, m2))
                else:
                    hi, low = om.backbone(xi)
                    a = om.aspp(hi, None)
                    votes = torch.stack([torch.argmax(O._bilinear(om.decoder(a * m1[t][:, :, None, None], low, m2[t])[0], xi.shape[2:]), dim=1)
                                         for t in range(T)], dim=1)
                S.vote_entropy_maps(votes, li, args.classes)
                nimg += 2
            dt = time.perf_counter() - t0
            out["mc_dropout_%s" % style] = {"value": round(nimg / dt, 4), "unit": "pool images/s", "T": T,
                                            "sample": "%d images in %.1f s" % (nimg, dt)}
        # ---- (iii) core-set: pooled decoder features + sklearn fp64 k-center (k = 125, 50 pre-selected) on [2975, 2736]
        om.return_features = True
        nimg, t0 = 0, time.perf_counter()
        feats = []
        while nimg < 8 and time.perf_counter() - t0 < 6.0:
            xi, _ = synthetic_batch(2, s, s, args.classes, 100000 + nimg)
            feats.append(S.coreset_features(om(xi)[1], 64))
            nimg += 2
        dtf = time.perf_counter() - t0
        om.return_features = False
    import numpy as np

    full = kcenter_matrix().astype(np.float64)
    t0 = time.perf_counter()
    cpu_picks, _ = S.kcenter_greedy(full, list(range(50)), 125)
    dtk = time.perf_counter() - t0
    out["core_set"] = {"feature_images_per_s": round(nimg / dtf, 4), "kcenter_seconds": round(dtk, 3), "kcenter_picks": [int(i) for i in cpu_picks],
                       "sample": "%d images of feature extraction in %.1f s; sklearn pairwise_distances fp64 k-center, k=125 on 2975x2736" % (nimg, dtf)}
    # ---- (iv) BASELINE config 0: U-Net(3,4) 128x128 batch 2, 3 SGD steps, stock PyTorch on the CPU (the reference's train.py
    # plumbing case; oracle/unet_cpu.py pinned against models/unet.py by tests/golden/unet_config0.npz) -- loss must decrease
    from oracle import unet_cpu as U

    torch.manual_seed(1234)
    net = U.OUNet(3, 4)
    U.config0_steps(U.OUNet(3, 4), steps=1)  # warm-up on a throw-away copy
    t0 = time.perf_counter()
    losses = U.config0_steps(net, steps=3, lr=0.01)
    dtu = time.perf_counter() - t0
    out["config0_unet"] = {"value": round(6 / dtu, 3), "unit": "images/s", "losses": [round(v, 6) for v in losses],
                           "loss_decreases": bool(losses[2] < losses[0]), "cores": cores,
                           "sample": "3 SGD steps of batch 2, U-Net(3,4) 128x128, %.2f s" % dtu}
    return out




def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None} if isinstance(d, dict) else None


def _compact_roofline(r):
    if not isinstance(r, dict):
        return None
    out = _pick(r, "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "train_step_frac")
    out.setdefault("traffic", None)
    if isinstance(r.get("dominant"), dict):
        out["dominant"] = _pick(r["dominant"], "kernel", "launches_per_step", "avg_us", "ms_per_step", "gflop_per_launch", "frac")
    if isinstance(r.get("conv_family"), dict):
        out["conv_family_frac"] = r["conv_family"].get("frac")
    if isinstance(r.get("mixed"), dict):
        out["t_lb_ms"] = r["mixed"].get("t_lb_ms")
        out["mixed_frac"] = r["mixed"].get("frac_of_step")
    if isinstance(r.get("bn_passes"), dict):
        out["bn"] = {"ms_per_step": r["bn_passes"].get("ms_per_step"), "tb_per_s": r["bn_passes"].get("tb_per_s")}
    return out


def compact_line(full):
    """the ONE JSON line rank 0 prints: the headline keys of the driver's contract, `roofline` and `cpu_baseline` with their contract
    keys, one number per informational leg -- below LINE_LIMIT bytes whatever the tables hold (they go to DETAIL_FILE: `full`)."""
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data")}
    line["config"] = _pick(full.get("config") or {}, "workload", "global_batch", "parallelism", "bn", "f32_mma", "hip_graph", "final_loss", "replicas_identical", "ddp")
    mc = full.get("mc_dropout")
    line["mc_dropout"] = _pick(mc, "value", "unit", "T", "pool_images", "scoring_batch", "frac_of_mfma_peak", "executed_frac_of_mfma_peak") if mc else None
    cs = full.get("core_set")
    if cs:
        line["core_set"] = _pick(cs, "value", "unit", "frac_of_mfma_peak")
        kc = cs.get("kcenter") or {}
        line["core_set"]["kcenter_s"] = kc.get("seconds")
        line["core_set"]["picks_equal_sklearn"] = kc.get("picks_equal_sklearn_on_same_matrix")
    else:
        line["core_set"] = None
    pr = full.get("pool_reader")
    line["pool_reader"] = {"value": pr.get("value"), "unit": pr.get("unit"), "cpu_1core": (pr.get("cpu_reference_style") or {}).get("value")} if pr else None
    line["roofline"] = _compact_roofline(full.get("roofline"))
    cpu = full.get("cpu_baseline")
    if cpu:
        c = _pick(cpu, "value", "unit", "cores", "kind")
        c["sample"] = str(cpu.get("sample", ""))[:160]
        c["mc_dropout_reference"] = (cpu.get("mc_dropout_reference") or {}).get("value")
        c["mc_dropout_hoisted"] = (cpu.get("mc_dropout_hoisted") or {}).get("value")
        c["core_set_features"] = (cpu.get("core_set") or {}).get("feature_images_per_s")
        c["kcenter_s"] = (cpu.get("core_set") or {}).get("kcenter_seconds")
        c["config0_unet"] = _pick(cpu.get("config0_unet") or {}, "value", "loss_decreases")
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    modes = {}
    for name in ("f32_mfma_mode", "bf16x1_perf_mode", "bf16_perf_mode", "f32_parity_mode"):
        m = full.get(name)
        if not isinstance(m, dict):
            continue
        if "error" in m:
            modes[name] = {"error": str(m["error"])[:120]}
            continue
        r = m.get("roofline") or {}
        modes[name] = {"train": m.get("train_images_per_s"), "graph": m.get("hip_graph"), "mc": (m.get("mc_dropout") or {}).get("value"),
                       "core_set": (m.get("core_set") or {}).get("value"), "dominant_frac": r.get("frac"), "train_step_frac": r.get("train_step_frac"),
                       "mixed_frac": (r.get("mixed") or {}).get("frac_of_step")}
    line["modes"] = modes or None
    line["detail"] = full.get("detail")
    text = json.dumps(line)
    if len(text) >= LINE_LIMIT:  # never let a long string cost the driver its line: drop the optional parts, longest first
        for k in ("modes", "pool_reader", "core_set", "detail"):
            line[k] = None
            if len(json.dumps(line)) < LINE_LIMIT:
                break
    return line


def write_detail(full):
    """every table of the run (by_symbol / by_kernel, bn_passes, step kernels, per-mode rooflines, the CPU legs): gpurun_out/ when present
    (it is merged back from the GPU box), else the repo root; -> the path written, relative to the repo root"""
    d = os.path.join(ROOT, "gpurun_out")
    path = os.path.join(d if os.path.isdir(d) else ROOT, DETAIL_FILE)
    try:
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError as exc:
        log("could not write %s: %r" % (path, exc))
        return None


def spawn_ranks(args):
    """`python bench.py --gpus N` without a torchrun environment: start N ranks (one per GPU) through torch.distributed.run as a
    CHILD process -- this parent has not touched the GPU and never does -- relay its output, exit with its code"""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log("--gpus %d without WORLD_SIZE: launching %s" % (args.gpus, " ".join(cmd[1:8])))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))       # (before any GPU call in this process)
    if world_env is not None and int(world_env) != args.gpus:
        print("bench.py: --gpus %d contradicts WORLD_SIZE=%s (launch with --nproc-per-node %d, or drop the launcher and let "
              "`--gpus N` start the ranks)" % (args.gpus, world_env, args.gpus), file=sys.stderr)
        sys.exit(2)
    env = Env()
    env.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    env.world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # rehearsal knobs for a 1-GPU box: DASS_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # DASS_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device); never set by the driver
    if os.environ.get("DASS_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    env.dev = torch.device("cuda", local_rank)
    env.dist = None
    # (DASS_DIST_FORCE=1 with one rank: the multi-process step -- graph A, RCCL all-reduce, graph B -- rehearsed on a one-GPU box with the real backend)
    if env.world > 1 or os.environ.get("DASS_DIST_FORCE") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend=os.environ.get("DASS_BENCH_BACKEND", "nccl"))  # nccl = RCCL on ROCm
        env.dist = dist

    head = run_mode(args, env, args.dtype, args.steps, args.warmup, args.f32_mma)
    others, failed = {}, {}

    def informational(name, *a):
        """an informational leg must not cost the headline its line: on one GPU a failure is recorded in the line instead of raised
        (with several ranks it is raised: a rank that skipped a leg would leave the others waiting in its collectives)"""
        if env.world > 1:
            others[name] = run_mode(args, env, *a)
            return
        try:
            others[name] = run_mode(args, env, *a)
        except Exception as exc:  # noqa: BLE001
            import traceback

            log("informational leg %s FAILED: %r" % (name, exc))
            traceback.print_exc(file=sys.stderr)
            failed[name] = repr(exc)[:400]
            torch.cuda.synchronize()
            torch.cuda.empty_cache()

    if not args.no_second_dtype:
        k2 = max(3, args.steps // 2)
        if args.dtype == "f32":
            if args.f32_mma != "f32":
                informational("f32_mfma_mode", "f32", k2, 4, "f32")
            if args.f32_mma != "bf16x1":
                informational("bf16x1_perf_mode", "f32", k2, 4, "bf16x1")  # one bf16 part per operand on the pipelined kernels, step as a hipGraph
            informational("bf16_perf_mode", "bf16", k2, 4)  # 4 warm-up steps: the allocator re-grows after empty_cache()
        else:
            informational("f32_parity_mode", "f32", k2, 4, args.f32_mma)
    cpu = reader = None
    if env.rank == 0 and env.world == 1 and not args.no_pool_reader:
        reader = pool_reader_leg(args, env.dev)
    if env.rank == 0 and env.world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    if cpu is not None and head["coreset"] is not None:
        # config E parity at full size: the device k-center loop and sklearn's saw the same matrix
        gpu_picks = head["coreset"]["kcenter"].pop("picks")
        head["coreset"]["kcenter"]["picks_equal_sklearn_on_same_matrix"] = gpu_picks == cpu["core_set"].pop("kcenter_picks")
    elif head["coreset"] is not None:
        head["coreset"]["kcenter"].pop("picks", None)
    for other in others.values():
        if other["coreset"] is not None:
            other["coreset"]["kcenter"].pop("picks", None)
    if env.rank == 0:
        b, s, world = args.batch, args.size, env.world
        line = {"metric": "train_images_per_s (DeepLab-v3+ %s %dx%d; + mc_dropout pool-images/s in 'mc_dropout')"
                          % ({"resnet101": "R101", "resnet": "R50", "mobilenet": "MobileNetV2"}.get(args.backbone, args.backbone), s, s),
                "value": round(head["train_ips"], 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(head["ms_per_step"], 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": "DeepLab-v3+ %s os16 %d-class %dx%d train step (fwd+CE+bwd+SGD), per-GPU batch %d"
                                       % (args.backbone, args.classes, s, s, b),
                           "global_batch": b * world, "parallelism": "dp%d" % world, "bn": "per-GPU",
                           "f32_mma": args.f32_mma if args.dtype == "f32" else None, "hip_graph": head.get("graph", False),
                           "final_loss": round(head["final_loss"], 5), "replicas_identical": head.get("replicas_identical"), "ddp": head.get("ddp")},
                "mc_dropout": head["mc"], "core_set": head["coreset"], "pool_reader": reader, "roofline": head["roofline"],
                "cpu_baseline": cpu}
        notes = {"bf16_perf_mode": "informational; bf16 storage does not meet the parity bar (tests/test_bf16_gpu.py measures the deviation)",
                 "bf16x1_perf_mode": "informational; f32 tensors, the pipelined pre-split kernels with ONE bf16 part per operand and one product "
                                     "(what autocast-bf16 multiplies); not parity-grade (tests/test_bf16_gpu.py)",
                 "f32_mfma_mode": "same f32 tensors, convs on v_mfma_f32_32x32x2_f32; parity-grade as well",
                 "f32_parity_mode": "the parity mode (f32 tensors)"}
        for name, other in others.items():
            line[name] = {"train_images_per_s": round(other["train_ips"], 3), "ms_per_step": round(other["ms_per_step"], 3),
                          "final_loss": round(other["final_loss"], 5), "hip_graph": other.get("graph", False), "mc_dropout": other["mc"], "core_set": other["coreset"],
                          "roofline": other["roofline"],
                          "note": notes[name]}
        for name, err in failed.items():
            line[name] = {"error": err}
        line["detail"] = write_detail(line)
        log("full tables: %s (%d bytes as one line)" % (line["detail"], len(json.dumps(line))))
        short = json.dumps(compact_line(line))
        assert len(short) < LINE_LIMIT, len(short)
        sys.stderr.flush()
        print(short, flush=True)
    if env.dist is not None:
        env.dist.destroy_process_group()


if __name__ == "__main__":
    main()
